// binning.hip -- depth sort, offset scan, instance emission, tile sort, tile ranges (gfx950).
//
// The reference builds R = sum(tiles_touched) 64-bit keys (tile << 32 | depth bits)
// in Gaussian order (duplicateWithKeys, rasterizer_impl.cu:71-112) and runs an
// 8-pass 64-bit CUB radix sort over all of them (rasterizer_impl.cu:325-330).
// The order it defines is (tile id, depth bits, Gaussian id) -- the last because the
// sort is stable and emission is in Gaussian-id order.
//
// This implementation produces the IDENTICAL permutation with far less HBM
// traffic by splitting the key:
//   1. stable 32-bit radix sort of the P (depth bits, Gaussian id) pairs   -- 4 passes over P
//   2. exclusive scan of tiles_touched in that depth order                 -- R, per-Gaussian offsets
//   3. instance emission in depth order, one lane per output slot          -- coalesced 8 B / instance
//   4. stable radix sort of the R (tile id, Gaussian id) pairs on the
//      ceil(log2 T) tile bits only                                         -- 2 passes over R for T <= 65536
//   5. tile ranges from the sorted tile ids (identifyTileRanges, rasterizer_impl.cu:117-139)
// Stability of (4) preserves the (depth bits, Gaussian id) order inside each tile.
//
// Radix pass = 3 launches: per-workgroup digit histogram -> per-digit scan over
// workgroups -> stable scatter (wave64 match-any ranking via 8 ballots, per-wave
// digit counters in LDS, LDS reorder so the global writes are contiguous per digit run).
// Small inputs (the depth sort) use 1024-key chunks so they still fill the chip.
// All integer work, HBM-bound; no MFMA.
#include "fdgs_common.h"

namespace fdgs
{
	// ------------------------------------------------------------------
	// radix sort
	// ------------------------------------------------------------------

	// hist[d * nblocks + b] = number of keys of workgroup-chunk b whose digit is d
	template <int ITEMS>
	__global__ void __launch_bounds__(SORT_THREADS) radix_hist_kernel(const uint32_t* __restrict__ keys, int n, int shift,
	                                                                  uint32_t* __restrict__ hist, int nblocks)
	{
		__shared__ uint32_t h[RADIX];
		h[threadIdx.x] = 0;
		__syncthreads();
		const int base = blockIdx.x * (SORT_THREADS * ITEMS);
#pragma unroll
		for (int i = 0; i < ITEMS; i++)
		{
			const int k = base + i * SORT_THREADS + threadIdx.x;
			if (k < n) atomicAdd(&h[(keys[k] >> shift) & (RADIX - 1)], 1u);
		}
		__syncthreads();
		hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
	}

	// One workgroup per digit: exclusive scan of that digit's counts over the workgroup
	// chunks, in place; the digit total goes to totals[d].
	__global__ void __launch_bounds__(256) radix_scan_kernel(uint32_t* __restrict__ hist, int nblocks, uint32_t* __restrict__ totals)
	{
		__shared__ uint32_t wave_sums[4];
		__shared__ uint32_t carry_s;
		uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		if (threadIdx.x == 0) carry_s = 0;
		__syncthreads();
		for (int base = 0; base < nblocks; base += 256)
		{
			const int i = base + threadIdx.x;
			const uint32_t v = (i < nblocks) ? row[i] : 0u;
			uint32_t incl = v;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1)
			{
				const uint32_t t = __shfl_up(incl, o);
				if (lane >= o) incl += t;
			}
			if (lane == 63) wave_sums[wave] = incl;
			__syncthreads();
			uint32_t wbase = 0;
			for (int w = 0; w < wave; w++) wbase += wave_sums[w];
			const uint32_t carry = carry_s;
			if (i < nblocks) row[i] = carry + wbase + incl - v;
			__syncthreads();
			if (threadIdx.x == 255) carry_s = carry + wbase + incl;
			__syncthreads();
		}
		if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
	}

	// Stable scatter.  Wave w of the workgroup owns the contiguous sub-chunk
	// [base + w*64*ITEMS, base + (w+1)*64*ITEMS) and walks it in ITEMS rounds of 64 consecutive keys:
	//   phase 1  rank of every key among the equal-digit keys of its wave (wave64 match-any: 8 ballots);
	//   phase 2  per-digit exclusive scans: over the 4 waves, and over the 256 digits (workgroup-local
	//            position of each digit's run); global base of the run from the scanned histograms;
	//   phase 3  keys / values are first written to LDS at their workgroup-local sorted position, then
	//            copied out in that order, so consecutive lanes write consecutive global addresses within
	//            each digit run instead of 4-byte stores scattered over 256 runs.
	template <int ITEMS>
	__global__ void __launch_bounds__(SORT_THREADS) radix_scatter_kernel(
		const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
		uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
		int n, int shift, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals, int nblocks)
	{
		constexpr int WAVES = SORT_THREADS / WAVE;           // 4
		constexpr int PER_WAVE = WAVE * ITEMS;
		constexpr int CHUNK = SORT_THREADS * ITEMS;
		__shared__ uint32_t cnt[WAVES][RADIX];               // per-wave digit counters, then per-wave local bases
		__shared__ uint32_t run_start[RADIX];                // workgroup-local start of each digit run
		__shared__ uint32_t run_gbase[RADIX];                // global position of each digit run
		__shared__ uint32_t ws[WAVES];
		__shared__ uint32_t s_key[CHUNK];
		__shared__ uint32_t s_val[CHUNK];

		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		for (int i = threadIdx.x; i < WAVES * RADIX; i += SORT_THREADS) (&cnt[0][0])[i] = 0;

		// exclusive scan of the 256 digit totals -> global base of each digit, plus this chunk's offset in the run
		{
			const uint32_t v = totals[threadIdx.x];
			uint32_t incl = v;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1)
			{
				const uint32_t t = __shfl_up(incl, o);
				if (lane >= o) incl += t;
			}
			if (lane == 63) ws[wave] = incl;
			__syncthreads();
			uint32_t wbase = 0;
			for (int w = 0; w < wave; w++) wbase += ws[w];
			run_gbase[threadIdx.x] = wbase + incl - v + hist[(size_t)threadIdx.x * nblocks + blockIdx.x];
		}
		__syncthreads();

		const int wbase_idx = blockIdx.x * CHUNK + wave * PER_WAVE;
		uint32_t key[ITEMS], val[ITEMS], rank[ITEMS];
		const unsigned long long lt_mask = (1ull << lane) - 1ull;

		// phase 1
#pragma unroll
		for (int r = 0; r < ITEMS; r++)
		{
			const int k = wbase_idx + r * WAVE + lane;
			const bool valid = k < n;
			key[r] = valid ? keys_in[k] : 0xFFFFFFFFu;
			val[r] = valid ? vals_in[k] : 0u;
			const uint32_t d = (key[r] >> shift) & (RADIX - 1);
			unsigned long long peers = __ballot(valid);
#pragma unroll
			for (int b = 0; b < RADIX_BITS; b++)
			{
				const unsigned long long bal = __ballot((d >> b) & 1u);
				peers &= ((d >> b) & 1u) ? bal : ~bal;
			}
			const int leader = __ffsll((long long)peers) - 1;
			uint32_t prev = 0;
			if (valid && lane == leader)
			{
				prev = cnt[wave][d];
				cnt[wave][d] = prev + (uint32_t)__popcll(peers);
			}
			prev = __shfl(prev, leader < 0 ? 0 : leader);
			rank[r] = prev + (uint32_t)__popcll(peers & lt_mask);
		}
		__syncthreads();

		// phase 2: thread d owns digit d
		{
			const int d = threadIdx.x;
			uint32_t run = 0;
#pragma unroll
			for (int w = 0; w < WAVES; w++)
			{
				const uint32_t c = cnt[w][d];
				cnt[w][d] = run;      // offset of wave w inside the digit run of this workgroup
				run += c;
			}
			// exclusive scan of the run lengths over the digits
			uint32_t incl = run;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1)
			{
				const uint32_t t = __shfl_up(incl, o);
				if (lane >= o) incl += t;
			}
			if (lane == 63) ws[wave] = incl;
			__syncthreads();
			uint32_t wb = 0;
			for (int w = 0; w < wave; w++) wb += ws[w];
			run_start[d] = wb + incl - run;
		}
		__syncthreads();

		// phase 3a: into LDS at the workgroup-local sorted position
#pragma unroll
		for (int r = 0; r < ITEMS; r++)
		{
			const int k = wbase_idx + r * WAVE + lane;
			if (k < n)
			{
				const uint32_t d = (key[r] >> shift) & (RADIX - 1);
				const uint32_t lp = run_start[d] + cnt[wave][d] + rank[r];
				s_key[lp] = key[r];
				s_val[lp] = val[r];
			}
		}
		__syncthreads();
		// phase 3b: out in sorted order (contiguous within each digit run)
		const int nvalid = min(CHUNK, n - blockIdx.x * CHUNK);
#pragma unroll
		for (int r = 0; r < ITEMS; r++)
		{
			const int i = r * SORT_THREADS + threadIdx.x;
			if (i < nvalid)
			{
				const uint32_t kk = s_key[i];
				const uint32_t d = (kk >> shift) & (RADIX - 1);
				const uint32_t pos = run_gbase[d] + ((uint32_t)i - run_start[d]);
				keys_out[pos] = kk;
				vals_out[pos] = s_val[i];
			}
		}
	}

	template <int ITEMS>
	static void radix_pass(uint32_t* keys[2], uint32_t* vals[2], int cur, int n, int bit, uint32_t* hist, hipStream_t stream)
	{
		const int nblocks = div_up(n, SORT_THREADS * ITEMS);
		uint32_t* totals = hist + (size_t)RADIX * nblocks; // block counts, then the RADIX digit totals
		hipLaunchKernelGGL(radix_hist_kernel<ITEMS>, dim3(nblocks), dim3(SORT_THREADS), 0, stream, keys[cur], n, bit, hist, nblocks);
		hipLaunchKernelGGL(radix_scan_kernel, dim3(RADIX), dim3(256), 0, stream, hist, nblocks, totals);
		hipLaunchKernelGGL(radix_scatter_kernel<ITEMS>, dim3(nblocks), dim3(SORT_THREADS), 0, stream,
		                   keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n, bit, hist, totals, nblocks);
	}

	hipError_t radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], int n, int bit_lo, int bit_hi,
	                            uint32_t* hist, hipStream_t stream, int* result)
	{
		int cur = 0;
		if (n > 0)
		{
			const int items = sort_items_for(n);
			for (int bit = bit_lo; bit < bit_hi; bit += RADIX_BITS)
			{
				if (items == SORT_ITEMS_SMALL) radix_pass<SORT_ITEMS_SMALL>(keys, vals, cur, n, bit, hist, stream);
				else radix_pass<SORT_ITEMS>(keys, vals, cur, n, bit, hist, stream);
				cur ^= 1;
			}
		}
		*result = cur;
		return hipGetLastError();
	}

	// ------------------------------------------------------------------
	// offsets[j] = exclusive scan over j of tiles_touched[order[j]]  (3 launches)
	// ------------------------------------------------------------------

	__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* total, uint32_t* smem /* >= 16 */)
	{
		// exclusive scan of one value per thread of a 256-thread workgroup
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		uint32_t incl = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1)
		{
			const uint32_t t = __shfl_up(incl, o);
			if (lane >= o) incl += t;
		}
		if (lane == 63) smem[wave] = incl;
		__syncthreads();
		uint32_t wbase = 0;
		for (int w = 0; w < wave; w++) wbase += smem[w];
		if (threadIdx.x == 255) *total = wbase + incl;
		return wbase + incl - v;
	}

	// Single pass with decoupled look-back: every workgroup scans its 1024-element chunk, publishes its total as an
	// AGGREGATE, looks back over its predecessors' published words (a whole wave at a time: 64 predecessors per load)
	// until it meets one that already carries an inclusive PREFIX, publishes its own prefix and writes its offsets.
	// One launch instead of three (reduce / scan of the chunk sums / final).  A word = flag << 62 | value, written and
	// read whole, so no fences are needed; the words must be zero on entry (the forward's first kernel clears them).
	// Safe only while all workgroups are resident at once (they spin on lower-numbered workgroups), hence the
	// three-launch fallback for very large P.
	constexpr unsigned long long LB_AGG = 1ull << 62, LB_PREFIX = 2ull << 62, LB_FLAGS = 3ull << 62;

	__global__ void __launch_bounds__(256) offsets_lookback_kernel(const uint32_t* __restrict__ tiles, const uint32_t* __restrict__ order,
	                                                              int P, unsigned long long* __restrict__ state, uint32_t* __restrict__ total_out,
	                                                              uint32_t* __restrict__ offsets)
	{
		constexpr int IT = SCAN_CHUNK / 256;   // consecutive elements per thread
		__shared__ uint32_t smem[16];
		__shared__ uint32_t total_s, excl_s;
		const int b = blockIdx.x, nblocks = gridDim.x;
		const int first_j = b * SCAN_CHUNK + threadIdx.x * IT;
		// serial scan of the thread's IT elements, one workgroup scan of the thread totals
		uint32_t ex[IT], run = 0;
#pragma unroll
		for (int i = 0; i < IT; i++)
		{
			const int j = first_j + i;
			const uint32_t v = (j < P) ? tiles[order[j]] : 0u;
			ex[i] = run;
			run += v;
		}
		const uint32_t tbase = block_excl_scan_256(run, &total_s, smem);
		__syncthreads();
		const uint32_t chunk_total = total_s;
		if (threadIdx.x < WAVE)
		{
			const int lane = threadIdx.x;
			if (lane == 0)
				__hip_atomic_store(&state[b], (b == 0 ? LB_PREFIX : LB_AGG) | chunk_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			uint32_t excl = 0;
			for (int j = b - 1; j >= 0; j -= WAVE)
			{
				const int idx = j - lane;   // lane 0 looks at the nearest predecessor
				unsigned long long st = LB_PREFIX; // virtual predecessor of workgroup 0: prefix 0
				if (idx >= 0)
				{
					do { st = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((st & LB_FLAGS) == 0ull);
				}
				const unsigned long long pmask = __ballot((st & LB_FLAGS) == LB_PREFIX);
				const int first = pmask ? __ffsll((long long)pmask) - 1 : WAVE;   // nearest predecessor that already has a prefix
				uint32_t v = (lane <= first) ? (uint32_t)(st & 0xFFFFFFFFull) : 0u;
#pragma unroll
				for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
				excl += v;
				if (pmask) break;
			}
			if (lane == 0)
			{
				if (b > 0) __hip_atomic_store(&state[b], LB_PREFIX | (unsigned long long)(excl + chunk_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				excl_s = excl;
				if (b == nblocks - 1) *total_out = excl + chunk_total;
			}
		}
		__syncthreads();
		const uint32_t base = excl_s + tbase;
#pragma unroll
		for (int i = 0; i < IT; i++)
		{
			const int j = first_j + i;
			if (j < P) offsets[j] = base + ex[i];
		}
	}

	// phase A: per-chunk (4096) sums of the gathered counts
	__global__ void __launch_bounds__(256) offsets_reduce_kernel(const uint32_t* __restrict__ tiles, const uint32_t* __restrict__ order,
	                                                            int P, uint32_t* __restrict__ block_sums)
	{
		__shared__ uint32_t ws[4];
		uint32_t s = 0;
		const int base = blockIdx.x * SCAN_CHUNK;
		for (int i = 0; i < SCAN_CHUNK / 256; i++)
		{
			const int j = base + i * 256 + threadIdx.x;
			if (j < P) s += tiles[order[j]];
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
		if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
		__syncthreads();
		if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
	}

	// phase B: one workgroup scans the chunk sums in place (exclusive); grand total -> *total_out
	__global__ void __launch_bounds__(256) offsets_scan_sums_kernel(uint32_t* __restrict__ block_sums, int nblocks, uint32_t* __restrict__ total_out)
	{
		__shared__ uint32_t smem[16];
		__shared__ uint32_t total_s, carry_s;
		if (threadIdx.x == 0) carry_s = 0;
		__syncthreads();
		for (int base = 0; base < nblocks; base += 256)
		{
			const int i = base + threadIdx.x;
			const uint32_t v = (i < nblocks) ? block_sums[i] : 0u;
			const uint32_t ex = block_excl_scan_256(v, &total_s, smem);
			const uint32_t carry = carry_s;
			if (i < nblocks) block_sums[i] = carry + ex;
			__syncthreads();
			if (threadIdx.x == 0) carry_s = carry + total_s;
			__syncthreads();
		}
		if (threadIdx.x == 0) *total_out = carry_s;
	}

	// phase C: final exclusive offsets
	__global__ void __launch_bounds__(256) offsets_final_kernel(const uint32_t* __restrict__ tiles, const uint32_t* __restrict__ order,
	                                                           int P, const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ offsets)
	{
		__shared__ uint32_t smem[16];
		__shared__ uint32_t total_s;
		uint32_t carry = block_sums[blockIdx.x];
		const int base = blockIdx.x * SCAN_CHUNK;
		for (int i = 0; i < SCAN_CHUNK / 256; i++)
		{
			const int j = base + i * 256 + threadIdx.x;
			const uint32_t v = (j < P) ? tiles[order[j]] : 0u;
			const uint32_t ex = block_excl_scan_256(v, &total_s, smem);
			if (j < P) offsets[j] = carry + ex;
			__syncthreads();
			carry += total_s;
			__syncthreads();
		}
	}

	hipError_t launch_offsets_scan(const uint32_t* tiles_touched, const uint32_t* order, int P,
	                               uint32_t* offsets, uint32_t* block_sums, hipStream_t stream)
	{
		const int nblocks = div_up(P, SCAN_CHUNK);
		uint32_t* total = scan_total_ptr(block_sums, P);
		if (nblocks <= LOOKBACK_MAX_BLOCKS)
		{
			// the state words were cleared by the forward's first kernel (preprocess_fwd.hip)
			hipLaunchKernelGGL(offsets_lookback_kernel, dim3(nblocks), dim3(256), 0, stream, tiles_touched, order, P,
			                   reinterpret_cast<unsigned long long*>(block_sums), total, offsets);
			return hipGetLastError();
		}
		uint32_t* sums = block_sums;   // fallback: reduce / scan / final; the chunk sums reuse the front of the state area
		hipLaunchKernelGGL(offsets_reduce_kernel, dim3(nblocks), dim3(256), 0, stream, tiles_touched, order, P, sums);
		hipLaunchKernelGGL(offsets_scan_sums_kernel, dim3(1), dim3(256), 0, stream, sums, nblocks, total);
		hipLaunchKernelGGL(offsets_final_kernel, dim3(nblocks), dim3(256), 0, stream, tiles_touched, order, P, sums, offsets);
		return hipGetLastError();
	}

	// ------------------------------------------------------------------
	// instance emission: one lane per output slot (coalesced key/value stores)
	// ------------------------------------------------------------------
	// Slot s belongs to the depth-ordered Gaussian j with offsets[j] <= s < offsets[j+1].  A workgroup owns
	// EMIT_SLOTS consecutive slots: one lane binary-searches the owner of the first slot in global memory, the
	// workgroup copies the next EMIT_SLOTS+1 offsets (every visible Gaussian owns >= 1 slot, so all owners of
	// the workgroup's slots are in that window) into LDS, and every lane then searches the LDS window -- 10
	// LDS steps instead of 18 dependent global loads per slot.  Within a Gaussian the slots enumerate its tile
	// rectangle y-major like the reference (rasterizer_impl.cu:99-109); the final order only depends on
	// (tile, depth order).
	constexpr int EMIT_THREADS = 256;
	constexpr int EMIT_PER_THREAD = 4;
	constexpr int EMIT_SLOTS = EMIT_THREADS * EMIT_PER_THREAD;

	__global__ void __launch_bounds__(EMIT_THREADS) emit_instances_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
	                                                                    const ushort4* __restrict__ rect, int P, int R, int grid_x,
	                                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
	{
		__shared__ uint32_t win[EMIT_SLOTS + 1];
		__shared__ int j0_s;
		const int s0 = blockIdx.x * EMIT_SLOTS;
		if (threadIdx.x < WAVE)
		{
			// largest j with offsets[j] <= s0 (offsets is non-decreasing; culled Gaussians sit at the end with offset R).
			// 64-ary search by one wave: 3 dependent loads for 300 k Gaussians instead of 18 for a binary search by one lane.
			const int lane = threadIdx.x;
			int lo = 0, n = P;                       // invariant: offsets[lo] <= s0 (offsets[0] = 0), answer in [lo, lo + n)
			while (n > 1)
			{
				const int step = (n + WAVE - 1) / WAVE;
				const int idx = lo + lane * step;
				const bool ok = idx < lo + n && offsets[idx] <= (uint32_t)s0;
				const unsigned long long m = __ballot(ok);   // a prefix of the lanes (monotone), lane 0 always set
				const int k = 63 - __clzll((long long)m);
				n = min(step, lo + n - (lo + k * step));
				lo += k * step;
			}
			if (lane == 0) j0_s = lo;
		}
		__syncthreads();
		const int j0 = j0_s;
		const int nwin = min(EMIT_SLOTS + 1, P - j0);
		for (int i = threadIdx.x; i < nwin; i += EMIT_THREADS) win[i] = offsets[j0 + i];
		__syncthreads();
#pragma unroll
		for (int k = 0; k < EMIT_PER_THREAD; k++)
		{
			const int s = s0 + k * EMIT_THREADS + threadIdx.x;
			if (s >= R) continue;
			int lo = 0, hi = nwin - 1;
			while (lo < hi)
			{
				const int mid = (lo + hi + 1) >> 1;
				if (win[mid] <= (uint32_t)s) lo = mid; else hi = mid - 1;
			}
			const uint32_t g = order[j0 + lo];
			const ushort4 rc = rect[g];
			const uint32_t local = (uint32_t)s - win[lo];
			const uint32_t w = (uint32_t)(rc.z - rc.x);
			const uint32_t ty = rc.y + local / w, tx = rc.x + local % w;
			keys[s] = ty * (uint32_t)grid_x + tx;
			vals[s] = g;
		}
	}

	hipError_t launch_emit_instances(const uint32_t* order, const uint32_t* offsets, const uint16_t* rect,
	                                 int P, int R, int grid_x, uint32_t* keys, uint32_t* vals, hipStream_t stream)
	{
		if (R <= 0) return hipSuccess;
		hipLaunchKernelGGL(emit_instances_kernel, dim3(div_up(R, EMIT_SLOTS)), dim3(EMIT_THREADS), 0, stream,
		                   order, offsets, reinterpret_cast<const ushort4*>(rect), P, R, grid_x, keys, vals);
		return hipGetLastError();
	}

	// ------------------------------------------------------------------
	// tile ranges (identifyTileRanges, rasterizer_impl.cu:117-139)
	// ------------------------------------------------------------------
	__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t* __restrict__ tile_keys, int R, uint2* __restrict__ ranges)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= R) return;
		const uint32_t cur = tile_keys[i];
		if (i == 0) ranges[cur].x = 0;
		else
		{
			const uint32_t prev = tile_keys[i - 1];
			if (cur != prev) { ranges[prev].y = (uint32_t)i; ranges[cur].x = (uint32_t)i; }
		}
		if (i == R - 1) ranges[cur].y = (uint32_t)R;
	}

	hipError_t launch_tile_ranges(const uint32_t* sorted_tile_keys, int R, int T, uint32_t* ranges, hipStream_t stream)
	{
		hipError_t e = hipMemsetAsync(ranges, 0, (size_t)T * 8, stream); // rasterizer_impl.cu:332
		if (e != hipSuccess) return e;
		if (R > 0)
			hipLaunchKernelGGL(tile_ranges_kernel, dim3(div_up(R, 256)), dim3(256), 0, stream, sorted_tile_keys, R, reinterpret_cast<uint2*>(ranges));
		return hipGetLastError();
	}
}

// tilebin.hip -- per-tile instance lists in the reference's order, without a global sort (gfx950).
//
// What the reference does: duplicateWithKeys builds R = sum(tiles_touched) 64-bit keys (tile << 32 | depth bits) in
// Gaussian order (rasterizer_impl.cu:71-112), an 8-pass 64-bit CUB radix sort orders all of them
// (rasterizer_impl.cu:325-330) and identifyTileRanges finds the tile boundaries (:117-139).  The order this defines
// inside a tile is (depth bits, Gaussian id): the sort is stable and emission is in id order.
//
// What this file does instead -- ONE pass of counting, ONE pass of scattering, ONE local sort per tile:
//   tile_bin<false>   every (Gaussian, covered tile) instance adds 1 to its tile's counter            (count)
//   tile_scan         exclusive scan of the tile counters -> list start of every tile, R, longest list (one workgroup)
//   [host: reads R and the longest list (8 bytes) -- the forward's one synchronisation, rasterizer_impl.cu:302 --
//    and gets the binning buffer from the allocator]
//   tile_bin<true>    every instance takes the next free slot of its tile (returning atomic on the scanned counter)
//                     and drops its (depth bits, id) pair there                                        (scatter)
//   tile_sort         one workgroup per tile brings the pairs of its list into (depth bits, id) order in LDS and
//                     writes point_list and the tile's range
// The arrival order inside a tile is whatever the atomics made it; (depth bits, id) is a total order (ids are unique),
// so point_list / ranges are deterministic and bit-identical to the reference's sorted values.
// R-sized traffic: 8 B scattered write + 8 B read + 4 B write per instance (the reference's sort moves ~200 B per
// instance), and 4 launches instead of CUB's ~20.  All integer work; no MFMA.
//
// Load balance of tile_bin: a wave owns 64 consecutive Gaussians, scans their tile counts and then walks the wave's
// instances 64 at a time (lane -> instance; owner found by a 6-step search in an LDS copy of the scan), so a Gaussian
// covering hundreds of tiles does not hold 63 idle lanes hostage (the reference loops per thread).
// Atomics: a workgroup bins a batch of 1024+ Gaussians into an LDS histogram over all tiles first and touches the
// global counters once per (workgroup, 32 consecutive tiles) -- see the note above tile_bin_lds_kernel.
//
// tile_sort, a list of n entries held as 64-bit keys (depth bits << 32 | id):
//   n <= 4096  one most-significant-digit step on the bits that vary inside THIS tile (1024 buckets between the
//              list's own min and max depth bits; LDS histogram with returning atomics, scan, scatter into LDS), then
//              every entry counts the smaller keys of its bucket (buckets hold ~1 entry).  When a bucket is crowded
//              (many equal or nearly equal depths) the tile falls back to a bitonic sort of the keys in LDS.
//   n  > 4096  bitonic sort in global scratch (R keys, requested from the allocator only when such a list exists).
#include "fdgs_common.h"

namespace fdgs
{
	// ------------------------------------------------------------------------------------------------
	// count / scatter
	// ------------------------------------------------------------------------------------------------
	// Device-scope atomics are the scarce resource here: MI355X retires only ~13 atomic REQUESTS per ns chip-wide
	// (tools/probe/README.md: 2.6 M single-lane requests take 55 us however the counters are laid out), so a workgroup
	// first bins its batch of Gaussians into an LDS histogram over all tiles and then adds the histogram to the global
	// counters with lane = tile: 32 consecutive counters are one 128-byte request.  In the scatter pass the same flush uses
	// returning atomics: the workgroup reserves, per tile, a contiguous piece of the tile's list, and hands out its
	// slots with LDS atomics.

	// A lane walks the tiles of its own Gaussian's rectangle (short independent loops: no cross-lane dependency, the
	// LDS atomics pipeline back to back); rectangles of more than BIN_BIG tiles are left out and walked afterwards by
	// the whole wave, so one huge splat does not keep 63 lanes idle for hundreds of trips.  f(tile, depth bits, id).
	constexpr int BIN_BIG = 48;
	template <typename F>
	__device__ __forceinline__ void walk_rect_tiles(const ushort4 r, const uint32_t key, const uint32_t gid, int grid_x, int lane, F f)
	{
		const uint32_t w = (uint32_t)(r.z - r.x), cnt = w * (uint32_t)(r.w - r.y);
		const bool big = cnt > (uint32_t)BIN_BIG;
		if (!big)
		{
			uint32_t x = r.x, row = (uint32_t)r.y * (uint32_t)grid_x;
			for (uint32_t k = 0; k < cnt; k++)
			{
				f(row + x, key, gid);
				x++;
				if (x == r.z) { x = r.x; row += (uint32_t)grid_x; }
			}
		}
		unsigned long long todo = __ballot(big);
		while (todo)
		{
			const int j = __ffsll((long long)todo) - 1;
			todo &= todo - 1;
			const uint32_t xy0 = __shfl((int)((uint32_t)r.x | ((uint32_t)r.y << 16)), j);
			const uint32_t wj = __shfl((int)w, j), cj = __shfl((int)cnt, j), kj = __shfl((int)key, j), gj = __shfl((int)gid, j);
			const float inv = __builtin_amdgcn_rcpf((float)wj);
			for (uint32_t s = lane; s < cj; s += WAVE)
			{
				// s / wj by a float reciprocal and one correction step either way (the quotient is < 65536)
				uint32_t q = (uint32_t)((float)s * inv);
				int rem = (int)s - (int)(q * wj);
				if (rem < 0) { q--; rem += (int)wj; }
				else if (rem >= (int)wj) { q++; rem -= (int)wj; }
				f(((xy0 >> 16) + q) * (uint32_t)grid_x + (xy0 & 0xFFFFu) + (uint32_t)rem, kj, gj);
			}
		}
	}

	constexpr int BIN_T = 1024;
	template <bool SCATTER>
	__global__ void __launch_bounds__(BIN_T) tile_bin_lds_kernel(const ushort4* __restrict__ rect, const float* __restrict__ depths, int P,
	                                                             int grid_x, int T, int rounds /* batch = rounds * BIN_T Gaussians per workgroup */,
	                                                             uint32_t* __restrict__ counters, uint2* __restrict__ pairs)
	{
		extern __shared__ uint32_t s_hist[];   // T words
		const int lane = threadIdx.x & 63;
		for (int t = threadIdx.x; t < T; t += BIN_T) s_hist[t] = 0u;
		__syncthreads();
		const int g_first = blockIdx.x * rounds * BIN_T + threadIdx.x;
		for (int c = 0; c < rounds; c++)
		{
			const int g = g_first + c * BIN_T;
			ushort4 r = make_ushort4(0, 0, 0, 0);   // culled Gaussians carry an empty rectangle (preprocess_fwd.hip)
			if (g < P) r = rect[g];
			walk_rect_tiles(r, 0u, 0u, grid_x, lane, [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&s_hist[tile], 1u); });
		}
		__syncthreads();
		// histogram -> global counters, lane = tile: 32 consecutive counters travel as one 128-byte atomic request
		for (int t0 = threadIdx.x; t0 < T; t0 += 4 * BIN_T)
		{
			uint32_t c[4], base[4];
#pragma unroll
			for (int u = 0; u < 4; u++) { const int t = t0 + u * BIN_T; c[u] = t < T ? s_hist[t] : 0u; }
#pragma unroll
			for (int u = 0; u < 4; u++)
			{
				base[u] = 0u;
				if (c[u] != 0u)
				{
					if (!SCATTER) atomicAdd(&counters[t0 + u * BIN_T], c[u]);
					else base[u] = atomicAdd(&counters[t0 + u * BIN_T], c[u]);   // scanned counter: start of this workgroup's piece of the list
				}
			}
			if (SCATTER)
			{
#pragma unroll
				for (int u = 0; u < 4; u++) if (c[u] != 0u) s_hist[t0 + u * BIN_T] = base[u];
			}
		}
		if (!SCATTER) return;
		__syncthreads();
		for (int c = 0; c < rounds; c++)
		{
			const int g = g_first + c * BIN_T;
			ushort4 r = make_ushort4(0, 0, 0, 0);
			uint32_t key = 0u;
			if (g < P) { r = rect[g]; key = __float_as_uint(depths[g]); }
#if defined(FDGS_PROBE_NO_PASS_B)   // timing probes only (tools/probe/bin_probe.hip)
			(void)r; (void)key;
#elif defined(FDGS_PROBE_NO_STORE)
			walk_rect_tiles(r, key, (uint32_t)g, grid_x, lane, [&](uint32_t tile, uint32_t k, uint32_t id) {
				const uint32_t slot = atomicAdd(&s_hist[tile], 1u);
				if (slot == 0xFFFFFFFFu) pairs[slot] = make_uint2(k, id);
			});
#else
			walk_rect_tiles(r, key, (uint32_t)g, grid_x, lane, [&](uint32_t tile, uint32_t k, uint32_t id) {
				const uint32_t slot = atomicAdd(&s_hist[tile], 1u);
				pairs[slot] = make_uint2(k, id);
			});
#endif
		}
	}

	// Images with more tiles than an LDS histogram holds: the same walk with one global atomic per instance.
	template <bool SCATTER>
	__global__ void __launch_bounds__(256) tile_bin_direct_kernel(const ushort4* __restrict__ rect, const float* __restrict__ depths, int P,
	                                                              int grid_x, uint32_t* __restrict__ counters, uint2* __restrict__ pairs)
	{
		const int g = blockIdx.x * blockDim.x + threadIdx.x;
		ushort4 r = make_ushort4(0, 0, 0, 0);
		uint32_t key = 0u;
		if (g < P) { r = rect[g]; if (SCATTER) key = __float_as_uint(depths[g]); }
		walk_rect_tiles(r, key, (uint32_t)g, grid_x, threadIdx.x & 63, [&](uint32_t tile, uint32_t k, uint32_t id) {
			if (!SCATTER) atomicAdd(&counters[tile], 1u);
			else
			{
				const uint32_t slot = atomicAdd(&counters[tile], 1u);
				pairs[slot] = make_uint2(k, id);
			}
		});
	}

	// ------------------------------------------------------------------------------------------------
	// exclusive scan of the tile counters by ONE workgroup; ctl[0] = R, ctl[1] = longest tile list
	// ------------------------------------------------------------------------------------------------
	constexpr int SCAN_T = 1024;
	__global__ void __launch_bounds__(SCAN_T) tile_scan_kernel(uint32_t* __restrict__ counters, int T, int per_thread /* multiple of 4 */,
	                                                           uint32_t* __restrict__ ctl)
	{
		__shared__ uint32_t s_w[SCAN_T / WAVE], s_m[SCAN_T / WAVE];
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		const int first = threadIdx.x * per_thread;
		// the counter array is padded to a multiple of 4 words (zeros), so the uint4 accesses stay inside it
		uint32_t sum = 0, m = 0;
		for (int i = 0; i < per_thread; i += 4)
		{
			if (first + i >= T) break;
			const uint4 v = *reinterpret_cast<const uint4*>(counters + first + i);
			sum += (v.x + v.y) + (v.z + v.w);
			m = max(max(m, max(v.x, v.y)), max(v.z, v.w));
		}
		uint32_t incl = sum;
#pragma unroll
		for (int o = 1; o < WAVE; o <<= 1)
		{
			const uint32_t t = __shfl_up(incl, o);
			if (lane >= o) incl += t;
			m = max(m, (uint32_t)__shfl_xor((int)m, o));
		}
		if (lane == WAVE - 1) { s_w[wave] = incl; s_m[wave] = m; }
		__syncthreads();
		uint32_t base = 0, gmax = 0, gtot = 0;
#pragma unroll
		for (int w2 = 0; w2 < SCAN_T / WAVE; w2++)
		{
			if (w2 < wave) base += s_w[w2];
			gtot += s_w[w2];
			gmax = max(gmax, s_m[w2]);
		}
		uint32_t run = base + incl - sum;
		for (int i = 0; i < per_thread; i += 4)
		{
			if (first + i >= T) break;
			uint4* p = reinterpret_cast<uint4*>(counters + first + i);
			const uint4 v = *p;
			uint4 o;
			o.x = run; run += v.x;
			o.y = run; run += v.y;
			o.z = run; run += v.z;
			o.w = run; run += v.w;
			*p = o;
		}
		if (threadIdx.x == 0) { ctl[0] = gtot; ctl[1] = gmax; }
	}

	// ------------------------------------------------------------------------------------------------
	// per-tile local sort
	// ------------------------------------------------------------------------------------------------
	// A list of n entries is held as 64-bit keys (depth bits << 32 | id): unique, so any correct sort gives the
	// reference's order.  Sample sort, because depth inside a tile is anything but uniform (surfaces: most of a list sits
	// in a sliver of its depth range; a far outlier stretches the range):
	//   1. 64 regularly spaced samples of the list are sorted across the lanes of one wave (bitonic on shuffles);
	//   2. every key finds its bucket among the 65 the splitters define (6-step search), counted with returning LDS
	//      atomics; one wave scans the 65 sizes; keys are written to LDS in bucket order;
	//   3. every key counts the smaller keys of its bucket (n / 65 entries on average, whatever the distribution, ties
	//      in depth included since the id is part of the key) -> final position -> point_list.
	// A bucket that still turns out crowded (> rank_max) sends the tile to a bitonic sort of its keys in LDS.
	// Two instances: one WAVE per tile for lists of up to 1024 entries (no workgroup barriers at all), 256 threads per
	// tile for up to 4096; longer lists are sorted by a bitonic network in global scratch.
	constexpr int TS_NS = 64;                   // splitters
	constexpr int TS_G = 4;                     // keys a thread handles side by side (independent LDS chains in flight)
	constexpr int TS_SMALL_T = 128, TS_SMALL_ITEMS = 8;    // lists of up to 1024 entries: two waves per tile
	constexpr int TS_LARGE_T = 256, TS_LARGE_ITEMS = 16;   // up to 4096: four waves per tile
	constexpr int TS_SMALL = TS_SMALL_T * TS_SMALL_ITEMS;
	constexpr int TS_LARGE = TS_LARGE_T * TS_LARGE_ITEMS;
	constexpr int TS_DIRECT = 96;               // lists this short skip the bucketing
	typedef unsigned long long u64;

	__device__ __forceinline__ int pow2_ceil(int n)
	{
		int p = 1;
		while (p < n) p <<= 1;
		return p;
	}
	__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int m)
	{
		const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
		return ((u64)hi << 32) | lo;
	}

	// In-place bitonic sort of n keys (LDS or global) by the THREADS threads of the workgroup.  The network is the
	// all-ascending form (first step of every merge mirrors the upper half), so the N - n padding keys of the
	// power-of-two network are +infinity that never moves: they are not stored, pairs that reach beyond n are skipped.
	template <int THREADS, typename PTR>
	__device__ __forceinline__ void bitonic_sort(PTR a, int n)
	{
		const int N = pow2_ceil(n);
		for (int k = 2; k <= N; k <<= 1)
			for (int j = k >> 1; j > 0; j >>= 1)
			{
				for (int t = threadIdx.x; t < (N >> 1); t += THREADS)
				{
					const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // element with bit j clear
					const int p = (j == (k >> 1)) ? (i ^ (k - 1)) : (i | j);   // mirror partner in the first step of a merge
					if (p < n)
					{
						const u64 x = a[i], y = a[p];
						if (x > y) { a[i] = y; a[p] = x; }
					}
				}
				__syncthreads();
			}
	}

	constexpr int TS_SUB = 4;                           // linear sub-buckets inside every splitter interval
	constexpr int TS_NBK = (TS_NS + 1) * TS_SUB;        // 260 buckets
	constexpr int TS_PAD = 128;                         // sentinel keys behind the list (>= the largest rank_max)

	template <int THREADS, int ITEMS>
	__global__ void __launch_bounds__(THREADS) tile_sort_kernel(const uint32_t* __restrict__ list_end, const uint2* __restrict__ pairs,
	                                                           uint32_t* __restrict__ point_list, uint2* __restrict__ ranges,
	                                                           u64* __restrict__ big_scratch, int n_lo /* handle lists longer than this */,
	                                                           int lds_cap, int rank_max)
	{
		// LDS: lds_cap + TS_PAD depth keys, then lds_cap ids, in bucket order (8 lds_cap + 4 TS_PAD bytes); the same
		// bytes hold the 64-bit keys of the bitonic fall-back and, at the end, the ids in final order
		extern __shared__ uint32_t s_dyn[];
		__shared__ uint32_t s_split[TS_NS];
		__shared__ uint32_t s_hist[TS_NBK + 4];    // bucket sizes -> starts; [TS_NBK] = n
		__shared__ uint32_t s_flag[2];             // largest bucket, depth ties seen
		constexpr int GROUPS = ITEMS / TS_G;
		const int tid = threadIdx.x, lane = tid & 63;
		// after the scatter pass a tile's counter holds the END of its list = the start of the next tile's
		const uint32_t start = blockIdx.x == 0 ? 0u : list_end[blockIdx.x - 1];
		const uint32_t end = list_end[blockIdx.x];
		const int n = (int)(end - start);
		if (n_lo == 0 && tid == 0) ranges[blockIdx.x] = n > 0 ? make_uint2(start, end) : make_uint2(0u, 0u);   // identifyTileRanges leaves empty tiles at the memset's (0,0)
		if (n <= n_lo) return;
		if (n == 1)
		{
			if (tid == 0) point_list[start] = pairs[start].y;
			return;
		}
		if (n > lds_cap)
		{
			if (big_scratch == nullptr) return;   // left to the next instance
			// a list longer than the LDS takes: bitonic sort in global scratch (slot s of the list = slot start + s)
			volatile u64* S = big_scratch + start;
			for (int i = tid; i < n; i += THREADS)
			{
				const uint2 p = pairs[start + i];
				S[i] = ((u64)p.x << 32) | p.y;
			}
			__syncthreads();
			bitonic_sort<THREADS>(S, n);
			for (int i = tid; i < n; i += THREADS) point_list[start + i] = (uint32_t)S[i];
			return;
		}
		uint32_t* s_key = s_dyn;                          // [lds_cap + TS_PAD]
		uint32_t* s_id = s_dyn + lds_cap + TS_PAD;        // [lds_cap]

		// the list: TS_G (depth bits, id) pairs side by side per thread and group
		const int ngroups = (n + THREADS * TS_G - 1) / (THREADS * TS_G);
		uint32_t key[ITEMS], id[ITEMS];
#pragma unroll
		for (int g = 0; g < GROUPS; g++)
		{
#pragma unroll
			for (int u = 0; u < TS_G; u++) { key[g * TS_G + u] = 0xFFFFFFFFu; id[g * TS_G + u] = 0xFFFFFFFFu; }
			if (g < ngroups)
			{
#pragma unroll
				for (int u = 0; u < TS_G; u++)
				{
					const int idx = (g * TS_G + u) * THREADS + tid;
					if (idx < n)
					{
						const uint2 p = pairs[start + idx];
						key[g * TS_G + u] = p.x;
						id[g * TS_G + u] = p.y;
					}
				}
			}
		}
		for (int i = tid; i < TS_NBK + 4; i += THREADS) s_hist[i] = 0u;
		if (tid < 2) s_flag[tid] = 0u;
		const bool direct = n <= TS_DIRECT;   // one bucket
		if (!direct && tid < WAVE)
		{
			// the depth bits of 64 regularly spaced entries, sorted across the lanes of wave 0: the splitters
			uint32_t v = pairs[start + (uint32_t)(((long long)lane * n) >> 6)].x;
#pragma unroll
			for (int k = 2; k <= WAVE; k <<= 1)
#pragma unroll
				for (int j = k >> 1; j > 0; j >>= 1)
				{
					const uint32_t o = (uint32_t)__shfl_xor((int)v, j);
					const bool keep_min = ((lane & j) == 0) == ((lane & k) == 0);
					v = keep_min ? min(v, o) : max(v, o);
				}
			s_split[lane] = v;
		}
		__syncthreads();

		// bucket = (number of splitters < depth bits) * TS_SUB + linear position inside the splitter interval: monotone in
		// the depth bits, so equal depths share a bucket; br = bucket << 16 | arrival index inside the bucket
		uint32_t br[ITEMS];
#pragma unroll
		for (int g = 0; g < GROUPS; g++)
		{
#pragma unroll
			for (int u = 0; u < TS_G; u++) br[g * TS_G + u] = 0u;
			if (g < ngroups)
			{
				int b[TS_G];
#pragma unroll
				for (int u = 0; u < TS_G; u++) b[u] = 0;
				if (!direct)
				{
					const uint32_t last = s_split[TS_NS - 1];
#pragma unroll
					for (int step = TS_NS / 2; step > 0; step >>= 1)
#pragma unroll
						for (int u = 0; u < TS_G; u++)
							if (s_split[b[u] + step - 1] < key[g * TS_G + u]) b[u] += step;
#pragma unroll
					for (int u = 0; u < TS_G; u++)
					{
						const uint32_t k = key[g * TS_G + u];
						if (last < k) b[u] = TS_NS;
						uint32_t sub = 0;
						if (b[u] > 0 && b[u] < TS_NS)
						{
							const uint32_t lo = s_split[b[u] - 1], hi = s_split[b[u]];   // lo < k <= hi
							sub = min((uint32_t)TS_SUB - 1u, (uint32_t)((float)(k - lo - 1u) * ((float)TS_SUB * __builtin_amdgcn_rcpf((float)(hi - lo)))));
						}
						b[u] = b[u] * TS_SUB + (int)sub;
					}
				}
#pragma unroll
				for (int u = 0; u < TS_G; u++)
					if ((g * TS_G + u) * THREADS + tid < n) br[g * TS_G + u] = ((uint32_t)b[u] << 16) | atomicAdd(&s_hist[b[u]], 1u);
			}
		}
		__syncthreads();
		if (tid < WAVE)
		{
			// exclusive scan of the bucket sizes by wave 0 (lane = splitter interval, TS_SUB sizes each); the largest bucket
			uint32_t c[TS_SUB + 1], sum = 0, mb = 0;
#pragma unroll
			for (int q = 0; q < TS_SUB; q++) { c[q] = s_hist[lane * TS_SUB + q]; sum += c[q]; mb = max(mb, c[q]); }
			uint32_t tail[TS_SUB];
#pragma unroll
			for (int q = 0; q < TS_SUB; q++) { tail[q] = s_hist[TS_NS * TS_SUB + q]; mb = max(mb, tail[q]); }
			uint32_t incl = sum;
#pragma unroll
			for (int o = 1; o < WAVE; o <<= 1)
			{
				const uint32_t t = __shfl_up(incl, o);
				if (lane >= o) incl += t;
				mb = max(mb, (uint32_t)__shfl_xor((int)mb, o));
			}
			uint32_t run = incl - sum;
#pragma unroll
			for (int q = 0; q < TS_SUB; q++) { s_hist[lane * TS_SUB + q] = run; run += c[q]; }
			if (lane == WAVE - 1)
			{
#pragma unroll
				for (int q = 0; q < TS_SUB; q++) { s_hist[TS_NS * TS_SUB + q] = run; run += tail[q]; }
				s_hist[TS_NBK] = (uint32_t)n;
				s_flag[0] = mb;
			}
		}
		__syncthreads();

		if ((int)s_flag[0] > rank_max && !direct)
		{
			// a crowded bucket (a pile of equal / nearly equal depths): bitonic sort of the 64-bit keys in LDS
			u64* s_a = reinterpret_cast<u64*>(s_dyn);
#pragma unroll
			for (int g = 0; g < GROUPS; g++)
			{
				if (g < ngroups)
				{
#pragma unroll
					for (int u = 0; u < TS_G; u++)
					{
						const int idx = (g * TS_G + u) * THREADS + tid;
						if (idx < n) s_a[idx] = ((u64)key[g * TS_G + u] << 32) | id[g * TS_G + u];
					}
				}
			}
			__syncthreads();
			bitonic_sort<THREADS>(s_a, n);
			for (int i = tid; i < n; i += THREADS) point_list[start + i] = (uint32_t)s_a[i];
			return;
		}

		// keys / ids into LDS in bucket order; TS_PAD sentinels behind the list
#pragma unroll
		for (int g = 0; g < GROUPS; g++)
		{
			if (g < ngroups)
			{
#pragma unroll
				for (int u = 0; u < TS_G; u++)
					if ((g * TS_G + u) * THREADS + tid < n)
					{
						const uint32_t pos = s_hist[br[g * TS_G + u] >> 16] + (br[g * TS_G + u] & 0xFFFFu);
						s_key[pos] = key[g * TS_G + u];
						s_id[pos] = id[g * TS_G + u];
					}
			}
		}
		for (int i = tid; i < TS_PAD; i += THREADS) s_key[n + i] = 0xFFFFFFFFu;
		__syncthreads();
		// Final position = start of the bucket + number of smaller keys in it.  TS_G keys walk their buckets side by side;
		// a key whose bucket is shorter than its neighbours' keeps reading: whatever follows its bucket has larger depth
		// bits (later bucket) or is a sentinel, and counts neither as smaller nor as equal.
		uint32_t fin[ITEMS];
#pragma unroll
		for (int g = 0; g < GROUPS; g++)
		{
#pragma unroll
			for (int u = 0; u < TS_G; u++) fin[g * TS_G + u] = 0xFFFFFFFFu;
			if (g < ngroups)
			{
				uint32_t bs[TS_G], lt[TS_G], eq[TS_G], maxlen = 0;
#pragma unroll
				for (int u = 0; u < TS_G; u++)
				{
					const uint32_t b = br[g * TS_G + u] >> 16;
					const bool valid = (g * TS_G + u) * THREADS + tid < n;
					bs[u] = valid ? s_hist[b] : (uint32_t)n;      // invalid slots read sentinels only
					const uint32_t len = valid ? s_hist[b + 1] - bs[u] : 0u;
					lt[u] = 0; eq[u] = 0;
					maxlen = max(maxlen, len);
				}
				for (uint32_t k = 0; k < maxlen; k++)
				{
#pragma unroll
					for (int u = 0; u < TS_G; u++)
					{
						const uint32_t kj = s_key[bs[u] + k];
						lt[u] += (kj < key[g * TS_G + u]) ? 1u : 0u;
						eq[u] += (kj == key[g * TS_G + u]) ? 1u : 0u;
					}
				}
#pragma unroll
				for (int u = 0; u < TS_G; u++)
				{
					const bool valid = (g * TS_G + u) * THREADS + tid < n;
					if (valid && eq[u] > 1u)
					{
						// equal depth bits: the Gaussian id decides (the reference's stable sort over id-ordered input)
						const uint32_t b = br[g * TS_G + u] >> 16;
						const uint32_t be = s_hist[b + 1];
						for (uint32_t j = bs[u]; j < be; j++)
							lt[u] += (s_key[j] == key[g * TS_G + u] && s_id[j] < id[g * TS_G + u]) ? 1u : 0u;
					}
					if (valid) fin[g * TS_G + u] = bs[u] + lt[u];
				}
			}
		}
		// ids through LDS in final order, so that point_list is written with contiguous stores
		__syncthreads();
		uint32_t* s_out = s_dyn;
#pragma unroll
		for (int g = 0; g < GROUPS; g++)
		{
			if (g < ngroups)
			{
#pragma unroll
				for (int u = 0; u < TS_G; u++)
					if (fin[g * TS_G + u] != 0xFFFFFFFFu) s_out[fin[g * TS_G + u]] = id[g * TS_G + u];
			}
		}
		__syncthreads();
		for (int i = tid; i < n; i += THREADS) point_list[start + i] = s_out[i];
	}

	// ------------------------------------------------------------------------------------------------
	// host side
	// ------------------------------------------------------------------------------------------------
	constexpr int BIN_LDS_MAX_TILES = 36 * 1024;   // a 144 KiB histogram stays inside the 160 KiB of a CU
	static inline int bin_rounds(int T) { return T <= 8192 ? 1 : 4; }   // bigger histograms: fewer, longer workgroups

	template <bool SCATTER>
	static hipError_t launch_tile_bin(const uint16_t* rect, const float* depths, int P, int grid_x, int T, uint32_t* counters, uint32_t* pairs,
	                                  hipStream_t stream)
	{
		if (P <= 0) return hipSuccess;
		const ushort4* r4 = reinterpret_cast<const ushort4*>(rect);
		uint2* p2 = reinterpret_cast<uint2*>(pairs);
		if (T <= BIN_LDS_MAX_TILES)
		{
			static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_bin_lds_kernel<SCATTER>),
			                                                   hipFuncAttributeMaxDynamicSharedMemorySize, BIN_LDS_MAX_TILES * 4);
			if (attr != hipSuccess) return attr;
			const int rounds = bin_rounds(T);
			hipLaunchKernelGGL(tile_bin_lds_kernel<SCATTER>, dim3(div_up(P, rounds * BIN_T)), dim3(BIN_T), (size_t)T * 4, stream, r4, depths, P,
			                   grid_x, T, rounds, counters, p2);
		}
		else
			hipLaunchKernelGGL(tile_bin_direct_kernel<SCATTER>, dim3(div_up(P, 256)), dim3(256), 0, stream, r4, depths, P, grid_x, counters, p2);
		return hipGetLastError();
	}

	hipError_t launch_tile_count(const uint16_t* rect, int P, int grid_x, int T, uint32_t* counters, hipStream_t stream)
	{
		return launch_tile_bin<false>(rect, nullptr, P, grid_x, T, counters, nullptr, stream);
	}

	hipError_t launch_tile_scan(uint32_t* counters, int T, uint32_t* ctl, hipStream_t stream)
	{
		const int per_thread = div_up(div_up(T, SCAN_T), 4) * 4;
		hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(SCAN_T), 0, stream, counters, T, per_thread, ctl);
		return hipGetLastError();
	}

	hipError_t launch_tile_scatter(const uint16_t* rect, const float* depths, int P, int grid_x, int T, uint32_t* counters, uint32_t* pairs,
	                               hipStream_t stream)
	{
		return launch_tile_bin<true>(rect, depths, P, grid_x, T, counters, pairs, stream);
	}

	// test hook: lower the list lengths at which the instances hand over, and the crowded-bucket threshold
	static std::atomic<int> g_small_cap{TS_SMALL}, g_large_cap{TS_LARGE}, g_rank_max{96};
	void tile_sort_debug_limits(int lds_cap, int rank_max)
	{
		g_large_cap.store(lds_cap > 0 && lds_cap < TS_LARGE ? lds_cap : TS_LARGE);
		g_small_cap.store(lds_cap > 0 && lds_cap < TS_SMALL ? lds_cap : TS_SMALL);
		g_rank_max.store(rank_max > 0 ? min(rank_max, TS_PAD) : 96);
	}
	int tile_sort_lds_cap() { return g_large_cap.load(); }

	hipError_t launch_tile_sort(const uint32_t* counters, int T, int max_count, const uint32_t* pairs, uint32_t* point_list, uint32_t* ranges,
	                            void* big_scratch, hipStream_t stream)
	{
		const int small_cap = g_small_cap.load(), large_cap = g_large_cap.load(), rank_max = g_rank_max.load();
		const uint2* p2 = reinterpret_cast<const uint2*>(pairs);
		uint2* r2 = reinterpret_cast<uint2*>(ranges);
		// LDS: the longest list this instance takes, in 64-key steps (a short longest list = more tiles in flight per CU)
		const int cap = min(small_cap, max(div_up(max_count, 64) * 64, 64));
		u64* big = reinterpret_cast<u64*>(big_scratch);
		const bool second = max_count > small_cap;
		hipLaunchKernelGGL((tile_sort_kernel<TS_SMALL_T, TS_SMALL_ITEMS>), dim3(T), dim3(TS_SMALL_T), (size_t)cap * 8 + TS_PAD * 4, stream, counters, p2,
		                   point_list, r2, (second && small_cap < large_cap) ? (u64*)nullptr : big, 0, cap, rank_max);
		if (second && small_cap < large_cap)
		{
			const int cap2 = min(large_cap, div_up(max_count, 64) * 64);
			hipLaunchKernelGGL((tile_sort_kernel<TS_LARGE_T, TS_LARGE_ITEMS>), dim3(T), dim3(TS_LARGE_T), (size_t)cap2 * 8 + TS_PAD * 4, stream, counters, p2,
			                   point_list, r2, big, cap, cap2, rank_max);
		}
		return hipGetLastError();
	}
}

// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs (gfx950).
//
// Used by knn.hip (Morton order of the points for distCUDA2).  The rasterizer's forward no longer sorts globally:
// its tile lists are built by counting / scattering / per-tile local sorts (tilebin.hip).
//
// Radix pass = 3 launches: per-workgroup digit histogram -> per-digit scan over
// workgroups -> stable scatter (wave64 match-any ranking via 8 ballots, per-wave
// digit counters in LDS, LDS reorder so the global writes are contiguous per digit run).
// Small inputs use 1024-key chunks so they still fill the chip.  All integer work, HBM-bound; no MFMA.
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
}

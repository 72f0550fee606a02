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
//   n <= 4096  sample sort in LDS (see above tile_sort_kernel): 64 splitters from the list itself, 8 linear sub-buckets
//              inside every splitter interval, then every entry counts the smaller keys of its bucket; 128 / 256 / 512
//              threads per tile for lists of up to 1024 / 2048 / 4096 entries.  When a bucket is still crowded (a pile
//              of equal depths) the tile falls back to a bitonic sort of the 64-bit keys in LDS.
//   n <= 16384 the same with 1024 threads per tile (8 or 16 keys per thread) for the few long lists of a clustered scene.
//   n  > 16384 bitonic sort in global scratch (R keys, requested from the allocator only when such a list exists).
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

	// Order of the tiles for the blend kernels and the per-tile sort (blend_common.h, block_of): ONE order over all tiles, longest lists
	// first -- position p of it goes to XCD p % 8 (workgroups are dealt round-robin to the XCDs), so that every XCD gets the same
	// share of the long and of the short lists and inside an XCD the launch ends on short tiles.  A counting sort over ORDER_BUCKETS
	// length classes of the longest list (class 0 = the longest); the rank of a tile inside its class is whatever the LDS atomic
	// hands out (the order only schedules work, no result depends on it).  Rounds 2-4 gave every XCD a contiguous eighth of the tiles
	// in row-major order (a band of tile rows, for its L2) with the longest-first order inside the band: the XCDs' shares of the WORK
	// then are the bands' shares of the scene -- on the bench's cameras the blend kernels ran 16 % longer than with the work dealt out
	// evenly (round 5, profiles/HISTORY.md; runs of 1 ... 85 tiles dealt round-robin and this global order are within 1 % of each other:
	// which XCD's L2 a tile's records pass through does not matter).
	// Run by ONE extra workgroup of the scatter launch (off the forward's critical path: the scan only leaves a copy of the tile counts)
	// or, with sparse lists, of the sort launch.
	// counts: [T] list lengths (left alone: a scatter pass that is launched a second time orders again); tmp: [T] scratch; order: [T];
	// s_cls: ORDER_BUCKETS words of LDS.  T < 2^24 (the launchers check).
	[[maybe_unused]] constexpr int NUM_XCDS_BIN = 8;     // blend_common.h NUM_XCDS
	constexpr int ORDER_BUCKETS = 64;   // = WAVE: one wave scans the classes
	__device__ __forceinline__ void tile_order_block(const uint32_t* __restrict__ counts, uint32_t* __restrict__ tmp, int T, uint32_t gmax,
	                                                 uint32_t* __restrict__ order, uint32_t* s_cls)
	{
		const int nthreads = (int)blockDim.x, lane = threadIdx.x & 63;
		for (int k = threadIdx.x; k < ORDER_BUCKETS; k += nthreads) s_cls[k] = 0u;
		__syncthreads();
		const float cls_scale = (float)ORDER_BUCKETS / ((float)gmax + 1.0f);
		for (int t = threadIdx.x; t < T; t += nthreads)
		{
			const uint32_t cls = (uint32_t)(ORDER_BUCKETS - 1) - min((uint32_t)(ORDER_BUCKETS - 1), (uint32_t)((float)counts[t] * cls_scale));
			const uint32_t rank = atomicAdd(&s_cls[cls], 1u);
			tmp[t] = (cls << 24) | rank;   // read back below by this same thread
		}
		__syncthreads();
		if (threadIdx.x < ORDER_BUCKETS)   // the first wave: exclusive scan of the class sizes (lane = class)
		{
			const uint32_t n = s_cls[threadIdx.x];
			uint32_t inc = n;
#pragma unroll
			for (int o = 1; o < WAVE; o <<= 1)
			{
				const uint32_t u = __shfl_up(inc, o);
				if (lane >= o) inc += u;
			}
			s_cls[threadIdx.x] = inc - n;
		}
		__syncthreads();
		for (int t = threadIdx.x; t < T; t += nthreads)
		{
			const uint32_t cr = tmp[t];
			order[s_cls[cr >> 24] + (cr & 0xFFFFFFu)] = (uint32_t)t;
		}
	}

	constexpr int BIN_T = 1024;
	template <bool SCATTER>
	__global__ void __launch_bounds__(BIN_T) tile_bin_lds_kernel(const ushort4* __restrict__ rect, const float* __restrict__ depths, int P,
	                                                             int grid_x, int T, int rounds /* batch = rounds * BIN_T Gaussians per workgroup */,
	                                                             uint32_t* __restrict__ counters, uint2* __restrict__ pairs,
	                                                             const uint32_t* __restrict__ ctl, uint32_t capacity,
	                                                             uint32_t* __restrict__ order /* [3 T + 16] or NULL */,
	                                                             uint32_t sparse_cap /* 0, or SPARSE lists: tile t's list lives at [t * sparse_cap, (t + 1) * sparse_cap) */)
	{
		extern __shared__ uint32_t s_hist[];   // max(T, ORDER_BUCKETS) words
		if (SCATTER && order != nullptr && blockIdx.x == gridDim.x - 1)
		{
			// the extra workgroup of the scatter launch: the blend kernels' tile order from the scan's copy of the counts
			tile_order_block(order + tile_order_counts_off(T), order + tile_order_tmp_off(T), T, ctl[1], order, s_hist);
			return;
		}
		// launched before the host knew num_rendered (capi.hip): `pairs` holds `capacity` instances -- more than that: leave everything alone
		// (sparse lists: nobody has counted yet -- the counters start from zero and a tile that outgrows its sparse_cap slots drops
		// what does not fit; its count still says so, and the host renders the view again)
		if (SCATTER && sparse_cap == 0u && ctl[0] > capacity) return;
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
				for (int u = 0; u < 4; u++) if (c[u] != 0u) s_hist[t0 + u * BIN_T] = base[u] + (uint32_t)(t0 + u * BIN_T) * sparse_cap;   // (compact lists: + 0)
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
				if (sparse_cap == 0u || slot < (tile + 1u) * sparse_cap) pairs[slot] = make_uint2(k, id);
			});
#endif
		}
	}

	// Images with more tiles than an LDS histogram holds: the same walk with one global atomic per instance.
	template <bool SCATTER>
	__global__ void __launch_bounds__(256) tile_bin_direct_kernel(const ushort4* __restrict__ rect, const float* __restrict__ depths, int P,
	                                                              int grid_x, uint32_t* __restrict__ counters, uint2* __restrict__ pairs,
	                                                              const uint32_t* __restrict__ ctl, uint32_t capacity,
	                                                              uint32_t* __restrict__ order, int T, uint32_t sparse_cap)
	{
		__shared__ uint32_t s_cls[ORDER_BUCKETS];
		if (SCATTER && order != nullptr && blockIdx.x == gridDim.x - 1)
		{
			tile_order_block(order + tile_order_counts_off(T), order + tile_order_tmp_off(T), T, ctl[1], order, s_cls);
			return;
		}
		if (SCATTER && sparse_cap == 0u && ctl[0] > capacity) return;
		const int g = blockIdx.x * blockDim.x + threadIdx.x;
		ushort4 r = make_ushort4(0, 0, 0, 0);
		uint32_t key = 0u;
		if (g < P) { r = rect[g]; if (SCATTER) key = __float_as_uint(depths[g]); }
		walk_rect_tiles(r, key, (uint32_t)g, grid_x, threadIdx.x & 63, [&](uint32_t tile, uint32_t k, uint32_t id) {
			if (!SCATTER) atomicAdd(&counters[tile], 1u);
			else
			{
				const uint32_t slot = atomicAdd(&counters[tile], 1u);
				if (sparse_cap == 0u) pairs[slot] = make_uint2(k, id);
				else if (slot < sparse_cap) pairs[tile * sparse_cap + slot] = make_uint2(k, id);
			}
		});
	}

	// ------------------------------------------------------------------------------------------------
	// exclusive scan of the tile counters by ONE workgroup; ctl[0] = R, ctl[1] = longest tile list
	// ------------------------------------------------------------------------------------------------
	constexpr int SCAN_T = 1024;
	__global__ void __launch_bounds__(SCAN_T) tile_scan_kernel(uint32_t* __restrict__ counters, int T, int per_thread /* multiple of 4 */,
	                                                           uint32_t* __restrict__ ctl, uint32_t* __restrict__ host_box, uint32_t ticket,
	                                                           uint32_t* __restrict__ counts_copy /* [T + 3] or NULL: the counts, for tile_order_block */)
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
			if (counts_copy) *reinterpret_cast<uint4*>(counts_copy + first + i) = v;   // 16-byte aligned (tile_order_counts_off), 3 words of slack behind T
		}
		if (threadIdx.x == 0)
		{
			ctl[0] = gtot; ctl[1] = gmax;
			if (host_box)
			{
				// straight into the caller's pinned, device-mapped mailbox: {R, longest list}, then the call's ticket with
				// system-scope release -- the host spins on the ticket instead of paying a copy kernel and a stream sync
				host_box[0] = gtot; host_box[1] = gmax;
				__hip_atomic_store(&host_box[2], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
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
	constexpr int TS_G = 2;                     // keys a thread handles side by side (independent LDS chains in flight); 4 wastes half of
	                                            // the lanes on the typical 470-entry list of a 256-thread instance, 1 serialises the chains
	constexpr int TS_ITEMS = 8;                 // keys per thread (four groups of TS_G) of the 128 / 256 / 512-thread instances
	constexpr int TS_ITEMS_LONG = 16;           // ... of the 1024-thread instance for the longest lists
	constexpr int TS_LARGE = 1024 * TS_ITEMS_LONG;   // 16384: the longest list sorted in LDS (128 KiB of keys + ids: one tile per CU)
	constexpr int TS_DIRECT = 96;               // lists this short skip the bucketing
	constexpr int TS_ENDS = 1024;               // lists longer than this subdivide the two end intervals too (tile_sort_one)
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

#ifdef FDGS_TS_TIMELINE   // probe only: cycles between the phases of tile_sort_kernel, summed over the tiles (wave 0, lane 0)
	__device__ unsigned int* g_tl;   // [T][16] cycles, written without atomics
#define TL_MARK(k) do { if (tid == 0) { const unsigned long long now__ = __builtin_readcyclecounter(); g_tl[blockIdx.x * 16 + (k)] = (unsigned int)(now__ - tl_prev); tl_prev = now__; } } while (0)
#else
#define TL_MARK(k) do { } while (0)
#endif
	constexpr int TS_PAD = 128;                         // sentinel keys behind the list (>= the largest rank_max)

	// One tile's list.  Returns true when the list is no longer than n_lo (not this instance's).
	template <int THREADS, int ITEMS>
	__device__ __forceinline__ bool tile_sort_one(const int tile, const uint32_t* __restrict__ list_end, const uint2* __restrict__ pairs,
	                                              uint32_t* __restrict__ point_list, uint2* __restrict__ ranges, u64* __restrict__ big_scratch,
	                                              const int n_lo, const int lds_cap, const int rank_max, const int last, const uint32_t sparse_cap)
	{
		// LDS: lds_cap + TS_PAD depth keys, then lds_cap ids, in bucket order (8 lds_cap + 4 TS_PAD bytes); the same
		// bytes hold the 64-bit keys of the bitonic fall-back and, at the end, the ids in final order
		extern __shared__ uint32_t s_dyn[];
		__shared__ uint32_t s_split[TS_NS];
		constexpr int GROUPS = ITEMS / TS_G;
		// linear sub-buckets inside every splitter interval: 8 for the 128- / 256-thread instances (520 buckets for up to 2048 keys), 32
		// for the long-list instances (measured on tiles of 3000 / 6000 / 12 000 keys with 8 | 16 | 32: 58 | 45 | 43, 176 | 126 | 100,
		// 648 | 479 | 291 us per 1024 tiles; the short lists of C3 lose 1 us with 16)
		constexpr int TS_SUB = THREADS >= 512 ? 32 : 8;
		constexpr int TS_NBK = (TS_NS + 1) * TS_SUB;
		__shared__ uint32_t s_hist[TS_NBK + 4];    // bucket sizes -> starts; [TS_NBK] = n
		__shared__ uint32_t s_flag[2];             // largest bucket, depth ties seen
		__shared__ uint32_t s_wmin[THREADS / WAVE], s_wmax[THREADS / WAVE];   // smallest / largest depth bits per wave (long lists)
		const int tid = threadIdx.x, lane = tid & 63;
		// after the scatter pass a tile's counter holds the END of its list = the start of the next tile's
#ifdef FDGS_TS_TIMELINE
		unsigned long long tl_prev = __builtin_readcyclecounter();
#endif
		// (sparse lists: the counter holds the tile's COUNT, its list starts at tile * sparse_cap; a count beyond sparse_cap = overflow,
		// reported through the longest list: what fitted is sorted so that everything queued behind reads valid ids)
		const uint32_t start = sparse_cap ? (uint32_t)tile * sparse_cap : (tile == 0 ? 0u : list_end[tile - 1]);
		const uint32_t end = sparse_cap ? start + min(list_end[tile], sparse_cap) : list_end[tile];
		const int n = (int)(end - start);
		if (n_lo == 0 && tid == 0) ranges[tile] = n > 0 ? make_uint2(start, end) : make_uint2(0u, 0u);   // identifyTileRanges leaves empty tiles at the memset's (0,0)
		if (n <= n_lo) return true;
		if (n == 1)
		{
			if (tid == 0) point_list[start] = pairs[start].y;
			return false;
		}
		if (n > lds_cap)
		{
			if (big_scratch == nullptr)
			{
				// left to the next instance.  If there is none -- a run-ahead launch sized by the previous view (capi.hip), which the
				// host repeats with the right instances -- the ids go out unsorted, so that the blend queued behind reads valid ones
				if (last)
					for (int i = tid; i < n; i += THREADS) point_list[start + i] = pairs[start + i].y;
				return false;
			}
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
			return false;
		}
		uint32_t* s_key = s_dyn;                          // [lds_cap + TS_PAD]
		uint32_t* s_id = s_dyn + lds_cap + TS_PAD;        // [lds_cap]

		// Per thread: ITEMS (depth bits, id) pairs in registers, in groups of TS_G; a list of n entries uses the first
		// ngroups groups.  Every phase below walks ALL of a thread's live keys side by side (independent LDS chains in flight).
#define FOR_ITEMS(...)                                                      \
		_Pragma("unroll") for (int g_ = 0; g_ < GROUPS; g_++)               \
			if (g_ < ngroups)                                               \
			{                                                               \
				_Pragma("unroll") for (int u_ = 0; u_ < TS_G; u_++)         \
				{                                                           \
					const int it = g_ * TS_G + u_;                          \
					const bool valid = it * THREADS + tid < n;              \
					(void)valid;                                            \
					__VA_ARGS__                                             \
				}                                                           \
			}
		const int ngroups = (n + THREADS * TS_G - 1) / (THREADS * TS_G);
		const bool direct = n <= TS_DIRECT;   // one bucket
		uint32_t key[ITEMS], id[ITEMS];
		uint32_t sample = 0;
		if (!direct && tid < WAVE) sample = pairs[start + (uint32_t)(((long long)lane * n) >> 6)].x;   // in flight together with the list
#pragma unroll
		for (int i = 0; i < ITEMS; i++) { key[i] = 0xFFFFFFFFu; id[i] = 0xFFFFFFFFu; }
		FOR_ITEMS(if (valid) { const uint2 p = pairs[start + it * THREADS + tid]; key[it] = p.x; id[it] = p.y; })
#ifdef FDGS_TS_TIMELINE
		if (key[0] == 0x12345678u) s_flag[0] = 1u;   // forces the wait for the loads before the mark
#endif
		TL_MARK(0);
		for (int i = tid; i < TS_NBK + 4; i += THREADS) s_hist[i] = 0u;
		if (tid < 2) s_flag[tid] = 0u;
		// The intervals below the first and above the last splitter have no second bound among the samples; each holds n / 65
		// keys on average -- 92 at n = 6000, where the bucket they used to be went over rank_max and sent every such tile to the
		// bitonic fall-back.  Lists beyond TS_ENDS keys take the smallest / largest key of the list as the missing bounds and
		// subdivide these two intervals like the others.
		const bool ends = n > TS_ENDS;
		if (ends)
		{
			uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
			for (int i = 0; i < ITEMS; i++) { mn = min(mn, key[i]); mx = max(mx, key[i] == 0xFFFFFFFFu ? 0u : key[i]); }   // (unused slots hold 0xFFFFFFFF)
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, o)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
			if (lane == 0) { s_wmin[tid >> 6] = mn; s_wmax[tid >> 6] = mx; }
		}
		if (!direct && tid < WAVE)
		{
			// the depth bits of 64 regularly spaced entries, sorted across the lanes of wave 0: the splitters
			uint32_t v = sample;
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
		TL_MARK(1);

		// bucket = (number of splitters < depth bits) * TS_SUB + linear position inside the splitter interval: monotone in
		// the depth bits, so equal depths share a bucket; br = bucket << 16 | arrival index inside the bucket
		uint32_t br[ITEMS];
#pragma unroll
		for (int i = 0; i < ITEMS; i++) br[i] = 0u;
		if (!direct)
		{
			const uint32_t last = s_split[TS_NS - 1];
			uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
			if (ends)
			{
#pragma unroll
				for (int w = 0; w < THREADS / WAVE; w++) { kmin = min(kmin, s_wmin[w]); kmax = max(kmax, s_wmax[w]); }
			}
			const uint32_t first = s_split[0];
#pragma unroll
			for (int step = TS_NS / 2; step > 0; step >>= 1)
			{
				FOR_ITEMS(if (s_split[br[it] + step - 1] < key[it]) br[it] += step;)
			}
			FOR_ITEMS(
				const uint32_t k = key[it];
				uint32_t b = br[it];
				if (last < k) b = TS_NS;
				uint32_t sub = 0;
				if ((b > 0 && b < TS_NS) || ends)
				{
					// lo < k <= hi (below the first splitter: kmin - 1 < k <= first; above the last: last < k <= kmax)
					const uint32_t lo = b == 0 ? kmin - 1u : s_split[b - 1];
					const uint32_t hi = b == 0 ? first : (b == TS_NS ? kmax : s_split[min(b, (uint32_t)TS_NS - 1u)]);
					sub = min((uint32_t)TS_SUB - 1u, (uint32_t)((float)(k - lo - 1u) * ((float)TS_SUB * __builtin_amdgcn_rcpf((float)(hi - lo)))));
				}
				br[it] = b * TS_SUB + sub;)
		}
		FOR_ITEMS(if (valid) br[it] = (br[it] << 16) | atomicAdd(&s_hist[br[it]], 1u);)
		__syncthreads();
		TL_MARK(2);
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
		TL_MARK(3);

		if ((int)s_flag[0] > rank_max && !direct)
		{
			// a crowded bucket (a pile of equal / nearly equal depths): bitonic sort of the 64-bit keys in LDS
			u64* s_a = reinterpret_cast<u64*>(s_dyn);
			FOR_ITEMS(if (valid) s_a[it * THREADS + tid] = ((u64)key[it] << 32) | id[it];)
			__syncthreads();
			bitonic_sort<THREADS>(s_a, n);
			for (int i = tid; i < n; i += THREADS) point_list[start + i] = (uint32_t)s_a[i];
			return false;
		}

		// keys / ids into LDS in bucket order; TS_PAD sentinels behind the list
		FOR_ITEMS(if (valid) { const uint32_t pos = s_hist[br[it] >> 16] + (br[it] & 0xFFFFu); s_key[pos] = key[it]; s_id[pos] = id[it]; })
		for (int i = tid; i < TS_PAD; i += THREADS) s_key[n + i] = 0xFFFFFFFFu;
		__syncthreads();
		TL_MARK(4);
		// Final position = start of the bucket + number of smaller keys in it.  All of a thread's keys walk their buckets
		// side by side; a key whose bucket is shorter than its neighbours' keeps reading: whatever follows its bucket has
		// larger depth bits (later bucket) or is a sentinel, and counts neither as smaller nor as equal.  Two instructions per
		// counter and comparison (compare into VCC, add with carry).
		uint32_t bs[ITEMS], lt[ITEMS], eq[ITEMS], maxlen = 0;
#pragma unroll
		for (int i = 0; i < ITEMS; i++) { bs[i] = (uint32_t)n; lt[i] = 0; eq[i] = 0; }
		FOR_ITEMS(if (valid) { const uint32_t b = br[it] >> 16; bs[it] = s_hist[b]; maxlen = max(maxlen, s_hist[b + 1] - bs[it]); })
		// Four positions of every bucket per trip: the trip is one LDS round trip (all of a thread's reads in flight together) plus
		// the counting, and the round trip was most of it (the loop was 30 k of the 54 k cycles a 2048-key tile takes, whatever the
		// bucket sizes).  Up to three positions beyond maxlen are read: later buckets or sentinels (rank_max + 3 <= TS_PAD).
		for (uint32_t k = 0; k < maxlen; k += 4)
		{
			FOR_ITEMS(
				const uint32_t* q = s_key + bs[it] + k;
				const uint32_t k0 = q[0]; const uint32_t k1 = q[1]; const uint32_t k2 = q[2]; const uint32_t k3 = q[3];
				asm("v_cmp_lt_u32 vcc, %2, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
				    "v_cmp_eq_u32 vcc, %2, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
				    "v_cmp_lt_u32 vcc, %3, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
				    "v_cmp_eq_u32 vcc, %3, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
				    "v_cmp_lt_u32 vcc, %4, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
				    "v_cmp_eq_u32 vcc, %4, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
				    "v_cmp_lt_u32 vcc, %5, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
				    "v_cmp_eq_u32 vcc, %5, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
				    : "+v"(lt[it]), "+v"(eq[it]) : "v"(k0), "v"(k1), "v"(k2), "v"(k3), "v"(key[it]) : "vcc");)
		}
		uint32_t fin[ITEMS];
#pragma unroll
		for (int i = 0; i < ITEMS; i++) fin[i] = 0xFFFFFFFFu;
		FOR_ITEMS(
			if (valid && eq[it] > 1u)
			{
				// equal depth bits: the Gaussian id decides (the reference's stable sort over id-ordered input)
				const uint32_t be = s_hist[(br[it] >> 16) + 1];
				for (uint32_t j = bs[it]; j < be; j++) lt[it] += (s_key[j] == key[it] && s_id[j] < id[it]) ? 1u : 0u;
			}
			if (valid) fin[it] = bs[it] + lt[it];)
		// ids through LDS in final order, so that point_list is written with contiguous stores
		TL_MARK(5);
		__syncthreads();
		TL_MARK(6);
		uint32_t* s_out = s_dyn;
		FOR_ITEMS(if (fin[it] != 0xFFFFFFFFu) s_out[fin[it]] = id[it];)
		__syncthreads();
		TL_MARK(7);
		for (int i = tid; i < n; i += THREADS) point_list[start + i] = s_out[i];
		TL_MARK(8);
#undef FOR_ITEMS
		return false;
	}

	// The main instance (n_lo == 0: every tile, also writes `ranges`): workgroup b takes ONE tile -- with `order` (the blend kernels'
	// tile order, written by the scatter launch: all tiles, the longest lists first) entry b of it, so that the long lists start first
	// -- dealt round-robin to the XCDs, as the workgroups are -- and the launch ends on short ones; without: tile b.
	template <int THREADS, int ITEMS>
	__global__ void __launch_bounds__(THREADS) tile_sort_kernel(const uint32_t* __restrict__ list_end, const uint2* __restrict__ pairs,
	                                                           uint32_t* __restrict__ point_list, uint2* __restrict__ ranges,
	                                                           u64* __restrict__ big_scratch, int n_lo /* handle lists longer than this */,
	                                                           int lds_cap, int rank_max, const uint32_t* __restrict__ ctl, uint32_t capacity,
	                                                           int last /* no further instance takes what this one leaves */,
	                                                           const uint32_t* __restrict__ order /* [T] or NULL */, int T,
	                                                           uint32_t sparse_cap, uint32_t* __restrict__ report_ctl /* sparse main instance: ctl (written) */,
	                                                           uint32_t* __restrict__ report_box, uint32_t ticket, uint32_t* __restrict__ order_out)
	{
		if (report_ctl != nullptr && blockIdx.x == 0)
		{
			// (workgroup 0, so that it is dispatched first and its ~7 us of serial work run next to the tiles' sorts, not behind them)
			// SPARSE lists, the extra workgroup of the main instance: what the scan kernel does for compact lists -- num_rendered and the
			// longest list from the tiles' counts (final since the scatter launch) into ctl and the caller's mailbox, and the blend
			// kernels' tile order
			__shared__ uint32_t s_rep[ORDER_BUCKETS];
			__shared__ uint32_t s_sum[THREADS / WAVE], s_max[THREADS / WAVE];
			uint32_t sum = 0u, mx = 0u;
			for (int t = threadIdx.x; t < T; t += THREADS) { const uint32_t c = list_end[t]; sum += c; mx = max(mx, c); }
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) { sum += (uint32_t)__shfl_xor((int)sum, o); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
			if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_max[threadIdx.x >> 6] = mx; }
			__syncthreads();
			uint32_t gtot = 0u, gmax = 0u;
#pragma unroll
			for (int w = 0; w < THREADS / WAVE; w++) { gtot += s_sum[w]; gmax = max(gmax, s_max[w]); }
			if (threadIdx.x == 0)
			{
				report_ctl[0] = gtot; report_ctl[1] = gmax;
				if (report_box)
				{
					report_box[0] = gtot; report_box[1] = gmax;
					__hip_atomic_store(&report_box[2], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
				}
			}
			if (order_out != nullptr) tile_order_block(list_end, order_out + tile_order_tmp_off(T), T, gmax, order_out, s_rep);
			return;
		}
		const int wg = (int)blockIdx.x - (report_ctl != nullptr ? 1 : 0);
		if (wg >= T) return;
		const int tile = order != nullptr ? (int)order[wg] : wg;
		// launched before the host knew num_rendered: if the buffers are too small the scatter pass did not run either (the counters
		// are not list ends) -- every tile is reported empty, so that whatever is queued behind reads nothing, and the host starts over
		if (sparse_cap == 0u && ctl[0] > capacity)
		{
			if (n_lo == 0 && threadIdx.x == 0) ranges[tile] = make_uint2(0u, 0u);
			return;
		}
		tile_sort_one<THREADS, ITEMS>(tile, list_end, pairs, point_list, ranges, big_scratch, n_lo, lds_cap, rank_max, last, sparse_cap);
	}

	// ------------------------------------------------------------------------------------------------
	// host side
	// ------------------------------------------------------------------------------------------------
	constexpr int BIN_LDS_MAX_TILES = 36 * 1024;   // a 144 KiB histogram stays inside the 160 KiB of a CU
	static inline int bin_rounds(int T) { return T <= 8192 ? 1 : 4; }   // bigger histograms: fewer, longer workgroups

	template <bool SCATTER>
	static hipError_t launch_tile_bin(const uint16_t* rect, const float* depths, int P, int grid_x, int T, uint32_t* counters, uint32_t* pairs,
	                                  const uint32_t* ctl, uint32_t capacity, uint32_t* order, hipStream_t stream, uint32_t sparse_cap = 0u)
	{
		if (P <= 0) return hipSuccess;
		const ushort4* r4 = reinterpret_cast<const ushort4*>(rect);
		uint2* p2 = reinterpret_cast<uint2*>(pairs);
		if (T >= (1 << 24)) order = nullptr;   // (tile_order_block packs a rank into 24 bits)
		const int extra = (SCATTER && order) ? 1 : 0;   // one more workgroup: tile_order_block
		if (T <= BIN_LDS_MAX_TILES)
		{
			// the attribute belongs to the (kernel, device) pair: set once per device this process launches on
			static std::atomic<unsigned long long> attr_done{0};   // bit d: set on device d
			int dev = 0;
			hipError_t e = hipGetDevice(&dev);
			if (e != hipSuccess) return e;
			if (dev >= 64 || !((attr_done.load(std::memory_order_acquire) >> dev) & 1ull))
			{
				e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_bin_lds_kernel<SCATTER>),
				                        hipFuncAttributeMaxDynamicSharedMemorySize, BIN_LDS_MAX_TILES * 4);
				if (e != hipSuccess) return e;
				if (dev < 64) attr_done.fetch_or(1ull << dev, std::memory_order_release);
			}
			const int rounds = bin_rounds(T);
			hipLaunchKernelGGL(tile_bin_lds_kernel<SCATTER>, dim3(div_up(P, rounds * BIN_T) + extra), dim3(BIN_T),
			                   (size_t)max(T, ORDER_BUCKETS) * 4, stream, r4, depths, P, grid_x, T, rounds, counters, p2, ctl, capacity, order, sparse_cap);
		}
		else
			hipLaunchKernelGGL(tile_bin_direct_kernel<SCATTER>, dim3(div_up(P, 256) + extra), dim3(256), 0, stream, r4, depths, P, grid_x, counters, p2,
			                   ctl, capacity, order, T, sparse_cap);
		return hipGetLastError();
	}

	hipError_t launch_tile_count(const uint16_t* rect, int P, int grid_x, int T, uint32_t* counters, hipStream_t stream)
	{
		return launch_tile_bin<false>(rect, nullptr, P, grid_x, T, counters, nullptr, nullptr, 0u, nullptr, stream);
	}

	hipError_t launch_tile_scan(uint32_t* counters, int T, uint32_t* ctl, uint32_t* host_box, uint32_t ticket, uint32_t* tile_order, hipStream_t stream)
	{
		const int per_thread = div_up(div_up(T, SCAN_T), 4) * 4;
		hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(SCAN_T), 0, stream, counters, T, per_thread, ctl, host_box, ticket,
		                   tile_order ? tile_order + tile_order_counts_off(T) : nullptr);
		return hipGetLastError();
	}

	hipError_t launch_tile_scatter(const uint16_t* rect, const float* depths, int P, int grid_x, int T, uint32_t* counters, uint32_t* pairs,
	                               const uint32_t* ctl, uint32_t capacity, uint32_t* tile_order, hipStream_t stream, uint32_t sparse_cap)
	{
		// sparse lists: the counters are still zero (no count / scan pass ran); the tile order is left to the sort launch (tile_sort_kernel)
		return launch_tile_bin<true>(rect, depths, P, grid_x, T, counters, pairs, ctl, capacity, sparse_cap ? nullptr : tile_order, stream, sparse_cap);
	}

	// test hook: cap the list length the LDS instances take, and the crowded-bucket threshold
	static std::atomic<int> g_lds_cap{TS_LARGE}, g_rank_max{96};
	void tile_sort_debug_limits(int lds_cap, int rank_max)
	{
		g_lds_cap.store(lds_cap > 0 && lds_cap < TS_LARGE ? lds_cap : TS_LARGE);
		g_rank_max.store(rank_max > 0 ? min(rank_max, TS_PAD - 4) : 96);
	}
	int tile_sort_lds_cap() { return g_lds_cap.load(); }

	template <int THREADS, int ITEMS = TS_ITEMS>
	static hipError_t launch_sort_instance(const uint32_t* counters, int T, const uint2* pairs, uint32_t* point_list, uint2* ranges, u64* big,
	                                 int n_lo, int cap, int rank_max, const uint32_t* ctl, uint32_t capacity, bool last, const uint32_t* order,
	                                 hipStream_t stream, uint32_t sparse_cap = 0u, uint32_t* report_ctl = nullptr, uint32_t* report_box = nullptr,
	                                 uint32_t ticket = 0u, uint32_t* order_out = nullptr)
	{
		const size_t lds = (size_t)cap * 8 + TS_PAD * 4;
		if (lds > 48 * 1024)
		{
			// more dynamic LDS than the default limit: raise it once per (instance, device)
			static std::atomic<unsigned long long> attr_done{0};   // bit d: set on device d
			int dev = 0;
			hipError_t e = hipGetDevice(&dev);
			if (e != hipSuccess) return e;
			if (dev >= 64 || !((attr_done.load(std::memory_order_acquire) >> dev) & 1ull))
			{
				e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_sort_kernel<THREADS, ITEMS>), hipFuncAttributeMaxDynamicSharedMemorySize,
				                        THREADS * ITEMS * 8 + TS_PAD * 4);
				if (e != hipSuccess) return e;
				if (dev < 64) attr_done.fetch_or(1ull << dev, std::memory_order_release);
			}
		}
		hipLaunchKernelGGL((tile_sort_kernel<THREADS, ITEMS>), dim3(T + (report_ctl ? 1 : 0)), dim3(THREADS), lds, stream,
		                   counters, pairs, point_list, ranges, big, n_lo, cap, rank_max, ctl, capacity, last ? 1 : 0, order, T,
		                   sparse_cap, report_ctl, report_box, ticket, order_out);
		return hipSuccess;
	}

	// sparse_cap != 0: SPARSE lists (see tile_bin_lds_kernel): `counters` hold the tiles' counts, tile t's list sits at t * sparse_cap;
	// the main instance carries one extra workgroup (its first) that reports num_rendered / the longest list into `report_ctl` and the mailbox
	// `report_box` (with `ticket`) and writes the blend kernels' tile order into `order_out`; max_count = the longest list PROVIDED FOR
	hipError_t launch_tile_sort(const uint32_t* counters, int T, int max_count, const uint32_t* pairs, uint32_t* point_list, uint32_t* ranges,
	                            void* big_scratch, const uint32_t* ctl, uint32_t capacity, const uint32_t* tile_order, hipStream_t stream,
	                            uint32_t sparse_cap, uint32_t* report_ctl, uint32_t* report_box, uint32_t ticket, uint32_t* order_out)
	{
		const int lds_cap = g_lds_cap.load(), rank_max = g_rank_max.load();
		const uint2* p2 = reinterpret_cast<const uint2*>(pairs);
		uint2* r2 = reinterpret_cast<uint2*>(ranges);
		u64* big = reinterpret_cast<u64*>(big_scratch);
		// The main instance takes every tile and is sized by the longest list: 128 threads per tile while no list
		// exceeds 1024 entries, else 256 (up to 2048 entries; measured at C3, where a quarter of the lists are longer than
		// 1024: one 256-thread launch 62 us, a 128-thread launch plus a 256-thread launch for the long ones 88 us).  Lists
		// beyond 2048 are rare on a uniform scene and the rule on a clustered one (a trained scene: most Gaussians on the
		// subject; C3-clustered: 700 of 5440 tiles).  They are taken by up to two more instances, all in LDS: 512 threads x
		// 8 keys for (2048, 4096], and for what is longer still 1024 threads x 8 keys while nothing exceeds 8192, else 1024 x
		// 16 (up to 16384).  Every instance is a launch over all tiles whose workgroups leave at once when the list is not
		// theirs (looping launches of a few hundred workgroups that walk the tile order from its long end were built and are
		// slower: 146 against 99 us for the three launches on C3-clustered -- the loop costs the body 40-80 VGPRs).  Only what is
		// longer than lds_cap (16384) still goes through the bitonic network in global scratch (91 rounds of global round trips
		// for 8192 keys: it used to take everything beyond 4096 -- 0.33 ms for the sort on C3-clustered).  An instance's LDS is
		// sized by the longest list it takes, in 64-key steps (a short longest list = more tiles in flight per CU).
		const int longest_lds = min(max_count, lds_cap);
		const int c1 = min(128 * TS_ITEMS, lds_cap), c2 = min(256 * TS_ITEMS, lds_cap), c3 = min(512 * TS_ITEMS, lds_cap);
		const int c4 = min(1024 * TS_ITEMS, lds_cap), c5 = min(1024 * TS_ITEMS_LONG, lds_cap);
		const auto lds_keys = [&](int c) { return min(c, max(64, div_up(longest_lds, 64) * 64)); };
		const bool overflow = max_count > lds_cap;   // somebody has to take the global path
		if (T >= (1 << 24)) { tile_order = nullptr; order_out = nullptr; }   // no order is written (launch_tile_bin; check_scene rejects such images anyway)
		if (sparse_cap) tile_order = nullptr;   // the order is only being written by this launch: tiles by index
		const uint32_t sc = sparse_cap;
		hipError_t e = hipSuccess;
		if (max_count <= c1 || c2 == c1)
			e = launch_sort_instance<128>(counters, T, p2, point_list, r2, overflow ? big : nullptr, 0, lds_keys(c1), rank_max, ctl, capacity, true, tile_order, stream,
			                              sc, report_ctl, report_box, ticket, order_out);
		else
		{
			const bool second = max_count > c2 && c3 > c2;
			const bool third = second && longest_lds > c3 && c4 > c3;
			e = launch_sort_instance<256>(counters, T, p2, point_list, r2, (overflow && !second) ? big : nullptr, 0, lds_keys(c2), rank_max, ctl, capacity, !second, tile_order, stream,
			                              sc, report_ctl, report_box, ticket, order_out);
			if (second && e == hipSuccess)
				e = launch_sort_instance<512>(counters, T, p2, point_list, r2, (overflow && !third) ? big : nullptr, c2, lds_keys(c3), rank_max, ctl, capacity, !third, tile_order,
				                              stream, sc);
			if (third && e == hipSuccess)
			{
				if (longest_lds <= c4 || c5 == c4)
					e = launch_sort_instance<1024>(counters, T, p2, point_list, r2, overflow ? big : nullptr, c3, lds_keys(c4), rank_max, ctl, capacity, true, tile_order, stream, sc);
				else
					e = launch_sort_instance<1024, TS_ITEMS_LONG>(counters, T, p2, point_list, r2, overflow ? big : nullptr, c3, lds_keys(c5), rank_max, ctl, capacity, true,
					                                              tile_order, stream, sc);
			}
		}
		if (e != hipSuccess) return e;
		return hipGetLastError();
	}
}

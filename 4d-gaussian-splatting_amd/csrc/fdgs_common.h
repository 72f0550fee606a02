// fdgs_common.h -- internal declarations shared by the HIP translation units of libfdgs.so.
//
// Scratch-buffer layouts, launch geometry constants and the host-side entry
// point of each stage.  Written for gfx950 only (wave64, 256 CUs / 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stddef.h>
#include <stdint.h>
#include "../../include/fdgs.h"

namespace fdgs
{
	constexpr int TILE_X = 16;          // reference config.h:16-17 (the tile grid defines key/range indexing)
	constexpr int TILE_Y = 16;
	constexpr int WAVE = 64;            // gfx950 wavefront
	constexpr int SORT_THREADS = 256;   // radix sort workgroup
	constexpr int SORT_ITEMS = 16;      // keys per thread (large inputs: 4096 keys per workgroup)
	constexpr int SORT_ITEMS_SMALL = 4; // keys per thread for inputs <= 1 M keys (1024 per workgroup: fills the chip)
	static inline int sort_items_for(int n) { return n <= (1 << 20) ? SORT_ITEMS_SMALL : SORT_ITEMS; }
	static inline int sort_blocks(int n) { return (n + SORT_THREADS * sort_items_for(n) - 1) / (SORT_THREADS * sort_items_for(n)); }
	constexpr int RADIX_BITS = 8;
	constexpr int RADIX = 1 << RADIX_BITS;
	constexpr int GRAD_ACC_WORDS = 16;  // packed per-Gaussian gradient accumulator record (64 B = one cache line)

	static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
	static inline int div_up(int a, int b) { return (a + b - 1) / b; }

	// Packed per-Gaussian blend record, 3 x float4 = 48 B, written by preprocess and
	// gathered by both blend kernels (one record instead of the reference's four arrays
	// means2D / conic_opacity / rgb / depths + the flow input):
	//   a = (x, y, conic.x, conic.y)   b = (conic.z, opacity, r, g)   c = (b, depth, flow.x, flow.y)
	// Loads / stores of data that is touched ONCE per optimizer step and is larger than the 256 MB Infinity Cache -- the two Adam moments
	// of the SH coefficients (0.69 GB at C3, 0.77 GB at C5): non-temporal, so that they neither wait for cache lines nor push out what
	// IS used again (the coefficients themselves -- the next forward reads them -- stay ordinary accesses).  Measured on MI355X
	// (profiles/HISTORY.md, round 6): fused SH flush + Adam 214 -> 180 us (1.08 GB: 6.0 TB/s), step +1.5 % at C3, +2.3 % at C5;
	// coefficients non-temporal as well: 190 us; the GEOMETRY parameters' moments (41 MB at C3: they live in that cache from step to
	// step) non-temporal: step -1 ... -2 %, left alone.  -DFDGS_STREAM_PLAIN: ordinary accesses (A/B).
#ifndef FDGS_STREAM_PLAIN
	typedef float fdgs_v4f __attribute__((ext_vector_type(4)));
	__device__ __forceinline__ float4 stream_ld(const float4* p)
	{
		const fdgs_v4f v = __builtin_nontemporal_load(reinterpret_cast<const fdgs_v4f*>(p));
		return make_float4(v.x, v.y, v.z, v.w);
	}
	__device__ __forceinline__ void stream_st(float4* p, float4 x)
	{
		fdgs_v4f v = { x.x, x.y, x.z, x.w };
		__builtin_nontemporal_store(v, reinterpret_cast<fdgs_v4f*>(p));
	}
#else
	__device__ __forceinline__ float4 stream_ld(const float4* p) { return *p; }
	__device__ __forceinline__ void stream_st(float4* p, float4 x) { *p = x; }
#endif

	struct GeomLayout
	{
		size_t records, depths, cov3D, tiles_touched, rect, clamped;
		size_t total;
	};
	static inline GeomLayout geom_layout(int P)
	{
		GeomLayout L;
		size_t o = 0;
		const size_t p = (size_t)(P > 0 ? P : 1);
		L.records = o; o = align_up(o + p * 48);
		L.depths = o; o = align_up(o + p * 4);
		L.cov3D = o; o = align_up(o + p * 24);
		L.tiles_touched = o; o = align_up(o + p * 4);
		L.rect = o; o = align_up(o + p * 8);
		L.clamped = o; o = align_up(o + p);
		L.total = o;
		return L;
	}

	// Tile binning (tilebin.hip): one instance counter per tile, padded to whole uint4s
	static inline size_t bin_counter_words(int T) { return ((size_t)T + 3) & ~(size_t)3; }
	// The tile-order area (ImageLayout::tile_order, 3 T + 16 words): [0, T) the order; at align4(T) the scan's copy of the tile
	// counts (whole uint4 stores: the offset is a multiple of 4 words for every T); behind it T words of scratch
	__host__ __device__ static inline int tile_order_counts_off(int T) { return (T + 3) & ~3; }
	__host__ __device__ static inline int tile_order_tmp_off(int T) { return 2 * ((T + 3) & ~3) + 4; }

	struct ImageLayout
	{
		size_t final_T, n_contrib, ranges;
		size_t tile_counters;   // [T] instance counts -> exclusive starts -> ends (count / scan / scatter passes)
		size_t bin_ctl;         // { R, longest tile list } written by the scan, read back by the host
		size_t tile_order;      // [T] the order in which the blend kernels take the tiles (all tiles, longest lists first; position p
		                        // goes to XCD p % 8), written by one workgroup of the scatter launch from the
		                        // [T] counts the scan leaves behind it (tile_order_counts_off); [T] words of scratch behind those
		size_t total;
	};
	static inline ImageLayout image_layout(int W, int H)
	{
		ImageLayout L;
		size_t o = 0;
		const size_t n = (size_t)W * H;
		const size_t t = (size_t)div_up(W, TILE_X) * div_up(H, TILE_Y);
		L.final_T = o; o = align_up(o + n * 4);
		L.n_contrib = o; o = align_up(o + n * 4);
		L.ranges = o; o = align_up(o + t * 8);
		L.tile_counters = o; o = align_up(o + bin_counter_words((int)t) * 4);
		L.bin_ctl = o; o = align_up(o + 16);
		L.tile_order = o; o = align_up(o + (3 * t + 16) * 4);
		L.total = o;
		return L;
	}

	// Binning buffer: the sorted instance list the blend kernels walk, the unsorted (depth bits, id) pairs of the
	// scatter pass, and -- only when some tile's list is longer than the LDS sort takes -- R keys of global scratch.
	struct BinLayout
	{
		size_t point_list, cull_bits, pairs, big_scratch, total;
		uint32_t cull_stride;   // 64-bit words per sub-block plane of cull_bits
	};
	// cull_bits: what the blend forward's per-block cull decided, one bit per (list entry, 8x8 block), kept for the blend backward
	// (which used to run the same test again: a tenth of its instructions).  Four planes (one per sub-block of a tile) of
	// cull_stride 64-bit words; the 64 entries [64 c, 64 c + 64) of tile t's list sit in word (list start >> 6) + t + c of the
	// plane -- consecutive tiles never share a word (the + t), and the index stays below (R >> 6) + T + 1.
	static inline BinLayout bin_layout(int R, bool with_big_scratch, int T)
	{
		BinLayout L;
		size_t o = 0;
		const size_t r = (size_t)(R > 0 ? R : 1);
		L.point_list = o; o = align_up(o + r * 4);
		L.cull_stride = (uint32_t)((r >> 6) + (size_t)(T > 0 ? T : 0) + 2);
		L.cull_bits = o; o = align_up(o + (size_t)L.cull_stride * 4 * 8);
		L.pairs = o; o = align_up(o + r * 8);
		L.big_scratch = o;
		if (with_big_scratch) o = align_up(o + r * 8);
		L.total = o;
		return L;
	}

	// ---- stage launchers (each enqueues on `stream`, returns hipError_t) ----

	hipError_t launch_preprocess_fwd(const fdgs_scene& s, const fdgs_forward_out& out, char* geom, uint32_t* bin_counters, int part,
	                                 hipStream_t stream);   // part 0: one launch; 1 / 2: geometry / colour halves

	// SH colours of several views of the same Gaussians in one pass over the coefficients (after the views' part-1 launches)
	hipError_t launch_colour_batch(int nviews, const fdgs_scene* const* views, const fdgs_forward_out* const* outs, char* const* geoms,
	                               hipStream_t stream);

	// Stable LSD radix sort of (key,value) u32 pairs on key bits [bit_lo, bit_hi) (radix_sort.hip; used by knn.hip).
	// keys[0]/vals[0] hold the input; *result receives the index (0/1) of the buffers holding the output.
	hipError_t radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], int n, int bit_lo, int bit_hi,
	                            uint32_t* hist, hipStream_t stream, int* result);

	// Tile binning (tilebin.hip).  counters: bin_counter_words(T) words, zero on entry of the count pass (cleared by
	// preprocess_fwd, the forward's first kernel); after the scan they hold every tile list's start, after the
	// scatter its end.  ctl[0] = R, ctl[1] = longest tile list.
	hipError_t launch_tile_count(const uint16_t* rect, int P, int grid_x, int T, uint32_t* counters, hipStream_t stream);
	// host_box (optional): device pointer of a pinned host mailbox {R, longest, ticket} the kernel writes directly
	// tile_order (optional, 3 T + 16 words, see ImageLayout): the scan leaves a copy of the tile counts at tile_order + tile_order_counts_off(T)
	hipError_t launch_tile_scan(uint32_t* counters, int T, uint32_t* ctl, uint32_t* host_box, uint32_t ticket, uint32_t* tile_order, hipStream_t stream);
	// scatter / sort may be launched before the host knows num_rendered: they compare ctl[0] with `capacity` (the instances
	// pairs / point_list hold) and leave everything alone -- the sort reports every tile empty -- when it does not fit
	// tile_order (optional): one extra workgroup of this launch turns the scan's copy of the counts into the blend kernels' tile order
	// sparse_cap != 0: SPARSE lists (fdgs_forward_out.sparse_lists): no count / scan pass ran, the counters are zero, tile t's list goes to
	// [t * sparse_cap, (t + 1) * sparse_cap) of `pairs` / point_list; afterwards the counters hold the tiles' counts
	hipError_t launch_tile_scatter(const uint16_t* rect, const float* depths, int P, int grid_x, int T, uint32_t* counters, uint32_t* pairs,
	                               const uint32_t* ctl, uint32_t capacity, uint32_t* tile_order, hipStream_t stream, uint32_t sparse_cap = 0u);
	// tile_order (optional): the order the scatter launch wrote -- the sort takes the tiles in it too (longest lists first)
	// sparse lists: report_ctl / report_box / ticket / order_out -- the sort's main instance reports num_rendered and the longest list and
	// writes the tile order (what the scan + scatter launches do for compact lists)
	hipError_t launch_tile_sort(const uint32_t* counters, int T, int max_count, const uint32_t* pairs, uint32_t* point_list, uint32_t* ranges,
	                            void* big_scratch, const uint32_t* ctl, uint32_t capacity, const uint32_t* tile_order, hipStream_t stream,
	                            uint32_t sparse_cap = 0u, uint32_t* report_ctl = nullptr, uint32_t* report_box = nullptr, uint32_t ticket = 0u,
	                            uint32_t* order_out = nullptr);
	int tile_sort_lds_cap();                                   // lists longer than this need the global scratch
	void tile_sort_debug_limits(int lds_cap, int rank_max);    // test hook (fdgs_debug_tile_sort_limits); <= 0 restores the default

	// tile_order: NULL = tiles in index order
	// cull_bits / cull_stride (BinLayout; NULL: not kept) and ctl (the image buffer's control words): the forward leaves the plane
	// stride in ctl[2] and the plane array's offset from the start of the binning buffer (in 64-bit words) in ctl[3] for the backward
	hipError_t launch_blend_fwd(const fdgs_scene& s, const fdgs_forward_out& out, const float* records,
	                            const uint32_t* point_list, const uint32_t* ranges, const uint32_t* tile_order,
	                            float* final_T, uint32_t* n_contrib, unsigned long long* cull_bits, uint32_t cull_stride, uint32_t cull_word_off,
	                            uint32_t* ctl, hipStream_t stream);

	hipError_t launch_blend_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                            const float* records, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* tile_order,
	                            const float* final_T, const uint32_t* n_contrib, const uint32_t* ctl /* cull planes: see launch_blend_fwd */,
	                            hipStream_t stream);

	// SH / 4D-SH backward (coalesced); must run after the blend backward and before launch_preprocess_bwd
	hipError_t launch_sh_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                         const char* geom, hipStream_t stream);

	// SH backward (deferred mode: stage records + mean / time gradient) of several views in one pass over the coefficients
	hipError_t launch_sh_bwd_batch(int nviews, const fdgs_scene* const* views, const fdgs_backward_in* const* ins,
	                               const fdgs_backward_out* const* outs, hipStream_t stream);

	// once per optimizer step: dL_dsh from the staged per-view records of the deferred SH backward (sh_bwd.hip)
	hipError_t launch_sh_flush(int P, int D, int D_t, int M, int gaussian_dim, int force_sh_3d, int analytic, int nviews,
	                           const float* stages, float* dL_dsh, int accumulate, hipStream_t stream);

	// once per optimizer step, instead of launch_sh_flush + Adam over the SH segment: the summed SH gradient is built in LDS and
	// consumed by the Adam update in the same kernel (sh_bwd.hip); dL_dsh (optional) also receives it
	struct AdamScalars { float lr_head_bc1, lr_bc1, b1, b2, eps, inv_sqrt_bc2; };
	hipError_t launch_sh_adam(int P, int D, int D_t, int M, int gaussian_dim, int force_sh_3d, int analytic, int nviews,
	                          const float* stages, float* params, float* exp_avg, float* exp_avg_sq, float* dL_dsh,
	                          const AdamScalars& k, hipStream_t stream);

	hipError_t launch_preprocess_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                                 const char* geom, hipStream_t stream);

	hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream);

	hipError_t launch_block_reaches_debug(int n, const float* tuples, uint8_t* out, hipStream_t stream);

	// One element of torch.optim.Adam (no amsgrad / weight decay; train.py:247-249), shared by adam.hip and the fused SH
	// update of sh_bwd.hip so that both round identically: explicit FMAs, independent of the translation unit's contraction mode.
	__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float lr_bc1, float b1, float b2, float eps,
	                                            float inv_sqrt_bc2)
	{
		m = fmaf(b1, m, (1.f - b1) * g);
		v = fmaf(b2, v, ((1.f - b2) * g) * g);
		const float denom = fmaf(sqrtf(v), inv_sqrt_bc2, eps);
		p = fmaf(-lr_bc1, m / denom, p);
	}

	hipError_t launch_activations(int P, const float* opacity_raw, const float* scales_raw, const float* scales_t_raw, const float* rot_raw,
	                              const float* rot_r_raw, float* opacity, float* scales, float* scales_t, float* rot, float* rot_r, hipStream_t stream);
}

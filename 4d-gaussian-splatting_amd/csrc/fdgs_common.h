// fdgs_common.h -- internal declarations shared by the HIP translation units of libfdgs.so.
//
// Scratch-buffer layouts, launch geometry constants and the host-side entry
// point of each stage.  Written for gfx950 only (wave64, 256 CUs / 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
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
	constexpr int SCAN_CHUNK = 4096;
	constexpr int GRAD_ACC_WORDS = 16;  // packed per-Gaussian gradient accumulator record (64 B = one cache line)    // elements per workgroup in the 3-phase scan

	static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
	static inline int div_up(int a, int b) { return (a + b - 1) / b; }

	// Packed per-Gaussian blend record, 3 x float4 = 48 B, written by preprocess and
	// gathered by both blend kernels (one record instead of the reference's four arrays
	// means2D / conic_opacity / rgb / depths + the flow input):
	//   a = (x, y, conic.x, conic.y)   b = (conic.z, opacity, r, g)   c = (b, depth, flow.x, flow.y)
	struct GeomLayout
	{
		size_t records, depths, cov3D, tiles_touched, rect, clamped;
		size_t sort_key[2], sort_val[2];   // depth sort ping-pong (keys = depth bits, vals = Gaussian id)
		size_t offsets;                    // exclusive scan of tiles_touched in depth order
		size_t scan_block;                 // per-chunk sums of the scan (+1 slot: grand total = R)
		size_t hist;                       // radix block histograms [RADIX][nblocks] + RADIX digit totals
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
		for (int i = 0; i < 2; i++) { L.sort_key[i] = o; o = align_up(o + p * 4); }
		for (int i = 0; i < 2; i++) { L.sort_val[i] = o; o = align_up(o + p * 4); }
		L.offsets = o; o = align_up(o + p * 4);
		L.scan_block = o; o = align_up(o + ((size_t)div_up((int)p, SCAN_CHUNK) + 2) * 8);   // 64-bit look-back words + the total
		L.hist = o; o = align_up(o + (size_t)RADIX * (sort_blocks((int)p) + 1) * 4);
		L.total = o;
		return L;
	}

	struct ImageLayout
	{
		size_t final_T, n_contrib, ranges, total;
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
		L.total = o;
		return L;
	}

	struct BinLayout
	{
		size_t key[2], val[2], hist, total;
	};
	static inline BinLayout bin_layout(int R)
	{
		BinLayout L;
		size_t o = 0;
		const size_t r = (size_t)(R > 0 ? R : 1);
		for (int i = 0; i < 2; i++) { L.key[i] = o; o = align_up(o + r * 4); }
		for (int i = 0; i < 2; i++) { L.val[i] = o; o = align_up(o + r * 4); }
		L.hist = o; o = align_up(o + (size_t)RADIX * (sort_blocks((int)r) + 1) * 4);
		L.total = o;
		return L;
	}

	// number of tile-id bits the instance sort has to look at, and its pass count
	static inline int tile_bits(int T)
	{
		int b = 1;
		while ((1 << b) < T) b++;
		return b;
	}
	static inline int tile_sort_passes(int T) { return div_up(tile_bits(T), RADIX_BITS); }

	// ---- stage launchers (each enqueues on `stream`, returns hipError_t) ----

	hipError_t launch_preprocess_fwd(const fdgs_scene& s, const fdgs_forward_out& out, char* geom, hipStream_t stream);

	// Stable LSD radix sort of (key,value) u32 pairs on key bits [bit_lo, bit_hi).
	// keys[0]/vals[0] hold the input; *result receives the index (0/1) of the buffers holding the output.
	hipError_t radix_sort_pairs(uint32_t* keys[2], uint32_t* vals[2], int n, int bit_lo, int bit_hi,
	                            uint32_t* hist, hipStream_t stream, int* result);

	// offsets[j] = exclusive prefix sum of tiles_touched[order[j]]; total (= R) written to scan_total_ptr(block_sums, P).
	// block_sums: one 64-bit look-back word per 1024-element chunk, zero on entry, then the 32-bit total.
	constexpr int LOOKBACK_MAX_BLOCKS = 2048;   // all workgroups of the single-pass scan must be resident at once
	static inline uint32_t* scan_total_ptr(uint32_t* block_sums, int P) { return block_sums + 2 * (size_t)div_up(P, SCAN_CHUNK); }
	static inline int scan_state_words(int P) { return div_up(P, SCAN_CHUNK) <= LOOKBACK_MAX_BLOCKS ? div_up(P, SCAN_CHUNK) : 0; }
	hipError_t launch_offsets_scan(const uint32_t* tiles_touched, const uint32_t* order, int P,
	                               uint32_t* offsets, uint32_t* block_sums, hipStream_t stream);

	// Emit one (tile id, Gaussian id) instance per covered tile, in depth order.
	hipError_t launch_emit_instances(const uint32_t* order, const uint32_t* offsets, const uint16_t* rect,
	                                 int P, int R, int grid_x, uint32_t* keys, uint32_t* vals, hipStream_t stream);

	hipError_t launch_tile_ranges(const uint32_t* sorted_tile_keys, int R, int T, uint32_t* ranges, hipStream_t stream);

	hipError_t launch_blend_fwd(const fdgs_scene& s, const fdgs_forward_out& out, const float* records,
	                            const uint32_t* point_list, const uint32_t* ranges,
	                            float* final_T, uint32_t* n_contrib, hipStream_t stream);

	hipError_t launch_blend_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                            const float* records, const uint32_t* point_list, const uint32_t* ranges,
	                            const float* final_T, const uint32_t* n_contrib, hipStream_t stream);

	// SH / 4D-SH backward (coalesced); must run after the blend backward and before launch_preprocess_bwd
	hipError_t launch_sh_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                         const char* geom, hipStream_t stream);

	hipError_t launch_preprocess_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                                 const char* geom, hipStream_t stream);

	hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream);
}

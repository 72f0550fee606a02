// TEST INFRASTRUCTURE ONLY.
// oracle_api.h entry points for the verbatim-reference oracle ("reference"
// kind).  The orchestration below follows CudaRasterizer::Rasterizer::forward /
// backward (rasterizer_impl.cu:199-364, 368-496); CUB's InclusiveSum and
// SortPairs (exact integer semantics) are replaced by a serial scan and
// std::stable_sort on the 64-bit keys.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#include <omp.h>
#include "ref_internal.h"

static inline dim3 tile_grid(const oracle_io* io)
{
	return dim3((io->W + 15) / 16, (io->H + 15) / 16, 1); // rasterizer_impl.cu:247
}

extern "C" int oracle_forward(oracle_io* io)
{
	const int P = io->P, W = io->W, H = io->H;
	const float focal_y = H / (2.0f * io->tan_fovy); // rasterizer_impl.cu:235-236
	const float focal_x = W / (2.0f * io->tan_fovx);
	const dim3 grid = tile_grid(io);
	const size_t N = (size_t)W * H, T = (size_t)grid.x * grid.y;

	// rasterize_points.cu:80-85: outputs are zero-filled, out_means3D = clone(means3D);
	// the geometry scratch is torch::empty in the reference -- zeroed here for determinism.
	memset(io->out_color, 0, 3 * N * 4); memset(io->out_flow, 0, 2 * N * 4);
	memset(io->out_depth, 0, N * 4); memset(io->out_T, 0, N * 4);
	memset(io->n_contrib, 0, N * 4);
	memset(io->radii, 0, (size_t)P * 4);
	memcpy(io->out_means3D, io->means3D, (size_t)P * 12);
	memset(io->means2D, 0, (size_t)P * 8); memset(io->depths, 0, (size_t)P * 4);
	memset(io->cov3D, 0, (size_t)P * 24); memset(io->rgb, 0, (size_t)P * 12);
	memset(io->conic_opacity, 0, (size_t)P * 16);
	memset(io->tiles_touched, 0, (size_t)P * 4); memset(io->point_offsets, 0, (size_t)P * 4);
	memset(io->clamped, 0, (size_t)P * 3);
	if (io->border) memset(io->border, 0, N);
	if (io->border_g) memset(io->border_g, 0, (size_t)P);
	io->R = 0; io->keys_sorted = nullptr; io->point_list = nullptr;
	if (P == 0) { memset(io->ranges, 0, T * 8); return 0; }

	ref_preprocess_fwd_all(io, focal_x, focal_y, grid);

	// rasterizer_impl.cu:298 InclusiveSum
	uint32_t run = 0;
	for (int i = 0; i < P; i++) { run += io->tiles_touched[i]; io->point_offsets[i] = run; }
	const int R = (int)run; // :302
	io->R = R;

	std::vector<uint64_t> keys_unsorted((size_t)R);
	std::vector<uint32_t> vals_unsorted((size_t)R);
	ref_duplicate_all(io, grid, keys_unsorted.data(), vals_unsorted.data()); // :310

	// :325 stable LSD radix sort over all 64 key bits == stable sort by key
	std::vector<uint32_t> perm((size_t)R);
	std::iota(perm.begin(), perm.end(), 0u);
	std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return keys_unsorted[a] < keys_unsorted[b]; });
	io->keys_sorted = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R > 0 ? R : 1));
	io->point_list = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
	for (int i = 0; i < R; i++) { io->keys_sorted[i] = keys_unsorted[perm[i]]; io->point_list[i] = vals_unsorted[perm[i]]; }

	memset(io->ranges, 0, T * 8); // :332
	if (R > 0) ref_ranges_all(io); // :335-339

	ref_render_fwd_all(io, grid); // :345 (writes accum_alpha straight into out_T; :362 is a copy)
	return R;
}

extern "C" int oracle_backward(oracle_io* io)
{
	const int P = io->P, M = io->M;
	const float focal_y = io->H / (2.0f * io->tan_fovy); // rasterizer_impl.cu:424-425
	const float focal_x = io->W / (2.0f * io->tan_fovx);
	// rasterize_points.cu:201-213: all gradients start at zero
	memset(io->dL_dmean2D, 0, (size_t)P * 12); memset(io->dL_dconic, 0, (size_t)P * 16);
	memset(io->dL_dopacity, 0, (size_t)P * 4); memset(io->dL_dcolor, 0, (size_t)P * 12);
	memset(io->dL_dmean3D, 0, (size_t)P * 12); memset(io->dL_dcov3D, 0, (size_t)P * 24);
	if (io->dL_dsh && M > 0) memset(io->dL_dsh, 0, (size_t)P * M * 12);
	memset(io->dL_dflows, 0, (size_t)P * 8); memset(io->dL_dts, 0, (size_t)P * 4);
	memset(io->dL_dscale, 0, (size_t)P * 12); memset(io->dL_dscale_t, 0, (size_t)P * 4);
	memset(io->dL_drot, 0, (size_t)P * 16); memset(io->dL_drot_r, 0, (size_t)P * 16);
	if (P == 0) return 0;
	ref_render_bwd_all(io, tile_grid(io));          // rasterizer_impl.cu:435
	ref_preprocess_bwd_all(io, focal_x, focal_y);   // rasterizer_impl.cu:462
	return 0;
}

extern "C" int oracle_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present)
{
	memset(present, 0, (size_t)P); // rasterize_points.cu:279
	ref_check_frustum_all(P, means3D, viewmatrix, projmatrix, present);
	return 0;
}

extern "C" void oracle_free(oracle_io* io)
{
	free(io->keys_sorted); free(io->point_list);
	io->keys_sorted = nullptr; io->point_list = nullptr;
}

#ifndef ORACLE_KIND_NAME
#define ORACLE_KIND_NAME "reference"
#endif
extern "C" const char* oracle_kind(void) { return ORACLE_KIND_NAME; }   // "reference_fma": the build with FP contraction on (build_ref.py --contract)
extern "C" int oracle_threads(void) { return omp_get_max_threads(); }

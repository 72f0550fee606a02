// TEST INFRASTRUCTURE ONLY.
// Wraps the reference's binning device code (rasterizer_impl.cu up to its
// first host function; pulled in verbatim by build_ref.py as impl_trunc.inc).
#include "impl_trunc.inc"
#include "ref_internal.h"

// Mirrors the launch at rasterizer_impl.cu:310.
void ref_duplicate_all(oracle_io* io, dim3 grid, uint64_t* keys_unsorted, uint32_t* values_unsorted)
{
	const int P = io->P;
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++)
	{
		refemu::g_ctx.grid_rank = (unsigned long long)idx;
		duplicateWithKeys(P, (const float2*)io->means2D, io->depths, io->point_offsets,
			keys_unsorted, values_unsorted, io->radii, grid);
	}
}

// Mirrors the launch at rasterizer_impl.cu:336.
void ref_ranges_all(oracle_io* io)
{
	const int R = io->R;
	for (int idx = 0; idx < R; idx++)
	{
		refemu::g_ctx.grid_rank = (unsigned long long)idx;
		identifyTileRanges(R, io->keys_sorted, (uint2*)io->ranges);
	}
}

// Mirrors the launch at rasterizer_impl.cu:149.
void ref_check_frustum_all(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present)
{
	for (int idx = 0; idx < P; idx++)
	{
		refemu::g_ctx.grid_rank = (unsigned long long)idx;
		checkFrustum(P, means3D, viewmatrix, projmatrix, (bool*)present);
	}
}

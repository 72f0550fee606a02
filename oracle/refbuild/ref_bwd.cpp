// TEST INFRASTRUCTURE ONLY.
// Wraps the reference's backward device code (backward.cu up to its host
// launchers; pulled in verbatim by build_ref.py as backward_trunc.inc).
#include "backward_trunc.inc"
#include "ref_internal.h"
#include "ref_emu.h"

int g_ref_acc_mode = 0;
extern "C" int oracle_set_accumulation(int mode)
{
	if (mode != 0 && mode != 1) return -1;   // the double-accumulation mode is the port oracle's
	g_ref_acc_mode = mode;
	return 0;
}

// Mirrors the launch at backward.cu:1254.
void ref_render_bwd_all(oracle_io* io, dim3 grid)
{
	const float* color_ptr = io->colors_precomp ? io->colors_precomp : io->rgb; // rasterizer_impl.cu:433
	const int ntiles = (int)(grid.x * grid.y);
	const bool reverse = g_ref_acc_mode == 1;   // oracle_set_accumulation(1): tiles and threads in descending order
#pragma omp parallel for schedule(dynamic, 4)
	for (int ti = 0; ti < ntiles; ti++)
	{
		const int t = reverse ? ntiles - 1 - ti : ti;
		refemu::runner().reverse = reverse;
		std::function<void()> body = [&]() {
			renderCUDA<NUM_CHANNELS>(
				(const uint2*)io->ranges, io->point_list, io->W, io->H, io->bg,
				(const float2*)io->means2D, (const float4*)io->conic_opacity, color_ptr,
				io->depths, io->flows, io->out_T, io->n_contrib,
				io->dL_dpix, io->dL_ddepth, io->dL_dmask, io->dL_dflow,
				(float3*)io->dL_dmean2D, (float4*)io->dL_dconic, io->dL_dopacity,
				io->dL_dcolor, io->dL_dflows);
		};
		refemu::runner().run(dim3(t % grid.x, t / grid.x, 0), dim3(BLOCK_X, BLOCK_Y, 1), body);
		refemu::runner().reverse = false;   // the forward / kNN kernels always run in ascending order
	}
}

// Mirrors BACKWARD::preprocess (backward.cu:1139-1229): computeCov2DCUDA then preprocessCUDA.
void ref_preprocess_bwd_all(oracle_io* io, float focal_x, float focal_y)
{
	const int P = io->P;
	const float* cov3D_ptr = io->cov3D_precomp ? io->cov3D_precomp : io->cov3D; // rasterizer_impl.cu:461
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++)
	{
		refemu::g_ctx.grid_rank = (unsigned long long)idx;
		computeCov2DCUDA(P, (const float3*)io->out_means3D, io->radii, cov3D_ptr,
			focal_x, focal_y, io->tan_fovx, io->tan_fovy, io->viewmatrix,
			io->dL_dconic, (const float3*)io->dL_dmean2D, (float3*)io->dL_dmean3D, io->dL_dcov3D);
	}
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++)
	{
		refemu::g_ctx.grid_rank = (unsigned long long)idx;
		preprocessCUDA<NUM_CHANNELS>(
			P, io->D, io->D_t, io->M,
			(const float3*)io->out_means3D, io->radii, io->shs, io->ts, io->opacities,
			(const bool*)io->clamped, io->tiles_touched,
			(const glm::vec3*)io->scales, io->scales_t,
			(const glm::vec4*)io->rotations, (const glm::vec4*)io->rotations_r,
			io->prefilter_var, io->scale_modifier, io->projmatrix,
			(const glm::vec3*)io->campos, io->timestamp, io->time_duration,
			io->rot_4d != 0, io->gaussian_dim, io->force_sh_3d != 0,
			(const float3*)io->dL_dmean2D, (glm::vec3*)io->dL_dmean3D, io->dL_dcolor,
			io->dL_dcov3D, io->dL_dsh, io->dL_dts,
			(glm::vec3*)io->dL_dscale, io->dL_dscale_t,
			(glm::vec4*)io->dL_drot, (glm::vec4*)io->dL_drot_r, io->dL_dopacity);
	}
}

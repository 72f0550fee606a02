// TEST INFRASTRUCTURE ONLY.
// Wraps the reference's forward device code (forward.cu up to its host
// launchers; pulled in verbatim from /root/reference by build_ref.py as
// forward_trunc.inc in a temporary build directory -- never copied into the repo).
#include "forward_trunc.inc"
#include "ref_internal.h"
#include "ref_emu.h"

// Mirrors the launch at forward.cu:696 (one CUDA thread per Gaussian).
void ref_preprocess_fwd_all(oracle_io* io, float focal_x, float focal_y, dim3 grid)
{
	const int P = io->P;
#pragma omp parallel for schedule(static)
	for (int idx = 0; idx < P; idx++)
	{
		refemu::g_ctx.grid_rank = (unsigned long long)idx;
		preprocessCUDA<NUM_CHANNELS>(
			P, io->D, io->D_t, io->M,
			io->means3D, io->out_means3D, io->ts,
			(const glm::vec3*)io->scales, io->scales_t, io->scale_modifier,
			(const glm::vec4*)io->rotations, (const glm::vec4*)io->rotations_r,
			io->opacities, io->shs, (bool*)io->clamped,
			io->cov3D_precomp, io->prefilter_var, io->colors_precomp,
			io->viewmatrix, io->projmatrix, (const glm::vec3*)io->campos,
			io->timestamp, io->time_duration,
			io->rot_4d != 0, io->gaussian_dim, io->force_sh_3d != 0,
			io->W, io->H,
			io->tan_fovx, io->tan_fovy, focal_x, focal_y,
			io->radii, (float2*)io->means2D, io->depths, io->cov3D, io->rgb,
			(float4*)io->conic_opacity, grid, io->tiles_touched, io->prefiltered != 0);
	}
}

// Mirrors the launch at forward.cu:645 (one 16x16 block per tile).
void ref_render_fwd_all(oracle_io* io, dim3 grid)
{
	const float* feature_ptr = io->colors_precomp ? io->colors_precomp : io->rgb; // rasterizer_impl.cu:343
	const int ntiles = (int)(grid.x * grid.y);
#pragma omp parallel for schedule(dynamic, 4)
	for (int t = 0; t < ntiles; t++)
	{
		std::function<void()> body = [&]() {
			renderCUDA<NUM_CHANNELS>(
				(const uint2*)io->ranges, io->point_list, io->W, io->H,
				(const float2*)io->means2D, feature_ptr, io->flows, io->depths,
				(const float4*)io->conic_opacity, io->out_T, io->n_contrib, io->bg,
				io->out_color, io->out_flow, io->out_depth);
		};
		refemu::runner().run(dim3(t % grid.x, t / grid.x, 0), dim3(BLOCK_X, BLOCK_Y, 1), body);
	}
}

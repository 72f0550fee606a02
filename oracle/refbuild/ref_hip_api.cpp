// TEST / BASELINE INFRASTRUCTURE ONLY -- never linked into libfdgs.so, never imported by the package.
//
// C ABI around the REFERENCE'S OWN rasterizer (CudaRasterizer::Rasterizer::forward / backward / markVisible,
// diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:21-113) compiled for gfx950 from the reference's sources where they lie
// under /root/reference (oracle/refbuild/build_ref_hip.py: torch.utils.hipify on a temporary copy + hipcc; no reference text enters
// the repository).  What rasterize_points.cu:28-270 does with torch tensors -- the three growable scratch buffers, the zero-filled
// outputs -- is done here with hipMalloc'ed buffers and raw device pointers, so that it can be driven through ctypes
// (oracle/ref_hip.py) next to the product's own C ABI: the same inputs, the same MI355X.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <functional>
#include "rasterizer.h"

namespace
{
	struct Growable
	{
		char* p = nullptr; size_t cap = 0;
		char* get(size_t n)
		{
			if (n > cap)
			{
				if (p) (void)hipFree(p);
				p = nullptr; cap = 0;
				const size_t want = n + n / 4 + 256;
				if (hipMalloc((void**)&p, want) != hipSuccess) return nullptr;
				cap = want;
			}
			return p;
		}
	};
	Growable g_geom, g_bin, g_img;   // rasterize_points.cu:28-34 (resizeFunctional): kept from the forward for the backward
}

struct refhip_scene   // device pointers (NULL = absent, as the reference's empty tensors: rasterize_points.cu:99-103) and scalars
{
	int32_t P, D, D_t, M, W, H;
	const float *background, *means3D, *shs, *colors_precomp, *flows, *opacities, *ts, *scales, *scales_t, *rotations, *rotations_r, *cov3D_precomp;
	const float *viewmatrix, *projmatrix, *campos;
	float scale_modifier, prefilter_var, tan_fovx, tan_fovy, timestamp, time_duration;
	int32_t rot_4d, gaussian_dim, force_sh_3d;
};

extern "C" const char* refhip_kind(void) { return "reference (hipified, gfx950)"; }

// rasterize_points.cu:36-149: outputs [3,H,W], [2,H,W], [1,H,W], [1,H,W], [P]; out_means3D [P,3] (a copy of means3D on entry: :89).
// Returns num_rendered, or < 0.
extern "C" int refhip_forward(const refhip_scene* s, float* out_means3D, float* out_color, float* out_flow, float* out_depth, float* out_T, int* radii)
{
	if (s->P == 0) return 0;
	std::function<char*(size_t)> geomF = [](size_t n) { return g_geom.get(n); };
	std::function<char*(size_t)> binF = [](size_t n) { return g_bin.get(n); };
	std::function<char*(size_t)> imgF = [](size_t n) { return g_img.get(n); };
	return CudaRasterizer::Rasterizer::forward(geomF, binF, imgF, s->P, s->D, s->D_t, s->M, s->background, s->W, s->H, s->means3D, out_means3D, s->shs,
	                                           s->colors_precomp, s->flows, s->opacities, s->ts, s->scales, s->scales_t, s->scale_modifier, s->rotations,
	                                           s->rotations_r, s->cov3D_precomp, s->prefilter_var, s->viewmatrix, s->projmatrix, s->campos, s->timestamp,
	                                           s->time_duration, s->rot_4d != 0, s->gaussian_dim, s->force_sh_3d != 0, s->tan_fovx, s->tan_fovy, false,
	                                           out_color, out_flow, out_depth, out_T, radii, false);
}

// rasterize_points.cu:151-270: every gradient array zero on entry (the caller's, as torch::zeros there); uses the forward's buffers
extern "C" int refhip_backward(const refhip_scene* s, int num_rendered, const float* out_means3D, const int* radii, const float* dL_dpix,
                               const float* dL_ddepth, const float* dL_dmask, const float* dL_dflow, float* dL_dmean2D, float* dL_dconic,
                               float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dflows, float* dL_dts,
                               float* dL_dscale, float* dL_dscale_t, float* dL_drot, float* dL_drot_r)
{
	if (s->P == 0) return 0;
	CudaRasterizer::Rasterizer::backward(s->P, s->D, s->D_t, s->M, num_rendered, s->background, s->W, s->H, out_means3D, s->shs, s->colors_precomp, s->flows,
	                                     s->opacities, s->ts, s->scales, s->scales_t, s->scale_modifier, s->rotations, s->rotations_r, s->cov3D_precomp,
	                                     s->prefilter_var, s->viewmatrix, s->projmatrix, s->campos, s->timestamp, s->time_duration, s->rot_4d != 0,
	                                     s->gaussian_dim, s->force_sh_3d != 0, s->tan_fovx, s->tan_fovy, radii, g_geom.p, g_bin.p, g_img.p, dL_dpix, dL_ddepth,
	                                     dL_dmask, dL_dflow, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dflows, dL_dts,
	                                     dL_dscale, dL_dscale_t, dL_drot, dL_drot_r, false);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" void refhip_free(void)
{
	for (Growable* g : { &g_geom, &g_bin, &g_img }) { if (g->p) (void)hipFree(g->p); g->p = nullptr; g->cap = 0; }
}

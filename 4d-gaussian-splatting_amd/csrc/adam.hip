// adam.hip -- fused Adam over the flat parameter bucket (gfx950).
//
// The optimizer step that closes every training iteration (reference train.py:247-249,
// torch.optim.Adam(lr=0, eps=1e-15) over 9 parameter groups, scene/gaussian_model.py:336-353).
// Parameters, gradients and both moments live in four flat fp32 buffers (161 floats per Gaussian
// at M = 48), so the whole step is ONE streaming pass: read p, g, m, v, write p, m, v
// (28 B / element -> 1.35 GB at 300 k Gaussians; PyTorch's multi-tensor path takes 11 launches).
// Learning rates come from a small segment table; a segment may give its first `head` elements
// of every `period` a different rate (SH DC vs. rest inside one [P, M, 3] tensor, so the
// reference's cat(features_dc, features_rest) disappears from the step).
// Same arithmetic as torch.optim.Adam (no amsgrad / weight decay):
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include "fdgs_common.h"

namespace fdgs
{
	constexpr int ADAM_MAX_SEG = 16;
	struct AdamSegs
	{
		long long begin[ADAM_MAX_SEG], end[ADAM_MAX_SEG];
		float lr[ADAM_MAX_SEG], lr_head[ADAM_MAX_SEG];
		int period[ADAM_MAX_SEG], head[ADAM_MAX_SEG];
		int n;
	};

	__device__ __forceinline__ float seg_lr(const AdamSegs& s, long long i)
	{
		float lr = 0.f;
#pragma unroll 1
		for (int k = 0; k < s.n; k++)
			if (i >= s.begin[k] && i < s.end[k])
			{
				lr = s.lr[k];
				if (s.period[k] > 0 && (int)((i - s.begin[k]) % s.period[k]) < s.head[k]) lr = s.lr_head[k];
			}
		return lr;
	}

	__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
	                                                 float* __restrict__ m, float* __restrict__ v, long long n,
	                                                 const AdamSegs segs, float b1, float b2, float eps,
	                                                 float inv_bc1, float inv_sqrt_bc2)
	{
		const long long nvec = n / 4;
		const long long stride = (long long)gridDim.x * blockDim.x;
		for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nvec; q += stride)
		{
			const float4 gg = reinterpret_cast<const float4*>(g)[q];
			float4 mm = reinterpret_cast<float4*>(m)[q], vv = reinterpret_cast<float4*>(v)[q], pp = reinterpret_cast<float4*>(p)[q];
			const float ge[4] = { gg.x, gg.y, gg.z, gg.w };
			float me[4] = { mm.x, mm.y, mm.z, mm.w }, ve[4] = { vv.x, vv.y, vv.z, vv.w }, pe[4] = { pp.x, pp.y, pp.z, pp.w };
#pragma unroll
			for (int e = 0; e < 4; e++)
			{
				const float lr = seg_lr(segs, 4 * q + e);
				adam_update(pe[e], me[e], ve[e], ge[e], lr * inv_bc1, b1, b2, eps, inv_sqrt_bc2);
			}
			reinterpret_cast<float4*>(m)[q] = make_float4(me[0], me[1], me[2], me[3]);
			reinterpret_cast<float4*>(v)[q] = make_float4(ve[0], ve[1], ve[2], ve[3]);
			reinterpret_cast<float4*>(p)[q] = make_float4(pe[0], pe[1], pe[2], pe[3]);
		}
		// tail (n not a multiple of 4)
		for (long long i = nvec * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
		{
			const float lr = seg_lr(segs, i);
			float pi = p[i], mi = m[i], vi = v[i];
			adam_update(pi, mi, vi, g[i], lr * inv_bc1, b1, b2, eps, inv_sqrt_bc2);
			m[i] = mi; v[i] = vi; p[i] = pi;
		}
	}
}

namespace fdgs
{
	// The same update with ONE segment per blockIdx.y (the usual case: the segments tile [0, n) -- the launcher checks): the learning
	// rate is a per-workgroup constant (two, for a segment with a DC head), so the per-element search through the segment table --
	// a dependent chain of scalar loads per element and segment in adam_kernel, which held the geometry bucket's step (17 floats per
	// Gaussian, 7 segments) at 2.5 TB/s -- disappears.  Same arithmetic per element (adam_update with lr * inv_bc1): bit-identical.
	__global__ void __launch_bounds__(256) adam_seg_kernel(float* __restrict__ p, const float* __restrict__ g,
	                                                     float* __restrict__ m, float* __restrict__ v, long long n,
	                                                     const AdamSegs segs, float b1, float b2, float eps,
	                                                     float inv_bc1, float inv_sqrt_bc2)
	{
		const int k = blockIdx.y;
		const long long sb = segs.begin[k];
		const long long b = sb > 0 ? sb : 0, e = segs.end[k] < n ? segs.end[k] : n;
		if (b >= e) return;
		const float lr = segs.lr[k] * inv_bc1, lr_head = segs.lr_head[k] * inv_bc1;
		const int period = segs.period[k], head = segs.head[k];
		const long long ab = (b + 3) & ~3ll, ae = e & ~3ll;   // the float4-aligned body (the buffers are 16-byte aligned at element 0)
		const long long stride = (long long)gridDim.x * blockDim.x;
		const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
		if (ab < ae)
		{
			for (long long q = ab / 4 + t; q < ae / 4; q += stride)
			{
				const float4 gg = reinterpret_cast<const float4*>(g)[q];
				float4 mm = reinterpret_cast<float4*>(m)[q], vv = reinterpret_cast<float4*>(v)[q], pp = reinterpret_cast<float4*>(p)[q];
				const float ge[4] = { gg.x, gg.y, gg.z, gg.w };
				float me[4] = { mm.x, mm.y, mm.z, mm.w }, ve[4] = { vv.x, vv.y, vv.z, vv.w }, pe[4] = { pp.x, pp.y, pp.z, pp.w };
				if (period > 0)
				{
					const int ph = (int)((4 * q - sb) % period);
#pragma unroll
					for (int c = 0; c < 4; c++)
					{
						int r = ph + c; while (r >= period) r -= period;   // (period may be < 4: segments of the public fdgs_adam_step)
						adam_update(pe[c], me[c], ve[c], ge[c], r < head ? lr_head : lr, b1, b2, eps, inv_sqrt_bc2);
					}
				}
				else
				{
#pragma unroll
					for (int c = 0; c < 4; c++) adam_update(pe[c], me[c], ve[c], ge[c], lr, b1, b2, eps, inv_sqrt_bc2);
				}
				reinterpret_cast<float4*>(m)[q] = make_float4(me[0], me[1], me[2], me[3]);
				reinterpret_cast<float4*>(v)[q] = make_float4(ve[0], ve[1], ve[2], ve[3]);
				reinterpret_cast<float4*>(p)[q] = make_float4(pe[0], pe[1], pe[2], pe[3]);
			}
		}
		// the (at most 3 + 3) elements in front of and behind the aligned body; a segment shorter than a float4: all of it
		if (t < 8)
		{
			const long long he = ab < e ? ab : e;                       // head: [b, he)
			const long long tb = ae > he ? ae : he;                     // tail: [tb, e)
			const long long i = t < 4 ? b + t : tb + (t - 4);
			if ((t < 4 && i < he) || (t >= 4 && i < e))
			{
				float l = lr;
				if (period > 0 && (int)((i - sb) % period) < head) l = lr_head;
				float pi = p[i], mi = m[i], vi = v[i];
				adam_update(pi, mi, vi, g[i], l, b1, b2, eps, inv_sqrt_bc2);
				m[i] = mi; v[i] = vi; p[i] = pi;
			}
		}
	}
}

extern "C" int fdgs_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                              const fdgs_adam_segment* segments, int32_t num_segments,
                              float beta1, float beta2, float eps, int32_t step, void* stream)
{
	using namespace fdgs;
	if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || !segments || num_segments <= 0 || num_segments > ADAM_MAX_SEG || step < 1)
		return FDGS_ERR_INVALID_ARG;
	if (n == 0) return FDGS_OK;
	AdamSegs s;
	s.n = num_segments;
	for (int k = 0; k < num_segments; k++)
	{
		s.begin[k] = segments[k].begin; s.end[k] = segments[k].end;
		s.lr[k] = segments[k].lr; s.lr_head[k] = segments[k].lr_head;
		s.period[k] = segments[k].period; s.head[k] = segments[k].head;
	}
	const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
	// Do the segments, clipped to [0, n), tile [0, n) -- every element in exactly one of them?  Then: one segment per blockIdx.y.
	// (Otherwise the general kernel: an element outside every segment keeps lr = 0 while its moments still update, an element
	// inside two takes the later one's rate.)
	int order[ADAM_MAX_SEG], m = 0;
	long long longest = 0;
	for (int k = 0; k < num_segments; k++)
	{
		const long long b = std::max<long long>(s.begin[k], 0), e = std::min<long long>(s.end[k], n);
		if (b < e) { order[m++] = k; longest = std::max(longest, e - b); }
	}
	std::sort(order, order + m, [&](int a, int b) { return s.begin[a] < s.begin[b]; });
	long long at = 0;
	bool tiles = m > 0, aligned = ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
	                                reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0;
	for (int i = 0; i < m && tiles; i++)
	{
		const long long b = std::max<long long>(s.begin[order[i]], 0), e = std::min<long long>(s.end[order[i]], n);
		if (b != at) tiles = false;
		at = e;
	}
	tiles = tiles && at == n;
	static const bool force_general = []() { const char* e = getenv("FDGS_ADAM_GENERAL"); return e && e[0] == '1'; }();   // A/B timing switch
	if (tiles && aligned && !force_general)
	{
		const int bx = (int)std::min<long long>((longest / 4 + 255) / 256 + 1, 256 * 8);
		hipLaunchKernelGGL(adam_seg_kernel, dim3(bx, num_segments), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
		                   (long long)n, s, beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)));
		return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
	}
	const int blocks = (int)std::min<long long>((n / 4 + 255) / 256 + 1, 256 * 16);
	hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
	                   (long long)n, s, beta1, beta2, eps, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)));
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_adam_step_sh(float* params, float* exp_avg, float* exp_avg_sq, float* dL_dsh,
                                 int32_t P, int32_t D, int32_t D_t, int32_t M, int32_t gaussian_dim, int32_t force_sh_3d,
                                 int32_t analytic_sh_grad, int32_t num_views, const float* stages,
                                 float lr, float lr_dc, float beta1, float beta2, float eps, int32_t step, void* stream)
{
	using namespace fdgs;
	if (!params || !exp_avg || !exp_avg_sq || !stages || P < 0 || M < 0 || num_views < 1 || step < 1) return FDGS_ERR_INVALID_ARG;
	if (P == 0 || M == 0) return FDGS_OK;
	const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
	const float inv_bc1 = (float)(1.0 / bc1);
	AdamScalars k;
	k.lr_bc1 = lr * inv_bc1; k.lr_head_bc1 = lr_dc * inv_bc1;
	k.b1 = beta1; k.b2 = beta2; k.eps = eps; k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
	const hipError_t e = launch_sh_adam(P, D, D_t, M, gaussian_dim, force_sh_3d, analytic_sh_grad, num_views, stages, params, exp_avg,
	                                    exp_avg_sq, dL_dsh, k, (hipStream_t)stream);
	if (e == hipErrorInvalidValue) return FDGS_ERR_INVALID_ARG;   // rows not a whole number of float4s, or arrays not 16-byte aligned
	return e == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

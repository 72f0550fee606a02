// densify.hip -- densification / pruning of the flat-bucket Gaussian model (gfx950).
//
// every `densification_interval` iterations the reference clones small high-gradient
// Gaussians, splits large ones into N samples, and prunes transparent / oversized ones
// (scene/gaussian_model.py:391-610).  It does so with ~60 boolean-mask gathers and torch.cat calls over 9 parameter
// tensors and their two Adam moments each.  Here the model is ONE flat buffer per role (parameters, exp_avg,
// exp_avg_sq; segments [P,3] [P,M,3] [P,1] ... back to back), so the whole re-layout is
//   1. classify : one pass over the per-Gaussian statistics -> clone / split / prune flags;
//   2. (host: three nonzero() calls turn the flags into the source index of every surviving row, in the reference's
//      order: kept originals, clones, split children copy 1..N);
//   3. gather   : ONE kernel builds the three new flat buffers (parameters copied, moments copied for kept originals
//                 and zeroed for new points);
//   4. split    : the children's xyz / t / scaling rows are overwritten: sample in the parent's local frame,
//                 rotate by build_rotation[_4d], shift by the parent's mean; scaling / (0.8 N).
// Pure streaming work (~3 * 161 * 4 B read + written per Gaussian at M = 48); HBM-bound, no MFMA.
#include "fdgs_common.h"

namespace fdgs
{
	constexpr int DEN_MAX_SEG = 16;
	struct DenSegs
	{
		int n;
		int row[DEN_MAX_SEG];        // floats per Gaussian
		int first[DEN_MAX_SEG + 1];  // prefix sums of row[] (position inside one Gaussian's concatenated row)
	};

	__global__ void __launch_bounds__(256) densify_classify_kernel(
		int P, const float* __restrict__ grad_accum, const float* __restrict__ denom, const float* __restrict__ scaling_raw,
		const float* __restrict__ opacity_raw, const float* __restrict__ max_radii2D,
		float max_grad, float min_opacity, float small_limit, float big_limit, float max_screen_size, float inv_split,
		int has_screen, int prune_only, uint8_t* __restrict__ flags)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= P) return;
		const float s0 = expf(scaling_raw[3 * (size_t)i]), s1 = expf(scaling_raw[3 * (size_t)i + 1]), s2 = expf(scaling_raw[3 * (size_t)i + 2]);
		const float smax = fmaxf(fmaxf(s0, s1), s2);
		uint8_t f = 0;
		if (!prune_only)
		{
			float g = grad_accum[i] / denom[i];          // gaussian_model.py:586-587
			if (g != g) g = 0.0f;
			if (g >= max_grad) f |= (smax <= small_limit) ? FDGS_DENSIFY_CLONE : FDGS_DENSIFY_SPLIT; // :547-549, :492-494
		}
		const float opacity = 1.0f / (1.0f + expf(-opacity_raw[i]));
		bool prune = opacity < min_opacity, prune_child = prune;     // :598
		if (has_screen)
		{
			// max_radii2D is reset by densification_postfix before this test unless prune_only (:481, :600)
			if (prune_only && max_radii2D[i] > max_screen_size) prune = true;
			if (smax > big_limit) prune = true;                        // :601
			// a child's scaling is log(s / (0.8 N)) and is exponentiated again by get_scaling
			const float c0 = expf(logf(s0 * inv_split)), c1 = expf(logf(s1 * inv_split)), c2 = expf(logf(s2 * inv_split));
			if (fmaxf(fmaxf(c0, c1), c2) > big_limit) prune_child = true;
		}
		if (prune) f |= FDGS_DENSIFY_PRUNE;
		if (prune_child) f |= FDGS_DENSIFY_PRUNE_CHILD;
		flags[i] = f;
	}

	// element e of new Gaussian j (all segments of one Gaussian numbered consecutively 0 .. first[n]-1)
	__global__ void __launch_bounds__(256) densify_gather_kernel(
		const DenSegs segs, long long P_old, long long P_new, const int32_t* __restrict__ src, const uint8_t* __restrict__ kind,
		const float* __restrict__ op, const float* __restrict__ om, const float* __restrict__ ov,
		float* __restrict__ np_, float* __restrict__ nm, float* __restrict__ nv)
	{
		const int per = segs.first[segs.n];
		const long long total = P_new * per;
		for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x)
		{
			const long long j = q / per;
			const int e = (int)(q - j * per);
			int k = 0;
#pragma unroll 1
			while (k + 1 < segs.n && e >= segs.first[k + 1]) k++;
			const int within = e - segs.first[k];
			const long long s = src[j];
			const long long o = (long long)segs.first[k] * P_old + s * segs.row[k] + within;
			const long long d = (long long)segs.first[k] * P_new + j * segs.row[k] + within;
			np_[d] = op[o];
			const bool keep_state = kind[j] == 0;   // new points start with zero moments (gaussian_model.py:441-442)
			nm[d] = keep_state ? om[o] : 0.0f;
			nv[d] = keep_state ? ov[o] : 0.0f;
		}
	}

	struct SplitArgs
	{
		int n, rot_4d, gaussian_dim;
		float inv_split;
		const int32_t* parent;      // [n] index into the OLD arrays
		const float* samples;       // [n, 4] (rot_4d) or [n, 3]: draws of N(0, std) in the parent's local frame
		const float* samples_t;     // [n] (gaussian_dim 4 without rot_4d) or NULL
		const float *xyz, *t, *scaling, *scaling_t, *rot, *rot_r;   // OLD raw parameters
		float *nxyz, *nt, *nscaling, *nscaling_t;                   // rows of the children in the NEW arrays
	};

	__global__ void __launch_bounds__(256) densify_split_kernel(const SplitArgs a)
	{
		const int c = blockIdx.x * blockDim.x + threadIdx.x;
		if (c >= a.n) return;
		const size_t p = (size_t)a.parent[c];
		// scaling_inverse_activation(get_scaling / (0.8 N)), gaussian_model.py:497
		for (int k = 0; k < 3; k++) a.nscaling[3 * (size_t)c + k] = logf(expf(a.scaling[3 * p + k]) * a.inv_split);
		if (a.gaussian_dim == 4) a.nscaling_t[c] = logf(expf(a.scaling_t[p]) * a.inv_split);
		if (a.rot_4d)
		{
			// build_rotation_4d (utils/general_utils.py:113-133): A = (M_l M_r) flipped in both axes
			float l[4], r[4];
			float nl = 0.f, nr = 0.f;
			for (int k = 0; k < 4; k++) { l[k] = a.rot[4 * p + k]; r[k] = a.rot_r[4 * p + k]; nl += l[k] * l[k]; nr += r[k] * r[k]; }
			nl = sqrtf(nl); nr = sqrtf(nr);
			for (int k = 0; k < 4; k++) { l[k] /= nl; r[k] /= nr; }
			const float ql[4][4] = { { l[0], -l[1], -l[2], -l[3] }, { l[1], l[0], -l[3], l[2] }, { l[2], l[3], l[0], -l[1] }, { l[3], -l[2], l[1], l[0] } };
			const float qr[4][4] = { { r[0], r[1], r[2], r[3] }, { -r[1], r[0], -r[3], r[2] }, { -r[2], r[3], r[0], -r[1] }, { -r[3], -r[2], r[1], r[0] } };
			float s[4];
			for (int k = 0; k < 4; k++) s[k] = a.samples[4 * (size_t)c + k];
			float out[4];
			for (int i = 0; i < 4; i++)
			{
				float acc = 0.f;
				for (int j = 0; j < 4; j++)
				{
					float m = 0.f;   // A[i][j] = (M_l M_r)[3-i][3-j]
					for (int k = 0; k < 4; k++) m += ql[3 - i][k] * qr[k][3 - j];
					acc += m * s[j];
				}
				out[i] = acc;
			}
			for (int k = 0; k < 3; k++) a.nxyz[3 * (size_t)c + k] = out[k] + a.xyz[3 * p + k];
			a.nt[c] = out[3] + a.t[p];
		}
		else
		{
			// build_rotation (utils/general_utils.py:79-100)
			float q[4], nq = 0.f;
			for (int k = 0; k < 4; k++) { q[k] = a.rot[4 * p + k]; nq += q[k] * q[k]; }
			nq = sqrtf(nq);
			for (int k = 0; k < 4; k++) q[k] /= nq;
			const float w = q[0], x = q[1], y = q[2], z = q[3];
			const float R[3][3] = { { 1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y) },
			                        { 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x) },
			                        { 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y) } };
			for (int i = 0; i < 3; i++)
			{
				float acc = 0.f;
				for (int j = 0; j < 3; j++) acc += R[i][j] * a.samples[3 * (size_t)c + j];
				a.nxyz[3 * (size_t)c + i] = acc + a.xyz[3 * p + i];
			}
			if (a.gaussian_dim == 4) a.nt[c] = a.samples_t[c] + a.t[p];
		}
	}
}

namespace fdgs
{
	constexpr int STATS_MAX_VIEWS = 16;
	struct StatsViews { const int32_t* radii[STATS_MAX_VIEWS]; const float* grad[STATS_MAX_VIEWS]; int n; };

	// train.py:164-172 for this rank's views: visibility count, max radius, sum of ||dL/dmean2D.xy||
	__global__ void __launch_bounds__(256) densify_stats_local_kernel(int P, const StatsViews v, float* __restrict__ count,
	                                                                  float* __restrict__ pgrad, float* __restrict__ radii_max)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= P) return;
		float c = 0.f, g = 0.f;
		int r = 0;
		for (int k = 0; k < v.n; k++)
		{
			const int rk = v.radii[k][i];
			c += rk > 0 ? 1.f : 0.f;
			r = max(r, rk);
			const float gx = v.grad[k][3 * (size_t)i], gy = v.grad[k][3 * (size_t)i + 1];
			g += sqrtf(gx * gx + gy * gy);
		}
		count[i] = c; pgrad[i] = g; radii_max[i] = (float)r;
	}

	// train.py:173-184, 229-236 + gaussian_model.py:637-642 on the (all-reduced) per-Gaussian sums
	__global__ void __launch_bounds__(256) densify_stats_apply_kernel(int P, const float* __restrict__ count, const float* __restrict__ pgrad,
	                                                                  const float* __restrict__ radii_max, const float* __restrict__ t_grad,
	                                                                  float global_batch, float* __restrict__ xyz_acc, float* __restrict__ t_acc,
	                                                                  float* __restrict__ denom, float* __restrict__ max_radii2D)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= P) return;
		const float c = count[i];
		if (!(c > 0.f)) return;
		max_radii2D[i] = fmaxf(max_radii2D[i], radii_max[i]);
		xyz_acc[i] += pgrad[i] * global_batch / c;
		denom[i] += 1.0f;
		if (t_grad) t_acc[i] += t_grad[i] * global_batch / c;
	}
}

extern "C" int fdgs_densify_stats_local(int32_t P, int32_t num_views, const int32_t* const* radii, const float* const* viewspace_grad,
                                        float* count, float* pgrad, float* radii_max, void* stream)
{
	using namespace fdgs;
	if (P < 0 || num_views < 1 || num_views > STATS_MAX_VIEWS || !radii || !viewspace_grad) return FDGS_ERR_INVALID_ARG;
	if (P == 0) return FDGS_OK;
	if (!count || !pgrad || !radii_max) return FDGS_ERR_INVALID_ARG;
	StatsViews v;
	v.n = num_views;
	for (int k = 0; k < num_views; k++)
	{
		if (!radii[k] || !viewspace_grad[k]) return FDGS_ERR_INVALID_ARG;
		v.radii[k] = radii[k]; v.grad[k] = viewspace_grad[k];
	}
	hipLaunchKernelGGL(densify_stats_local_kernel, dim3(div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, P, v, count, pgrad, radii_max);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_densify_stats_apply(int32_t P, const float* count, const float* pgrad, const float* radii_max, const float* t_grad,
                                        float global_batch, float* xyz_gradient_accum, float* t_gradient_accum, float* denom,
                                        float* max_radii2D, void* stream)
{
	using namespace fdgs;
	if (P < 0) return FDGS_ERR_INVALID_ARG;
	if (P == 0) return FDGS_OK;
	if (!count || !pgrad || !radii_max || !xyz_gradient_accum || !denom || !max_radii2D || (t_grad && !t_gradient_accum)) return FDGS_ERR_INVALID_ARG;
	hipLaunchKernelGGL(densify_stats_apply_kernel, dim3(div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, P, count, pgrad, radii_max, t_grad,
	                   global_batch, xyz_gradient_accum, t_gradient_accum, denom, max_radii2D);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_densify_classify(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* scaling_raw,
                                     const float* opacity_raw, const float* max_radii2D, float max_grad, float min_opacity,
                                     float extent, float max_screen_size, float percent_dense, int32_t N, int32_t prune_only,
                                     uint8_t* flags, void* stream)
{
	using namespace fdgs;
	if (P < 0 || N < 1) return FDGS_ERR_INVALID_ARG;
	if (P == 0) return FDGS_OK;
	if (!xyz_gradient_accum || !denom || !scaling_raw || !opacity_raw || !max_radii2D || !flags) return FDGS_ERR_INVALID_ARG;
	const int has_screen = max_screen_size > 0.0f ? 1 : 0;   // `if max_screen_size:` (None or 0 -> no size test)
	hipLaunchKernelGGL(densify_classify_kernel, dim3(div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, P, xyz_gradient_accum, denom,
	                   scaling_raw, opacity_raw, max_radii2D, max_grad, min_opacity, (float)((double)percent_dense * (double)extent),
	                   (float)(0.1 * (double)extent), max_screen_size, (float)(1.0 / (0.8 * (double)N)), has_screen, prune_only, flags);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_densify_gather(int32_t num_segments, const int32_t* row_floats, int64_t P_old, int64_t P_new,
                                   const int32_t* src, const uint8_t* kind, const float* old_params, const float* old_exp_avg,
                                   const float* old_exp_avg_sq, float* new_params, float* new_exp_avg, float* new_exp_avg_sq, void* stream)
{
	using namespace fdgs;
	if (num_segments <= 0 || num_segments > DEN_MAX_SEG || !row_floats || P_old < 0 || P_new < 0) return FDGS_ERR_INVALID_ARG;
	if (P_new == 0) return FDGS_OK;
	if (!src || !kind || !old_params || !old_exp_avg || !old_exp_avg_sq || !new_params || !new_exp_avg || !new_exp_avg_sq) return FDGS_ERR_INVALID_ARG;
	DenSegs s;
	s.n = num_segments;
	s.first[0] = 0;
	for (int k = 0; k < num_segments; k++)
	{
		if (row_floats[k] <= 0) return FDGS_ERR_INVALID_ARG;
		s.row[k] = row_floats[k];
		s.first[k + 1] = s.first[k] + row_floats[k];
	}
	const long long total = (long long)P_new * s.first[num_segments];
	const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 32);
	hipLaunchKernelGGL(densify_gather_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, (long long)P_old, (long long)P_new, src, kind,
	                   old_params, old_exp_avg, old_exp_avg_sq, new_params, new_exp_avg, new_exp_avg_sq);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_densify_split(int32_t n_children, int32_t N, int32_t rot_4d, int32_t gaussian_dim, const int32_t* parent,
                                  const float* samples, const float* samples_t, const float* xyz, const float* t, const float* scaling,
                                  const float* scaling_t, const float* rotation, const float* rotation_r,
                                  float* new_xyz, float* new_t, float* new_scaling, float* new_scaling_t, void* stream)
{
	using namespace fdgs;
	if (n_children < 0 || N < 1) return FDGS_ERR_INVALID_ARG;
	if (n_children == 0) return FDGS_OK;
	if (!parent || !samples || !xyz || !scaling || !rotation || !new_xyz || !new_scaling) return FDGS_ERR_INVALID_ARG;
	if (gaussian_dim == 4 && (!t || !scaling_t || !new_t || !new_scaling_t)) return FDGS_ERR_INVALID_ARG;
	if (rot_4d && (!rotation_r || gaussian_dim != 4)) return FDGS_ERR_INVALID_ARG;
	if (!rot_4d && gaussian_dim == 4 && !samples_t) return FDGS_ERR_INVALID_ARG;
	SplitArgs a;
	a.n = n_children; a.rot_4d = rot_4d; a.gaussian_dim = gaussian_dim; a.inv_split = (float)(1.0 / (0.8 * (double)N));
	a.parent = parent; a.samples = samples; a.samples_t = samples_t;
	a.xyz = xyz; a.t = t; a.scaling = scaling; a.scaling_t = scaling_t; a.rot = rotation; a.rot_r = rotation_r;
	a.nxyz = new_xyz; a.nt = new_t; a.nscaling = new_scaling; a.nscaling_t = new_scaling_t;
	hipLaunchKernelGGL(densify_split_kernel, dim3(div_up(n_children, 256)), dim3(256), 0, (hipStream_t)stream, a);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

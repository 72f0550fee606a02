// preprocess_bwd.hip -- fused per-Gaussian backward (gfx950).
//
// One kernel does what the reference does in two (computeCov2DCUDA backward.cu:486-617,
// then preprocessCUDA backward.cu:839-923 with computeCov3D :621-684 and
// computeCov3D_conditional :689-834): the conic gradient -> cov2D -> cov3D + view-space mean
// gradient, the projection Jacobian, and the covariance backward to scales / scale_t /
// rotations / rotation_r / t, including the marginal-opacity chain.  The SH / 4D-SH backward
// (80 % of the bytes) runs before it as its own coalesced kernel (sh_bwd.hip) and hands over
// four floats per Gaussian.  Fusing keeps dL_dcov3D
// and dL_dmean3D in registers between the two halves, and every output is
// written for every Gaussian (zeros for culled ones), so the caller never
// pre-zeroes the ~180 floats per Gaussian the reference memsets
// (rasterize_points.cu:201-213).
//
// Bug-compatible with the reference backward:
//   Q1  dL_dsh[1] uses the degree-0 basis in the 4D path      (backward.cu:190)
//   Q2  d cos / dt has the wrong sign                          (backward.cu:303,384)
//   Q3  the k=2 time term overwrites the k=1 term in dRGBdt    (backward.cu:403)
//   Q4  SH view direction from the SHIFTED mean (forward uses the input mean)
//   Q5  the whole dL_dmean (incl. SH part) is fed back as dL_d(delta_mean)
//   Q6  no marginal-opacity backward for gaussian_dim == 4 without rot_4d
//   Q7  cov12 read as Sigma[3][k] here, Sigma[k][3] in the forward
#pragma clang fp contract(off)
#include "fdgs_common.h"
#include "fdgs_math.h"

namespace fdgs
{
	struct BwdArgs
	{
		int P, D, D_t, M;
		const float *shs, *opacities, *ts, *scales, *scales_t, *rotations, *rotations_r, *cov3D_precomp;
		const float *viewmatrix, *projmatrix, *campos;
		float scale_modifier, prefilter_var, tan_fovx, tan_fovy, focal_x, focal_y, timestamp, time_duration;
		int rot_4d, gaussian_dim, force_sh_3d, raw, accum;
		const int32_t* radii; const float* means; /* out_means3D */
		const float* cov3D; const uint8_t* clamped;
		float* gacc; /* packed blend-backward accumulators [P,16], see blend_bwd.hip */
		const float4* records; /* the forward's packed blend records (conic, effective opacity): geometry buffer */
		float half_w, half_h;
		int rezero; /* leave the record zero for the next backward (fdgs_backward_out.grad_accum_clean) */
		float *dL_dmean2D, *dL_dcolor, *dL_dflows;
		float *dL_dopacity, *dL_dmeans, *dL_dcov3D, *dL_dts, *dL_dscale, *dL_dscale_t, *dL_drot, *dL_drot_r;
		// fdgs_backward_out.adam: the Adam step of the geometry parameters with the gradient completed here
		int adam;
		const float* means_in;                   // the means3D PARAMETER (a.means is the forward's shifted out_means3D)
		float *ad_flat, *ad_m, *ad_v;
		float ad_lr[7];                          // x 1 / (1 - beta1^step): means3D, opacities, ts, scales, scales_t, rotations, rotations_r
		float ad_b1, ad_b2, ad_eps, ad_inv_sqrt_bc2;
	};

	__device__ __forceinline__ float3 b_ld3(const float* p, size_t i) { return make_float3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
	__device__ __forceinline__ void b_st3(float* p, size_t i, float3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
	__device__ __forceinline__ float3 b_add(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
	__device__ __forceinline__ float3 b_scl(float s, float3 a) { return make_float3(s * a.x, s * a.y, s * a.z); }
	__device__ __forceinline__ float b_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

	// auxiliary.h:108-118
	__device__ __forceinline__ float3 dnormvdv(float3 v, float3 dv)
	{
		const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
		const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
		float3 r;
		r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
		r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
		r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
		return r;
	}

	__device__ __forceinline__ void preprocess_bwd_body(const BwdArgs& a)
	{
		const int idx = blockIdx.x * blockDim.x + threadIdx.x;
		if (idx >= a.P) return;
		const bool visible = a.radii[idx] > 0; // backward.cu:499, 873-875

		float3 dmean = make_float3(0, 0, 0);
		float dcov[6] = { 0, 0, 0, 0, 0, 0 };
		float dts = 0.f;
		float3 dscale = make_float3(0, 0, 0);
		float dscale_t = 0.f;
		float4 drot = make_float4(0, 0, 0, 0), drot_r = make_float4(0, 0, 0, 0);
		// unpack the accumulator record (blend_bwd.hip): colour 0-2, depth 3, flow 4-5, mean2D x,y 6-7,
		// conic xx 8, yy 9, xy 10, opacity 11, SH-backward mean/time 12-15
		float4* rec = reinterpret_cast<float4*>(a.gacc + (size_t)idx * GRAD_ACC_WORDS);
		const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
		if (a.rezero)
		{
			// only a record that holds something: most records are all zero on entry (culled Gaussians, and the visible ones no pixel
			// took a contribution from: 59 % of all on C3, 69 % on C5), and 64 bytes of zeros over 64 bytes of zeros are 64 bytes of traffic.
			// Bit test: -0.0f and NaN count as "something".
			const uint4 *u0 = reinterpret_cast<const uint4*>(&r0), *u1 = reinterpret_cast<const uint4*>(&r1), *u2 = reinterpret_cast<const uint4*>(&r2),
			            *u3 = reinterpret_cast<const uint4*>(&r3);
			const uint32_t any = (u0->x | u0->y | u0->z | u0->w) | (u1->x | u1->y | u1->z | u1->w) | (u2->x | u2->y | u2->z | u2->w) | (u3->x | u3->y | u3->z | u3->w);
			if (any != 0u) { const float4 z = make_float4(0.f, 0.f, 0.f, 0.f); rec[0] = z; rec[1] = z; rec[2] = z; rec[3] = z; }
		}
		const float3 g_color = make_float3(r0.x, r0.y, r0.z);
		const float2 g_flow = make_float2(r1.x, r1.y);
		// words 6-11 hold the pixel sums of q d^n (q = G dL/dalpha, d = mean2D - pixel); with the conic (A, B, C) and the
		// effective opacity o of the forward's blend record (backward.cu:1107-1133):
		//   dL/dmean2D = -o (W/2, H/2) * (A Mx + B My, B Mx + C My);  dL/dconic = -o/2 (Mxx, Mxy, Myy);  dL/do = M0
		float3 g_mean2D = make_float3(0.f, 0.f, r0.w), g_conic = make_float3(0.f, 0.f, 0.f);
		if (visible)   // culled Gaussians have no blend record (and all-zero moments)
		{
			const float4 ra = a.records[3 * (size_t)idx + 0], rb = a.records[3 * (size_t)idx + 1];
			const float cA = ra.z, cB = ra.w, cC = rb.x, o_eff = rb.y;
			const float Mx = r1.z, My = r1.w, Mxx = r2.x, Myy = r2.y, Mxy = r2.z;
			g_mean2D.x = -o_eff * a.half_w * (cA * Mx + cB * My);
			g_mean2D.y = -o_eff * a.half_h * (cB * Mx + cC * My);
			g_conic = make_float3(-0.5f * o_eff * Mxx, -0.5f * o_eff * Mxy, -0.5f * o_eff * Myy);
		}
		float g_opacity = r2.w;

		if (visible)
		{
			const float3 mean = b_ld3(a.means, idx);
			const float* cov3D = (a.cov3D_precomp ? a.cov3D_precomp : a.cov3D) + 6 * (size_t)idx;
			float c3[6];
#pragma unroll
			for (int k = 0; k < 6; k++) c3[k] = cov3D[k];

			// ---------------- cov2D backward (backward.cu:486-617) ----------------
			{
				const float dcx = g_conic.x, dcy = g_conic.y, dcz = g_conic.z;
				const Cov2D p = project_cov(mean, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, c3, a.viewmatrix);
				const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
				const float x_grad_mul = (p.txtz < -limx || p.txtz > limx) ? 0.f : 1.f;
				const float y_grad_mul = (p.tytz < -limy || p.tytz > limy) ? 0.f : 1.f;
				const float ca = p.a + 0.3f, cb = p.b, cc = p.c + 0.3f;
				const float denom = ca * cc - cb * cb;
				float dL_da = 0, dL_db = 0, dL_dc = 0;
				const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
#define TT(i, j) p.T.c[i][j]
#define VV(i, j) p.Vrk.c[i][j]
#define WW(i, j) p.W.c[i][j]
				if (denom2inv != 0)
				{
					dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
					dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
					dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
					dcov[0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
					dcov[3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
					dcov[5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
					dcov[1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
					dcov[2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
					dcov[4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
				}
				const float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da + (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
				const float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da + (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
				const float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da + (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
				const float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc + (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
				const float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc + (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
				const float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc + (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
				const float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
				const float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
				const float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
				const float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef TT
#undef VV
#undef WW
				const float tz = 1.f / p.t.z, tz2 = tz * tz, tz3 = tz2 * tz;
				const float h_x = a.focal_x, h_y = a.focal_y;
				const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
				const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
				const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * p.t.x) * tz3 * dL_dJ02 + (2 * h_y * p.t.y) * tz3 * dL_dJ12;
				const float vz = dL_dtz + g_mean2D.z; // Q10: depth-gradient carrier
				const float* m = a.viewmatrix; // transformVec4x3Transpose, auxiliary.h:90-98
				dmean.x = m[0] * dL_dtx + m[1] * dL_dty + m[2] * vz;
				dmean.y = m[4] * dL_dtx + m[5] * dL_dty + m[6] * vz;
				dmean.z = m[8] * dL_dtx + m[9] * dL_dty + m[10] * vz;
			}

			// ---------------- projection (backward.cu:877-894) ----------------
			{
				const float* proj = a.projmatrix;
				const float4 m_hom = xform4x4(mean, proj);
				const float m_w = 1.0f / (m_hom.w + 0.0000001f);
				const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
				const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
				const float gx = g_mean2D.x, gy = g_mean2D.y;
				dmean.x += (proj[0] * m_w - proj[3] * mul1) * gx + (proj[1] * m_w - proj[3] * mul2) * gy;
				dmean.y += (proj[4] * m_w - proj[7] * mul1) * gx + (proj[5] * m_w - proj[7] * mul2) * gy;
				dmean.z += (proj[8] * m_w - proj[11] * mul1) * gx + (proj[9] * m_w - proj[11] * mul2) * gy;
			}

			// ---------------- SH (backward.cu:897-906) ----------------
			// done by sh_bwd_kernel (sh_bwd.hip), which left its mean / time gradient in record words 12..15
			if (a.shs)
			{
				dmean.x += r3.x; dmean.y += r3.y; dmean.z += r3.z;
				dts += r3.w;
			}

			// ---------------- covariance (backward.cu:907-922) ----------------
			if (a.scales)
			{
				const float mod = a.scale_modifier;
				float3 sc = b_ld3(a.scales, idx);
				float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
				float inv_nq = 1.f, inv_nqr = 1.f;
				if (a.raw)
				{
					sc = make_float3(expf(sc.x), expf(sc.y), expf(sc.z));
					q = act_normalize(q, &inv_nq);
				}
				if (a.rot_4d)
				{
					// backward.cu:689-834
					float scale_t = a.scales_t[idx];
					float4 qr = reinterpret_cast<const float4*>(a.rotations_r)[idx];
					float opac = a.opacities[idx];
					if (a.raw)
					{
						scale_t = expf(scale_t);
						qr = act_normalize(qr, &inv_nqr);
						opac = act_sigmoid(opac);
					}
					const float dt = a.timestamp - a.ts[idx];
					const M4 S = diag4(mod * sc.x, mod * sc.y, mod * sc.z, mod * scale_t);
					M4 Ml, Mr;
					build_Ml_Mr(q, qr, Ml, Mr);
					const M4 R = mul(Mr, Ml);
					const M4 M = mul(S, R);
					const M4 Sigma = mul(transpose(M), M);
					const float cov_t = Sigma.c[3][3];
					const float cov_t_pre = (a.prefilter_var > 0.0) ? (a.prefilter_var + cov_t) : cov_t;
					const float marginal_t = expf((float)(-0.5 * dt * dt / cov_t_pre));
					if (marginal_t > 0.05)
					{
						const float c12[3] = { Sigma.c[3][0], Sigma.c[3][1], Sigma.c[3][2] }; // Q7
						float d12[3];
						d12[0] = -(float)(dcov[0] * c12[0] + dcov[1] * c12[1] * 0.5 + dcov[2] * c12[2] * 0.5) * 2.0f / cov_t;
						d12[1] = -(float)(dcov[1] * c12[0] * 0.5 + dcov[3] * c12[1] + dcov[4] * c12[2] * 0.5) * 2.0f / cov_t;
						d12[2] = -(float)(dcov[2] * c12[0] * 0.5 + dcov[4] * c12[1] * 0.5 + dcov[5] * c12[2]) * 2.0f / cov_t;
						float dL_dcovt = (c12[0] * c12[0] * dcov[0] + c12[0] * c12[1] * dcov[1] +
						                  c12[0] * c12[2] * dcov[2] + c12[1] * c12[1] * dcov[3] +
						                  c12[1] * c12[2] * dcov[4] + c12[2] * c12[2] * dcov[5]) / (cov_t * cov_t);
						const float dL_dmarginal_t = g_opacity * opac;
						g_opacity *= marginal_t;
						const float dmarg_dcovt = marginal_t * dt * dt / 2 / (cov_t_pre * cov_t_pre);
						const float dmarg_dt = marginal_t * dt / cov_t_pre;
						dL_dcovt += dmarg_dcovt * dL_dmarginal_t;
						float dL_dt = dL_dmarginal_t * dmarg_dt;
						// Q5: the whole mean gradient is treated as the gradient of delta_mean
						d12[0] += dmean.x / cov_t * dt; d12[1] += dmean.y / cov_t * dt; d12[2] += dmean.z / cov_t * dt;
						const float ddot = dmean.x * c12[0] + dmean.y * c12[1] + dmean.z * c12[2];
						dL_dcovt += -ddot / (cov_t * cov_t) * dt;
						dL_dt += -ddot / cov_t;
						dts += dL_dt;
						M4 dSig;
						dSig.c[0][0] = dcov[0]; dSig.c[0][1] = 0.5f * dcov[1]; dSig.c[0][2] = 0.5f * dcov[2]; dSig.c[0][3] = 0.5f * d12[0];
						dSig.c[1][0] = 0.5f * dcov[1]; dSig.c[1][1] = dcov[3]; dSig.c[1][2] = 0.5f * dcov[4]; dSig.c[1][3] = 0.5f * d12[1];
						dSig.c[2][0] = 0.5f * dcov[2]; dSig.c[2][1] = 0.5f * dcov[4]; dSig.c[2][2] = dcov[5]; dSig.c[2][3] = 0.5f * d12[2];
						dSig.c[3][0] = 0.5f * d12[0]; dSig.c[3][1] = 0.5f * d12[1]; dSig.c[3][2] = 0.5f * d12[2]; dSig.c[3][3] = dL_dcovt;
						M4 M2;
#pragma unroll
						for (int j = 0; j < 4; j++)
#pragma unroll
							for (int i = 0; i < 4; i++) M2.c[j][i] = 2.0f * M.c[j][i];
						const M4 dM = mul(M2, dSig);
						const M4 Rt = transpose(R);
						M4 dMt = transpose(dM);
						dscale.x = dot4(Rt.c[0], dMt.c[0]);
						dscale.y = dot4(Rt.c[1], dMt.c[1]);
						dscale.z = dot4(Rt.c[2], dMt.c[2]);
						dscale_t = dot4(Rt.c[3], dMt.c[3]);
						const float scl[4] = { mod * sc.x, mod * sc.y, mod * sc.z, mod * scale_t };
#pragma unroll
						for (int k = 0; k < 4; k++)
#pragma unroll
							for (int i = 0; i < 4; i++) dMt.c[k][i] *= scl[k];
						const M4 A = mul(dMt, Mr);
						drot.x = A.c[0][0] + A.c[1][1] + A.c[2][2] + A.c[3][3];
						drot.y = -A.c[0][1] + A.c[1][0] - A.c[2][3] + A.c[3][2];
						drot.z = A.c[0][2] - A.c[1][3] - A.c[2][0] + A.c[3][1];
						drot.w = -A.c[0][3] - A.c[1][2] + A.c[2][1] + A.c[3][0];
						const M4 B = mul(Ml, dMt);
						drot_r.x = B.c[0][0] + B.c[1][1] + B.c[2][2] + B.c[3][3];
						drot_r.y = -B.c[0][1] + B.c[1][0] + B.c[2][3] - B.c[3][2];
						drot_r.z = B.c[0][2] + B.c[1][3] - B.c[2][0] - B.c[3][1];
						drot_r.w = B.c[0][3] - B.c[1][2] + B.c[2][1] - B.c[3][0];
						if (a.raw)
						{
							dscale_t *= scale_t;                       // d exp
							drot_r = act_normalize_bwd(qr, inv_nqr, drot_r);
						}
					}
				}
				else
				{
					// backward.cu:621-684
					const float r = q.x, x = q.y, y = q.z, z = q.w;
					const M3 R = quat_to_R(q);
					const float s3[3] = { mod * sc.x, mod * sc.y, mod * sc.z };
					M3 S;
#pragma unroll
					for (int j = 0; j < 3; j++)
#pragma unroll
						for (int i = 0; i < 3; i++) S.c[j][i] = 0.f;
					S.c[0][0] = s3[0]; S.c[1][1] = s3[1]; S.c[2][2] = s3[2];
					const M3 M = mul(S, R);
					M3 dSig;
					dSig.c[0][0] = dcov[0]; dSig.c[0][1] = 0.5f * dcov[1]; dSig.c[0][2] = 0.5f * dcov[2];
					dSig.c[1][0] = 0.5f * dcov[1]; dSig.c[1][1] = dcov[3]; dSig.c[1][2] = 0.5f * dcov[4];
					dSig.c[2][0] = 0.5f * dcov[2]; dSig.c[2][1] = 0.5f * dcov[4]; dSig.c[2][2] = dcov[5];
					M3 M2;
#pragma unroll
					for (int j = 0; j < 3; j++)
#pragma unroll
						for (int i = 0; i < 3; i++) M2.c[j][i] = 2.0f * M.c[j][i];
					const M3 dM = mul(M2, dSig);
					const M3 Rt = transpose(R);
					M3 dMt = transpose(dM);
					dscale.x = dot3(Rt.c[0][0], Rt.c[0][1], Rt.c[0][2], dMt.c[0][0], dMt.c[0][1], dMt.c[0][2]);
					dscale.y = dot3(Rt.c[1][0], Rt.c[1][1], Rt.c[1][2], dMt.c[1][0], dMt.c[1][1], dMt.c[1][2]);
					dscale.z = dot3(Rt.c[2][0], Rt.c[2][1], Rt.c[2][2], dMt.c[2][0], dMt.c[2][1], dMt.c[2][2]);
#pragma unroll
					for (int k = 0; k < 3; k++)
#pragma unroll
						for (int i = 0; i < 3; i++) dMt.c[k][i] *= s3[k];
#define DD(i, j) dMt.c[i][j]
					drot.x = 2 * z * (DD(0, 1) - DD(1, 0)) + 2 * y * (DD(2, 0) - DD(0, 2)) + 2 * x * (DD(1, 2) - DD(2, 1));
					drot.y = 2 * y * (DD(1, 0) + DD(0, 1)) + 2 * z * (DD(2, 0) + DD(0, 2)) + 2 * r * (DD(1, 2) - DD(2, 1)) - 4 * x * (DD(2, 2) + DD(1, 1));
					drot.z = 2 * x * (DD(1, 0) + DD(0, 1)) + 2 * r * (DD(2, 0) - DD(0, 2)) + 2 * z * (DD(1, 2) + DD(2, 1)) - 4 * y * (DD(2, 2) + DD(0, 0));
					drot.w = 2 * r * (DD(0, 1) - DD(1, 0)) + 2 * x * (DD(2, 0) + DD(0, 2)) + 2 * y * (DD(1, 2) + DD(2, 1)) - 4 * z * (DD(1, 1) + DD(0, 0));
#undef DD
					// Q6: gaussian_dim == 4 without rot_4d has no marginal-opacity backward
				}
				if (a.raw)
				{
					dscale = make_float3(dscale.x * sc.x, dscale.y * sc.y, dscale.z * sc.z); // d exp
					drot = act_normalize_bwd(q, inv_nq, drot);
				}
			}
		}

		if (a.raw)
		{
			// d sigmoid: the blend accumulated dL/d(activated opacity) (already rescaled by marginal_t where rot_4d)
			const float o = act_sigmoid(a.opacities[idx]);
			g_opacity *= o * (1.0f - o);
		}
		// ---- stores (every output written for every Gaussian) ----
		b_st3(a.dL_dmean2D, idx, g_mean2D);
		// (the three per-view outputs nobody downstream of a training step reads may be NULL: fdgs_backward_out)
		if (a.dL_dcolor) b_st3(a.dL_dcolor, idx, g_color);
		if (a.dL_dflows) { a.dL_dflows[2 * (size_t)idx] = g_flow.x; a.dL_dflows[2 * (size_t)idx + 1] = g_flow.y; }
		if (a.dL_dcov3D)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) a.dL_dcov3D[6 * (size_t)idx + k] = dcov[k];
		}
		if (a.adam)
		{
			// The last view of an optimizer step: complete the gradient (this view's, plus what the earlier views left when accumulating)
			// and take the parameter's Adam step with it -- every Gaussian, visible here or not (torch.optim.Adam moves a parameter whose
			// gradient is zero by its first moment).  Same device function as adam.hip: bit-identical to the separate launch.
			const auto step = [&](const float* param, float* grad, const size_t e, const float g_view, const float lr)
			{
				const float total = a.accum ? grad[e] + g_view : g_view;
				if (!a.accum || visible) grad[e] = total;
				if (param == nullptr) return;   // (a gradient array without its parameter: a 3D scene's dL_dts / dL_dscale_t / dL_drot_r of the common sink)
				float* p = const_cast<float*>(param) + e;
				const size_t off = (size_t)(p - a.ad_flat);
				float pv = *p, m = a.ad_m[off], v = a.ad_v[off];
				adam_update(pv, m, v, total, lr, a.ad_b1, a.ad_b2, a.ad_eps, a.ad_inv_sqrt_bc2);
				a.ad_m[off] = m; a.ad_v[off] = v; *p = pv;
			};
			const size_t i = (size_t)idx;
			step(a.opacities, a.dL_dopacity, i, g_opacity, a.ad_lr[1]);
			step(a.means_in, a.dL_dmeans, 3 * i, dmean.x, a.ad_lr[0]); step(a.means_in, a.dL_dmeans, 3 * i + 1, dmean.y, a.ad_lr[0]); step(a.means_in, a.dL_dmeans, 3 * i + 2, dmean.z, a.ad_lr[0]);
			if (a.dL_dts) step(a.ts, a.dL_dts, i, dts, a.ad_lr[2]);
			if (a.dL_dscale) { step(a.scales, a.dL_dscale, 3 * i, dscale.x, a.ad_lr[3]); step(a.scales, a.dL_dscale, 3 * i + 1, dscale.y, a.ad_lr[3]); step(a.scales, a.dL_dscale, 3 * i + 2, dscale.z, a.ad_lr[3]); }
			if (a.dL_dscale_t) step(a.scales_t, a.dL_dscale_t, i, dscale_t, a.ad_lr[4]);
			if (a.dL_drot) { step(a.rotations, a.dL_drot, 4 * i, drot.x, a.ad_lr[5]); step(a.rotations, a.dL_drot, 4 * i + 1, drot.y, a.ad_lr[5]); step(a.rotations, a.dL_drot, 4 * i + 2, drot.z, a.ad_lr[5]); step(a.rotations, a.dL_drot, 4 * i + 3, drot.w, a.ad_lr[5]); }
			if (a.dL_drot_r) { step(a.rotations_r, a.dL_drot_r, 4 * i, drot_r.x, a.ad_lr[6]); step(a.rotations_r, a.dL_drot_r, 4 * i + 1, drot_r.y, a.ad_lr[6]); step(a.rotations_r, a.dL_drot_r, 4 * i + 2, drot_r.z, a.ad_lr[6]); step(a.rotations_r, a.dL_drot_r, 4 * i + 3, drot_r.w, a.ad_lr[6]); }
			return;
		}
		if (a.accum)
		{
			// gradient accumulation over the views of one optimizer step: add into the parameter gradients
			if (!visible) return; // nothing to add
			a.dL_dopacity[idx] += g_opacity;
			a.dL_dmeans[3 * (size_t)idx] += dmean.x; a.dL_dmeans[3 * (size_t)idx + 1] += dmean.y; a.dL_dmeans[3 * (size_t)idx + 2] += dmean.z;
			if (a.dL_dts) a.dL_dts[idx] += dts;
			if (a.dL_dscale) { a.dL_dscale[3 * (size_t)idx] += dscale.x; a.dL_dscale[3 * (size_t)idx + 1] += dscale.y; a.dL_dscale[3 * (size_t)idx + 2] += dscale.z; }
			if (a.dL_dscale_t) a.dL_dscale_t[idx] += dscale_t;
			if (a.dL_drot) { float4* d = reinterpret_cast<float4*>(a.dL_drot) + idx; const float4 o = *d; *d = make_float4(o.x + drot.x, o.y + drot.y, o.z + drot.z, o.w + drot.w); }
			if (a.dL_drot_r) { float4* d = reinterpret_cast<float4*>(a.dL_drot_r) + idx; const float4 o = *d; *d = make_float4(o.x + drot_r.x, o.y + drot_r.y, o.z + drot_r.z, o.w + drot_r.w); }
			return;
		}
		a.dL_dopacity[idx] = g_opacity;
		b_st3(a.dL_dmeans, idx, dmean);
		if (a.dL_dts) a.dL_dts[idx] = dts;
		if (a.dL_dscale) b_st3(a.dL_dscale, idx, dscale);
		if (a.dL_dscale_t) a.dL_dscale_t[idx] = dscale_t;
		if (a.dL_drot) reinterpret_cast<float4*>(a.dL_drot)[idx] = drot;
		if (a.dL_drot_r) reinterpret_cast<float4*>(a.dL_drot_r)[idx] = drot_r;
	}

	// Two builds of the same body (round 5): 132 VGPRs = three waves per SIMD, and -- for scenes of a million Gaussians and more -- held
	// to 128 VGPRs = four waves per SIMD at the price of three 8-byte spills (C5: 151 -> 141 us; C3 unchanged).  Round 6: the translation
	// unit is built WITHOUT the SLP vectorizer (build.sh) -- its packed fp32 operations (725 of them here) need register PAIRS and 523
	// moves to line operands up: 2524 -> 2642 instructions but 132 -> 97 VGPRs, no spills, five waves per SIMD for both builds (C5:
	// 141 -> 131 us; the same IEEE operations one at a time: bit-identical results).
	__global__ void __launch_bounds__(256) preprocess_bwd_kernel(const BwdArgs a) { preprocess_bwd_body(a); }
	__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) preprocess_bwd_kernel_w4(const BwdArgs a) { preprocess_bwd_body(a); }

	hipError_t launch_preprocess_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                                 const char* geom, hipStream_t stream)
	{
		const GeomLayout L = geom_layout(s.P);
		BwdArgs a;
		a.P = s.P; a.D = s.D; a.D_t = s.D_t; a.M = s.M;
		a.shs = s.shs; a.opacities = s.opacities; a.ts = s.ts; a.scales = s.scales; a.scales_t = s.scales_t;
		a.rotations = s.rotations; a.rotations_r = s.rotations_r; a.cov3D_precomp = s.cov3D_precomp;
		a.viewmatrix = s.viewmatrix; a.projmatrix = s.projmatrix; a.campos = s.campos;
		a.scale_modifier = s.scale_modifier; a.prefilter_var = s.prefilter_var;
		a.tan_fovx = s.tan_fovx; a.tan_fovy = s.tan_fovy;
		a.focal_y = s.H / (2.0f * s.tan_fovy); // rasterizer_impl.cu:424-425
		a.focal_x = s.W / (2.0f * s.tan_fovx);
		a.timestamp = s.timestamp; a.time_duration = s.time_duration;
		a.rot_4d = s.rot_4d; a.gaussian_dim = s.gaussian_dim; a.force_sh_3d = s.force_sh_3d; a.raw = s.raw_params; a.accum = out.accumulate;
		a.radii = in.radii; a.means = in.out_means3D;
		a.cov3D = reinterpret_cast<const float*>(geom + L.cov3D);
		a.clamped = reinterpret_cast<const uint8_t*>(geom + L.clamped);
		a.gacc = out.grad_accum; a.rezero = out.grad_accum_clean;
		a.records = reinterpret_cast<const float4*>(geom + L.records);
		a.half_w = 0.5f * s.W; a.half_h = 0.5f * s.H;   // ddelx_dx, ddely_dy (backward.cu:1010-1011)
		a.dL_dmean2D = out.dL_dmeans2D; a.dL_dcolor = out.dL_dcolors; a.dL_dflows = out.dL_dflows;
		a.dL_dopacity = out.dL_dopacity; a.dL_dmeans = out.dL_dmeans3D; a.dL_dcov3D = out.dL_dcov3D;
		a.dL_dts = out.dL_dts; a.dL_dscale = out.dL_dscales; a.dL_dscale_t = out.dL_dscales_t;
		a.dL_drot = out.dL_drotations; a.dL_drot_r = out.dL_drotations_r;
		a.adam = 0; a.means_in = s.means3D; a.ad_flat = a.ad_m = a.ad_v = nullptr;
		if (out.adam != nullptr)
		{
			const fdgs_geometry_adam& g = *out.adam;
			const double bc1 = 1.0 - pow((double)g.beta1, (double)g.step), bc2 = 1.0 - pow((double)g.beta2, (double)g.step);   // as fdgs_adam_step
			const float inv_bc1 = (float)(1.0 / bc1);
			a.adam = 1; a.ad_flat = g.flat; a.ad_m = g.exp_avg; a.ad_v = g.exp_avg_sq;
			const float lr[7] = { g.lr_means3D, g.lr_opacities, g.lr_ts, g.lr_scales, g.lr_scales_t, g.lr_rotations, g.lr_rotations_r };
			for (int k = 0; k < 7; k++) a.ad_lr[k] = lr[k] * inv_bc1;
			a.ad_b1 = g.beta1; a.ad_b2 = g.beta2; a.ad_eps = g.eps; a.ad_inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
		}
		if (s.P >= (1 << 20)) hipLaunchKernelGGL(preprocess_bwd_kernel_w4, dim3(div_up(s.P, 256)), dim3(256), 0, stream, a);
		else hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(div_up(s.P, 256)), dim3(256), 0, stream, a);
		return hipGetLastError();
	}
}

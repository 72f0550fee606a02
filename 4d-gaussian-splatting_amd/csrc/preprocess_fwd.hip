// preprocess_fwd.hip -- per-Gaussian forward preprocessing for gfx950.
//
// What it computes (reference preprocessCUDA, forward.cu:355-496): 4D -> 3D
// conditional covariance + conditional mean shift + temporal marginal opacity
// (forward.cu:279-352) or 3D covariance with the 1-D temporal marginal
// (forward.cu:242-276, 431-437), near cull (auxiliary.h:140-163), EWA projection
// (forward.cu:198-237), conic, radius, tile rectangle (auxiliary.h:47-57),
// SH / 4D-SH -> RGB (forward.cu:20-195).
//
// What is different from the reference's kernel: the SH coefficients (12*M bytes per Gaussian, the bulk of
// the input) are staged per wave through LDS with fully coalesced loads instead of a 12*M-byte stride between
// threads; culled Gaussians are never fetched.  It emits ONE packed 48-byte
// blend record per Gaussian (position, conic, opacity, colour, depth, flow)
// instead of four separate arrays, the tile rectangle (so later stages never
// recompute it), and it writes every output unconditionally so no buffer needs
// pre-zeroing; as the forward's first kernel it also clears the tile counters
// of the binning passes (tilebin.hip).
//
// Bit-exactness: radius, rectangle, tile count and depth bits must equal the
// oracle's exactly, so this file is compiled with FP contraction OFF and follows
// the evaluation order fixed in fdgs_math.h; double promotions of the
// reference (ndc2Pix, the -0.5*dt*dt/var exponent, 2.0*zz, 2*pi*...) are kept.
// FP contraction must be off for everything below, including the inline helpers of fdgs_math.h.
#pragma clang fp contract(off)
#include "fdgs_common.h"
#include "fdgs_math.h"

namespace fdgs
{
	struct PreArgs
	{
		int P, D, D_t, M, W, H;
		const float *means3D, *shs, *colors_precomp, *flows, *opacities, *ts, *scales, *scales_t;
		const float *rotations, *rotations_r, *cov3D_precomp, *viewmatrix, *projmatrix, *campos;
		float scale_modifier, prefilter_var, tan_fovx, tan_fovy, focal_x, focal_y, timestamp, time_duration;
		int rot_4d, gaussian_dim, force_sh_3d, raw, sh_vec_ok;
		int grid_x, grid_y;
		int tile_cull;   // fdgs_forward_out.tile_cull: list the Gaussian only in the tiles it can reach with alpha >= 1/255
		// outputs
		int32_t* radii; float* out_means3D; float* covs_com;
		float4* records; float* depths; float* cov3D; uint32_t* tiles_touched; ushort4* rect; uint8_t* clamped;
		uint32_t* bin_counters; int bin_counter_words;   // tile instance counters of the binning passes, cleared here (tilebin.hip)
	};

	__device__ __forceinline__ float3 ld3(const float* p, size_t i) { return make_float3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
	__device__ __forceinline__ float3 add3(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
	__device__ __forceinline__ float3 scl3(float s, float3 a) { return make_float3(s * a.x, s * a.y, s * a.z); }
	__device__ __forceinline__ float3 sub3(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }

	// 3D SH (forward.cu:20-71); returns the un-clamped colour + 0.5
	__device__ float3 sh_color_3d(int deg, const float* __restrict__ sh, float3 dir)
	{
		float3 result = scl3(SH_C0, ld3(sh, 0));
		if (deg > 0)
		{
			const float x = dir.x, y = dir.y, z = dir.z;
			result = sub3(add3(sub3(result, scl3(SH_C1 * y, ld3(sh, 1))), scl3(SH_C1 * z, ld3(sh, 2))), scl3(SH_C1 * x, ld3(sh, 3)));
			if (deg > 1)
			{
				const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
				result = add3(result, scl3(SH_C2[0] * xy, ld3(sh, 4)));
				result = add3(result, scl3(SH_C2[1] * yz, ld3(sh, 5)));
				result = add3(result, scl3(SH_C2[2] * (2.0f * zz - xx - yy), ld3(sh, 6)));
				result = add3(result, scl3(SH_C2[3] * xz, ld3(sh, 7)));
				result = add3(result, scl3(SH_C2[4] * (xx - yy), ld3(sh, 8)));
				if (deg > 2)
				{
					result = add3(result, scl3(SH_C3[0] * y * (3.0f * xx - yy), ld3(sh, 9)));
					result = add3(result, scl3(SH_C3[1] * xy * z, ld3(sh, 10)));
					result = add3(result, scl3(SH_C3[2] * y * (4.0f * zz - xx - yy), ld3(sh, 11)));
					result = add3(result, scl3(SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), ld3(sh, 12)));
					result = add3(result, scl3(SH_C3[4] * x * (4.0f * zz - xx - yy), ld3(sh, 13)));
					result = add3(result, scl3(SH_C3[5] * z * (xx - yy), ld3(sh, 14)));
					result = add3(result, scl3(SH_C3[6] * x * (xx - 3.0f * yy), ld3(sh, 15)));
				}
			}
		}
		return make_float3(result.x + 0.5f, result.y + 0.5f, result.z + 0.5f);
	}

	// basis values of the 4D path (forward.cu:87-131; note the double promotion in l2m0)
	__device__ __forceinline__ void sh_basis_4d(int deg, float x, float y, float z, float* l)
	{
		l[0] = SH_C0;
		if (deg > 0)
		{
			l[1] = -1 * SH_C1 * y; l[2] = SH_C1 * z; l[3] = -1 * SH_C1 * x;
			if (deg > 1)
			{
				const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
				l[4] = SH_C2[0] * xy;
				l[5] = SH_C2[1] * yz;
				l[6] = (float)(SH_C2[2] * (2.0 * zz - xx - yy));
				l[7] = SH_C2[3] * xz;
				l[8] = SH_C2[4] * (xx - yy);
				if (deg > 2)
				{
					l[9] = SH_C3[0] * y * (3 * xx - yy);
					l[10] = SH_C3[1] * xy * z;
					l[11] = SH_C3[2] * y * (4 * zz - xx - yy);
					l[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
					l[13] = SH_C3[4] * x * (4 * zz - xx - yy);
					l[14] = SH_C3[5] * z * (xx - yy);
					l[15] = SH_C3[6] * x * (xx - 3 * yy);
				}
			}
		}
	}
	__device__ __forceinline__ float3 sh_weighted(const float* l, const float* __restrict__ sh, int lo, int hi, int off)
	{
		float3 acc = scl3(l[lo - off], ld3(sh, lo));
#pragma unroll
		for (int k = lo + 1; k <= hi; k++) acc = add3(acc, scl3(l[k - off], ld3(sh, k)));
		return acc;
	}
	// 4D SH (forward.cu:73-195), split by coefficient block so each block of 16 coefficients can be staged through
	// LDS on its own: block 0 = the plain SH sum, blocks 1 / 2 = the same basis times cos(2 pi k dt / T), k = 1, 2
	// (only when deg > 2).  Evaluation order inside and across blocks is the reference's.
	__device__ __forceinline__ float3 sh4d_block0(int deg, const float* l, const float* __restrict__ sh)
	{
		float3 result = scl3(l[0], ld3(sh, 0));
		if (deg > 0)
		{
			result = add3(result, sh_weighted(l, sh, 1, 3, 0));
			if (deg > 1)
			{
				result = add3(result, sh_weighted(l, sh, 4, 8, 0));
				if (deg > 2) result = add3(result, sh_weighted(l, sh, 9, 15, 0));
			}
		}
		return result;
	}

	// Coalesced staging of one coefficient block of 64 consecutive Gaussians into a wave-private LDS tile
	// (row stride SH_STRIDE floats: odd, so the lane-per-Gaussian reads that follow are bank-conflict free).
	// The reference reads these 12*M bytes per Gaussian with a 12*M-byte stride between threads (forward.cu:85).
	constexpr int SH_STRIDE = 49;
#ifdef FDGS_PRE_WG_SYNC   // A/B: the workgroup barriers of rounds 1-4
#define FDGS_TILE_SYNC() __syncthreads()
#else
#define FDGS_TILE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
	// generic path (any coefficient count / alignment): one float per lane and trip
	__device__ __forceinline__ void stage_sh_block_scalar(float* __restrict__ tile, const float* __restrict__ shs, int g0, int P, int M,
	                                                      int first_coeff, int ncoeff, unsigned long long alive_mask, int lane)
	{
		const int nf = 3 * ncoeff;                 // floats per Gaussian in this block
		const int total = WAVE * nf;
		const size_t row_floats = (size_t)3 * M;
		int g = lane / nf, pos = lane - g * nf;    // element e = lane, lane + 64, ...  ->  (Gaussian, float)
		const int dg = WAVE / nf, dpos = WAVE - dg * nf;
		for (int e = lane; e < total; e += WAVE)
		{
			if (g0 + g < P && ((alive_mask >> g) & 1ull))
				tile[g * SH_STRIDE + pos] = shs[(size_t)(g0 + g) * row_floats + (size_t)first_coeff * 3 + pos];
			g += dg; pos += dpos;
			if (pos >= nf) { pos -= nf; g++; }
		}
	}
	// full 16-coefficient block, rows 16-byte aligned: each lane issues its 12 dwordx4 loads back to back
	// (64 lanes x 16 B = 1 KiB per instruction, 192-byte runs), then scatters them into the tile
	__device__ __forceinline__ void stage_sh_block16(float* __restrict__ tile, const float* __restrict__ shs, int g0, int P, int M,
	                                                 int first_coeff, unsigned long long alive_mask, int lane)
	{
		constexpr int CH = 12;                     // float4 chunks per Gaussian: 16 coefficients x 3 floats / 4
#ifndef FDGS_PRE_BATCH
#define FDGS_PRE_BATCH 12                      // float4 per lane in flight (A/B: 6 = two batches per block, fewer staging registers)
#endif
		constexpr int BATCH = FDGS_PRE_BATCH;
		const size_t row_floats = (size_t)3 * M;
#pragma unroll
		for (int b = 0; b < CH / BATCH; b++)
		{
			float4 v[BATCH];
#pragma unroll
			for (int j = 0; j < BATCH; j++)
			{
				const int i = b * BATCH + j;
				const int c = i * WAVE + lane, g = c / CH, q = c - g * CH;
				const bool ok = g0 + g < P && ((alive_mask >> g) & 1ull);
				v[j] = ok ? *reinterpret_cast<const float4*>(shs + (size_t)(g0 + g) * row_floats + (size_t)first_coeff * 3 + 4 * q)
				          : make_float4(0.f, 0.f, 0.f, 0.f);
			}
#pragma unroll
			for (int j = 0; j < BATCH; j++)
			{
				const int i = b * BATCH + j;
				const int c = i * WAVE + lane, g = c / CH, q = c - g * CH;
				float* d = tile + g * SH_STRIDE + 4 * q;
				d[0] = v[j].x; d[1] = v[j].y; d[2] = v[j].z; d[3] = v[j].w;
			}
		}
	}
	__device__ __forceinline__ void stage_sh_block(float* __restrict__ tile, const float* __restrict__ shs, int g0, int P, int M,
	                                               int first_coeff, int ncoeff, unsigned long long alive_mask, int lane, bool vec_ok)
	{
		if (vec_ok && ncoeff == 16) stage_sh_block16(tile, shs, g0, P, M, first_coeff, alive_mask, lane);
		else stage_sh_block_scalar(tile, shs, g0, P, M, first_coeff, ncoeff, alive_mask, lane);
	}

	// fdgs_forward_out.tile_cull: the tiles of the reference's rectangle (the square of 3 sigma_max around the mean,
	// auxiliary.h:46-57) that the Gaussian can actually reach.  alpha = min(0.99, opacity * exp(-q(d))) with q(d) = 0.5 d^T K d,
	// K = the conic AS STORED (fp32), passes the blend's test alpha >= 1/255 (forward.cu:590) only where q(d) <= ln(255 opacity).
	// The ellipse q <= tau has the axis-aligned extents |dx| <= sqrt(2 tau S_xx), |dy| <= sqrt(2 tau S_yy) with S = K^-1 =
	// (K_yy, -K_xy, K_xx) / D_K, D_K = K_xx K_yy - K_xy^2 evaluated in DOUBLE from the three stored fp32 numbers (their products
	// are exact in double, so the difference keeps every bit an elongated splat cancels): S is the exact inverse of the matrix the
	// blend kernels evaluate, whatever rounding the conic's entries went through on their way into the record.  Slack: 0.05 in tau
	// (5 % in alpha, as the blend kernels' block test: covers every rounding of the per-pixel evaluation), 0.1 % + half a pixel on
	// the extents.  Opacity below 1/255: alpha < 1/255 everywhere, listed nowhere.  Anything odd: the reference's rectangle.
	__device__ __forceinline__ ushort4 reachable_rect(const ushort4 ref, const float2 pix, const float3 conic, const float opacity)
	{
		if (opacity < 1.0f / 255.0f) return make_ushort4(0, 0, 0, 0);
		if (!(conic.x > 0.0f && conic.z > 0.0f)) return ref;
		const double DK = (double)conic.x * (double)conic.z - (double)conic.y * (double)conic.y;
		if (!(DK > 0.0)) return ref;
		const double tau2 = 2.0 * ((double)logf(255.0f * opacity) + 0.05);
		const float hx = (float)sqrt(tau2 * (double)conic.z / DK) * 1.001f + 0.5f, hy = (float)sqrt(tau2 * (double)conic.x / DK) * 1.001f + 0.5f;
		if (!(hx < 1.0e6f && hy < 1.0e6f)) return ref;   // also NaN
		// tile t holds the pixel centres TILE * t .. TILE * t + TILE - 1: those with a centre inside [pix - h, pix + h]
		const int tx0 = max((int)ref.x, (int)ceilf((pix.x - hx - (float)(TILE_X - 1)) / (float)TILE_X));
		const int tx1 = min((int)ref.z, (int)floorf((pix.x + hx) / (float)TILE_X) + 1);
		const int ty0 = max((int)ref.y, (int)ceilf((pix.y - hy - (float)(TILE_Y - 1)) / (float)TILE_Y));
		const int ty1 = min((int)ref.w, (int)floorf((pix.y + hy) / (float)TILE_Y) + 1);
		if (tx1 <= tx0 || ty1 <= ty0) return make_ushort4(0, 0, 0, 0);
		return make_ushort4((unsigned short)tx0, (unsigned short)ty0, (unsigned short)tx1, (unsigned short)ty1);
	}

	// The geometry of one Gaussian (everything of preprocessCUDA but the colour): forward.cu:279-352 / 242-276 / 431-437, the near cull,
	// the EWA projection, conic, radius and tile rectangle.  ``in``: the inputs AS STORED (raw parameters when a.raw: the activations run
	// here); shared by the one-launch kernel, the geometry half of the split forward and the streaming kernel below.
	struct GeoIn { float3 p; float opacity; float3 sc; float sct; float4 q, qr; float t; };
	struct GeoOut
	{
		bool alive; int radius; uint32_t tiles; ushort4 rect; float depth; float2 pix; float3 conic; float opacity; float3 p_orig; float cov[6];
	};
	__device__ __forceinline__ void pre_geometry(const PreArgs& a, const GeoIn& in, const float* __restrict__ cov_precomp, const bool valid, GeoOut& o,
	                                             const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix)
	{
		float3 p_orig = in.p;
		float opacity = in.opacity;
		if (a.raw) opacity = act_sigmoid(opacity);
		float cov[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
		bool alive = valid;
		int radius = 0;
		uint32_t tiles = 0;
		ushort4 rect = make_ushort4(0, 0, 0, 0);
		float depth = 0.0f;
		float2 pix = make_float2(0.f, 0.f);
		float3 conic = make_float3(0.f, 0.f, 0.f);
		if (cov_precomp != nullptr)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) cov[k] = cov_precomp[k];
		}
		else if (a.rot_4d)
		{
			// forward.cu:279-352
			float3 sc = in.sc;
			float sct = in.sct;
			float4 q = in.q, qr = in.qr;
			if (a.raw)
			{
				float unused;
				sc = make_float3(expf(sc.x), expf(sc.y), expf(sc.z));
				sct = expf(sct);
				q = act_normalize(q, &unused);
				qr = act_normalize(qr, &unused);
			}
			const float mod = a.scale_modifier;
			const float dt = a.timestamp - in.t;
			const M4 S = diag4(mod * sc.x, mod * sc.y, mod * sc.z, mod * sct);
			M4 Ml, Mr;
			build_Ml_Mr(q, qr, Ml, Mr);
			const M4 M = mul(S, mul(Mr, Ml));
			const M4 Sigma = mul(transpose(M), M);
			const float cov_t = Sigma.c[3][3];
			const float marginal_t = expf((float)(-0.5 * dt * dt / ((a.prefilter_var > 0.0) ? (a.prefilter_var + cov_t) : cov_t)));
			alive = marginal_t > 0.05;
			if (alive)
			{
				opacity *= marginal_t;
				const float c12[3] = { Sigma.c[0][3], Sigma.c[1][3], Sigma.c[2][3] };
				cov[0] = Sigma.c[0][0] - (c12[0] * c12[0]) / cov_t;
				cov[1] = Sigma.c[0][1] - (c12[1] * c12[0]) / cov_t;
				cov[2] = Sigma.c[0][2] - (c12[2] * c12[0]) / cov_t;
				cov[3] = Sigma.c[1][1] - (c12[1] * c12[1]) / cov_t;
				cov[4] = Sigma.c[1][2] - (c12[2] * c12[1]) / cov_t;
				cov[5] = Sigma.c[2][2] - (c12[2] * c12[2]) / cov_t;
				p_orig.x += c12[0] / cov_t * dt;
				p_orig.y += c12[1] / cov_t * dt;
				p_orig.z += c12[2] / cov_t * dt;
			}
		}
		else
		{
			// forward.cu:242-276
			float3 sc = in.sc;
			float4 q = in.q;
			if (a.raw)
			{
				float unused;
				sc = make_float3(expf(sc.x), expf(sc.y), expf(sc.z));
				q = act_normalize(q, &unused);
			}
			const float mod = a.scale_modifier;
			M3 S;
#pragma unroll
			for (int j = 0; j < 3; j++)
#pragma unroll
				for (int i = 0; i < 3; i++) S.c[j][i] = 0.0f;
			S.c[0][0] = mod * sc.x; S.c[1][1] = mod * sc.y; S.c[2][2] = mod * sc.z;
			const M3 M = mul(S, quat_to_R(q));
			const M3 Sigma = mul(transpose(M), M);
			cov[0] = Sigma.c[0][0]; cov[1] = Sigma.c[0][1]; cov[2] = Sigma.c[0][2];
			cov[3] = Sigma.c[1][1]; cov[4] = Sigma.c[1][2]; cov[5] = Sigma.c[2][2];
			if (a.gaussian_dim == 4)
			{
				// forward.cu:431-437 (scales_t used as a variance)
				const float dt = in.t - a.timestamp;
				const float sigma = (a.raw ? expf(in.sct) : in.sct) * mod;
				const float marginal_t = expf((float)(-0.5 * dt * dt / ((a.prefilter_var > 0.0) ? (a.prefilter_var + sigma) : sigma)));
				if (marginal_t <= 0.05) alive = false;
				else opacity *= marginal_t;
			}
		}

		if (alive)
		{
			const float3 p_view = xform4x3(p_orig, viewmatrix);
			alive = !(p_view.z <= 0.2f); // auxiliary.h:153
			if (alive)
			{
				const float4 p_hom = xform4x4(p_orig, projmatrix);
				const float p_w = 1.0f / (p_hom.w + 0.0000001f);
				const float p_proj_x = p_hom.x * p_w, p_proj_y = p_hom.y * p_w;
				const Cov2D c2 = project_cov(p_orig, a.focal_x, a.focal_y, a.tan_fovx, a.tan_fovy, cov, viewmatrix);
				const float cx = c2.a + 0.3f, cy = c2.b, cz = c2.c + 0.3f;
				const float det = (cx * cz - cy * cy);
				if (det == 0.0f) alive = false;
				else
				{
					const float det_inv = 1.f / det;
					conic = make_float3(cz * det_inv, -cy * det_inv, cx * det_inv);
					const float mid = 0.5f * (cx + cz);
					const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
					const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
					const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
					// auxiliary.h:42-45 (double)
					pix.x = (float)(((p_proj_x + 1.0) * a.W - 1.0) * 0.5);
					pix.y = (float)(((p_proj_y + 1.0) * a.H - 1.0) * 0.5);
					// auxiliary.h:47-57
					const int r = (int)my_radius;
					const int x0 = min(a.grid_x, max(0, (int)((pix.x - r) / TILE_X)));
					const int y0 = min(a.grid_y, max(0, (int)((pix.y - r) / TILE_Y)));
					const int x1 = min(a.grid_x, max(0, (int)((pix.x + r + TILE_X - 1) / TILE_X)));
					const int y1 = min(a.grid_y, max(0, (int)((pix.y + r + TILE_Y - 1) / TILE_Y)));
					if ((x1 - x0) * (y1 - y0) == 0 || r <= 0) alive = false; // forward.cu:471
					else
					{
						radius = r;
						tiles = (uint32_t)((y1 - y0) * (x1 - x0));   // the reference's count, whatever the lists hold
						rect = make_ushort4((unsigned short)x0, (unsigned short)y0, (unsigned short)x1, (unsigned short)y1);
						depth = p_view.z;
						if (a.tile_cull) rect = reachable_rect(rect, pix, conic, opacity);
					}
				}
			}
		}

		o.alive = alive; o.radius = radius; o.tiles = tiles; o.rect = rect; o.depth = depth; o.pix = pix; o.conic = conic;
		o.opacity = opacity; o.p_orig = p_orig;
#pragma unroll
		for (int k = 0; k < 6; k++) o.cov[k] = cov[k];
	}

	// PART 0: everything.  The forward can also run it in two launches (fdgs_forward_out.split_colour): PART 1 = the geometry
	// (everything the tile binning needs; colour left at zero), PART 2 = the SH colour of the Gaussians PART 1 kept (radius > 0),
	// on a second stream next to the binning -- same arithmetic, same results.
#ifdef FDGS_PRE_WAVES   // A/B: hold the kernel to 512 / FDGS_PRE_WAVES registers (3: 168 VGPRs and 66 spills; default: 235 VGPRs, two waves per SIMD)
#define FDGS_PRE_OCC __attribute__((amdgpu_waves_per_eu(FDGS_PRE_WAVES, FDGS_PRE_WAVES)))
#else
#define FDGS_PRE_OCC
#endif
	template <int PART>
	__global__ void __launch_bounds__(256) FDGS_PRE_OCC preprocess_fwd_kernel(const PreArgs a)
	{
		// every lane stays until the end: the SH blocks are staged cooperatively per wave
		const int tid_g = blockIdx.x * blockDim.x + threadIdx.x;
		// first kernel of the forward: clears the tile counters of the binning passes (more cells than Gaussians: stride)
		if (PART != 2)
			for (int c = tid_g; c < a.bin_counter_words; c += gridDim.x * blockDim.x) a.bin_counters[c] = 0u;
		const bool valid = tid_g < a.P;
		const int idx = valid ? tid_g : a.P - 1;   // out-of-range lanes shadow the last Gaussian and store nothing

		float3 p_orig = ld3(a.means3D, idx);
		const float3 p_in = p_orig;
		float opacity = a.opacities[idx];   // (as stored: pre_geometry applies the activation of a raw parameter)
		float cov[6] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
		bool alive = valid;
		int radius = 0;
		uint32_t tiles = 0;
		ushort4 rect = make_ushort4(0, 0, 0, 0);
		float depth = 0.0f;
		float2 pix = make_float2(0.f, 0.f);
		float3 conic = make_float3(0.f, 0.f, 0.f);
		float3 rgb = make_float3(0.f, 0.f, 0.f);
		uint8_t clampbits = 0;

		if constexpr (PART == 2) alive = valid && a.radii[idx] > 0;   // what the geometry launch kept
		else
		{
			GeoIn in;
			in.p = p_orig; in.opacity = opacity;
			in.sc = make_float3(0.f, 0.f, 0.f); in.sct = 0.f; in.t = 0.f;
			in.q = make_float4(1.f, 0.f, 0.f, 0.f); in.qr = in.q;
			if (a.cov3D_precomp == nullptr)
			{
				in.sc = ld3(a.scales, idx);
				in.q = reinterpret_cast<const float4*>(a.rotations)[idx];
				if (a.rot_4d)
				{
					in.sct = a.scales_t[idx];
					in.qr = reinterpret_cast<const float4*>(a.rotations_r)[idx];
					in.t = a.ts[idx];
				}
				else if (a.gaussian_dim == 4) { in.t = a.ts[idx]; in.sct = a.scales_t[idx]; }
			}
			GeoOut o;
			pre_geometry(a, in, a.cov3D_precomp != nullptr ? a.cov3D_precomp + 6 * (size_t)idx : nullptr, valid, o, a.viewmatrix, a.projmatrix);
			alive = o.alive; radius = o.radius; tiles = o.tiles; rect = o.rect; depth = o.depth; pix = o.pix; conic = o.conic;
			opacity = o.opacity; p_orig = o.p_orig;
#pragma unroll
			for (int k = 0; k < 6; k++) cov[k] = o.cov[k];
		}

		if (a.colors_precomp != nullptr)
		{
			if (alive) rgb = ld3(a.colors_precomp, idx);
		}
		else if constexpr (PART != 1)
		{
			__shared__ float s_sh[256 / WAVE][WAVE * SH_STRIDE];
			const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
			float* tile = s_sh[wave];
			const float* row = tile + lane * SH_STRIDE;
			const unsigned long long amask = __ballot(alive);
			const int g0 = blockIdx.x * blockDim.x + wave * WAVE;
			const bool sh3d = (a.gaussian_dim == 3 || a.force_sh_3d);
			const int ncoef0 = min(16, (a.D + 1) * (a.D + 1));
			const int nblocks = (!sh3d && a.D > 2) ? 1 + min(max(a.D_t, 0), 2) : 1;
			// Q4: the forward view direction uses the UN-shifted input mean (forward.cu:480-482)
			float3 dir = sub3(p_in, make_float3(a.campos[0], a.campos[1], a.campos[2]));
			const float len = sqrtf(dot3(dir.x, dir.y, dir.z, dir.x, dir.y, dir.z));
			dir = make_float3(dir.x / len, dir.y / len, dir.z / len);
			float l[16];
			if (!sh3d) sh_basis_4d(a.D, dir.x, dir.y, dir.z, l);
			const float dir_t = (!sh3d) ? a.ts[idx] - a.timestamp : 0.f;
			float3 c = make_float3(0.f, 0.f, 0.f);
			for (int blk = 0; blk < nblocks; blk++)
			{
				stage_sh_block(tile, a.shs, g0, a.P, a.M, 16 * blk, blk == 0 ? ncoef0 : 16, amask, lane, a.sh_vec_ok != 0);
				// the tile is wave-private and the LDS executes one wave's operations in order: no workgroup barrier (which held the four
				// waves of the workgroup in lockstep: all loading, then all evaluating) -- only the compiler must keep the order
				FDGS_TILE_SYNC();
				if (alive)
				{
					if (blk == 0) c = sh3d ? sh_color_3d(a.D, row, dir) : sh4d_block0(a.D, l, row);
					else
					{
						const float tk = (blk == 1) ? (float)cos(2 * REF_PI * dir_t / a.time_duration)
						                            : (float)cos(2 * REF_PI * dir_t * 2 / a.time_duration);
						c = add3(c, scl3(tk, sh_weighted(l, row, 0, 15, 0)));
					}
				}
				FDGS_TILE_SYNC();
			}
			if (alive)
			{
				if (!sh3d) c = make_float3(c.x + 0.5f, c.y + 0.5f, c.z + 0.5f); // sh_color_3d already added it
				clampbits = (uint8_t)((c.x < 0 ? 1 : 0) | (c.y < 0 ? 2 : 0) | (c.z < 0 ? 4 : 0));
				rgb = make_float3(fmaxf(c.x, 0.0f), fmaxf(c.y, 0.0f), fmaxf(c.z, 0.0f));
			}
		}

		if (!valid) return;
		if constexpr (PART == 2)
		{
			// the colour words of the blend record (zero so far) and the clamp bits
			if (alive)
			{
				float* rec = reinterpret_cast<float*>(a.records + 3 * (size_t)idx);
				rec[6] = rgb.x; rec[7] = rgb.y; rec[8] = rgb.z;
				a.clamped[idx] = clampbits;
			}
			return;
		}
		// ---- stores (every output written for every Gaussian) ----
		a.radii[idx] = radius;
		a.tiles_touched[idx] = tiles;
		a.rect[idx] = rect;
		a.depths[idx] = depth;
		a.clamped[idx] = clampbits;
		a.out_means3D[3 * (size_t)idx + 0] = p_orig.x;
		a.out_means3D[3 * (size_t)idx + 1] = p_orig.y;
		a.out_means3D[3 * (size_t)idx + 2] = p_orig.z;
#pragma unroll
		for (int k = 0; k < 6; k++) a.cov3D[6 * (size_t)idx + k] = cov[k];
		if (a.covs_com != nullptr)
		{
#pragma unroll
			for (int k = 0; k < 6; k++) a.covs_com[6 * (size_t)idx + k] = cov[k];
		}
		float2 flow = make_float2(0.f, 0.f);
		if (a.flows != nullptr) flow = reinterpret_cast<const float2*>(a.flows)[idx];
		a.records[3 * (size_t)idx + 0] = make_float4(pix.x, pix.y, conic.x, conic.y);
		a.records[3 * (size_t)idx + 1] = make_float4(conic.z, radius > 0 ? opacity : 0.0f, rgb.x, rgb.y);
		a.records[3 * (size_t)idx + 2] = make_float4(rgb.z, depth, flow.x, flow.y);
	}


	// part: 0 = one launch; 1 / 2 = the geometry / colour halves (see the kernel)
	hipError_t launch_preprocess_fwd(const fdgs_scene& s, const fdgs_forward_out& out, char* geom, uint32_t* bin_counters, int part,
	                                 hipStream_t stream)
	{
		const GeomLayout L = geom_layout(s.P);
		PreArgs a;
		a.P = s.P; a.D = s.D; a.D_t = s.D_t; a.M = s.M; a.W = s.W; a.H = s.H;
		a.means3D = s.means3D; a.shs = s.shs; a.colors_precomp = s.colors_precomp; a.flows = s.flows;
		a.opacities = s.opacities; a.ts = s.ts; a.scales = s.scales; a.scales_t = s.scales_t;
		a.rotations = s.rotations; a.rotations_r = s.rotations_r; a.cov3D_precomp = s.cov3D_precomp;
		a.viewmatrix = s.viewmatrix; a.projmatrix = s.projmatrix; a.campos = s.campos;
		a.scale_modifier = s.scale_modifier; a.prefilter_var = s.prefilter_var;
		a.tan_fovx = s.tan_fovx; a.tan_fovy = s.tan_fovy;
		a.focal_y = s.H / (2.0f * s.tan_fovy); // rasterizer_impl.cu:235-236
		a.focal_x = s.W / (2.0f * s.tan_fovx);
		a.timestamp = s.timestamp; a.time_duration = s.time_duration;
		a.rot_4d = s.rot_4d; a.gaussian_dim = s.gaussian_dim; a.force_sh_3d = s.force_sh_3d; a.raw = s.raw_params;
		a.sh_vec_ok = (s.shs != nullptr && (reinterpret_cast<uintptr_t>(s.shs) & 15) == 0 && (3 * s.M) % 4 == 0) ? 1 : 0;
		a.grid_x = div_up(s.W, TILE_X); a.grid_y = div_up(s.H, TILE_Y);
		a.tile_cull = out.tile_cull != 0 ? 1 : 0;
		a.radii = out.radii; a.out_means3D = out.out_means3D; a.covs_com = out.covs_com;
		a.records = reinterpret_cast<float4*>(geom + L.records);
		a.depths = reinterpret_cast<float*>(geom + L.depths);
		a.cov3D = reinterpret_cast<float*>(geom + L.cov3D);
		a.tiles_touched = reinterpret_cast<uint32_t*>(geom + L.tiles_touched);
		a.rect = reinterpret_cast<ushort4*>(geom + L.rect);
		a.clamped = reinterpret_cast<uint8_t*>(geom + L.clamped);
		a.bin_counters = bin_counters;
		a.bin_counter_words = (int)bin_counter_words(a.grid_x * a.grid_y);
		if (part == 1) hipLaunchKernelGGL(preprocess_fwd_kernel<1>, dim3(div_up(s.P, 256)), dim3(256), 0, stream, a);
		else if (part == 2) hipLaunchKernelGGL(preprocess_fwd_kernel<2>, dim3(div_up(s.P, 256)), dim3(256), 0, stream, a);
		else hipLaunchKernelGGL(preprocess_fwd_kernel<0>, dim3(div_up(s.P, 256)), dim3(256), 0, stream, a);
		return hipGetLastError();
	}

	// ------------------------------------------------------------------------------------------------
	// SH colours of SEVERAL views in one pass over the coefficients (fdgs_preprocess_batch).
	// Within one optimizer step the parameters are constant, and the 12 M bytes of SH coefficients per Gaussian are the bulk of
	// what the preprocess reads (173 of ~200 MB per view at C3): the views' geometry runs per view (PART 1 above), their
	// colours here -- every coefficient block is staged through LDS ONCE and evaluated for each view's direction and timestamp.
	// Same device functions, same order of operations as PART 2: colours and clamp bits are bit-identical to the per-view path.
	// ------------------------------------------------------------------------------------------------
	constexpr int COLOUR_BATCH_MAX = 8;
	struct ColourBatchArgs
	{
		int P, D, D_t, M, nviews;
		const float *means3D, *shs, *ts;
		float time_duration;
		int gaussian_dim, force_sh_3d, sh_vec_ok;
		struct View
		{
			const float* campos; float timestamp;
			const int32_t* radii;       // of the view's geometry launch: colour only where radius > 0
			float4* records; uint8_t* clamped;
		} v[COLOUR_BATCH_MAX];
	};

	__global__ void __launch_bounds__(256) colour_batch_kernel(const ColourBatchArgs a)
	{
		__shared__ float s_sh[256 / WAVE][WAVE * SH_STRIDE];
		// the views' running colours of this thread's Gaussian: [view * 3 + channel][thread] (the view loop is a real loop -- unrolled
		// over 8 views the SH evaluation needs more registers than the file has -- so the per-view state lives here)
		__shared__ float s_c[COLOUR_BATCH_MAX * 3][256];
		const int tid_g = blockIdx.x * blockDim.x + threadIdx.x;
		const bool valid = tid_g < a.P;
		const int idx = valid ? tid_g : a.P - 1;   // every lane stays: the SH blocks are staged cooperatively per wave
		const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
		float* tile = s_sh[wave];
		const float* row = tile + lane * SH_STRIDE;
		const int g0 = blockIdx.x * blockDim.x + wave * WAVE;
		const bool sh3d = (a.gaussian_dim == 3 || a.force_sh_3d);
		const int ncoef0 = min(16, (a.D + 1) * (a.D + 1));
		const int nblocks = (!sh3d && a.D > 2) ? 1 + min(max(a.D_t, 0), 2) : 1;
		const float3 p_in = ld3(a.means3D, idx);   // Q4: the forward view direction uses the UN-shifted input mean (forward.cu:480-482)
		const float t_in = (!sh3d) ? a.ts[idx] : 0.f;

		bool any = false;
#pragma unroll 1
		for (int v = 0; v < a.nviews; v++) any = any || (valid && a.v[v].radii[idx] > 0);
		const unsigned long long amask = __ballot(any);   // rows of Gaussians no view kept are never fetched
		for (int blk = 0; blk < nblocks; blk++)
		{
			stage_sh_block(tile, a.shs, g0, a.P, a.M, 16 * blk, blk == 0 ? ncoef0 : 16, amask, lane, a.sh_vec_ok != 0);
			__syncthreads();
#pragma unroll 1
			for (int v = 0; v < a.nviews; v++)
			{
				if (!(valid && a.v[v].radii[idx] > 0)) continue;
				const float* cp = a.v[v].campos;
				float3 dir = sub3(p_in, make_float3(cp[0], cp[1], cp[2]));
				const float len = sqrtf(dot3(dir.x, dir.y, dir.z, dir.x, dir.y, dir.z));
				dir = make_float3(dir.x / len, dir.y / len, dir.z / len);
				float l[16];
				if (!sh3d) sh_basis_4d(a.D, dir.x, dir.y, dir.z, l);
				float3 c;
				if (blk == 0) c = sh3d ? sh_color_3d(a.D, row, dir) : sh4d_block0(a.D, l, row);
				else
				{
					const float dir_t = t_in - a.v[v].timestamp;
					const float tk = (blk == 1) ? (float)cos(2 * REF_PI * dir_t / a.time_duration)
					                            : (float)cos(2 * REF_PI * dir_t * 2 / a.time_duration);
					c = make_float3(s_c[3 * v][threadIdx.x], s_c[3 * v + 1][threadIdx.x], s_c[3 * v + 2][threadIdx.x]);
					c = add3(c, scl3(tk, sh_weighted(l, row, 0, 15, 0)));
				}
				if (blk == nblocks - 1)
				{
					if (!sh3d) c = make_float3(c.x + 0.5f, c.y + 0.5f, c.z + 0.5f); // sh_color_3d already added it
					float* rec = reinterpret_cast<float*>(a.v[v].records + 3 * (size_t)idx);
					rec[6] = fmaxf(c.x, 0.0f); rec[7] = fmaxf(c.y, 0.0f); rec[8] = fmaxf(c.z, 0.0f);
					a.v[v].clamped[idx] = (uint8_t)((c.x < 0 ? 1 : 0) | (c.y < 0 ? 2 : 0) | (c.z < 0 ? 4 : 0));
				}
				else { s_c[3 * v][threadIdx.x] = c.x; s_c[3 * v + 1][threadIdx.x] = c.y; s_c[3 * v + 2][threadIdx.x] = c.z; }
			}
			__syncthreads();
		}
	}

	// views[v] / geoms[v]: the scene and the geometry buffer of view v; all views share P, M, degrees and the Gaussian tensors
	hipError_t launch_colour_batch(int nviews, const fdgs_scene* const* views, const fdgs_forward_out* const* outs, char* const* geoms,
	                               hipStream_t stream)
	{
		const fdgs_scene& s = *views[0];
		if (s.P <= 0 || s.shs == nullptr) return hipSuccess;
		const GeomLayout L = geom_layout(s.P);
		for (int v0 = 0; v0 < nviews; v0 += COLOUR_BATCH_MAX)
		{
			ColourBatchArgs a;
			a.P = s.P; a.D = s.D; a.D_t = s.D_t; a.M = s.M; a.nviews = min(COLOUR_BATCH_MAX, nviews - v0);
			a.means3D = s.means3D; a.shs = s.shs; a.ts = s.ts; a.time_duration = s.time_duration;
			a.gaussian_dim = s.gaussian_dim; a.force_sh_3d = s.force_sh_3d;
			a.sh_vec_ok = ((reinterpret_cast<uintptr_t>(s.shs) & 15) == 0 && (3 * s.M) % 4 == 0) ? 1 : 0;
			for (int v = 0; v < COLOUR_BATCH_MAX; v++)
			{
				const int w = min(v0 + v, nviews - 1);
				a.v[v].campos = views[w]->campos; a.v[v].timestamp = views[w]->timestamp;
				a.v[v].radii = outs[w]->radii;
				a.v[v].records = reinterpret_cast<float4*>(geoms[w] + L.records);
				a.v[v].clamped = reinterpret_cast<uint8_t*>(geoms[w] + L.clamped);
			}
			hipLaunchKernelGGL(colour_batch_kernel, dim3(div_up(s.P, 256)), dim3(256), 0, stream, a);
		}
		return hipGetLastError();
	}

	// checkFrustum (rasterizer_impl.cu:54-67)
	__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ vm, uint8_t* present)
	{
		const int idx = blockIdx.x * blockDim.x + threadIdx.x;
		if (idx >= P) return;
		const float3 pv = xform4x3(ld3(means3D, idx), vm);
		present[idx] = !(pv.z <= 0.2f);
	}
	hipError_t launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t stream)
	{
		hipLaunchKernelGGL(mark_visible_kernel, dim3(div_up(P, 256)), dim3(256), 0, stream, P, means3D, viewmatrix, present);
		return hipGetLastError();
	}

	// Parity-test introspection (fdgs_debug_activations): the activated tensors preprocess derives from RAW parameters
	// when fdgs_scene.raw_params != 0, produced by the very same device functions in the same translation unit
	// (same compiler flags), so they are bit-identical to what the kernels above and preprocess_bwd compute in flight.
	__global__ void activations_kernel(int P, const float* __restrict__ opacity_raw, const float* __restrict__ scales_raw,
	                                   const float* __restrict__ scales_t_raw, const float* __restrict__ rot_raw,
	                                   const float* __restrict__ rot_r_raw, float* __restrict__ opacity, float* __restrict__ scales,
	                                   float* __restrict__ scales_t, float* __restrict__ rot, float* __restrict__ rot_r)
	{
		const int idx = blockIdx.x * blockDim.x + threadIdx.x;
		if (idx >= P) return;
		float unused;
		if (opacity_raw && opacity) opacity[idx] = act_sigmoid(opacity_raw[idx]);
		if (scales_raw && scales)
		{
			const float3 sc = ld3(scales_raw, idx);
			scales[3 * (size_t)idx + 0] = expf(sc.x); scales[3 * (size_t)idx + 1] = expf(sc.y); scales[3 * (size_t)idx + 2] = expf(sc.z);
		}
		if (scales_t_raw && scales_t) scales_t[idx] = expf(scales_t_raw[idx]);
		if (rot_raw && rot) reinterpret_cast<float4*>(rot)[idx] = act_normalize(reinterpret_cast<const float4*>(rot_raw)[idx], &unused);
		if (rot_r_raw && rot_r) reinterpret_cast<float4*>(rot_r)[idx] = act_normalize(reinterpret_cast<const float4*>(rot_r_raw)[idx], &unused);
	}
	hipError_t launch_activations(int P, const float* opacity_raw, const float* scales_raw, const float* scales_t_raw, const float* rot_raw,
	                              const float* rot_r_raw, float* opacity, float* scales, float* scales_t, float* rot, float* rot_r, hipStream_t stream)
	{
		hipLaunchKernelGGL(activations_kernel, dim3(div_up(P, 256)), dim3(256), 0, stream, P, opacity_raw, scales_raw, scales_t_raw,
		                   rot_raw, rot_r_raw, opacity, scales, scales_t, rot, rot_r);
		return hipGetLastError();
	}
}

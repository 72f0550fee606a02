// sh_bwd.hip -- SH / 4D-SH backward with coalesced HBM traffic (gfx950).
//
// The reference evaluates this inside its per-Gaussian backward kernel (computeColorFromSH / _4D,
// backward.cu:20-139, 144-481): every thread reads its own 12*M bytes of coefficients and writes its own
// 12*M bytes of dL_dsh with a 12*M-byte stride between threads.  At M = 48 that is 576 B in and 576 B out per
// Gaussian -- 80 % of all bytes the whole per-Gaussian backward moves -- so it gets its own kernel here:
//   * a wave owns 32 consecutive Gaussians (64 measured 1 % slower: half the waves in flight); their rows are staged block by block (16
//     coefficients = 192 B per Gaussian) through a wave-private LDS tile with fully coalesced dwordx4 loads;
//   * each lane then walks ITS Gaussian's row in LDS (row stride 49 floats: conflict free), accumulates the
//     direction / time dot products in registers and overwrites the row with dL_dsh;
//   * the tile is written back with coalesced dwordx4 stores (zeros for culled Gaussians, so dL_dsh never
//     needs a memset).
// The four numbers the geometry backward needs from here -- the mean gradient through the view direction
// (3) and the time gradient (1) -- travel in the spare words 12..15 of the Gaussian's packed accumulator record.
//
// One table-driven formulation serves both the 3D and the 4D path: basis values l[k] and their x/y/z
// derivatives for k < 16; blocks 1 and 2 reuse them times cos(2 pi j dt / T).  Bug-compatible with the
// reference: Q1 dL_dsh[1] = l[0] * dRGB in the 4D path, Q2 sign of d cos/dt,
// Q3 the last time block overwrites dRGB/dt, Q4 view direction from the SHIFTED mean.  fdgs_scene.analytic_sh_grad
// (opt-in) switches Q1-Q3 to the analytic gradient of the forward pass.
#pragma clang fp contract(off)
#include "fdgs_common.h"
#include "fdgs_math.h"

namespace fdgs
{
	constexpr int SHB_STRIDE = 49;   // LDS row stride in floats (odd: lane-per-row access is conflict free)
	constexpr int SHB_CH = 12;       // float4 chunks per Gaussian and block (16 coefficients x 3 / 4)
#ifndef FDGS_SHB_GPW
#define FDGS_SHB_GPW 32
#endif
	constexpr int SHB_GPW = FDGS_SHB_GPW;   // Gaussians per wave: 32 halves the LDS tile (6.3 KB) -> twice the waves per CU in flight
	constexpr int SHB_IT = SHB_CH * SHB_GPW / WAVE;  // float4 per lane and block

	struct ShBwdArgs
	{
		int P, D, D_t, M;
		const float *shs, *ts, *campos;
		float timestamp, time_duration;
		int gaussian_dim, force_sh_3d, vec_ok, accum, analytic;
		const int32_t* radii; const float* means; const uint8_t* clamped;
		float* gacc; float* dL_dsh;
		float4* stage;   // deferred mode: [P][2] = (dRGB.xyz, dir_t) (dir.xyz, 0) per Gaussian instead of dL_dsh (see sh_flush_kernel)
	};

	__device__ __forceinline__ float3 s_ld3(const float* p, int k) { return make_float3(p[3 * k], p[3 * k + 1], p[3 * k + 2]); }
	__device__ __forceinline__ float3 s_add(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
	__device__ __forceinline__ float3 s_scl(float s, float3 a) { return make_float3(s * a.x, s * a.y, s * a.z); }
	__device__ __forceinline__ float s_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

	// basis values and derivatives (backward.cu:172-263); entries the reference has no term for stay zero
	__device__ __forceinline__ void sh_tables(int deg, float x, float y, float z, bool promote, float* l, float* dX, float* dY, float* dZ)
	{
#pragma unroll
		for (int k = 0; k < 16; k++) { l[k] = 0.f; dX[k] = 0.f; dY[k] = 0.f; dZ[k] = 0.f; }
		l[0] = SH_C0;
		if (deg > 0)
		{
			l[1] = -1 * SH_C1 * y; l[2] = SH_C1 * z; l[3] = -1 * SH_C1 * x;
			dY[1] = -1 * SH_C1; dZ[2] = SH_C1; dX[3] = -1 * SH_C1;
			if (deg > 1)
			{
				const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
				l[4] = SH_C2[0] * xy; l[5] = SH_C2[1] * yz;
				l[6] = promote ? (float)(SH_C2[2] * (2.0 * zz - xx - yy)) : SH_C2[2] * (2.f * zz - xx - yy);
				l[7] = SH_C2[3] * xz; l[8] = SH_C2[4] * (xx - yy);
				dX[4] = SH_C2[0] * y; dY[4] = SH_C2[0] * x;
				dY[5] = SH_C2[1] * z; dZ[5] = SH_C2[1] * y;
				dX[6] = -2 * SH_C2[2] * x; dY[6] = -2 * SH_C2[2] * y; dZ[6] = 4 * SH_C2[2] * z;
				dX[7] = SH_C2[3] * z; dZ[7] = SH_C2[3] * x;
				dX[8] = 2 * SH_C2[4] * x; dY[8] = -2 * SH_C2[4] * y;
				if (deg > 2)
				{
					l[9] = SH_C3[0] * y * (3 * xx - yy);
					l[10] = SH_C3[1] * xy * z;
					l[11] = SH_C3[2] * y * (4 * zz - xx - yy);
					l[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
					l[13] = SH_C3[4] * x * (4 * zz - xx - yy);
					l[14] = SH_C3[5] * z * (xx - yy);
					l[15] = SH_C3[6] * x * (xx - 3 * yy);
					dX[9] = SH_C3[0] * y * 6 * x;               dY[9] = SH_C3[0] * (3 * xx - 3 * yy);
					dX[10] = SH_C3[1] * yz;                     dY[10] = SH_C3[1] * xz;                     dZ[10] = SH_C3[1] * xy;
					dX[11] = -SH_C3[2] * y * 2 * x;             dY[11] = SH_C3[2] * (4 * zz - xx - 3 * yy);  dZ[11] = SH_C3[2] * y * 8 * z;
					dX[12] = -SH_C3[3] * z * 6 * x;             dY[12] = -SH_C3[3] * z * 6 * y;             dZ[12] = SH_C3[3] * (6 * zz - 3 * xx - 3 * yy);
					dX[13] = SH_C3[4] * (4 * zz - 3 * xx - yy);  dY[13] = -SH_C3[4] * x * 2 * y;             dZ[13] = SH_C3[4] * x * 8 * z;
					dX[14] = SH_C3[5] * z * 2 * x;              dY[14] = -SH_C3[5] * z * 2 * y;             dZ[14] = SH_C3[5] * (xx - yy);
					dX[15] = SH_C3[6] * (3 * xx - 3 * yy);      dY[15] = -SH_C3[6] * x * 6 * y;
				}
			}
		}
	}

	// ---- tile <-> global, coalesced ----
	__device__ __forceinline__ void tile_load16(float* __restrict__ tile, const float* __restrict__ src, int g0, int P, size_t row_floats,
	                                            int first_float, unsigned long long mask, int lane)
	{
		float4 v[SHB_IT];
#pragma unroll
		for (int i = 0; i < SHB_IT; i++)
		{
			const int c = i * WAVE + lane, g = c / SHB_CH, q = c - g * SHB_CH;
			const bool ok = g0 + g < P && ((mask >> g) & 1ull);
			v[i] = ok ? *reinterpret_cast<const float4*>(src + (size_t)(g0 + g) * row_floats + first_float + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
#pragma unroll
		for (int i = 0; i < SHB_IT; i++)
		{
			const int c = i * WAVE + lane, g = c / SHB_CH, q = c - g * SHB_CH;
			float* d = tile + g * SHB_STRIDE + 4 * q;
			d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
		}
	}
	__device__ __forceinline__ void tile_store16(const float* __restrict__ tile, float* __restrict__ dst, int g0, int P, size_t row_floats,
	                                             int first_float, unsigned long long mask, int lane, bool accum)
	{
#pragma unroll
		for (int i = 0; i < SHB_IT; i++)
		{
			const int c = i * WAVE + lane, g = c / SHB_CH, q = c - g * SHB_CH;
			if (g0 + g < P)
			{
				const float* s = tile + g * SHB_STRIDE + 4 * q;
				const bool live = (mask >> g) & 1ull;
				float4* d = reinterpret_cast<float4*>(dst + (size_t)(g0 + g) * row_floats + first_float + 4 * q);
				float4 v = live ? make_float4(s[0], s[1], s[2], s[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
				if (accum)
				{
					if (!live) continue; // adding zero
					const float4 o = *d;
					v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
				}
				*d = v;
			}
		}
	}
	// generic (any float count per row / alignment)
	__device__ __forceinline__ void tile_load_any(float* __restrict__ tile, const float* __restrict__ src, int g0, int P, size_t row_floats,
	                                              int first_float, int nf, unsigned long long mask, int lane)
	{
		int g = lane / nf, pos = lane - g * nf;
		const int dg = WAVE / nf, dpos = WAVE - dg * nf;
		for (int e = lane; e < SHB_GPW * nf; e += WAVE)
		{
			if (g0 + g < P && ((mask >> g) & 1ull)) tile[g * SHB_STRIDE + pos] = src[(size_t)(g0 + g) * row_floats + first_float + pos];
			g += dg; pos += dpos;
			if (pos >= nf) { pos -= nf; g++; }
		}
	}
	__device__ __forceinline__ void tile_store_any(const float* __restrict__ tile, float* __restrict__ dst, int g0, int P, size_t row_floats,
	                                               int first_float, int nf, unsigned long long mask, int lane, bool accum)
	{
		int g = lane / nf, pos = lane - g * nf;
		const int dg = WAVE / nf, dpos = WAVE - dg * nf;
		for (int e = lane; e < SHB_GPW * nf; e += WAVE)
		{
			if (g0 + g < P)
			{
				float* d = dst + (size_t)(g0 + g) * row_floats + first_float + pos;
				const bool live = (mask >> g) & 1ull;
				if (!accum) *d = live ? tile[g * SHB_STRIDE + pos] : 0.f;
				else if (live) *d += tile[g * SHB_STRIDE + pos];
			}
			g += dg; pos += dpos;
			if (pos >= nf) { pos -= nf; g++; }
		}
	}

	// One wave per workgroup: the tile is wave-private, so no workgroup barrier is needed anywhere (LDS
	// operations of one wave execute in order) and waves of different phases (load / compute / store) overlap freely.
	template <bool STAGE>
	__global__ void __launch_bounds__(WAVE) sh_bwd_kernel(const ShBwdArgs a)
	{
		__shared__ float tile[SHB_GPW * SHB_STRIDE];
		const int lane = threadIdx.x;
		float* row = tile + (lane < SHB_GPW ? lane : 0) * SHB_STRIDE;
		const int g0 = blockIdx.x * SHB_GPW;
		const int tid_g = g0 + lane;
		const bool valid = lane < SHB_GPW && tid_g < a.P;
		const int idx = valid ? tid_g : a.P - 1;
		const bool visible = valid && a.radii[idx] > 0; // backward.cu:873
		const size_t row_floats = (size_t)3 * a.M;

		const bool sh3d = (a.gaussian_dim == 3 || a.force_sh_3d);
		const int ncoef0 = min(16, (a.D + 1) * (a.D + 1));
		const int nblocks = (!sh3d && a.D > 2) ? 1 + min(max(a.D_t, 0), 2) : 1;

		// per-Gaussian prologue
		const float3 campos = make_float3(a.campos[0], a.campos[1], a.campos[2]);
		const float3 mean = make_float3(a.means[3 * (size_t)idx], a.means[3 * (size_t)idx + 1], a.means[3 * (size_t)idx + 2]);
		const float3 dir_orig = make_float3(mean.x - campos.x, mean.y - campos.y, mean.z - campos.z); // Q4: shifted mean
		const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
		const float3 dir = make_float3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
		float3 dRGB = make_float3(a.gacc[(size_t)idx * GRAD_ACC_WORDS + 0], a.gacc[(size_t)idx * GRAD_ACC_WORDS + 1], a.gacc[(size_t)idx * GRAD_ACC_WORDS + 2]);
		const uint8_t cl = a.clamped[idx];
		if (cl & 1) dRGB.x = 0.f;   // clamped channels get no gradient (backward.cu:158-161)
		if (cl & 2) dRGB.y = 0.f;
		if (cl & 4) dRGB.z = 0.f;
		// Every output of this kernel is linear in dRGB.  A Gaussian that is visible but contributed to no pixel
		// (occluded, or too faint everywhere: 58 % of the visible ones on the C3 workload) has dRGB == 0 exactly:
		// its coefficients are not read and its gradient row is zero (not touched at all when accumulating).
		const bool live = visible && (dRGB.x != 0.f || dRGB.y != 0.f || dRGB.z != 0.f);
		const unsigned long long vmask = __ballot(live);
		const float dir_t = sh3d ? 0.f : a.ts[idx] - a.timestamp;
		float l[16], dX[16], dY[16], dZ[16];
		sh_tables(a.D, dir.x, dir.y, dir.z, !sh3d, l, dX, dY, dZ);

		float3 gx = make_float3(0.f, 0.f, 0.f), gy = gx, gz = gx, gt = gx;
		for (int blk = 0; blk < nblocks; blk++)
		{
			const int nk = (blk == 0) ? ncoef0 : 16;
			const int first_float = 48 * blk;
			const bool vec = a.vec_ok && nk == 16;
			if (vec) tile_load16(tile, a.shs, g0, a.P, row_floats, first_float, vmask, lane);
			else tile_load_any(tile, a.shs, g0, a.P, row_floats, first_float, 3 * nk, vmask, lane);
			__builtin_amdgcn_wave_barrier();
			if (live)
			{
				float tk = 1.f, dtk_dt = 0.f;
				if (blk == 1)
				{
					tk = (float)cos(2 * REF_PI * dir_t / a.time_duration);
					dtk_dt = (float)(sin(2 * REF_PI * dir_t / a.time_duration) * 2 * REF_PI / a.time_duration); // Q2
				}
				else if (blk == 2)
				{
					tk = (float)cos(2 * REF_PI * dir_t * 2 / a.time_duration);
					dtk_dt = (float)(sin(2 * REF_PI * dir_t * 2 / a.time_duration) * 2 * REF_PI * 2 / a.time_duration);
				}
				if (a.analytic) dtk_dt = -dtk_dt;   // d cos(u) / dt = -sin(u) du/dt
				float3 st = make_float3(0.f, 0.f, 0.f), sx = st, sy = st, sz = st;
#pragma unroll
				for (int k = 0; k < 16; k++)   // fully unrolled: the tables stay in registers (no dynamic indexing)
				{
					if (k >= nk) break;
					const float3 s = s_ld3(row, k);
					float basis = l[k];
					if (blk == 0 && k == 1 && !sh3d && !a.analytic) basis = l[0]; // Q1
					if (!STAGE)
					{
						const float3 d = s_scl(blk == 0 ? basis : tk * basis, dRGB);
						row[3 * k] = d.x; row[3 * k + 1] = d.y; row[3 * k + 2] = d.z;
					}
					st = s_add(st, s_scl(l[k], s));
					sx = s_add(sx, s_scl(dX[k], s));
					sy = s_add(sy, s_scl(dY[k], s));
					sz = s_add(sz, s_scl(dZ[k], s));
				}
				if (blk == 0) { gx = sx; gy = sy; gz = sz; }
				else
				{
					gx = s_add(gx, s_scl(tk, sx)); gy = s_add(gy, s_scl(tk, sy)); gz = s_add(gz, s_scl(tk, sz));
					gt = a.analytic ? s_add(gt, s_scl(dtk_dt, st)) : s_scl(dtk_dt, st); // Q3: overwrite, not accumulate
				}
			}
			__builtin_amdgcn_wave_barrier();
			if (!STAGE)
			{
				if (vec) tile_store16(tile, a.dL_dsh, g0, a.P, row_floats, first_float, vmask, lane, a.accum != 0);
				else tile_store_any(tile, a.dL_dsh, g0, a.P, row_floats, first_float, 3 * nk, vmask, lane, a.accum != 0);
				__builtin_amdgcn_wave_barrier();
			}
		}
		if (STAGE)
		{
			// what the flush needs to rebuild this view's contribution basis(dir, dir_t) x dRGB to dL_dsh
			if (valid)
			{
				a.stage[2 * (size_t)idx] = live ? make_float4(dRGB.x, dRGB.y, dRGB.z, dir_t) : make_float4(0.f, 0.f, 0.f, 0.f);
				a.stage[2 * (size_t)idx + 1] = make_float4(dir.x, dir.y, dir.z, a.analytic ? 1.f : 0.f);
			}
		}
		// coefficients above the active degree get a zero gradient (nothing to add when accumulating)
		else if (!a.accum)
		{
			const int written = (nblocks - 1) * 48 + 3 * (nblocks > 1 ? 16 : ncoef0);
			const int rest = (int)row_floats - written;
			if (rest > 0) tile_store_any(tile, a.dL_dsh, g0, a.P, row_floats, written, rest > 48 ? 48 : rest, 0ull, lane, false);
			for (int done = 48; done < rest; done += 48)
				tile_store_any(tile, a.dL_dsh, g0, a.P, row_floats, written + done, min(48, rest - done), 0ull, lane, false);
		}
		if (valid)
		{
			float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
			if (live)
			{
				const float3 ddir = make_float3(s_dot(gx, dRGB), s_dot(gy, dRGB), s_dot(gz, dRGB));
				// dnormvdv, auxiliary.h:108-118
				const float3 v = dir_orig;
				const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
				const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
				o.x = ((+sum2 - v.x * v.x) * ddir.x - v.y * v.x * ddir.y - v.z * v.x * ddir.z) * invsum32;
				o.y = (-v.x * v.y * ddir.x + (sum2 - v.y * v.y) * ddir.y - v.z * v.y * ddir.z) * invsum32;
				o.z = (-v.x * v.z * ddir.x - v.y * v.z * ddir.y + (sum2 - v.z * v.z) * ddir.z) * invsum32;
				o.w = sh3d ? 0.f : s_dot(gt, dRGB);
			}
			reinterpret_cast<float4*>(a.gacc + (size_t)idx * GRAD_ACC_WORDS)[3] = o;
		}
	}

	hipError_t launch_sh_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out, const char* geom, hipStream_t stream)
	{
		if (s.shs == nullptr || s.M <= 0) return hipSuccess;
		const GeomLayout L = geom_layout(s.P);
		ShBwdArgs a;
		a.P = s.P; a.D = s.D; a.D_t = s.D_t; a.M = s.M;
		a.shs = s.shs; a.ts = s.ts; a.campos = s.campos;
		a.timestamp = s.timestamp; a.time_duration = s.time_duration;
		a.gaussian_dim = s.gaussian_dim; a.force_sh_3d = s.force_sh_3d; a.analytic = s.analytic_sh_grad;
		a.vec_ok = ((reinterpret_cast<uintptr_t>(s.shs) & 15) == 0 && (reinterpret_cast<uintptr_t>(out.dL_dsh) & 15) == 0 && (3 * s.M) % 4 == 0) ? 1 : 0;
		a.radii = in.radii; a.means = in.out_means3D;
		a.clamped = reinterpret_cast<const uint8_t*>(geom + L.clamped);
		a.gacc = out.grad_accum; a.dL_dsh = out.dL_dsh; a.accum = out.accumulate;
		a.stage = reinterpret_cast<float4*>(out.sh_stage);
		if (out.sh_stage) hipLaunchKernelGGL(sh_bwd_kernel<true>, dim3(div_up(s.P, SHB_GPW)), dim3(WAVE), 0, stream, a);
		else hipLaunchKernelGGL(sh_bwd_kernel<false>, dim3(div_up(s.P, SHB_GPW)), dim3(WAVE), 0, stream, a);
		return hipGetLastError();
	}

	// ------------------------------------------------------------------------------------------------
	// Deferred SH gradient (gradient accumulation over the views of one optimizer step).
	// dL_dsh of a view is basis(view direction, time) (x) dL_dRGB: M x 3 floats written (or read-modified-written, from the
	// second view on) per Gaussian and view, although the view only contributes 7 numbers.  In deferred mode
	// (fdgs_backward_out.sh_stage) sh_bwd_kernel stores those 7 numbers per view and this kernel, once per step, rebuilds
	// the views' contributions in registers, sums them in view order -- the same additions in the same order as the
	// accumulating path, so the result is bit-identical -- and writes dL_dsh ONCE: at C3 / 4 views 1.7 KB of traffic per
	// live Gaussian and view become 0.6 KB + 0.6 KB / 4.
	// ------------------------------------------------------------------------------------------------
	struct ShFlushArgs
	{
		int P, D, D_t, M, nviews, sh3d, vec_ok, accum;
		float time_duration;
		const float4* stages;   // [nviews][P][2]
		float* dL_dsh;
	};

	// basis values only (sh_tables without the derivative tables)
	__device__ __forceinline__ void sh_values(int deg, float x, float y, float z, bool promote, float* l)
	{
		float dX[16], dY[16], dZ[16];
		sh_tables(deg, x, y, z, promote, l, dX, dY, dZ);
	}

	__global__ void __launch_bounds__(WAVE) sh_flush_kernel(const ShFlushArgs a)
	{
		__shared__ float tile[SHB_GPW * SHB_STRIDE];
		const int lane = threadIdx.x;
		float* row = tile + (lane < SHB_GPW ? lane : 0) * SHB_STRIDE;
		const int g0 = blockIdx.x * SHB_GPW;
		const int tid_g = g0 + lane;
		const bool valid = lane < SHB_GPW && tid_g < a.P;
		const int idx = valid ? tid_g : a.P - 1;
		const size_t row_floats = (size_t)3 * a.M;
		const bool sh3d = a.sh3d != 0;
		const int ncoef0 = min(16, (a.D + 1) * (a.D + 1));
		const int nblocks = (!sh3d && a.D > 2) ? 1 + min(max(a.D_t, 0), 2) : 1;
		bool any = false;
		for (int v = 0; v < a.nviews; v++)
		{
			const float4 s0 = a.stages[2 * ((size_t)v * a.P + idx)];
			any = any || (valid && (s0.x != 0.f || s0.y != 0.f || s0.z != 0.f));
		}
		const unsigned long long vmask = __ballot(any);
		for (int blk = 0; blk < nblocks; blk++)
		{
			const int nk = (blk == 0) ? ncoef0 : 16;
			const int first_float = 48 * blk;
			const bool vec = a.vec_ok && nk == 16;
			if (any)
			{
				// the wave-private LDS row is the accumulator: the first live view writes, the following ones add (same
				// additions, same order as backward calls accumulating into dL_dsh view after view)
				bool first = true;
				for (int v = 0; v < a.nviews; v++)
				{
					const float4 s0 = a.stages[2 * ((size_t)v * a.P + idx)], s1 = a.stages[2 * ((size_t)v * a.P + idx) + 1];
					const float3 dRGB = make_float3(s0.x, s0.y, s0.z);
					if (dRGB.x == 0.f && dRGB.y == 0.f && dRGB.z == 0.f) continue;   // this view added nothing
					float l[16];
					sh_values(a.D, s1.x, s1.y, s1.z, !sh3d, l);
					const float dir_t = s0.w;
					float tk = 1.f;
					if (blk == 1) tk = (float)cos(2 * REF_PI * dir_t / a.time_duration);
					else if (blk == 2) tk = (float)cos(2 * REF_PI * dir_t * 2 / a.time_duration);
#pragma unroll
					for (int k = 0; k < 16; k++)
					{
						if (k < nk)
						{
							float basis = l[k];
							if (blk == 0 && k == 1 && !sh3d && s1.w == 0.f) basis = l[0]; // Q1 (s1.w: the view ran with analytic_sh_grad)
							float3 d = s_scl(blk == 0 ? basis : tk * basis, dRGB);
							if (!first) d = s_add(s_ld3(row, k), d);
							row[3 * k] = d.x; row[3 * k + 1] = d.y; row[3 * k + 2] = d.z;
						}
					}
					first = false;
				}
			}
			__builtin_amdgcn_wave_barrier();
			if (vec) tile_store16(tile, a.dL_dsh, g0, a.P, row_floats, first_float, vmask, lane, a.accum != 0);
			else tile_store_any(tile, a.dL_dsh, g0, a.P, row_floats, first_float, 3 * nk, vmask, lane, a.accum != 0);
			__builtin_amdgcn_wave_barrier();
		}
		if (!a.accum)
		{
			const int written = (nblocks - 1) * 48 + 3 * (nblocks > 1 ? 16 : ncoef0);
			const int rest = (int)row_floats - written;
			if (rest > 0) tile_store_any(tile, a.dL_dsh, g0, a.P, row_floats, written, rest > 48 ? 48 : rest, 0ull, lane, false);
			for (int done = 48; done < rest; done += 48)
				tile_store_any(tile, a.dL_dsh, g0, a.P, row_floats, written + done, min(48, rest - done), 0ull, lane, false);
		}
	}

	hipError_t launch_sh_flush(int P, int D, int D_t, int M, int gaussian_dim, int force_sh_3d, float time_duration, int nviews,
	                           const float* stages, float* dL_dsh, int accumulate, hipStream_t stream)
	{
		if (P <= 0 || M <= 0 || nviews <= 0) return hipSuccess;
		ShFlushArgs a;
		a.P = P; a.D = D; a.D_t = D_t; a.M = M; a.nviews = nviews;
		a.sh3d = (gaussian_dim == 3 || force_sh_3d) ? 1 : 0;
		a.vec_ok = ((reinterpret_cast<uintptr_t>(dL_dsh) & 15) == 0 && (3 * M) % 4 == 0) ? 1 : 0;
		a.accum = accumulate; a.time_duration = time_duration;
		a.stages = reinterpret_cast<const float4*>(stages);
		a.dL_dsh = dL_dsh;
		hipLaunchKernelGGL(sh_flush_kernel, dim3(div_up(P, SHB_GPW)), dim3(WAVE), 0, stream, a);
		return hipGetLastError();
	}
}

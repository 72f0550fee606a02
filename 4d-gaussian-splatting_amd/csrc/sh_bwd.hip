// sh_bwd.hip -- SH / 4D-SH backward with coalesced HBM traffic (gfx950).
//
// The reference evaluates this inside its per-Gaussian backward kernel (computeColorFromSH / _4D,
// backward.cu:20-139, 144-481): every thread reads its own 12*M bytes of coefficients and writes its own
// 12*M bytes of dL_dsh with a 12*M-byte stride between threads.  At M = 48 that is 576 B in and 576 B out per
// Gaussian -- 80 % of all bytes the whole per-Gaussian backward moves -- so it gets its own kernel here:
//   * every output of the kernel is linear in the Gaussian's dL_dRGB, and a Gaussian that is visible but contributed to no
//     pixel (occluded, or too faint everywhere: 58 % of the visible ones on C3) has dL_dRGB == 0 exactly.  A wave walks
//     its range of consecutive Gaussians 64 at a time and ballot-compacts the LIVE ones (those with a colour gradient) into a list;
//     the others only get their zeros (the stage record / the dL_dsh row; their accumulator words 12..15 are zero already);
//   * whenever 64 live Gaussians are waiting (and at the end of the range) they are evaluated, one per lane: their rows are staged block by block (16 coefficients
//     = 192 B per Gaussian) through a wave-private LDS tile with coalesced dwordx4 loads (gathered rows, contiguous runs);
//   * each lane walks ITS Gaussian's row in LDS (row stride 49 floats: conflict free), accumulates the
//     direction / time dot products in registers and overwrites the row with dL_dsh;
//   * the tile is written back with coalesced dwordx4 stores.
//   (One lane per Gaussian of the span instead -- the first version -- left 13 of 64 lanes with work in the evaluation.)
// The four numbers the geometry backward needs from here -- the mean gradient through the view direction
// (3) and the time gradient (1) -- travel in the spare words 12..15 of the Gaussian's packed accumulator record.
//
// One table-driven formulation serves both the 3D and the 4D path: basis values l[k] and their x/y/z
// derivatives for k < 16; blocks 1 and 2 reuse them times cos(2 pi j dt / T).  Bug-compatible with the
// reference: Q1 dL_dsh[1] = l[0] * dRGB in the 4D path, Q2 sign of d cos/dt,
// Q3 the last time block overwrites dRGB/dt, Q4 view direction from the SHIFTED mean.  fdgs_scene.analytic_sh_grad
// (opt-in) switches Q1-Q3 to the analytic gradient of the forward pass.
#pragma clang fp contract(off)
#include <algorithm>
#include "fdgs_common.h"
#include "fdgs_math.h"

namespace fdgs
{
	constexpr int SHB_STRIDE = 49;   // LDS row stride in floats (odd: lane-per-row access is conflict free)
	constexpr int SHB_CH = 12;       // float4 chunks per Gaussian and block (16 coefficients x 3 / 4)
	constexpr int SHB_ROWS = WAVE;   // live Gaussians evaluated per round, one per lane (tile: 12.5 KB)

	struct ShBwdArgs
	{
		int P, D, D_t, M;
		const float *shs, *ts, *campos;
		float timestamp, time_duration;
		int gaussian_dim, force_sh_3d, vec_ok, accum, analytic;
		const int32_t* radii; const float* means; const uint8_t* clamped;
		float* gacc; float* dL_dsh;
		float4* stage;   // deferred mode: [P][2] = (dRGB.xyz, cos factor of time block 1) (dir.xyz, cos factor of block 2) per Gaussian instead of dL_dsh (see sh_flush_kernel)
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

	// ---- tile <-> global, coalesced; tile row r belongs to Gaussian g0 + list[r], r < nrows ----
	constexpr int SHB_BATCH = 6;     // float4 per lane in flight while staging (2 batches per block)
	__device__ __forceinline__ void rows_load16(float* __restrict__ tile, const float* __restrict__ src, const uint32_t* list, int nrows,
	                                            int g0, size_t row_floats, int first_float, int lane)
	{
#pragma unroll
		for (int b = 0; b < SHB_CH * SHB_ROWS / WAVE / SHB_BATCH; b++)
		{
			float4 v[SHB_BATCH];
#pragma unroll
			for (int i = 0; i < SHB_BATCH; i++)
			{
				const int c = (b * SHB_BATCH + i) * WAVE + lane, r = c / SHB_CH, q = c - r * SHB_CH;
				v[i] = r < nrows ? *reinterpret_cast<const float4*>(src + (size_t)(g0 + list[r]) * row_floats + first_float + 4 * q)
				                 : make_float4(0.f, 0.f, 0.f, 0.f);
			}
#pragma unroll
			for (int i = 0; i < SHB_BATCH; i++)
			{
				const int c = (b * SHB_BATCH + i) * WAVE + lane, r = c / SHB_CH, q = c - r * SHB_CH;
				float* d = tile + r * SHB_STRIDE + 4 * q;
				d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
			}
		}
	}
	__device__ __forceinline__ void rows_store16(const float* __restrict__ tile, float* __restrict__ dst, const uint32_t* list, int nrows,
	                                             int g0, size_t row_floats, int first_float, int lane, bool accum)
	{
#pragma unroll
		for (int i = 0; i < SHB_CH * SHB_ROWS / WAVE; i++)
		{
			const int c = i * WAVE + lane, r = c / SHB_CH, q = c - r * SHB_CH;
			if (r < nrows)
			{
				const float* t = tile + r * SHB_STRIDE + 4 * q;
				float4* d = reinterpret_cast<float4*>(dst + (size_t)(g0 + list[r]) * row_floats + first_float + 4 * q);
				float4 v = make_float4(t[0], t[1], t[2], t[3]);
				if (accum)
				{
					const float4 o = *d;
					v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
				}
				*d = v;
			}
		}
	}
	// generic (any float count per row / alignment)
	__device__ __forceinline__ void rows_load_any(float* __restrict__ tile, const float* __restrict__ src, const uint32_t* list, int nrows,
	                                              int g0, size_t row_floats, int first_float, int nf, int lane)
	{
		int r = lane / nf, pos = lane - r * nf;
		const int dr = WAVE / nf, dpos = WAVE - dr * nf;
		for (int e = lane; e < nrows * nf; e += WAVE)
		{
			tile[r * SHB_STRIDE + pos] = src[(size_t)(g0 + list[r]) * row_floats + first_float + pos];
			r += dr; pos += dpos;
			if (pos >= nf) { pos -= nf; r++; }
		}
	}
	__device__ __forceinline__ void rows_store_any(const float* __restrict__ tile, float* __restrict__ dst, const uint32_t* list, int nrows,
	                                               int g0, size_t row_floats, int first_float, int nf, int lane, bool accum)
	{
		int r = lane / nf, pos = lane - r * nf;
		const int dr = WAVE / nf, dpos = WAVE - dr * nf;
		for (int e = lane; e < nrows * nf; e += WAVE)
		{
			float* d = dst + (size_t)(g0 + list[r]) * row_floats + first_float + pos;
			if (accum) *d += tile[r * SHB_STRIDE + pos];
			else *d = tile[r * SHB_STRIDE + pos];
			r += dr; pos += dpos;
			if (pos >= nf) { pos -= nf; r++; }
		}
	}

	// dL_dRGB of a Gaussian as the blend backward left it, clamped channels zeroed (backward.cu:158-161)
	__device__ __forceinline__ float3 colour_gradient(const ShBwdArgs& a, int idx)
	{
		const float4 w = *reinterpret_cast<const float4*>(a.gacc + (size_t)idx * GRAD_ACC_WORDS);
		float3 dRGB = make_float3(w.x, w.y, w.z);
		const uint8_t cl = a.clamped[idx];
		if (cl & 1) dRGB.x = 0.f;
		if (cl & 2) dRGB.y = 0.f;
		if (cl & 4) dRGB.z = 0.f;
		return dRGB;
	}

	// One wave per workgroup: the tile is wave-private, so no workgroup barrier is needed anywhere (LDS
	// operations of one wave execute in order) and waves of different phases (load / compute / store) overlap freely.
	// The launch is 1.5 waves per wave slot the device has for this kernel (launcher), wave w takes the CHUNKS of 64 Gaussians
	// w, w + G, w + 2 G, ... (G = waves of the launch): chunks that far apart have nothing to do with each other even when
	// the model is stored in spatial order (where whole neighbourhoods are live or dead), so every wave gets about the same
	// number of live Gaussians.  It collects them and evaluates whenever 64 are waiting (and what is left at the end): the
	// evaluation's latency chain is paid per batch, and most waves need one.  (Fixed spans of 128 Gaussians per wave, the
	// previous version, made 1.14 rounds of two batches each, 64 + 9 rows.)
	// (A ticket counter handing out the chunks was built and measured at 0.19 ms: ~6700 returning atomics on one address.)
	template <bool STAGE>
	__global__ void __launch_bounds__(WAVE) sh_bwd_kernel(const ShBwdArgs a)
	{
		__shared__ float tile[SHB_ROWS * SHB_STRIDE];
		__shared__ uint32_t s_list[2 * WAVE];     // the live Gaussians waiting for evaluation
		const int lane = threadIdx.x;
		const size_t row_floats = (size_t)3 * a.M;
		const bool sh3d = (a.gaussian_dim == 3 || a.force_sh_3d);
		const int ncoef0 = min(16, (a.D + 1) * (a.D + 1));
		const int nblocks = (!sh3d && a.D > 2) ? 1 + min(max(a.D_t, 0), 2) : 1;
		const unsigned long long lt_mask = (1ull << lane) - 1ull;
		const float3 campos = make_float3(a.campos[0], a.campos[1], a.campos[2]);
		const int nchunks = (a.P + WAVE - 1) / WAVE;

		int n = 0;      // live Gaussians waiting in s_list
		int next = blockIdx.x;
		bool more = true;
		while (more || n > 0)
		{
			if (more)
			{
				const int base = next * WAVE;
				if (next >= nchunks) { more = false; if (n == 0) break; }
				else
				{
				next += gridDim.x;
				const int g_end = a.P;
				// ---- which of the next 64 Gaussians carry a colour gradient (visible: backward.cu:873) ----
				const int idx = base + lane;
				const bool valid = idx < g_end;
				bool live = false;
				if (valid && a.radii[idx] > 0)
				{
					const float3 dRGB = colour_gradient(a, idx);
					live = dRGB.x != 0.f || dRGB.y != 0.f || dRGB.z != 0.f;
				}
				const unsigned long long live_mask = __ballot(live);
				if (live) s_list[n + __popcll(live_mask & lt_mask)] = (uint32_t)idx;
				n += __popcll(live_mask);
				// a Gaussian without a colour gradient: all of its outputs are zero.  Record words 12..15 already are (the record is
				// all zero when the blend backward starts and nobody else writes them); dL_dsh: below; the flush looks at dRGB only
				if (STAGE && valid && !live) a.stage[2 * (size_t)idx] = make_float4(0.f, 0.f, 0.f, 0.f);
				__builtin_amdgcn_wave_barrier();

				if (!STAGE && !a.accum)
				{
					// dL_dsh is fully written by this call: zero rows for the Gaussians without a colour gradient, zeros beyond the
					// active degrees for the others (nothing to add when accumulating)
					const int written = (nblocks - 1) * 48 + 3 * (nblocks > 1 ? 16 : ncoef0);
					const int span = min(WAVE, g_end - base);
					if (a.vec_ok && (written & 3) == 0)
					{
						const int RC = (int)row_floats / 4, total = span * RC;
						const int dg = WAVE / RC, dq = WAVE - dg * RC;
						int g = lane / RC, q = lane - g * RC;
						for (int c = lane; c < total; c += WAVE)
						{
							const bool lv = (live_mask >> g) & 1ull;
							if (!lv || 4 * q >= written)
								*reinterpret_cast<float4*>(a.dL_dsh + (size_t)(base + g) * row_floats + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
							g += dg; q += dq;
							if (q >= RC) { q -= RC; g++; }
						}
					}
					else
					{
						const int rf = (int)row_floats, total = span * rf;
						const int dg = WAVE / rf, dpos = WAVE - dg * rf;
						int g = lane / rf, pos = lane - g * rf;
						for (int e = lane; e < total; e += WAVE)
						{
							const bool lv = (live_mask >> g) & 1ull;
							if (!lv || pos >= written) a.dL_dsh[(size_t)(base + g) * row_floats + pos] = 0.f;
							g += dg; pos += dpos;
							if (pos >= rf) { pos -= rf; g++; }
						}
					}
				}
				if (n < SHB_ROWS) continue;   // keep collecting
				}
			}

			// ---- up to 64 waiting live Gaussians, one per lane ----
			{
			constexpr int g0 = 0;    // the list holds Gaussian indices
			const int nrows = min(SHB_ROWS, n);
			const uint32_t* list = s_list;
			const bool live = lane < nrows;
			const int idx = g0 + (int)list[live ? lane : 0];
			float* row = tile + lane * SHB_STRIDE;
			// per-Gaussian prologue
			const float3 mean = make_float3(a.means[3 * (size_t)idx], a.means[3 * (size_t)idx + 1], a.means[3 * (size_t)idx + 2]);
			const float3 dir_orig = make_float3(mean.x - campos.x, mean.y - campos.y, mean.z - campos.z); // Q4: shifted mean
			const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
			const float3 dir = make_float3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
			const float3 dRGB = colour_gradient(a, idx);
			const float dir_t = sh3d ? 0.f : a.ts[idx] - a.timestamp;
			float l[16], dX[16], dY[16], dZ[16];
			sh_tables(a.D, dir.x, dir.y, dir.z, !sh3d, l, dX, dY, dZ);

			float3 gx = make_float3(0.f, 0.f, 0.f), gy = gx, gz = gx, gt = gx;
			float tk_stage[2] = { 0.f, 0.f };   // cos factors of the two time blocks (deferred mode hands them to the flush)
			for (int blk = 0; blk < nblocks; blk++)
			{
				const int nk = (blk == 0) ? ncoef0 : 16;
				const int first_float = 48 * blk;
				const bool vec = a.vec_ok && nk == 16;
				if (vec) rows_load16(tile, a.shs, list, nrows, g0, row_floats, first_float, lane);
				else rows_load_any(tile, a.shs, list, nrows, g0, row_floats, first_float, 3 * nk, lane);
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
					if (blk == 1) tk_stage[0] = tk;
					if (blk == 2) tk_stage[1] = tk;
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
					if (vec) rows_store16(tile, a.dL_dsh, list, nrows, g0, row_floats, first_float, lane, a.accum != 0);
					else rows_store_any(tile, a.dL_dsh, list, nrows, g0, row_floats, first_float, 3 * nk, lane, a.accum != 0);
					__builtin_amdgcn_wave_barrier();
				}
			}
			if (live)
			{
				if (STAGE)
				{
					// what the flush needs to rebuild this view's contribution basis(dir) x time factor x dRGB to dL_dsh
					a.stage[2 * (size_t)idx] = make_float4(dRGB.x, dRGB.y, dRGB.z, tk_stage[0]);
					a.stage[2 * (size_t)idx + 1] = make_float4(dir.x, dir.y, dir.z, tk_stage[1]);
				}
				float4 o;
				const float3 ddir = make_float3(s_dot(gx, dRGB), s_dot(gy, dRGB), s_dot(gz, dRGB));
				// dnormvdv, auxiliary.h:108-118
				const float3 v = dir_orig;
				const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
				const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
				o.x = ((+sum2 - v.x * v.x) * ddir.x - v.y * v.x * ddir.y - v.z * v.x * ddir.z) * invsum32;
				o.y = (-v.x * v.y * ddir.x + (sum2 - v.y * v.y) * ddir.y - v.z * v.y * ddir.z) * invsum32;
				o.z = (-v.x * v.z * ddir.x - v.y * v.z * ddir.y + (sum2 - v.z * v.z) * ddir.z) * invsum32;
				o.w = sh3d ? 0.f : s_dot(gt, dRGB);
				reinterpret_cast<float4*>(a.gacc + (size_t)idx * GRAD_ACC_WORDS)[3] = o;
			}
			// the Gaussians that did not fit into this batch move to the front of the list
			const int left = n - nrows;
			const uint32_t moved = lane < left ? s_list[nrows + lane] : 0u;
			__builtin_amdgcn_wave_barrier();
			if (lane < left) s_list[lane] = moved;
			n = left;
			__builtin_amdgcn_wave_barrier();
			}
		}
	}

	// wave slots of the device for a one-wave kernel (occupancy x CUs), per device
	template <typename K>
	static int resident_waves(K kernel)
	{
		int dev = 0, per_cu = 0, cus = 0;
		if (hipGetDevice(&dev) != hipSuccess) return 2048;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, WAVE, 0) != hipSuccess || per_cu <= 0) per_cu = 8;
		if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
		return per_cu * cus;
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
		// one wave per wave slot the device has for this kernel (cached per device)
		constexpr int MAXDEV = 64;
		static int slots_of[2][MAXDEV] = {};   // same value whoever writes it first
		int dev = 0;
		(void)hipGetDevice(&dev);
		const int k = out.sh_stage ? 1 : 0;
		int& slots = slots_of[k][(unsigned)dev % MAXDEV];
		if (slots == 0) slots = k ? resident_waves(sh_bwd_kernel<true>) : resident_waves(sh_bwd_kernel<false>);
		// 1.5 waves per slot: ~1.5 chunks = ~55 live Gaussians per wave on C3, i.e. ONE evaluation batch for most waves and two
		// rounds of them (C3, random / Morton order: 41 / 42 us; one wave per slot 43 / 50; spans of 128 as before 52 / 43)
		const int grid = min(slots + slots / 2, div_up(s.P, WAVE));
		if (out.sh_stage) hipLaunchKernelGGL(sh_bwd_kernel<true>, dim3(grid), dim3(WAVE), 0, stream, a);
		else hipLaunchKernelGGL(sh_bwd_kernel<false>, dim3(grid), dim3(WAVE), 0, stream, a);
		return hipGetLastError();
	}

	// ------------------------------------------------------------------------------------------------
	// SH backward of SEVERAL views in one pass over the coefficients (fdgs_sh_backward_batch; deferred mode only).
	// What a view's SH backward has to read is the Gaussian's coefficient row (12 M bytes) -- the same row for every view of an
	// optimizer step.  Here a wave scans 64 consecutive Gaussians, ballot-compacts those that carry a colour gradient in ANY
	// view, and takes them 32 at a time: their WHOLE rows (the prefix the active degrees use) are staged once through a
	// wave-private LDS tile with linear float4 loads, and lane = (Gaussian, view parity) evaluates its Gaussian for the views
	// v = parity, parity + 2, ...: per view the 8 staged numbers for the flush (dL_dRGB, direction, the two cosine factors) and
	// the mean / time gradient in words 12..15 of the view's accumulator record.  Same arithmetic, operation by operation, as
	// sh_bwd_kernel<true> run view by view (tests compare them bit for bit).
	// ------------------------------------------------------------------------------------------------
	constexpr int SBB_MAX = 8;        // views per launch
	constexpr int SBB_ROWS = 32;
	constexpr int SBB_SPAN = 64;
	constexpr int SBB_ACT = 144;      // floats of the three coefficient blocks that can be active
	constexpr int SBB_STRIDE = SBB_ACT + 1;
	struct ShBwdBatchArgs
	{
		int P, D, D_t, M, nviews;
		const float *shs, *ts;
		float time_duration;
		int gaussian_dim, force_sh_3d, vec_ok, analytic;
		struct View
		{
			const float* campos; float timestamp;
			const int32_t* radii; const float* means; const uint8_t* clamped;
			float* gacc; float4* stage;
		} v[SBB_MAX];
	};

	__device__ __forceinline__ float3 colour_gradient_of(const float* gacc, const uint8_t* clamped, int idx)
	{
		const float4 w = *reinterpret_cast<const float4*>(gacc + (size_t)idx * GRAD_ACC_WORDS);
		float3 dRGB = make_float3(w.x, w.y, w.z);
		const uint8_t cl = clamped[idx];
		if (cl & 1) dRGB.x = 0.f;
		if (cl & 2) dRGB.y = 0.f;
		if (cl & 4) dRGB.z = 0.f;
		return dRGB;
	}

	__global__ void __launch_bounds__(WAVE) sh_bwd_batch_kernel(const ShBwdBatchArgs a)
	{
		__shared__ float tile[SBB_ROWS * SBB_STRIDE];
		__shared__ uint32_t s_list[SBB_SPAN];
		const int lane = threadIdx.x;
		const int g0 = blockIdx.x * SBB_SPAN;
		const int row_floats = 3 * a.M;
		const bool sh3d = (a.gaussian_dim == 3 || a.force_sh_3d);
		const int ncoef0 = min(16, (a.D + 1) * (a.D + 1));
		const int nblocks = (!sh3d && a.D > 2) ? 1 + min(max(a.D_t, 0), 2) : 1;
		const int act = (nblocks - 1) * 48 + 3 * (nblocks > 1 ? 16 : ncoef0);   // floats of a row the active degrees read
		const unsigned long long lt_mask = (1ull << lane) - 1ull;

		// ---- which Gaussians of the span carry a colour gradient in some view; the others get their zero stage records ----
		int n = 0;
		{
			const int idx = g0 + lane;
			const bool valid = idx < a.P;
			bool any = false;
#pragma unroll 1
			for (int v = 0; v < a.nviews; v++)
			{
				bool live = false;
				if (valid && a.v[v].radii[idx] > 0)
				{
					const float3 d = colour_gradient_of(a.v[v].gacc, a.v[v].clamped, idx);
					live = d.x != 0.f || d.y != 0.f || d.z != 0.f;
				}
				if (valid && !live) a.v[v].stage[2 * (size_t)idx] = make_float4(0.f, 0.f, 0.f, 0.f);
				any = any || live;
			}
			const unsigned long long m = __ballot(any);
			if (any) s_list[__popcll(m & lt_mask)] = (uint32_t)lane;
			n = __popcll(m);
		}
		__builtin_amdgcn_wave_barrier();

		const int g = lane & (SBB_ROWS - 1), vpar = lane >> 5;
		for (int r0 = 0; r0 < n; r0 += SBB_ROWS)
		{
			const int nrows = min(SBB_ROWS, n - r0);
			// ---- whole rows into the tile ----
			if (a.vec_ok && (act & 3) == 0)
			{
				const int RC = act / 4, total = nrows * RC;
				const int dg = WAVE / RC, dq = WAVE - dg * RC;
				int cg = lane / RC, cq = lane - cg * RC;
				for (int c0 = 0; c0 < total; c0 += 6 * WAVE)
				{
					float4 val[6];
					int og[6], oq[6];
#pragma unroll
					for (int i = 0; i < 6; i++)
					{
						og[i] = cg; oq[i] = cq;
						cg += dg; cq += dq;
						if (cq >= RC) { cq -= RC; cg++; }
						if (c0 + i * WAVE + lane >= total) og[i] = -1;
						else val[i] = *reinterpret_cast<const float4*>(a.shs + (size_t)(g0 + (int)s_list[r0 + og[i]]) * row_floats + 4 * oq[i]);
					}
#pragma unroll
					for (int i = 0; i < 6; i++)
					{
						if (og[i] < 0) continue;
						float* d = tile + og[i] * SBB_STRIDE + 4 * oq[i];
						d[0] = val[i].x; d[1] = val[i].y; d[2] = val[i].z; d[3] = val[i].w;
					}
				}
			}
			else
			{
				const int total = nrows * act;
				const int dg = WAVE / act, dpos = WAVE - dg * act;
				int cg = lane / act, pos = lane - cg * act;
				for (int e = lane; e < total; e += WAVE)
				{
					tile[cg * SBB_STRIDE + pos] = a.shs[(size_t)(g0 + (int)s_list[r0 + cg]) * row_floats + pos];
					cg += dg; pos += dpos;
					if (pos >= act) { pos -= act; cg++; }
				}
			}
			__builtin_amdgcn_wave_barrier();

			// ---- lane = (row g, view parity): the views v = vpar, vpar + 2, ... of Gaussian list[r0 + g] ----
			if (g < nrows)
			{
				const int idx = g0 + (int)s_list[r0 + g];
				const float* row = tile + g * SBB_STRIDE;
				const float t_in = sh3d ? 0.f : a.ts[idx];
#pragma unroll 1
				for (int v = vpar; v < a.nviews; v += 2)
				{
					if (!(a.v[v].radii[idx] > 0)) continue;
					const float3 dRGB = colour_gradient_of(a.v[v].gacc, a.v[v].clamped, idx);
					if (dRGB.x == 0.f && dRGB.y == 0.f && dRGB.z == 0.f) continue;
					const float* cp = a.v[v].campos;
					const float* mp = a.v[v].means + 3 * (size_t)idx;
					const float3 dir_orig = make_float3(mp[0] - cp[0], mp[1] - cp[1], mp[2] - cp[2]); // Q4: shifted mean
					const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
					const float3 dir = make_float3(dir_orig.x / len, dir_orig.y / len, dir_orig.z / len);
					const float dir_t = sh3d ? 0.f : t_in - a.v[v].timestamp;
					float l[16], dX[16], dY[16], dZ[16];
					sh_tables(a.D, dir.x, dir.y, dir.z, !sh3d, l, dX, dY, dZ);
					float3 gx = make_float3(0.f, 0.f, 0.f), gy = gx, gz = gx, gt = gx;
					float tk_stage[2] = { 0.f, 0.f };
					for (int blk = 0; blk < nblocks; blk++)
					{
						const int nk = (blk == 0) ? ncoef0 : 16;
						const float* brow = row + 48 * blk;
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
						if (a.analytic) dtk_dt = -dtk_dt;
						if (blk == 1) tk_stage[0] = tk;
						if (blk == 2) tk_stage[1] = tk;
						float3 st = make_float3(0.f, 0.f, 0.f), sx = st, sy = st, sz = st;
#pragma unroll
						for (int k = 0; k < 16; k++)
						{
							if (k >= nk) break;
							const float3 sv = s_ld3(brow, k);
							st = s_add(st, s_scl(l[k], sv));
							sx = s_add(sx, s_scl(dX[k], sv));
							sy = s_add(sy, s_scl(dY[k], sv));
							sz = s_add(sz, s_scl(dZ[k], sv));
						}
						if (blk == 0) { gx = sx; gy = sy; gz = sz; }
						else
						{
							gx = s_add(gx, s_scl(tk, sx)); gy = s_add(gy, s_scl(tk, sy)); gz = s_add(gz, s_scl(tk, sz));
							gt = a.analytic ? s_add(gt, s_scl(dtk_dt, st)) : s_scl(dtk_dt, st); // Q3: overwrite, not accumulate
						}
					}
					a.v[v].stage[2 * (size_t)idx] = make_float4(dRGB.x, dRGB.y, dRGB.z, tk_stage[0]);
					a.v[v].stage[2 * (size_t)idx + 1] = make_float4(dir.x, dir.y, dir.z, tk_stage[1]);
					float4 o;
					const float3 ddir = make_float3(s_dot(gx, dRGB), s_dot(gy, dRGB), s_dot(gz, dRGB));
					// dnormvdv, auxiliary.h:108-118
					const float3 w = dir_orig;
					const float sum2 = w.x * w.x + w.y * w.y + w.z * w.z;
					const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
					o.x = ((+sum2 - w.x * w.x) * ddir.x - w.y * w.x * ddir.y - w.z * w.x * ddir.z) * invsum32;
					o.y = (-w.x * w.y * ddir.x + (sum2 - w.y * w.y) * ddir.y - w.z * w.y * ddir.z) * invsum32;
					o.z = (-w.x * w.z * ddir.x - w.y * w.z * ddir.y + (sum2 - w.z * w.z) * ddir.z) * invsum32;
					o.w = sh3d ? 0.f : s_dot(gt, dRGB);
					reinterpret_cast<float4*>(a.v[v].gacc + (size_t)idx * GRAD_ACC_WORDS)[3] = o;
				}
			}
			__builtin_amdgcn_wave_barrier();
		}
	}

	// views / ins / outs / geoms of the batch: all views share P, M, the degrees and the Gaussian tensors; outs[v]->sh_stage and
	// outs[v]->grad_accum are the view's own
	hipError_t launch_sh_bwd_batch(int nviews, const fdgs_scene* const* views, const fdgs_backward_in* const* ins,
	                               const fdgs_backward_out* const* outs, hipStream_t stream)
	{
		const fdgs_scene& s = *views[0];
		if (s.shs == nullptr || s.M <= 0 || s.P <= 0) return hipSuccess;
		const GeomLayout L = geom_layout(s.P);
		for (int v0 = 0; v0 < nviews; v0 += SBB_MAX)
		{
			ShBwdBatchArgs a;
			a.P = s.P; a.D = s.D; a.D_t = s.D_t; a.M = s.M; a.nviews = min(SBB_MAX, nviews - v0);
			a.shs = s.shs; a.ts = s.ts; a.time_duration = s.time_duration;
			a.gaussian_dim = s.gaussian_dim; a.force_sh_3d = s.force_sh_3d; a.analytic = s.analytic_sh_grad;
			a.vec_ok = ((reinterpret_cast<uintptr_t>(s.shs) & 15) == 0 && (3 * s.M) % 4 == 0) ? 1 : 0;
			for (int v = 0; v < SBB_MAX; v++)
			{
				const int w = min(v0 + v, nviews - 1);
				a.v[v].campos = views[w]->campos; a.v[v].timestamp = views[w]->timestamp;
				a.v[v].radii = ins[w]->radii; a.v[v].means = ins[w]->out_means3D;
				a.v[v].clamped = reinterpret_cast<const uint8_t*>(reinterpret_cast<const char*>(ins[w]->geom_buffer) + L.clamped);
				a.v[v].gacc = outs[w]->grad_accum; a.v[v].stage = reinterpret_cast<float4*>(outs[w]->sh_stage);
			}
			hipLaunchKernelGGL(sh_bwd_batch_kernel, dim3(div_up(s.P, SBB_SPAN)), dim3(WAVE), 0, stream, a);
		}
		return hipGetLastError();
	}

	// ------------------------------------------------------------------------------------------------
	// Deferred SH gradient (gradient accumulation over the views of one optimizer step).
	// dL_dsh of a view is basis(view direction) x time factor (x) dL_dRGB: M x 3 floats written (or read-modified-written,
	// from the second view on) per Gaussian and view, although the view only contributes 8 numbers.  In deferred mode
	// (fdgs_backward_out.sh_stage) sh_bwd_kernel stores those 8 numbers per view and this kernel, once per step, rebuilds
	// the views' contributions in registers and sums them in view order -- the same additions in the same order as the
	// accumulating path.  Two consumers:
	//   MODE 0 (fdgs_sh_flush)      dL_dsh is written once: at C3 / 4 views 1.7 KB of traffic per live Gaussian and view
	//                               become 0.6 KB + 0.6 KB / 4;
	//   MODE 1 (fdgs_adam_step_sh)  the only reader of the summed dL_dsh is the Adam update of the SH coefficients
	//                               (train.py:247-249; 89 % of all parameters at M = 48), so the gradient goes from the LDS
	//                               tile straight into it: read p, m, v, write p, m, v (24 B per coefficient) and dL_dsh
	//                               never travels through memory (it is also written when the caller asks for it).
	// A wave owns 32 consecutive Gaussians; lane = (Gaussian, half of the 16 coefficients of a block), the three possible
	// coefficient blocks accumulate in registers and land in a wave-private LDS tile of whole rows, so that the output phase
	// walks memory strictly linearly (32 x 12 M bytes per array) -- the block-by-block sweeps of sh_bwd_kernel touch every
	// 128-byte line of the 576-byte rows up to three times.
	// ------------------------------------------------------------------------------------------------
	constexpr int SHF_GPW = 32;
	// LDS row stride of the flush tile: the active blocks' floats + 1 (odd: lane-per-row accesses are conflict free)
	static inline int shf_tile_stride(int D, int D_t, bool sh3d) { return 48 * ((!sh3d && D > 2) ? 1 + std::min(std::max(D_t, 0), 2) : 1) + 1; }
#ifndef FDGS_SHF_BATCH
#define FDGS_SHF_BATCH 6
#endif
	constexpr int SHF_BATCH = FDGS_SHF_BATCH;   // float4 chunks per lane whose p / m / v loads are in flight together

	struct ShFlushArgs
	{
		int P, D, D_t, M, nviews, sh3d, analytic, accum, vec_ok;
		const float4* stages;   // [nviews][P][2]
		float* dL_dsh;
		float *p, *m, *v;       // MODE 1
		AdamScalars k;
	};

	// basis values only (sh_tables without the derivative tables)
	__device__ __forceinline__ void sh_values(int deg, float x, float y, float z, bool promote, float* l)
	{
		float dX[16], dY[16], dZ[16];
		sh_tables(deg, x, y, z, promote, l, dX, dY, dZ);
	}

	template <int MODE>
	__global__ void __launch_bounds__(WAVE) sh_flush_kernel(const ShFlushArgs a)
	{
		// the tile holds the coefficient blocks that can carry a gradient: 48 floats per active block and Gaussian, + 1 (odd stride); its
		// size is the launch's dynamic LDS (one active block -- 3D SH, M = 16, or the first iterations of the degree ramp -- takes a
		// third of the three-block tile: more waves per CU for the streaming phase)
		extern __shared__ float tile[];
		const int lane = threadIdx.x, g = lane & (SHF_GPW - 1), half = lane >> 5;
		const int g0 = blockIdx.x * SHF_GPW;
		const bool valid = g0 + g < a.P;
		const int idx = valid ? g0 + g : a.P - 1;
		const int row_floats = 3 * a.M;
		const bool sh3d = a.sh3d != 0;
		const int ncoef0 = min(16, (a.D + 1) * (a.D + 1));
		const int nblocks = (!sh3d && a.D > 2) ? 1 + min(max(a.D_t, 0), 2) : 1;
		const int act_floats = min(48 * nblocks, row_floats);   // row prefix that can carry a gradient
		const int stride = 48 * nblocks + 1;                    // = shf_tile_stride() of the launcher

		float3 acc[3][8];
#pragma unroll
		for (int b = 0; b < 3; b++)
#pragma unroll
			for (int j = 0; j < 8; j++) acc[b][j] = make_float3(0.f, 0.f, 0.f);
		bool any = false;
		for (int v = 0; v < a.nviews; v++)
		{
			const float4 s0 = a.stages[2 * ((size_t)v * a.P + idx)];
			const float3 dRGB = make_float3(s0.x, s0.y, s0.z);
			if (!valid || (dRGB.x == 0.f && dRGB.y == 0.f && dRGB.z == 0.f)) continue;   // this view added nothing
			const float4 s1 = a.stages[2 * ((size_t)v * a.P + idx) + 1];
			any = true;
			float l[16];
			sh_values(a.D, s1.x, s1.y, s1.z, !sh3d, l);
#pragma unroll
			for (int j = 0; j < 8; j++)
			{
				const int k = 8 * half + j;
				const float lk = half ? l[8 + j] : l[j];
				float basis = lk;
				if (j == 1 && half == 0 && !sh3d && !a.analytic) basis = l[0]; // Q1
				if (k >= ncoef0) basis = 0.f;
				acc[0][j] = s_add(acc[0][j], s_scl(basis, dRGB));
				if (nblocks > 1) acc[1][j] = s_add(acc[1][j], s_scl(s0.w * lk, dRGB));
				if (nblocks > 2) acc[2][j] = s_add(acc[2][j], s_scl(s1.w * lk, dRGB));
			}
		}
		const unsigned long long vmask = __ballot(any);   // bit g (and g + 32)
		if (any)
		{
			float* row = tile + g * stride + 24 * half;
#pragma unroll
			for (int b = 0; b < 3; b++)
			{
				if (b >= nblocks) break;
#pragma unroll
				for (int j = 0; j < 8; j++)
				{
					row[48 * b + 3 * j] = acc[b][j].x; row[48 * b + 3 * j + 1] = acc[b][j].y; row[48 * b + 3 * j + 2] = acc[b][j].z;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();

		if (MODE == 1 || a.vec_ok)
		{
			// whole rows, float4 by float4, linearly through memory: chunk c of the wave = (Gaussian c / RC, float4 c % RC of its row)
			const int RC = row_floats / 4, total = SHF_GPW * RC;
			const int dg = WAVE / RC, dq = WAVE - dg * RC;
			int cg = lane / RC, cq = lane - cg * RC;
			for (int c0 = 0; c0 < total; c0 += SHF_BATCH * WAVE)
			{
				float4 pp[SHF_BATCH], mm[SHF_BATCH], vv[SHF_BATCH];
				int og[SHF_BATCH], oq[SHF_BATCH];
#pragma unroll
				for (int i = 0; i < SHF_BATCH; i++)
				{
					og[i] = cg; oq[i] = cq;
					cg += dg; cq += dq;
					if (cq >= RC) { cq -= RC; cg++; }
					const bool in = c0 + i * WAVE + lane < total && g0 + og[i] < a.P;
					if (!in) og[i] = -1;
					if (MODE == 1 && in)
					{
						const size_t o = (size_t)(g0 + og[i]) * row_floats + 4 * oq[i];
						pp[i] = *reinterpret_cast<const float4*>(a.p + o);
						mm[i] = stream_ld(reinterpret_cast<const float4*>(a.m + o));   // the moments: once per step, non-temporal (fdgs_common.h)
						vv[i] = stream_ld(reinterpret_cast<const float4*>(a.v + o));
					}
				}
#pragma unroll
				for (int i = 0; i < SHF_BATCH; i++)
				{
					if (og[i] < 0) continue;
					const int e0 = 4 * oq[i];
					const bool live = ((vmask >> og[i]) & 1ull) && e0 < act_floats;
					const float* t = tile + og[i] * stride + e0;
					float ge[4] = { 0.f, 0.f, 0.f, 0.f };
					if (live) { ge[0] = t[0]; ge[1] = t[1]; ge[2] = t[2]; ge[3] = t[3]; }
					const size_t o = (size_t)(g0 + og[i]) * row_floats + e0;
					if (MODE == 1)
					{
						float pe[4] = { pp[i].x, pp[i].y, pp[i].z, pp[i].w }, me[4] = { mm[i].x, mm[i].y, mm[i].z, mm[i].w };
						float ve[4] = { vv[i].x, vv[i].y, vv[i].z, vv[i].w };
#pragma unroll
						for (int e = 0; e < 4; e++)
						{
							// the DC coefficient (first 3 floats of the row) has its own learning rate (gaussian_model.py:339-340)
							const float lr = (e0 == 0 && e < 3) ? a.k.lr_head_bc1 : a.k.lr_bc1;
							adam_update(pe[e], me[e], ve[e], ge[e], lr, a.k.b1, a.k.b2, a.k.eps, a.k.inv_sqrt_bc2);
						}
						stream_st(reinterpret_cast<float4*>(a.m + o), make_float4(me[0], me[1], me[2], me[3]));
						stream_st(reinterpret_cast<float4*>(a.v + o), make_float4(ve[0], ve[1], ve[2], ve[3]));
						*reinterpret_cast<float4*>(a.p + o) = make_float4(pe[0], pe[1], pe[2], pe[3]);
						if (a.dL_dsh) *reinterpret_cast<float4*>(a.dL_dsh + o) = make_float4(ge[0], ge[1], ge[2], ge[3]);
					}
					else
					{
						float4* d = reinterpret_cast<float4*>(a.dL_dsh + o);
						if (!a.accum) *d = make_float4(ge[0], ge[1], ge[2], ge[3]);
						else if (live)
						{
							const float4 old = *d;
							*d = make_float4(old.x + ge[0], old.y + ge[1], old.z + ge[2], old.w + ge[3]);
						}
					}
				}
			}
		}
		else
		{
			// any row length / alignment, float by float
			const int total = SHF_GPW * row_floats;
			const int dg = WAVE / row_floats, dpos = WAVE - dg * row_floats;
			int cg = lane / row_floats, pos = lane - cg * row_floats;
			for (int e = lane; e < total; e += WAVE)
			{
				if (g0 + cg < a.P)
				{
					const bool live = ((vmask >> cg) & 1ull) && pos < act_floats;
					float* d = a.dL_dsh + (size_t)(g0 + cg) * row_floats + pos;
					if (!a.accum) *d = live ? tile[cg * stride + pos] : 0.f;
					else if (live) *d += tile[cg * stride + pos];
				}
				cg += dg; pos += dpos;
				if (pos >= row_floats) { pos -= row_floats; cg++; }
			}
		}
	}

	static void fill_flush_args(ShFlushArgs& a, int P, int D, int D_t, int M, int gaussian_dim, int force_sh_3d, int analytic, int nviews,
	                            const float* stages, float* dL_dsh)
	{
		a.P = P; a.D = D; a.D_t = D_t; a.M = M; a.nviews = nviews;
		a.sh3d = (gaussian_dim == 3 || force_sh_3d) ? 1 : 0;
		a.analytic = analytic; a.accum = 0;
		a.vec_ok = ((reinterpret_cast<uintptr_t>(dL_dsh) & 15) == 0 && (3 * M) % 4 == 0) ? 1 : 0;
		a.stages = reinterpret_cast<const float4*>(stages);
		a.dL_dsh = dL_dsh;
		a.p = a.m = a.v = nullptr;
		a.k = AdamScalars{};
	}

	hipError_t launch_sh_flush(int P, int D, int D_t, int M, int gaussian_dim, int force_sh_3d, int analytic, int nviews,
	                           const float* stages, float* dL_dsh, int accumulate, hipStream_t stream)
	{
		if (P <= 0 || M <= 0 || nviews <= 0) return hipSuccess;
		ShFlushArgs a;
		fill_flush_args(a, P, D, D_t, M, gaussian_dim, force_sh_3d, analytic, nviews, stages, dL_dsh);
		a.accum = accumulate;
		hipLaunchKernelGGL(sh_flush_kernel<0>, dim3(div_up(P, SHF_GPW)), dim3(WAVE), (size_t)SHF_GPW * shf_tile_stride(D, D_t, a.sh3d != 0) * sizeof(float), stream, a);
		return hipGetLastError();
	}

	hipError_t launch_sh_adam(int P, int D, int D_t, int M, int gaussian_dim, int force_sh_3d, int analytic, int nviews,
	                          const float* stages, float* params, float* exp_avg, float* exp_avg_sq, float* dL_dsh,
	                          const AdamScalars& k, hipStream_t stream)
	{
		if (P <= 0 || M <= 0) return hipSuccess;
		const uintptr_t align = reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(exp_avg) |
		                        reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(dL_dsh);
		// the DC learning rate is applied to the first 3 floats of a row: rows must start on a float4 boundary
		if (nviews <= 0 || (3 * M) % 4 != 0 || (align & 15) != 0) return hipErrorInvalidValue;
		ShFlushArgs a;
		fill_flush_args(a, P, D, D_t, M, gaussian_dim, force_sh_3d, analytic, nviews, stages, dL_dsh);
		a.p = params; a.m = exp_avg; a.v = exp_avg_sq; a.k = k;
		hipLaunchKernelGGL(sh_flush_kernel<1>, dim3(div_up(P, SHF_GPW)), dim3(WAVE), (size_t)SHF_GPW * shf_tile_stride(D, D_t, a.sh3d != 0) * sizeof(float), stream, a);
		return hipGetLastError();
	}
}

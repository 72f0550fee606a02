// ssim.hip -- fused photometric loss (1-l) L1 + l (1 - SSIM), forward and backward (gfx950).
//
// The step right after the rasterizer in every training iteration (reference train.py:115-117,
// utils/loss_utils.py:17-64): L1 + SSIM with an 11x11 Gaussian window (sigma 1.5, zero padding,
// C1 = 0.01^2, C2 = 0.03^2, mean over all pixels and channels).  The reference runs it as five
// grouped 11x11 convolutions forward plus their backward through the DL library; on ROCm that
// is ~7.6 ms per 1352x1014 image, four times the whole rasterizer.  Here:
//   forward : one pass.  A 32x32 output tile loads its 42x42 halo of both images into LDS as pairs (u, v) = (x + y, x - y), does
//             the separable window (horizontal, then vertical) for FOUR moments (u, v, u^2, v^2: round 6 -- the window statistics SSIM
//             needs follow from them, see the kernel; rounds 1-5: five, x, y, x^2, y^2, xy), evaluates SSIM and the three partial derivatives
//             d ssim/d mu1, d ssim/d E[x^2], d ssim/d E[xy] per pixel (kept for the backward),
//             and writes per-tile partial sums of |x-y| and ssim (summed by the host: deterministic).
//             The kernel is VALU-issue bound (SQ counters), so it is written for instruction count: the
//             moments travel as pairs (u,v) (u^2,v^2) -> two packed-fp32 FMAs per tap (rounds 2-5: (x,y) (x^2,y^2) + xy: three; before: five);
//             a thread produces 4 adjacent outputs of the horizontal pass (14 b128-loaded inputs instead of
//             44 scalar reads) and 4 adjacent rows of the vertical pass (14 row reads per 4 outputs).  Round 4: 32x32 tile
//             instead of 32x16 (halo 1.72 x instead of 2.13 x the tile, 1.31 instead of 1.63 rows of horizontal pass per
//             output row) with the input tile's LDS bytes reused for the filtered arrays: forward + backward 85.5 -> 76 us
//             per 1352x1014 image.
//   backward: dL/dx(p) = w_l1 sign(x-y) + w_ssim [ (W * dmu1)(p) + 2 x(p) (W * dE11)(p) + y(p) (W * dE12)(p) ]
//             -- three more separable windows over the stored derivative maps (W symmetric).
// fp32, ~200 VALU instructions per pixel-channel forward, ~100 backward; no MFMA (11-tap separable stencil).
#include <cstdlib>
#include "fdgs_common.h"

namespace fdgs
{
#ifndef FDGS_SSIM_STY
#define FDGS_SSIM_STY 32   // 16: the tile of rounds 1-3 (A/B: FDGS_EXTRA_FLAGS=-DFDGS_SSIM_STY=16 csrc/build.sh)
#endif
	constexpr int STX = 32, STY = FDGS_SSIM_STY;    // output tile (rows: 16 or 32)
	constexpr int SROWS = STY / 8;       // adjacent output rows a thread finishes in the vertical pass (256 threads = 32 columns x 8 row groups)
	constexpr int SR = 5;                // window radius (11 taps)
	constexpr int SW = STX + 2 * SR;     // 42: tile + halo, columns
	constexpr int SHH = STY + 2 * SR;    // 26: rows
	// LDS row strides (16-byte aligned rows for the b128 accesses), chosen for the banks.  In the horizontal pass 8 threads take a
	// row and a wave 8 rows; a thread's b128 touches 4 of every 8 dwords of a pair-array row, so consecutive rows must be offset by
	// 4 (mod 8) dwords or all 8 rows fall on the same half of the banks (measured with strides 44 / 36: 41 % of the backward's LDS
	// cycles were bank conflicts): 46 and 38 pairs = 92 and 76 dwords.  Scalar arrays read / written 32 dwords per row: 48 (rows 0,
	// 48, 32, 16 mod 64: every bank four times per b128, the minimum).  Effect: 86 -> 85 us for forward + backward: the conflicts were not what holds the
	// kernels at 74 % / 52 % of their VALU issue bound.
	constexpr int SSTR = 46;             // input tile, in (x,y) pairs
	constexpr int SSTR1 = 48;            // input tile of the backward's scalar map
	constexpr int HSTR = 38;             // horizontally filtered moments, pair arrays
	constexpr int HSTR1B = 36;           // ... of the backward (48 would cost its sixth workgroup per CU)
	constexpr int STHREADS = 256;
	typedef float v2f __attribute__((ext_vector_type(2)));
	typedef float v4f __attribute__((ext_vector_type(4)));

	// Workgroup -> tile.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs, each with its own L2: with the
	// plain (x, y, channel) grid the tiles that share a halo -- horizontal neighbours -- always sit on DIFFERENT XCDs and every halo
	// byte is fetched from memory once per tile that needs it (FETCH_SIZE: 1.6 x the tile bytes).  Here XCD j takes the j-th eighth of
	// the tiles in row-major order -- a band of tile rows of one channel -- so that neighbours meet in one L2.
	struct TileId { int tx, ty, c, index; bool valid; };
	__device__ __forceinline__ TileId ssim_tile_of(int gx, int gy, int C)
	{
		const int total = gx * gy * C, chunk = (total + 7) / 8;
		const int wg = (int)blockIdx.x, xcd = wg & 7, k = wg >> 3;
		TileId t;
		t.index = xcd * chunk + k;
		t.valid = k < chunk && t.index < total;
		const int i = t.valid ? t.index : 0;
		t.c = i / (gx * gy);
		const int r = i - t.c * (gx * gy);
		t.ty = r / gx; t.tx = r - t.ty * gx;
		return t;
	}
	static inline int ssim_grid(int gx, int gy, int C) { return ((gx * gy * C + 7) / 8) * 8; }

	// gaussian(11, 1.5) normalised, as utils/loss_utils.py:23-25
	__device__ constexpr float GW[11] = {
		0.0010283801f, 0.0075987582f, 0.0360007733f, 0.1093606874f, 0.2130055279f, 0.2660117149f,
		0.2130055279f, 0.1093606874f, 0.0360007733f, 0.0075987582f, 0.0010283801f };

	__global__ void __launch_bounds__(STHREADS) ssim_fwd_kernel(
		const float* __restrict__ img1, const float* __restrict__ img2, int C, int H, int W,
		float* __restrict__ dm_dmu1, float* __restrict__ dm_de11, float* __restrict__ dm_de12,
		float* __restrict__ partial_l1, float* __restrict__ partial_ssim)
	{
		// The input tile shares its bytes with two of the three horizontally filtered arrays: those results wait in registers until
		// every task of the horizontal pass has read its inputs (a barrier in between).  A workgroup then holds 34 KB instead of 49
		// (32-row tile): four workgroups per CU instead of three.  (All three behind the barrier: 16 more VGPRs for nothing -- the
		// results are larger than the inputs.)
		// Round 6: FOUR moment channels instead of five, all of them packed pairs.  With u = x + y, v = x - y the window statistics SSIM
		// needs are E[u], E[v], E[u^2], E[v^2]:  mu1, mu2 = (E[u] +- E[v]) / 2,  E[x^2] + E[y^2] = (E[u^2] + E[v^2]) / 2,  E[xy] = (E[u^2] -
		// E[v^2]) / 4 -- the scalar fifth channel x y (a plain FMA per tap next to two packed ones, and per pair of outputs four register moves
		// the compiler needed to feed it to a packed multiply) is gone: 3 -> 2 instructions per tap, one LDS array less (28 instead of 34 KB:
		// five workgroups per CU instead of four).  sigma_1^2 and sigma_2^2 only ever enter SSIM as their sum.
		constexpr int IN_BYTES = SHH * SSTR * 8, HM_BYTES = SHH * HSTR * 8;
		constexpr int SH_BYTES = IN_BYTES > HM_BYTES ? IN_BYTES : HM_BYTES;
		__shared__ __attribute__((aligned(16))) char s_raw[HM_BYTES + SH_BYTES];
		v2f (*h_m)[HSTR] = reinterpret_cast<v2f (*)[HSTR]>(s_raw);                                   // horizontally filtered (u, v)
		v2f (*s_in)[SSTR] = reinterpret_cast<v2f (*)[SSTR]>(s_raw + HM_BYTES);                       // (u, v) = (x + y, x - y)
		v2f (*h_s)[HSTR] = reinterpret_cast<v2f (*)[HSTR]>(s_raw + HM_BYTES);                        // (u^2, v^2): over the input tile
		__shared__ float red[2][STHREADS / WAVE];

		const TileId tile = ssim_tile_of((W + STX - 1) / STX, (H + STY - 1) / STY, C);
		if (!tile.valid) return;
		const int c = tile.c;
		const int x0 = tile.tx * STX, y0 = tile.ty * STY;
		const int tid = threadIdx.x;
		const size_t plane = (size_t)c * H * W;

		// halo: thread -> one column of the 42 and rows tid / 42, + 6, + 12, ... (252 of the 256 threads; the column, its bounds test
		// and the address are computed once, a trip only moves down six rows).  All loads of the thread are issued before the first
		// one is waited for (a rolled loop paid one global round trip per trip)
		float l1 = 0.f;   // |x - y| over the tile's own pixels: summed where they are loaded (the input tile is gone after the horizontal pass)
		{
			constexpr int RPT = STHREADS / SW, TRIPS = (SHH + RPT - 1) / RPT;
			const int lyb = tid / SW, hx = tid - lyb * SW;
			const int gxh = x0 + hx - SR;
			const bool col_in = tid < RPT * SW && (unsigned)gxh < (unsigned)W;
			const bool col_own = (unsigned)(hx - SR) < (unsigned)STX;
			v2f p[TRIPS];
#pragma unroll
			for (int t = 0; t < TRIPS; t++)
			{
				const int ly = lyb + t * RPT, gy = y0 + ly - SR;
				const bool in = col_in && ly < SHH && (unsigned)gy < (unsigned)H;
				const size_t o = in ? plane + (size_t)gy * W + gxh : plane;   // branch-free: outside lanes read a valid address
				const float vx = img1[o], vy = img2[o];
				p[t] = in ? v2f{ vx + vy, vx - vy } : v2f{ 0.0f, 0.0f };     // (u, v); zero padding (F.conv2d padding = 5)
			}
#pragma unroll
			for (int t = 0; t < TRIPS; t++)
			{
				const int ly = lyb + t * RPT;
				if (tid < RPT * SW && ly < SHH) s_in[ly][hx] = p[t];
				const bool own = col_own && (unsigned)(ly - SR) < (unsigned)STY;   // (outside the image: 0 - 0)
				l1 += own ? fabsf(p[t].y) : 0.0f;                                  // |x - y|
			}
		}
		__syncthreads();

		// horizontal pass: task -> (row, 4 adjacent columns); SHH * 8 tasks over the 256 threads in HR rounds; results stay in
		// registers until every task has read its inputs
		constexpr int HR = (SHH * (STX / 4) + STHREADS - 1) / STHREADS;
		v2f as[HR][4];
#pragma unroll
		for (int r = 0; r < HR; r++)
		{
			const int task = tid + r * STHREADS;
			if (task < SHH * (STX / 4))
			{
				const int ly = task >> 3, cx = (task & 7) * 4;
				v2f p[16], sq[14];
				const v4f* src = reinterpret_cast<const v4f*>(&s_in[ly][cx]);
#pragma unroll
				for (int i = 0; i < 7; i++) { const v4f q = src[i]; p[2 * i] = v2f{ q.x, q.y }; p[2 * i + 1] = v2f{ q.z, q.w }; }
#pragma unroll
				for (int i = 0; i < 14; i++) sq[i] = p[i] * p[i];
				v2f am[4];
#pragma unroll
				for (int j = 0; j < 4; j++)
				{
					am[j] = GW[0] * p[j]; as[r][j] = GW[0] * sq[j];
#pragma unroll
					for (int k = 1; k < 11; k++) { am[j] += GW[k] * p[j + k]; as[r][j] += GW[k] * sq[j + k]; }
				}
				v4f* dm = reinterpret_cast<v4f*>(&h_m[ly][cx]);   // (its own bytes: written at once)
				dm[0] = v4f{ am[0].x, am[0].y, am[1].x, am[1].y }; dm[1] = v4f{ am[2].x, am[2].y, am[3].x, am[3].y };
			}
			__builtin_amdgcn_sched_barrier(0);   // one round's inputs at a time in registers
		}
		__syncthreads();   // the input tile has been read: its bytes become the filtered arrays
#pragma unroll
		for (int r = 0; r < HR; r++)
		{
			const int task = tid + r * STHREADS;
			if (task < SHH * (STX / 4))
			{
				const int ly = task >> 3, cx = (task & 7) * 4;
				v4f* ds = reinterpret_cast<v4f*>(&h_s[ly][cx]);
				ds[0] = v4f{ as[r][0].x, as[r][0].y, as[r][1].x, as[r][1].y }; ds[1] = v4f{ as[r][2].x, as[r][2].y, as[r][3].x, as[r][3].y };
			}
		}
		__syncthreads();

		// vertical pass: thread -> (column, SROWS adjacent rows)
		const int lx = tid & (STX - 1), ly0 = (tid >> 5) * SROWS;
		v2f vm[10 + SROWS], vs[10 + SROWS];
#pragma unroll
		for (int r = 0; r < 10 + SROWS; r++) { vm[r] = h_m[ly0 + r][lx]; vs[r] = h_s[ly0 + r][lx]; }
		float sv = 0.f;
		const int gx = x0 + lx;
#pragma unroll
		for (int j = 0; j < SROWS; j++)
		{
			v2f mu = GW[0] * vm[j], e2 = GW[0] * vs[j];
#pragma unroll
			for (int k = 1; k < 11; k++) { mu += GW[k] * vm[j + k]; e2 += GW[k] * vs[j + k]; }
			const int gy = y0 + ly0 + j;
			if (gx < W && gy < H)
			{
				const float mu1 = 0.5f * (mu.x + mu.y), mu2 = 0.5f * (mu.x - mu.y);   // from E[u], E[v]
				const float e_sum = 0.5f * (e2.x + e2.y), e12 = 0.25f * (e2.x - e2.y);   // E[x^2] + E[y^2], E[xy] from E[u^2], E[v^2]
				const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
				const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
				const float sg12 = e12 - mu12;
				const float A = 2.f * mu12 + C1, B = 2.f * sg12 + C2, Cc = mu1_sq + mu2_sq + C1, D = (e_sum - (mu1_sq + mu2_sq)) + C2;
				// 1 / Cc and 1 / D by v_rcp_f32 (1 ulp): three IEEE divisions were a seventh of the kernel's instructions
				const float rC = __builtin_amdgcn_rcpf(Cc), rD = __builtin_amdgcn_rcpf(D);
				const float inv = rC * rD;
				const float m = A * B * inv;
				// total derivative w.r.t. mu1 (through A, B, Cc, D), and w.r.t. the raw moments E[x^2], E[xy]
				const float dm_dA = B * inv, dm_dB = A * inv, dm_dC = -m * rC, dm_dD = -m * rD;
				const size_t o = plane + (size_t)gy * W + gx;
				dm_dmu1[o] = dm_dA * 2.f * mu2 - dm_dB * 2.f * mu2 + dm_dC * 2.f * mu1 - dm_dD * 2.f * mu1;
				dm_de11[o] = dm_dD;
				dm_de12[o] = 2.f * dm_dB;
				sv += m;
			}
		}
		// per-tile partial sums (wave shuffle + 4 partials)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { l1 += __shfl_down(l1, o); sv += __shfl_down(sv, o); }
		if ((tid & 63) == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = sv; }
		__syncthreads();
		if (tid == 0)
		{
			const int b = tile.index;
			partial_l1[b] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
			partial_ssim[b] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
		}
	}

	__global__ void __launch_bounds__(STHREADS) ssim_bwd_kernel(
		const float* __restrict__ img1, const float* __restrict__ img2, int C, int H, int W,
		const float* __restrict__ dm_dmu1, const float* __restrict__ dm_de11, const float* __restrict__ dm_de12,
		const float* __restrict__ upstream, float w_l1, float w_ssim, float* __restrict__ dL_dimg1)
	{
		// (inputs and horizontally filtered maps share their bytes, as in the forward kernel)
		constexpr int SP_BYTES = SHH * SSTR * 8, SQ_BYTES = SHH * SSTR1 * 4, HP_BYTES = SHH * HSTR * 8, HQ_BYTES = SHH * HSTR1B * 4;
		constexpr int LDS_BYTES = SP_BYTES + SQ_BYTES > HP_BYTES + HQ_BYTES ? SP_BYTES + SQ_BYTES : HP_BYTES + HQ_BYTES;
		__shared__ __attribute__((aligned(16))) char s_raw[LDS_BYTES];
		v2f (*s_p)[SSTR] = reinterpret_cast<v2f (*)[SSTR]>(s_raw);                       // (dm/dmu1, dm/dE11)
		float (*s_q)[SSTR1] = reinterpret_cast<float (*)[SSTR1]>(s_raw + SP_BYTES);      // dm/dE12
		v2f (*h_p)[HSTR] = reinterpret_cast<v2f (*)[HSTR]>(s_raw);
		float (*h_q)[HSTR1B] = reinterpret_cast<float (*)[HSTR1B]>(s_raw + HP_BYTES);

		const TileId tile = ssim_tile_of((W + STX - 1) / STX, (H + STY - 1) / STY, C);
		if (!tile.valid) return;
		const int c = tile.c;
		const int x0 = tile.tx * STX, y0 = tile.ty * STY;
		const int tid = threadIdx.x;
		const size_t plane = (size_t)c * H * W;

		{
			constexpr int RPT = STHREADS / SW, TRIPS = (SHH + RPT - 1) / RPT;   // (thread -> column + every sixth row, as in the forward kernel)
			const int lyb = tid / SW, hx = tid - lyb * SW;
			const int gxh = x0 + hx - SR;
			const bool col_in = tid < RPT * SW && (unsigned)gxh < (unsigned)W;
			v2f p[TRIPS];
			float q[TRIPS];
#pragma unroll
			for (int t = 0; t < TRIPS; t++)       // every load in flight before the first wait
			{
				const int ly = lyb + t * RPT, gy = y0 + ly - SR;
				const bool in = col_in && ly < SHH && (unsigned)gy < (unsigned)H;
				const size_t o = in ? plane + (size_t)gy * W + gxh : plane;   // branch-free: outside lanes read a valid address
				const float va = dm_dmu1[o], vb = dm_de11[o], vc = dm_de12[o];
				p[t] = in ? v2f{ va, vb } : v2f{ 0.0f, 0.0f };
				q[t] = in ? vc : 0.0f;
			}
#pragma unroll
			for (int t = 0; t < TRIPS; t++)
			{
				const int ly = lyb + t * RPT;
				if (tid < RPT * SW && ly < SHH) { s_p[ly][hx] = p[t]; s_q[ly][hx] = q[t]; }
			}
		}
		// the pixels this thread finishes below: their image values travel while the windows are computed
		const int lx = tid & (STX - 1), ly0 = (tid >> 5) * SROWS;
		const int gx = x0 + lx;
		float px[SROWS], py[SROWS];
#pragma unroll
		for (int j = 0; j < SROWS; j++)
		{
			const int gy = y0 + ly0 + j;
			const size_t o = (gx < W && gy < H) ? plane + (size_t)gy * W + gx : plane;
			px[j] = img1[o]; py[j] = img2[o];
		}
		__syncthreads();
		constexpr int HR = (SHH * (STX / 4) + STHREADS - 1) / STHREADS;
		v2f ap[HR][4];
		float aq[HR][4];
#pragma unroll
		for (int r = 0; r < HR; r++)
		{
			const int task = tid + r * STHREADS;
			if (task < SHH * (STX / 4))
			{
				const int ly = task >> 3, cx = (task & 7) * 4;
				v2f p[16];
				float q[16];
				const v4f* sp = reinterpret_cast<const v4f*>(&s_p[ly][cx]);
				const v4f* sq = reinterpret_cast<const v4f*>(&s_q[ly][cx]);
#pragma unroll
				for (int i = 0; i < 7; i++) { const v4f t = sp[i]; p[2 * i] = v2f{ t.x, t.y }; p[2 * i + 1] = v2f{ t.z, t.w }; }
#pragma unroll
				for (int i = 0; i < 4; i++) { const v4f t = sq[i]; q[4 * i] = t.x; q[4 * i + 1] = t.y; q[4 * i + 2] = t.z; q[4 * i + 3] = t.w; }
#pragma unroll
				for (int j = 0; j < 4; j++)
				{
					ap[r][j] = GW[0] * p[j]; aq[r][j] = GW[0] * q[j];
#pragma unroll
					for (int k = 1; k < 11; k++) { ap[r][j] += GW[k] * p[j + k]; aq[r][j] += GW[k] * q[j + k]; }
				}
			}
			__builtin_amdgcn_sched_barrier(0);   // one round's inputs at a time in registers
		}
		__syncthreads();   // the inputs have been read: their bytes become the filtered arrays
#pragma unroll
		for (int r = 0; r < HR; r++)
		{
			const int task = tid + r * STHREADS;
			if (task < SHH * (STX / 4))
			{
				const int ly = task >> 3, cx = (task & 7) * 4;
				v4f* dp = reinterpret_cast<v4f*>(&h_p[ly][cx]);
				dp[0] = v4f{ ap[r][0].x, ap[r][0].y, ap[r][1].x, ap[r][1].y }; dp[1] = v4f{ ap[r][2].x, ap[r][2].y, ap[r][3].x, ap[r][3].y };
				*reinterpret_cast<v4f*>(&h_q[ly][cx]) = v4f{ aq[r][0], aq[r][1], aq[r][2], aq[r][3] };
			}
		}
		__syncthreads();
		v2f vp[10 + SROWS];
		float vq[10 + SROWS];
#pragma unroll
		for (int r = 0; r < 10 + SROWS; r++) { vp[r] = h_p[ly0 + r][lx]; vq[r] = h_q[ly0 + r][lx]; }
		const float up = upstream[0];
#pragma unroll
		for (int j = 0; j < SROWS; j++)
		{
			v2f ab = GW[0] * vp[j];
			float d = GW[0] * vq[j];
#pragma unroll
			for (int k = 1; k < 11; k++) { ab += GW[k] * vp[j + k]; d += GW[k] * vq[j + k]; }
			const int gy = y0 + ly0 + j;
			if (gx < W && gy < H)
			{
				const size_t o = plane + (size_t)gy * W + gx;
				const float x = px[j], y = py[j];
				const float diff = x - y;
				const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
				dL_dimg1[o] = up * (w_l1 * sgn + w_ssim * (ab.x + 2.f * x * ab.y + y * d));
			}
		}
	}
}

namespace fdgs
{
	// ------------------------------------------------------------------------------------------------------------------------
	// Value AND gradient in ONE kernel (fdgs_l1_ssim_value_and_grad: what a training step needs -- the gradient does not depend
	// on anything but the two images and a scalar).  The two-kernel path writes three fp32 derivative maps (49 MB per 1352x1014
	// image) that the backward reads back with a 1.72 x halo: 367 MB of HBM traffic per image (rocprofv3 FETCH_SIZE / WRITE_SIZE)
	// for ~115 MB of images in and gradient out.  Here a workgroup rebuilds what it needs instead: for a 32x32 output tile the
	// derivative maps on the 42x42 pixels around it (window radius 5), from the moments of the 52x52 input pixels around those --
	// nothing but the two images is read, nothing but the gradient and two partial sums per tile is written.  The price is
	// arithmetic: the forward's windows and SSIM algebra run on 1.72 x the pixels (every derivative pixel is computed by the up to
	// four tiles whose halo it lies in).  Same arithmetic per pixel, operation by operation, as ssim_fwd_kernel / ssim_bwd_kernel:
	// the results are bit-identical to the two-kernel path.
	//   phase 0  A = 52x52 inputs (x, y) -> LDS (zero padding outside the image); |x - y| over the own 32x32
	//   phase 1  horizontal windows of (x, y), (x^2, y^2), xy on the 52 rows x 44 columns (42 used)
	//   phase 2  vertical windows + SSIM algebra on B = 42x42: ssim (summed over the own 32x32) and d ssim / d mu1, d E[x^2], d E[xy]
	//            (zero outside the image: the maps end there); results wait in registers for the barrier, then take the LDS bytes over
	//   phase 3  horizontal windows of the three derivative maps: 42 rows x 32 columns
	//   phase 4  vertical windows -> dL/dx on the own 32x32
	// LDS: 48.3 KB per workgroup (three per CU).
	// ------------------------------------------------------------------------------------------------------------------------
	constexpr int FT = 32;                  // output tile edge
	constexpr int FB = FT + 2 * SR;         // 42: derivative region edge
	constexpr int FA = FB + 2 * SR;         // 52: input region edge
	constexpr int FBC = 44;                 // columns of the phase-1 results (42 used; whole groups of 4)
	constexpr int FSA = 54;                 // s_in row stride in pairs (>= 44 + 10, 108 dwords = 4 mod 8)
	constexpr int FSH = 46;                 // h_m / h_s row stride in pairs (92 dwords)
	constexpr int FSX = 48;                 // h_x row stride in floats
	constexpr int FSD = 46;                 // d_p row stride in pairs
	constexpr int FSQ = 48;                 // d_q row stride in floats
	constexpr int FRV = 7;                  // rows per thread in phase 2 (42 columns x 6 row groups = 252 threads)

	__device__ __forceinline__ void ssim_fused_body(
		const float* __restrict__ img1, const float* __restrict__ img2, int C, int H, int W,
		const float* __restrict__ upstream, float w_l1, float w_ssim, float* __restrict__ dL_dimg1,
		float* __restrict__ partial_l1, float* __restrict__ partial_ssim)
	{
		constexpr int R0_BYTES = FA * FSH * 8;                                  // h_m; later h_p + h_q
		constexpr int IN_BYTES = FA * FSA * 8, HS_BYTES = FA * FSH * 8, HX_BYTES = FA * FSX * 4;
		constexpr int R1_BYTES = IN_BYTES > HS_BYTES + HX_BYTES ? IN_BYTES : HS_BYTES + HX_BYTES;   // s_in; then h_s + h_x; then d_p + d_q
		static_assert(FB * HSTR * 8 + FB * HSTR1B * 4 <= R0_BYTES, "h_p + h_q must fit where h_m was");
		static_assert(FB * FSD * 8 + FB * FSQ * 4 <= R1_BYTES, "d_p + d_q must fit where h_s + h_x were");
		__shared__ __attribute__((aligned(16))) char s_raw[R0_BYTES + R1_BYTES];
		v2f (*h_m)[FSH] = reinterpret_cast<v2f (*)[FSH]>(s_raw);
		v2f (*s_in)[FSA] = reinterpret_cast<v2f (*)[FSA]>(s_raw + R0_BYTES);
		v2f (*h_s)[FSH] = reinterpret_cast<v2f (*)[FSH]>(s_raw + R0_BYTES);
		float (*h_x)[FSX] = reinterpret_cast<float (*)[FSX]>(s_raw + R0_BYTES + HS_BYTES);
		v2f (*d_p)[FSD] = reinterpret_cast<v2f (*)[FSD]>(s_raw + R0_BYTES);                     // (d ssim/d mu1, d ssim/d E11)
		float (*d_q)[FSQ] = reinterpret_cast<float (*)[FSQ]>(s_raw + R0_BYTES + FB * FSD * 8);  // d ssim/d E12
		v2f (*h_p)[HSTR] = reinterpret_cast<v2f (*)[HSTR]>(s_raw);
		float (*h_q)[HSTR1B] = reinterpret_cast<float (*)[HSTR1B]>(s_raw + FB * HSTR * 8);
		__shared__ float red[2][STHREADS / WAVE];

		const TileId tile = ssim_tile_of((W + FT - 1) / FT, (H + FT - 1) / FT, C);
		if (!tile.valid) return;
		const int c = tile.c;
		const int x0 = tile.tx * FT, y0 = tile.ty * FT;
		const int tid = threadIdx.x;
		const size_t plane = (size_t)c * H * W;

		// ---- phase 0: the 52x52 inputs; thread -> one column and rows tid / 52, + 4, + 8, ... (208 of the 256 threads) ----
		float l1 = 0.f;
		{
			constexpr int RPT = STHREADS / FA, TRIPS = (FA + RPT - 1) / RPT;   // 4 rows per trip, 13 trips
			const int lyb = tid / FA, hx = tid - lyb * FA;
			const int gxh = x0 + hx - 2 * SR;
			const bool col_in = tid < RPT * FA && (unsigned)gxh < (unsigned)W;
			const bool col_own = (unsigned)(hx - 2 * SR) < (unsigned)FT;
			v2f p[TRIPS];
#pragma unroll
			for (int t = 0; t < TRIPS; t++)
			{
				const int ly = lyb + t * RPT, gy = y0 + ly - 2 * SR;
				const bool in = col_in && ly < FA && (unsigned)gy < (unsigned)H;
				const size_t o = in ? plane + (size_t)gy * W + gxh : plane;
				const float vx = img1[o], vy = img2[o];
				p[t] = in ? v2f{ vx, vy } : v2f{ 0.0f, 0.0f };
			}
#pragma unroll
			for (int t = 0; t < TRIPS; t++)
			{
				const int ly = lyb + t * RPT;
				if (tid < RPT * FA && ly < FA) s_in[ly][hx] = p[t];
				const bool own = col_own && (unsigned)(ly - 2 * SR) < (unsigned)FT;
				l1 += own ? fabsf(p[t].x - p[t].y) : 0.0f;
			}
			// the two pad columns the last group of four of phase 1 reads (its results, columns 42 and 43, are never used)
			if (tid < FA) { s_in[tid][FA] = v2f{ 0.0f, 0.0f }; s_in[tid][FA + 1] = v2f{ 0.0f, 0.0f }; }
		}
		// the own pixels this thread finishes in phase 4: their values travel meanwhile
		const int lx = tid & (FT - 1), ly0 = (tid >> 5) * SROWS;
		const int gx = x0 + lx;
		float px[SROWS], py[SROWS];
#pragma unroll
		for (int j = 0; j < SROWS; j++)
		{
			const int gy = y0 + ly0 + j;
			const size_t o = (gx < W && gy < H) ? plane + (size_t)gy * W + gx : plane;
			px[j] = img1[o]; py[j] = img2[o];
		}
		__syncthreads();

		// ---- phase 1: horizontal windows; task -> (row, 4 adjacent columns): 52 x 11 tasks in 3 rounds ----
		{
			constexpr int GPR = FBC / 4, NT = FA * GPR, HR = (NT + STHREADS - 1) / STHREADS;
			v2f as[HR][4];
			float ax[HR][4];
#pragma unroll
			for (int r = 0; r < HR; r++)
			{
				const int task = tid + r * STHREADS;
				if (task < NT)
				{
					const int ly = task / GPR, cx = (task - ly * GPR) * 4;
					v2f p[16], sq[14];
					float xy[14];
					const v4f* src = reinterpret_cast<const v4f*>(&s_in[ly][cx]);
#pragma unroll
					for (int i = 0; i < 7; i++) { const v4f q = src[i]; p[2 * i] = v2f{ q.x, q.y }; p[2 * i + 1] = v2f{ q.z, q.w }; }
#pragma unroll
					for (int i = 0; i < 14; i++) { sq[i] = p[i] * p[i]; xy[i] = p[i].x * p[i].y; }
					v2f am[4];
#pragma unroll
					for (int j = 0; j < 4; j++)
					{
						am[j] = GW[0] * p[j]; as[r][j] = GW[0] * sq[j]; ax[r][j] = GW[0] * xy[j];
#pragma unroll
						for (int k = 1; k < 11; k++) { am[j] += GW[k] * p[j + k]; as[r][j] += GW[k] * sq[j + k]; ax[r][j] += GW[k] * xy[j + k]; }
					}
					v4f* dm = reinterpret_cast<v4f*>(&h_m[ly][cx]);
					dm[0] = v4f{ am[0].x, am[0].y, am[1].x, am[1].y }; dm[1] = v4f{ am[2].x, am[2].y, am[3].x, am[3].y };
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			__syncthreads();   // the input tile has been read: its bytes become h_s / h_x
#pragma unroll
			for (int r = 0; r < HR; r++)
			{
				const int task = tid + r * STHREADS;
				if (task < NT)
				{
					const int ly = task / GPR, cx = (task - ly * GPR) * 4;
					v4f* ds = reinterpret_cast<v4f*>(&h_s[ly][cx]);
					ds[0] = v4f{ as[r][0].x, as[r][0].y, as[r][1].x, as[r][1].y }; ds[1] = v4f{ as[r][2].x, as[r][2].y, as[r][3].x, as[r][3].y };
					*reinterpret_cast<v4f*>(&h_x[ly][cx]) = v4f{ ax[r][0], ax[r][1], ax[r][2], ax[r][3] };
				}
			}
		}
		__syncthreads();

		// ---- phase 2: vertical windows + SSIM algebra on the 42x42 derivative region; thread -> (column, 7 adjacent rows) ----
		float sv = 0.f;
		{
			const int bc = tid % FB, rg = tid / FB;          // rg 0..5 (threads 252..255: rg = 6, no work)
			const bool work = rg < FB / FRV;
			const int br0 = work ? rg * FRV : 0;
			v2f dp[FRV];
			float dq[FRV];
			{
				v2f vm[10 + FRV], vs[10 + FRV];
				float vx[10 + FRV];
#pragma unroll
				for (int r = 0; r < 10 + FRV; r++) { vm[r] = h_m[br0 + r][bc]; vs[r] = h_s[br0 + r][bc]; vx[r] = h_x[br0 + r][bc]; }
				const int gxb = x0 + bc - SR;
				const bool col_own = (unsigned)(bc - SR) < (unsigned)FT;
#pragma unroll
				for (int j = 0; j < FRV; j++)
				{
					v2f mu = GW[0] * vm[j], e2 = GW[0] * vs[j];
					float e12 = GW[0] * vx[j];
#pragma unroll
					for (int k = 1; k < 11; k++) { mu += GW[k] * vm[j + k]; e2 += GW[k] * vs[j + k]; e12 += GW[k] * vx[j + k]; }
					const int br = br0 + j, gyb = y0 + br - SR;
					dp[j] = v2f{ 0.0f, 0.0f }; dq[j] = 0.0f;
					if (work && (unsigned)gxb < (unsigned)W && (unsigned)gyb < (unsigned)H)
					{
						const float mu1 = mu.x, mu2 = mu.y, e11 = e2.x, e22 = e2.y;
						const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
						const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
						const float sg1 = e11 - mu1_sq, sg2 = e22 - mu2_sq, sg12 = e12 - mu12;
						const float A = 2.f * mu12 + C1, B = 2.f * sg12 + C2, Cc = mu1_sq + mu2_sq + C1, D = sg1 + sg2 + C2;
						const float rC = __builtin_amdgcn_rcpf(Cc), rD = __builtin_amdgcn_rcpf(D);
						const float inv = rC * rD;
						const float m = A * B * inv;
						const float dm_dA = B * inv, dm_dB = A * inv, dm_dC = -m * rC, dm_dD = -m * rD;
						dp[j] = v2f{ dm_dA * 2.f * mu2 - dm_dB * 2.f * mu2 + dm_dC * 2.f * mu1 - dm_dD * 2.f * mu1, dm_dD };
						dq[j] = 2.f * dm_dB;
						if (col_own && (unsigned)(br - SR) < (unsigned)FT) sv += m;   // the tile's own pixels: each counted by exactly one tile
					}
				}
			}
			__syncthreads();   // every thread has read h_m / h_s / h_x: their bytes become the derivative maps
			if (work)
			{
#pragma unroll
				for (int j = 0; j < FRV; j++) { d_p[br0 + j][bc] = dp[j]; d_q[br0 + j][bc] = dq[j]; }
			}
			// pad columns 42..45 of d_q (the b128 reads of the last group reach column 43) -- d_p needs none (14 pairs from column 28: 41)
			if (tid < FB) { d_q[tid][FB] = 0.0f; d_q[tid][FB + 1] = 0.0f; d_q[tid][FB + 2] = 0.0f; d_q[tid][FB + 3] = 0.0f; }
		}
		__syncthreads();

		// ---- phase 3: horizontal windows of the derivative maps: 42 rows x 8 groups of 4 columns, 2 rounds; results go where h_m was ----
		{
			constexpr int NT = FB * (FT / 4), HR = (NT + STHREADS - 1) / STHREADS;
#pragma unroll
			for (int r = 0; r < HR; r++)
			{
				const int task = tid + r * STHREADS;
				if (task < NT)
				{
					const int ly = task >> 3, cx = (task & 7) * 4;
					v2f p[16];
					float q[16];
					const v4f* sp = reinterpret_cast<const v4f*>(&d_p[ly][cx]);
					const v4f* sq = reinterpret_cast<const v4f*>(&d_q[ly][cx]);
#pragma unroll
					for (int i = 0; i < 7; i++) { const v4f t = sp[i]; p[2 * i] = v2f{ t.x, t.y }; p[2 * i + 1] = v2f{ t.z, t.w }; }
#pragma unroll
					for (int i = 0; i < 4; i++) { const v4f t = sq[i]; q[4 * i] = t.x; q[4 * i + 1] = t.y; q[4 * i + 2] = t.z; q[4 * i + 3] = t.w; }
					v2f ap[4];
					float aq[4];
#pragma unroll
					for (int j = 0; j < 4; j++)
					{
						ap[j] = GW[0] * p[j]; aq[j] = GW[0] * q[j];
#pragma unroll
						for (int k = 1; k < 11; k++) { ap[j] += GW[k] * p[j + k]; aq[j] += GW[k] * q[j + k]; }
					}
					v4f* dpo = reinterpret_cast<v4f*>(&h_p[ly][cx]);
					dpo[0] = v4f{ ap[0].x, ap[0].y, ap[1].x, ap[1].y }; dpo[1] = v4f{ ap[2].x, ap[2].y, ap[3].x, ap[3].y };
					*reinterpret_cast<v4f*>(&h_q[ly][cx]) = v4f{ aq[0], aq[1], aq[2], aq[3] };
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		__syncthreads();

		// ---- phase 4: vertical windows -> the gradient of the tile's own pixels ----
		{
			v2f vp[10 + SROWS];
			float vq[10 + SROWS];
#pragma unroll
			for (int r = 0; r < 10 + SROWS; r++) { vp[r] = h_p[ly0 + r][lx]; vq[r] = h_q[ly0 + r][lx]; }
			const float up = upstream[0];
#pragma unroll
			for (int j = 0; j < SROWS; j++)
			{
				v2f ab = GW[0] * vp[j];
				float d = GW[0] * vq[j];
#pragma unroll
				for (int k = 1; k < 11; k++) { ab += GW[k] * vp[j + k]; d += GW[k] * vq[j + k]; }
				const int gy = y0 + ly0 + j;
				if (gx < W && gy < H)
				{
					const size_t o = plane + (size_t)gy * W + gx;
					const float x = px[j], y = py[j];
					const float diff = x - y;
					const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
					dL_dimg1[o] = up * (w_l1 * sgn + w_ssim * (ab.x + 2.f * x * ab.y + y * d));
				}
			}
		}
		// per-tile partial sums of |x - y| and ssim (fixed order: deterministic)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { l1 += __shfl_down(l1, o); sv += __shfl_down(sv, o); }
		if ((tid & 63) == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = sv; }
		__syncthreads();
		if (tid == 0)
		{
			const int b = tile.index;
			partial_l1[b] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
			partial_ssim[b] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
		}
	}
}

namespace fdgs
{
	// two register allocations of the same body: held to 3 waves per SIMD (168 VGPRs, a few spills: three workgroups per CU, what the
	// LDS allows) or free (176 VGPRs, no spills, two workgroups per CU); FDGS_SSIM_FUSED_WPE=2 in the environment selects the second (A/B)
#define FDGS_SSIM_FUSED_PARAMS const float* __restrict__ img1, const float* __restrict__ img2, int C, int H, int W, const float* __restrict__ upstream, \
		float w_l1, float w_ssim, float* __restrict__ dL_dimg1, float* __restrict__ partial_l1, float* __restrict__ partial_ssim
	__global__ void __launch_bounds__(STHREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) ssim_fused_kernel(FDGS_SSIM_FUSED_PARAMS)
	{
		ssim_fused_body(img1, img2, C, H, W, upstream, w_l1, w_ssim, dL_dimg1, partial_l1, partial_ssim);
	}
	__global__ void __launch_bounds__(STHREADS) ssim_fused_kernel_wpe2(FDGS_SSIM_FUSED_PARAMS)
	{
		ssim_fused_body(img1, img2, C, H, W, upstream, w_l1, w_ssim, dL_dimg1, partial_l1, partial_ssim);
	}
#undef FDGS_SSIM_FUSED_PARAMS
}

extern "C" int fdgs_l1_ssim_value_and_grad(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                           const float* upstream, float lambda_dssim, float* dL_dimg,
                                           float* partial_l1, float* partial_ssim, void* stream)
{
	using namespace fdgs;
	static_assert(STY == 32 && SROWS == 4, "the fused kernel shares the 32-row tile's partial-sum layout with the two-kernel path");
	if (!img || !gt || !upstream || !dL_dimg || !partial_l1 || !partial_ssim || C <= 0 || H <= 0 || W <= 0) return FDGS_ERR_INVALID_ARG;
	const float n = (float)C * (float)H * (float)W;
	const float w_l1 = (1.0f - lambda_dssim) / n, w_ssim = -lambda_dssim / n;
	const dim3 grid(ssim_grid(div_up(W, FT), div_up(H, FT), C)), block(STHREADS, 1, 1);
	static const bool wpe2 = []() { const char* e = getenv("FDGS_SSIM_FUSED_WPE"); return e && e[0] == '2'; }();
	if (wpe2) hipLaunchKernelGGL(ssim_fused_kernel_wpe2, grid, block, 0, (hipStream_t)stream, img, gt, C, H, W, upstream, w_l1, w_ssim, dL_dimg, partial_l1, partial_ssim);
	else hipLaunchKernelGGL(ssim_fused_kernel, grid, block, 0, (hipStream_t)stream, img, gt, C, H, W, upstream, w_l1, w_ssim, dL_dimg, partial_l1, partial_ssim);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_l1_ssim_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                    float* dm_dmu1, float* dm_de11, float* dm_de12,
                                    float* partial_l1, float* partial_ssim, void* stream)
{
	using namespace fdgs;
	if (!img || !gt || !dm_dmu1 || !dm_de11 || !dm_de12 || !partial_l1 || !partial_ssim || C <= 0 || H <= 0 || W <= 0) return FDGS_ERR_INVALID_ARG;
	const dim3 grid(ssim_grid(div_up(W, STX), div_up(H, STY), C)), block(STHREADS, 1, 1);
	hipLaunchKernelGGL(ssim_fwd_kernel, grid, block, 0, (hipStream_t)stream, img, gt, C, H, W, dm_dmu1, dm_de11, dm_de12, partial_l1, partial_ssim);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_l1_ssim_backward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                     const float* dm_dmu1, const float* dm_de11, const float* dm_de12,
                                     const float* upstream, float lambda_dssim, float* dL_dimg, void* stream)
{
	using namespace fdgs;
	if (!img || !gt || !dm_dmu1 || !dm_de11 || !dm_de12 || !upstream || !dL_dimg || C <= 0 || H <= 0 || W <= 0) return FDGS_ERR_INVALID_ARG;
	const float n = (float)C * (float)H * (float)W;
	const float w_l1 = (1.0f - lambda_dssim) / n, w_ssim = -lambda_dssim / n;
	const dim3 grid(ssim_grid(div_up(W, STX), div_up(H, STY), C)), block(STHREADS, 1, 1);
	hipLaunchKernelGGL(ssim_bwd_kernel, grid, block, 0, (hipStream_t)stream, img, gt, C, H, W, dm_dmu1, dm_de11, dm_de12, upstream, w_l1, w_ssim, dL_dimg);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

namespace fdgs
{
	// loss = (1 - lambda) * sum(l1) / n + lambda * (1 - sum(ssim) / n), fixed summation order (deterministic)
	// (blockIdx.x = view: the batch call below reduces every view of an optimizer step in ONE launch; view_stride = floats between the
	// partial arrays of consecutive views, 0 for the single call)
	__global__ void __launch_bounds__(1024) l1_ssim_finish_kernel(const float* __restrict__ partial_l1, const float* __restrict__ partial_ssim,
	                                                              int nparts, float inv_n, float lambda_dssim, float* __restrict__ out, long long view_stride)
	{
		partial_l1 += (size_t)blockIdx.x * view_stride; partial_ssim += (size_t)blockIdx.x * view_stride; out += 3 * (size_t)blockIdx.x;
		// 1024 threads, up to 8 partials per thread and sum in flight at once (the kernel is pure latency), then a fixed
		// shuffle tree per wave and a fixed order over the 16 waves: deterministic
		__shared__ float r0[16], r1[16];
		float a = 0.f, b = 0.f;
		for (int base = 0; base < nparts; base += 8 * 1024)
		{
			float va[8], vb[8];
#pragma unroll
			for (int k = 0; k < 8; k++)
			{
				const int i = base + k * 1024 + (int)threadIdx.x;
				va[k] = i < nparts ? partial_l1[i] : 0.f;
				vb[k] = i < nparts ? partial_ssim[i] : 0.f;
			}
			a += ((va[0] + va[1]) + (va[2] + va[3])) + ((va[4] + va[5]) + (va[6] + va[7]));
			b += ((vb[0] + vb[1]) + (vb[2] + vb[3])) + ((vb[4] + vb[5]) + (vb[6] + vb[7]));
		}
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
		if ((threadIdx.x & 63) == 0) { r0[threadIdx.x >> 6] = a; r1[threadIdx.x >> 6] = b; }
		__syncthreads();
		if (threadIdx.x == 0)
		{
			float s0 = 0.f, s1 = 0.f;
			for (int w = 0; w < 16; w++) { s0 += r0[w]; s1 += r1[w]; }
			const float l1 = s0 * inv_n, ss = s1 * inv_n;
			out[0] = (1.0f - lambda_dssim) * l1 + lambda_dssim * (1.0f - ss);
			out[1] = l1;
			out[2] = ss;
		}
	}
}

extern "C" int fdgs_l1_ssim_loss(const float* partial_l1, const float* partial_ssim, int32_t num_partials, int32_t C, int32_t H, int32_t W,
                                 float lambda_dssim, float* loss_l1_ssim, void* stream)
{
	using namespace fdgs;
	if (!partial_l1 || !partial_ssim || !loss_l1_ssim || num_partials <= 0 || C <= 0 || H <= 0 || W <= 0) return FDGS_ERR_INVALID_ARG;
	const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
	hipLaunchKernelGGL(l1_ssim_finish_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial_l1, partial_ssim, num_partials, inv_n, lambda_dssim, loss_l1_ssim, 0ll);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_l1_ssim_loss_batch(const float* partials, int32_t num_views, int32_t num_partials, int32_t C, int32_t H, int32_t W,
                                       float lambda_dssim, float* losses, void* stream)
{
	using namespace fdgs;
	if (!partials || !losses || num_views <= 0 || num_partials <= 0 || C <= 0 || H <= 0 || W <= 0) return FDGS_ERR_INVALID_ARG;
	const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
	hipLaunchKernelGGL(l1_ssim_finish_kernel, dim3(num_views), dim3(1024), 0, (hipStream_t)stream, partials, partials + num_partials, num_partials, inv_n,
	                   lambda_dssim, losses, 2ll * num_partials);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_l1_ssim_num_partials(int32_t C, int32_t H, int32_t W)
{
	using namespace fdgs;
	return div_up(W, STX) * div_up(H, STY) * C;
}

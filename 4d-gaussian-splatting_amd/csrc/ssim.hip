// ssim.hip -- fused photometric loss (1-l) L1 + l (1 - SSIM), forward and backward (gfx950).
//
// The step right after the rasterizer in every training iteration (reference train.py:115-117,
// utils/loss_utils.py:17-64): L1 + SSIM with an 11x11 Gaussian window (sigma 1.5, zero padding,
// C1 = 0.01^2, C2 = 0.03^2, mean over all pixels and channels).  The reference runs it as five
// grouped 11x11 convolutions forward plus their backward through the DL library; on ROCm that
// is ~7.6 ms per 1352x1014 image, four times the whole rasterizer.  Here:
//   forward : one pass.  A 16x16 output tile loads its 26x26 halo of both images into LDS, does
//             the separable window (horizontal, then vertical) for the five moments
//             (x, y, x^2, y^2, xy), evaluates SSIM and the three partial derivatives
//             d ssim/d mu1, d ssim/d E[x^2], d ssim/d E[xy] per pixel (kept for the backward),
//             and writes per-tile partial sums of |x-y| and ssim (summed by the host: deterministic).
//   backward: dL/dx(p) = w_l1 sign(x-y) + w_ssim [ (W * dmu1)(p) + 2 x(p) (W * dE11)(p) + y(p) (W * dE12)(p) ]
//             -- three more separable windows over the stored derivative maps (W symmetric).
// Pure streaming fp32 work: ~70 B/pixel-channel forward, ~60 B backward; HBM-bound, no MFMA.
#include "fdgs_common.h"

namespace fdgs
{
	constexpr int ST = 16;             // output tile edge
	constexpr int SR = 5;              // window radius (11 taps)
	constexpr int SH = ST + 2 * SR;    // 26: tile + halo

	// gaussian(11, 1.5) normalised, as utils/loss_utils.py:23-25
	__device__ constexpr float GW[11] = {
		0.0010283801f, 0.0075987582f, 0.0360007733f, 0.1093606874f, 0.2130055279f, 0.2660117149f,
		0.2130055279f, 0.1093606874f, 0.0360007733f, 0.0075987582f, 0.0010283801f };

	__global__ void __launch_bounds__(ST * ST) ssim_fwd_kernel(
		const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
		float* __restrict__ dm_dmu1, float* __restrict__ dm_de11, float* __restrict__ dm_de12,
		float* __restrict__ partial_l1, float* __restrict__ partial_ssim)
	{
		__shared__ float s1[SH][SH + 1];
		__shared__ float s2[SH][SH + 1];
		__shared__ float h[5][SH][ST + 1];   // horizontally filtered moments
		__shared__ float red[2][ST * ST / WAVE];

		const int c = blockIdx.z;
		const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
		const int tid = threadIdx.y * ST + threadIdx.x;
		const size_t plane = (size_t)c * H * W;

		for (int i = tid; i < SH * SH; i += ST * ST)
		{
			const int ly = i / SH, lx = i % SH;
			const int gy = y0 + ly - SR, gx = x0 + lx - SR;
			const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
			const size_t o = plane + (size_t)gy * W + gx;
			s1[ly][lx] = in ? img1[o] : 0.0f;   // zero padding (F.conv2d padding = 5)
			s2[ly][lx] = in ? img2[o] : 0.0f;
		}
		__syncthreads();

		for (int i = tid; i < SH * ST; i += ST * ST)
		{
			const int ly = i / ST, lx = i % ST;
			float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
			for (int k = 0; k < 11; k++)
			{
				const float w = GW[k], u = s1[ly][lx + k], v = s2[ly][lx + k];
				a += w * u; b += w * v; aa += w * u * u; bb += w * v * v; ab += w * u * v;
			}
			h[0][ly][lx] = a; h[1][ly][lx] = b; h[2][ly][lx] = aa; h[3][ly][lx] = bb; h[4][ly][lx] = ab;
		}
		__syncthreads();

		const int lx = threadIdx.x, ly = threadIdx.y;
		const int gx = x0 + lx, gy = y0 + ly;
		float l1 = 0.f, sv = 0.f;
		if (gx < W && gy < H)
		{
			float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
			for (int k = 0; k < 11; k++)
			{
				const float w = GW[k];
				mu1 += w * h[0][ly + k][lx]; mu2 += w * h[1][ly + k][lx];
				e11 += w * h[2][ly + k][lx]; e22 += w * h[3][ly + k][lx]; e12 += w * h[4][ly + k][lx];
			}
			const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
			const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
			const float sg1 = e11 - mu1_sq, sg2 = e22 - mu2_sq, sg12 = e12 - mu12;
			const float A = 2.f * mu12 + C1, B = 2.f * sg12 + C2, Cc = mu1_sq + mu2_sq + C1, D = sg1 + sg2 + C2;
			const float inv = 1.0f / (Cc * D);
			const float m = A * B * inv;
			// total derivative w.r.t. mu1 (through A, B, Cc, D), and w.r.t. the raw moments E[x^2], E[xy]
			const float dm_dA = B * inv, dm_dB = A * inv, dm_dC = -m / Cc, dm_dD = -m / D;
			const size_t o = plane + (size_t)gy * W + gx;
			dm_dmu1[o] = dm_dA * 2.f * mu2 - dm_dB * 2.f * mu2 + dm_dC * 2.f * mu1 - dm_dD * 2.f * mu1;
			dm_de11[o] = dm_dD;
			dm_de12[o] = 2.f * dm_dB;
			sv = m;
			l1 = fabsf(s1[ly + SR][lx + SR] - s2[ly + SR][lx + SR]);
		}
		// per-tile partial sums (wave shuffle + 4 partials)
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { l1 += __shfl_down(l1, o); sv += __shfl_down(sv, o); }
		if ((tid & 63) == 0) { red[0][tid >> 6] = l1; red[1][tid >> 6] = sv; }
		__syncthreads();
		if (tid == 0)
		{
			const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
			partial_l1[b] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
			partial_ssim[b] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
		}
	}

	__global__ void __launch_bounds__(ST * ST) ssim_bwd_kernel(
		const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
		const float* __restrict__ dm_dmu1, const float* __restrict__ dm_de11, const float* __restrict__ dm_de12,
		const float* __restrict__ upstream, float w_l1, float w_ssim, float* __restrict__ dL_dimg1)
	{
		__shared__ float s[3][SH][SH + 1];
		__shared__ float h[3][SH][ST + 1];

		const int c = blockIdx.z;
		const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
		const int tid = threadIdx.y * ST + threadIdx.x;
		const size_t plane = (size_t)c * H * W;

		for (int i = tid; i < SH * SH; i += ST * ST)
		{
			const int ly = i / SH, lx = i % SH;
			const int gy = y0 + ly - SR, gx = x0 + lx - SR;
			const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
			const size_t o = plane + (size_t)gy * W + gx;
			s[0][ly][lx] = in ? dm_dmu1[o] : 0.0f;
			s[1][ly][lx] = in ? dm_de11[o] : 0.0f;
			s[2][ly][lx] = in ? dm_de12[o] : 0.0f;
		}
		__syncthreads();
		for (int i = tid; i < SH * ST; i += ST * ST)
		{
			const int ly = i / ST, lx = i % ST;
			float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
			for (int k = 0; k < 11; k++)
			{
				const float w = GW[k];
				a += w * s[0][ly][lx + k]; b += w * s[1][ly][lx + k]; d += w * s[2][ly][lx + k];
			}
			h[0][ly][lx] = a; h[1][ly][lx] = b; h[2][ly][lx] = d;
		}
		__syncthreads();
		const int lx = threadIdx.x, ly = threadIdx.y;
		const int gx = x0 + lx, gy = y0 + ly;
		if (gx < W && gy < H)
		{
			float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
			for (int k = 0; k < 11; k++)
			{
				const float w = GW[k];
				a += w * h[0][ly + k][lx]; b += w * h[1][ly + k][lx]; d += w * h[2][ly + k][lx];
			}
			const size_t o = plane + (size_t)gy * W + gx;
			const float x = img1[o], y = img2[o];
			const float up = upstream[0];
			const float diff = x - y;
			const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
			dL_dimg1[o] = up * (w_l1 * sgn + w_ssim * (a + 2.f * x * b + y * d));
		}
	}
}

extern "C" int fdgs_l1_ssim_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                    float* dm_dmu1, float* dm_de11, float* dm_de12,
                                    float* partial_l1, float* partial_ssim, void* stream)
{
	using namespace fdgs;
	if (!img || !gt || !dm_dmu1 || !dm_de11 || !dm_de12 || !partial_l1 || !partial_ssim || C <= 0 || H <= 0 || W <= 0) return FDGS_ERR_INVALID_ARG;
	const dim3 grid(div_up(W, ST), div_up(H, ST), C), block(ST, ST, 1);
	hipLaunchKernelGGL(ssim_fwd_kernel, grid, block, 0, (hipStream_t)stream, img, gt, H, W, dm_dmu1, dm_de11, dm_de12, partial_l1, partial_ssim);
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
	const dim3 grid(div_up(W, ST), div_up(H, ST), C), block(ST, ST, 1);
	hipLaunchKernelGGL(ssim_bwd_kernel, grid, block, 0, (hipStream_t)stream, img, gt, H, W, dm_dmu1, dm_de11, dm_de12, upstream, w_l1, w_ssim, dL_dimg);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

extern "C" int fdgs_l1_ssim_num_partials(int32_t C, int32_t H, int32_t W)
{
	using namespace fdgs;
	return div_up(W, ST) * div_up(H, ST) * C;
}

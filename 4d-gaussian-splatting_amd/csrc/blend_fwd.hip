// blend_fwd.hip -- front-to-back alpha blend, one wave64 per 8x8 pixel block (gfx950).
//
// Semantics: reference renderCUDA forward (forward.cu:501-626) -- per pixel, walk
// the tile's depth-sorted list; skip power > 0 and alpha < 1/255; stop (and do
// not count) when T*(1-alpha) < 1e-4; accumulate RGB(3) + flow(2) + depth(1);
// out = C + T*bg; n_contrib = 1-based list position of the last contributor.
//
// Structure (see blend_common.h): per 64 list entries, each lane gathers one packed
// 48-byte record (the reference gathers colour / flow / depth per contribution from
// three arrays, forward.cu:601-604), culls it against the wave's pixel block, and the
// survivors are compacted into LDS; the inner loop then reads only wave-uniform LDS
// words.  No workgroup barriers: a wave is its own workgroup and stops as soon as
// its 64 pixels are saturated (the reference needs the whole 256-pixel tile to be).
// fp32 with FMA contraction and the hardware exp2 (exp(x) = exp2(x * log2 e)).
#include "blend_common.h"

namespace fdgs
{
	__global__ void __launch_bounds__(WAVE) blend_fwd_kernel(
		const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ records,
		int W, int H, int grid_x, int ntiles, const float* __restrict__ bg,
		float* __restrict__ out_color, float* __restrict__ out_flow, float* __restrict__ out_depth, float* __restrict__ out_T,
		float* __restrict__ final_T, uint32_t* __restrict__ n_contrib)
	{
		// wave-private queue of the surviving entries of the current 64-entry chunk (+2: inert padding entry)
		__shared__ float4 s_a[WAVE + 2];
		__shared__ float4 s_b[WAVE + 2];
		__shared__ float4 s_c[WAVE + 2];
		__shared__ uint32_t s_pos[WAVE + 2];

		const BlockId blk = block_of(blockIdx.x, ntiles);
		if (blk.tile >= ntiles) return;
		const int lane = threadIdx.x;
		const int bx0 = (blk.tile % grid_x) * TILE_X + (blk.sub & 1) * BLK;
		const int by0 = (blk.tile / grid_x) * TILE_Y + (blk.sub >> 1) * BLK;
		if (bx0 >= W || by0 >= H) return; // block entirely outside the image
		const int px = bx0 + (lane & (BLK - 1)), py = by0 + (lane >> 3);
		const bool inside = px < W && py < H;
		const float pixfx = (float)px, pixfy = (float)py;
		const float rx0 = (float)bx0, rx1 = (float)min(bx0 + BLK - 1, W - 1);
		const float ry0 = (float)by0, ry1 = (float)min(by0 + BLK - 1, H - 1);

		const uint2 range = ranges[blk.tile];
		const int n = (int)(range.y - range.x);
		const unsigned long long lt_mask = (1ull << lane) - 1ull;

		bool done = !inside;
		float T = 1.0f;
		uint32_t last_contributor = 0;
		float C0 = 0.f, C1 = 0.f, C2 = 0.f, F0 = 0.f, F1 = 0.f, D = 0.f;

		for (int base = 0; base < n; base += WAVE)
		{
			if (__ballot(!done) == 0ull) break; // all 64 pixels saturated
			const int pos = base + lane;
			bool keep = false;
			uint32_t id = 0;
			float4 a, b;
			if (pos < n)
			{
				id = point_list[range.x + pos];
				a = records[3 * (size_t)id + 0];
				b = records[3 * (size_t)id + 1];
				keep = block_reaches(a, b, rx0, rx1, ry0, ry1);
			}
			const unsigned long long mask = __ballot(keep);
			const int cnt = __popcll(mask);
			if (keep)
			{
				const int slot = __popcll(mask & lt_mask);
				s_a[slot] = a;
				s_b[slot] = b;
				s_c[slot] = records[3 * (size_t)id + 2];
				s_pos[slot] = (uint32_t)pos;
			}
			if (lane == 0 && (cnt & 1))
			{
				// inert padding so the 2x unrolled loop below needs no tail: opacity 0 -> alpha 0 -> rejected
				s_a[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
				s_b[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
				s_c[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
				s_pos[cnt] = 0u;
			}
			__syncthreads(); // single-wave workgroup: orders the LDS writes before the cross-lane reads

			// Branch-free inner loop, two entries per trip with all LDS reads issued up front: a wave's
			// progress here is bound by its dependent-latency chain (LDS -> exp -> compares), not by issue
			// rate, so the per-pixel tests of the reference (forward.cu:582-597) become predicates + selects.
			for (int j = 0; j < cnt; j += 2)
			{
				const float4 a0 = s_a[j], b0 = s_b[j], c0 = s_c[j];
				const float4 a1 = s_a[j + 1], b1 = s_b[j + 1], c1 = s_c[j + 1];
				const uint32_t p0 = s_pos[j], p1 = s_pos[j + 1];
#define FDGS_BLEND_ONE(ea, eb, ec, epos)                                                                  \
				{                                                                                         \
					const float dx = ea.x - pixfx, dy = ea.y - pixfy;                                     \
					const float power = -0.5f * (ea.z * dx * dx + eb.x * dy * dy) - ea.w * dx * dy;      \
					const float alpha = fminf(0.99f, eb.y * fast_exp(power));                             \
					const float test_T = T * (1.0f - alpha);                                              \
					const bool valid = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);             \
					const bool stop = valid && (test_T < 0.0001f);                                        \
					const bool contrib = valid && !stop;                                                  \
					const float w = contrib ? alpha * T : 0.0f;                                           \
					C0 += eb.z * w; C1 += eb.w * w; C2 += ec.x * w;                                       \
					D += ec.y * w;                                                                        \
					F0 += ec.z * w; F1 += ec.w * w;                                                       \
					T = contrib ? test_T : T;                                                             \
					last_contributor = contrib ? epos + 1u : last_contributor;                            \
					done = done || stop;                                                                  \
				}
				FDGS_BLEND_ONE(a0, b0, c0, p0)
				FDGS_BLEND_ONE(a1, b1, c1, p1)
#undef FDGS_BLEND_ONE
				if (__ballot(!done) == 0ull) break;
			}
			__syncthreads(); // the queue is rewritten by the next chunk
		}

		if (inside)
		{
			const size_t pix_id = (size_t)W * py + px, HW = (size_t)H * W;
			final_T[pix_id] = T;
			out_T[pix_id] = T;
			n_contrib[pix_id] = last_contributor;
			out_color[0 * HW + pix_id] = C0 + T * bg[0];
			out_color[1 * HW + pix_id] = C1 + T * bg[1];
			out_color[2 * HW + pix_id] = C2 + T * bg[2];
			out_flow[0 * HW + pix_id] = F0;
			out_flow[1 * HW + pix_id] = F1;
			out_depth[pix_id] = D;
		}
	}

	hipError_t launch_blend_fwd(const fdgs_scene& s, const fdgs_forward_out& out, const float* records,
	                            const uint32_t* point_list, const uint32_t* ranges,
	                            float* final_T, uint32_t* n_contrib, hipStream_t stream)
	{
		const int gx = div_up(s.W, TILE_X), gy = div_up(s.H, TILE_Y);
		const int ntiles = gx * gy;
		hipLaunchKernelGGL(blend_fwd_kernel, dim3(blend_grid(ntiles)), dim3(WAVE), 0, stream,
		                   reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(records),
		                   s.W, s.H, gx, ntiles, s.bg,
		                   out.out_color, out.out_flow, out.out_depth, out.out_T, final_T, n_contrib);
		return hipGetLastError();
	}
}

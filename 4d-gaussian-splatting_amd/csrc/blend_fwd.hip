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
	typedef float v2f __attribute__((ext_vector_type(2)));

	// FLOW = false: the scene has no flow_2d input (fdgs_scene.flows == NULL, the training default): every record's flow is zero, so
	// is the flow image -- its two accumulators, their queue row and its LDS traffic are left out and zeros are stored.
	template <bool FLOW>
	__global__ void __launch_bounds__(WAVE) blend_fwd_kernel(
		const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ records,
		const uint32_t* __restrict__ tile_order, int W, int H, int grid_x, int ntiles, const float* __restrict__ bg,
		float* __restrict__ out_color, float* __restrict__ out_flow, float* __restrict__ out_depth, float* __restrict__ out_T,
		float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
		unsigned long long* __restrict__ cull_bits /* BinLayout::cull_bits or NULL */, uint32_t cull_stride, uint32_t cull_word_off,
		uint32_t* __restrict__ ctl)
	{
		// Wave-private queue of the surviving entries of the current 64-entry chunk, stored as PAIRS of entries with the
		// two entries interleaved word by word: (x0,x1,y0,y1) (A0,A1,B0,B1) (C0,C1,o0,o1) (r0,r1,g0,g1) (b0,b1,d0,d1)
		// (fx0,fx1,fy0,fy1).  One b128 read then delivers the same quantity of both entries in an aligned register
		// pair, so the per-entry arithmetic up to alpha runs as packed fp32 (v_pk_*: two entries per instruction).
		constexpr int QP = WAVE / 2 + 1;
		// row 6: list position + 1 of the two entries (= n_contrib if it is the last contributor) in .x / .y -- same stride as the
		// other rows, so the loop addresses the whole queue with one base register and immediate offsets
		__shared__ float4 s_q[7][QP];

		const int lane = threadIdx.x;
		// where the backward finds the cull planes: stride and offset (64-bit words from the start of the binning buffer)
		if (blockIdx.x == 0 && lane == 0 && ctl != nullptr) { ctl[2] = cull_bits ? cull_stride : 0u; ctl[3] = cull_word_off; }
		const BlockId blk = block_of(blockIdx.x, ntiles, tile_order);
		if (blk.tile >= ntiles) return;
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

		lanemask done = mask_of(!inside);   // pixels that take no further contribution
		float T = 1.0f;
		uint32_t last_contributor = 0;
		v2f acc_r = { 0.f, 0.f }, acc_g = acc_r, acc_b = acc_r, acc_d = acc_r, acc_fx = acc_r, acc_fy = acc_r;

		for (int base = 0; base < n; base += WAVE)
		{
			if (done == ~0ull) break; // all 64 pixels saturated
			const int pos = base + lane;
			bool keep = false;
			uint32_t id = 0;
			float4 a, b;
			if (pos < n)
			{
				id = point_list[range.x + pos];
				a = record_word(records, id, 0);
				b = record_word(records, id, 1);
				keep = block_reaches(a, b, rx0, rx1, ry0, ry1);
			}
			const unsigned long long mask = __ballot(keep);
			const int cnt = __popcll(mask);
			// kept for the backward (bit i = list position base + i): it takes the same 64-entry chunks and skips the test
			if (cull_bits != nullptr && lane == 0)
				cull_bits[(size_t)blk.sub * cull_stride + (range.x >> 6) + (uint32_t)blk.tile + (uint32_t)(base >> 6)] = mask;
			if (keep)
			{
				const int slot = __popcll(mask & lt_mask);
				const float4 c = record_word(records, id, 2);
				const int pr = slot >> 1, h = slot & 1;
				float* q0 = reinterpret_cast<float*>(&s_q[0][pr]) + h;
				float* q1 = reinterpret_cast<float*>(&s_q[1][pr]) + h;
				float* q2 = reinterpret_cast<float*>(&s_q[2][pr]) + h;
				float* q3 = reinterpret_cast<float*>(&s_q[3][pr]) + h;
				float* q4 = reinterpret_cast<float*>(&s_q[4][pr]) + h;
				float* q5 = reinterpret_cast<float*>(&s_q[5][pr]) + h;
				q0[0] = a.x; q0[2] = a.y; q1[0] = a.z; q1[2] = a.w; q2[0] = b.x; q2[2] = b.y;
				q3[0] = b.z; q3[2] = b.w; q4[0] = c.x; q4[2] = c.y;
				if (FLOW) { q5[0] = c.z; q5[2] = c.w; }
				(reinterpret_cast<uint32_t*>(&s_q[6][pr]))[h] = (uint32_t)pos + 1u;
			}
			if (lane == 0 && (cnt & 1))
			{
				// inert second half of the last pair: opacity 0 -> alpha 0 -> rejected
				const int pr = cnt >> 1;
#pragma unroll
				for (int k = 0; k < (FLOW ? 6 : 5); k++) { float* q = reinterpret_cast<float*>(&s_q[k][pr]) + 1; q[0] = 0.f; q[2] = 0.f; }
				(reinterpret_cast<uint32_t*>(&s_q[6][pr]))[1] = 0u;
			}
			__syncthreads(); // single-wave workgroup: orders the LDS writes before the cross-lane reads

			// Branch-free inner loop, one PAIR of entries per trip.  Everything up to alpha is packed arithmetic on the two
			// entries at once, with the reference's association per element (forward.cu:585-590); the transmittance
			// logic (forward.cu:591-597) is sequential by nature and runs entry by entry on the halves; the six
			// accumulators are kept as (even entries, odd entries) pairs and folded after the loop.
			const int npairs = (cnt + 1) >> 1;
			for (int i = 0; i < npairs; i++)
			{
				const float4 Q0 = s_q[0][i], Q1 = s_q[1][i], Q2 = s_q[2][i], Q3 = s_q[3][i], Q4 = s_q[4][i];
				float4 Q5 = make_float4(0.f, 0.f, 0.f, 0.f);
				if (FLOW) Q5 = s_q[5][i];
				const uint2 pp = *reinterpret_cast<const uint2*>(&s_q[6][i]);
				const v2f dx = v2f{ Q0.x, Q0.y } - pixfx, dy = v2f{ Q0.z, Q0.w } - pixfy;
				const v2f cA = { Q1.x, Q1.y }, cB = { Q1.z, Q1.w }, cC = { Q2.x, Q2.y }, op = { Q2.z, Q2.w };
				const v2f s2 = __builtin_elementwise_fma(cC * dy, dy, (cA * dx) * dx);
				const v2f power = __builtin_elementwise_fma(v2f{ -0.5f, -0.5f }, s2, -((cB * dx) * dy));
				const v2f e2 = power * 1.4426950408889634f;
				const v2f al = op * v2f{ __builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y) };
				const float alpha0 = fminf(0.99f, al.x), alpha1 = fminf(0.99f, al.y);
				v2f w;
				// forward.cu:588-597 on lane masks: valid = not finished, power <= 0, alpha >= 1/255; a valid entry that would take T
				// below 1e-4 finishes the pixel (and is not counted), any other valid entry contributes
#define FDGS_BLEND_STEP(alpha, pw, pos1, wout)                                                            \
				{                                                                                         \
					const float test_T = T * (1.0f - alpha);                                              \
					const lanemask valid = ~done & mask_of(!(pw > 0.0f)) & mask_of(!(alpha < 1.0f / 255.0f)); \
					const lanemask low = mask_of(test_T < 0.0001f);                                       \
					const lanemask contrib = valid & ~low;                                                \
					wout = mask_select(contrib, alpha * T, 0.0f);                                         \
					T = mask_select(contrib, test_T, T);                                                  \
					last_contributor = mask_select(contrib, pos1, last_contributor);                      \
					done |= valid & low;                                                                  \
				}
				FDGS_BLEND_STEP(alpha0, power.x, pp.x, w.x)
				FDGS_BLEND_STEP(alpha1, power.y, pp.y, w.y)
#undef FDGS_BLEND_STEP
				acc_r = __builtin_elementwise_fma(v2f{ Q3.x, Q3.y }, w, acc_r);
				acc_g = __builtin_elementwise_fma(v2f{ Q3.z, Q3.w }, w, acc_g);
				acc_b = __builtin_elementwise_fma(v2f{ Q4.x, Q4.y }, w, acc_b);
				acc_d = __builtin_elementwise_fma(v2f{ Q4.z, Q4.w }, w, acc_d);
				if (FLOW)
				{
					acc_fx = __builtin_elementwise_fma(v2f{ Q5.x, Q5.y }, w, acc_fx);
					acc_fy = __builtin_elementwise_fma(v2f{ Q5.z, Q5.w }, w, acc_fy);
				}
				if (done == ~0ull) break;
			}
			__syncthreads(); // the queue is rewritten by the next chunk
		}

		if (inside)
		{
			const size_t pix_id = (size_t)W * py + px, HW = (size_t)H * W;
			final_T[pix_id] = T;
			out_T[pix_id] = T;
			n_contrib[pix_id] = last_contributor;
			out_color[0 * HW + pix_id] = (acc_r.x + acc_r.y) + T * bg[0];
			out_color[1 * HW + pix_id] = (acc_g.x + acc_g.y) + T * bg[1];
			out_color[2 * HW + pix_id] = (acc_b.x + acc_b.y) + T * bg[2];
			out_flow[0 * HW + pix_id] = FLOW ? acc_fx.x + acc_fx.y : 0.0f;
			out_flow[1 * HW + pix_id] = FLOW ? acc_fy.x + acc_fy.y : 0.0f;
			out_depth[pix_id] = acc_d.x + acc_d.y;
		}
	}

	hipError_t launch_blend_fwd(const fdgs_scene& s, const fdgs_forward_out& out, const float* records,
	                            const uint32_t* point_list, const uint32_t* ranges, const uint32_t* tile_order,
	                            float* final_T, uint32_t* n_contrib, unsigned long long* cull_bits, uint32_t cull_stride, uint32_t cull_word_off,
	                            uint32_t* ctl, hipStream_t stream)
	{
		const int gx = div_up(s.W, TILE_X), gy = div_up(s.H, TILE_Y);
		const int ntiles = gx * gy;
		if (s.P >= (1 << 26)) return hipErrorInvalidValue;   // 32-bit byte offsets into the 48-byte records
#define LAUNCH_FWD(FLOW) hipLaunchKernelGGL(blend_fwd_kernel<FLOW>, dim3(blend_grid(ntiles)), dim3(WAVE), 0, stream, \
		                   reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(records), \
		                   tile_order, s.W, s.H, gx, ntiles, s.bg, \
		                   out.out_color, out.out_flow, out.out_depth, out.out_T, final_T, n_contrib, cull_bits, cull_stride, cull_word_off, ctl)
		if (s.flows != nullptr) LAUNCH_FWD(true);
		else LAUNCH_FWD(false);
#undef LAUNCH_FWD
		return hipGetLastError();
	}

	// Parity-test introspection (fdgs_debug_block_reaches): the block cull of blend_common.h next to the brute force it
	// must never contradict -- the per-pixel test of the blend kernels above (same arithmetic, same translation unit),
	// evaluated on every pixel centre of the rectangle.  tuples: [n][10] = x, y, conic xx / xy / yy, opacity, rx0, rx1, ry0,
	// ry1 (pixel-centre bounds of the block, already clamped to the image).  out: [n][2] = block_reaches, any pixel passes.
	__global__ void block_reaches_debug_kernel(int n, const float* __restrict__ t, uint8_t* __restrict__ out)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= n) return;
		const float* p = t + 10 * (size_t)i;
		const float4 a = make_float4(p[0], p[1], p[2], p[3]);
		const float4 b = make_float4(p[4], p[5], 0.f, 0.f);
		const float rx0 = p[6], rx1 = p[7], ry0 = p[8], ry1 = p[9];
		out[2 * (size_t)i] = block_reaches(a, b, rx0, rx1, ry0, ry1) ? 1 : 0;
		bool any = false;
		for (float py = ry0; py <= ry1; py += 1.0f)
			for (float px = rx0; px <= rx1; px += 1.0f)
			{
				// blend_fwd_kernel's inner loop, one entry
				const float dx = a.x - px, dy = a.y - py;
				const float s2 = fmaf(b.x * dy, dy, (a.z * dx) * dx);
				const float power = fmaf(-0.5f, s2, -((a.w * dx) * dy));
				const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power * 1.4426950408889634f));
				any = any || (!(power > 0.0f) && !(alpha < 1.0f / 255.0f));
			}
		out[2 * (size_t)i + 1] = any ? 1 : 0;
	}
	hipError_t launch_block_reaches_debug(int n, const float* tuples, uint8_t* out, hipStream_t stream)
	{
		hipLaunchKernelGGL(block_reaches_debug_kernel, dim3(div_up(n, 256)), dim3(256), 0, stream, n, tuples, out);
		return hipGetLastError();
	}
}

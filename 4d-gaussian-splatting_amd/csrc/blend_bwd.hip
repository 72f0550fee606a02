// blend_bwd.hip -- back-to-front blend backward, one wave64 per 8x8 pixel block (gfx950).
//
// Semantics: reference renderCUDA backward (backward.cu:926-1137): per pixel,
// walk the tile list from the pixel's last contributor to the front, recompute
// G / alpha, unwind T, keep the running "colour behind" recurrences for
// RGB / flow / depth / mask, and produce per-(pixel, Gaussian) contributions to
// dL_dcolor(3), dL_dflow(2), dL_dmean2D(x, y, z = depth carrier), dL_dconic(xx, xy, yy
// with the reference's 1/2-of-xy convention, Q12), dL_dopacity.  The alpha clamp is
// ignored in the backward (Q13), as in the reference.
//
// What is different from the reference, which issues 12 global float atomics per
// contributing (pixel, Gaussian) pair (backward.cu:1076-1134):
//   * list entries are culled against the wave's 8x8 block and compacted
//     (blend_common.h), and the wave starts at ITS deepest last contributor;
//   * the 12 per-lane values are summed over the 64 pixels with a butterfly
//     "transpose-reduce" on DPP lane permutes: at every halving step a lane keeps one
//     half of its slots and hands the other half to its partner, so after 4 steps
//     each lane of a 16-lane row owns ONE fully row-reduced slot (24+12+12+3 VALU ops
//     instead of 12 x 6), two cross-row shuffles finish the sum, and lanes 0..15 then
//     issue ONE global_atomic_add_f32 instruction for the 12 live slots, all inside the Gaussian's
//     packed 64-byte accumulator record (one cache line = one memory-side atomic request);
//   => one atomic request per surviving (block, Gaussian) instead of 12 per (pixel, Gaussian).
//   preprocess_bwd unpacks the records into the user-visible gradient tensors.
#include "blend_common.h"

namespace fdgs
{
	constexpr int NG = 12; // gradient words per Gaussian: colour 3, flow 2, mean2D 3, conic 3, opacity 1

	template <int CTRL>
	__device__ __forceinline__ float dpp_mov(float v)
	{
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
	}

	// Sums 16 slots (v[12..15] are zero) over the wave.  On return lane L (any row) holds the total of
	// slot  s(L) = 8*b0 + 4*b1 + 2*b2 + b3  (b_i = bit i of L & 15).
	__device__ __forceinline__ float transpose_reduce16(const float (&v)[16], int lane)
	{
		const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
		float w[8], u[4], t[2];
#pragma unroll
		for (int i = 0; i < 8; i++)   // partner = lane ^ 1  (quad_perm [1,0,3,2])
		{
			const float keep = b0 ? v[i + 8] : v[i];
			const float send = b0 ? v[i] : v[i + 8];
			w[i] = keep + dpp_mov<0xB1>(send);
		}
#pragma unroll
		for (int i = 0; i < 4; i++)   // partner = lane ^ 2  (quad_perm [2,3,0,1])
		{
			const float keep = b1 ? w[i + 4] : w[i];
			const float send = b1 ? w[i] : w[i + 4];
			u[i] = keep + dpp_mov<0x4E>(send);
		}
#pragma unroll
		for (int i = 0; i < 2; i++)   // partner = lane ^ 4  (row_shr:4 for bit2 lanes, row_shl:4 otherwise)
		{
			const float keep = b2 ? u[i + 2] : u[i];
			const float send = b2 ? u[i] : u[i + 2];
			const float from_lo = dpp_mov<0x114>(send); // row_shr:4: lane i <- lane i-4
			const float from_hi = dpp_mov<0x104>(send); // row_shl:4: lane i <- lane i+4
			t[i] = keep + (b2 ? from_lo : from_hi);
		}
		float r;
		{                             // partner = lane ^ 8  (row_ror:8)
			const float keep = b3 ? t[1] : t[0];
			const float send = b3 ? t[0] : t[1];
			r = keep + dpp_mov<0x128>(send);
		}
		r += __shfl_xor(r, 16);       // across the four 16-lane rows
		r += __shfl_xor(r, 32);
		return r;
	}

	__global__ void __launch_bounds__(WAVE) blend_bwd_kernel(
		const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ records,
		int W, int H, int grid_x, int ntiles, const float* __restrict__ bg,
		const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
		const float* __restrict__ dL_dpixels, const float* __restrict__ dL_depths, const float* __restrict__ dL_masks,
		const float* __restrict__ dL_dpix_flow,
		float* __restrict__ gacc)
	{
		// wave-private queue of the surviving entries of the current chunk (+1: inert padding entry for the prefetch)
		__shared__ float4 s_a[WAVE + 1];
		__shared__ float4 s_b[WAVE + 1];
		__shared__ float4 s_c[WAVE + 1];
		__shared__ uint32_t s_pos[WAVE + 1];
		__shared__ uint32_t s_id[WAVE + 1];

		const BlockId blk = block_of(blockIdx.x, ntiles);
		if (blk.tile >= ntiles) return;
		const int lane = threadIdx.x;
		const int bx0 = (blk.tile % grid_x) * TILE_X + (blk.sub & 1) * BLK;
		const int by0 = (blk.tile / grid_x) * TILE_Y + (blk.sub >> 1) * BLK;
		if (bx0 >= W || by0 >= H) return;
		const int px = bx0 + (lane & (BLK - 1)), py = by0 + (lane >> 3);
		const bool inside = px < W && py < H;
		const float pixfx = (float)px, pixfy = (float)py;
		const float rx0 = (float)bx0, rx1 = (float)min(bx0 + BLK - 1, W - 1);
		const float ry0 = (float)by0, ry1 = (float)min(by0 + BLK - 1, H - 1);
		const size_t pix_id = (size_t)W * py + px, HW = (size_t)H * W;
		const uint2 range = ranges[blk.tile];
		const unsigned long long lt_mask = (1ull << lane) - 1ull;

		const float T_final = inside ? final_Ts[pix_id] : 0.f;
		const int last_contributor = inside ? (int)n_contrib[pix_id] : 0;
		int wave_last = last_contributor; // deepest list position (exclusive) any pixel of the block reached
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, o));
		if (wave_last == 0) return;

		float T = T_final;
		float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLf0 = 0.f, dLf1 = 0.f, dL_depth = 0.f, dL_mask = 0.f;
		if (inside)
		{
			dLp0 = dL_dpixels[0 * HW + pix_id]; dLp1 = dL_dpixels[1 * HW + pix_id]; dLp2 = dL_dpixels[2 * HW + pix_id];
			dLf0 = dL_dpix_flow[0 * HW + pix_id]; dLf1 = dL_dpix_flow[1 * HW + pix_id];
			dL_depth = dL_depths[pix_id];
			dL_mask = dL_masks[pix_id];
		}
		const float bg_dot_dpixel = bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2;
		const float ddelx_dx = 0.5f * W, ddely_dy = 0.5f * H; // backward.cu:1010-1011

		// The slot this lane owns after transpose_reduce16 (12 live slots in lanes 0..15) is word `slot` of the
		// Gaussian's packed 64-byte accumulator record: colour 0-2, flow 3-4, mean2D 5-7, conic xx/xy/yy 8-10,
		// opacity 11.  One record = one 64-B segment, so the 12-lane atomic instruction is ONE memory-side
		// request instead of five (the reference scatters into five arrays, backward.cu:1116-1133).
		const int slot = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
		float* const slot_ptr = gacc + slot;
		const bool slot_writer = lane < 16 && slot < NG;

		float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accf0 = 0.f, accf1 = 0.f, acc_depth = 0.f, acc_mask = 0.f;
		float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, lf0 = 0.f, lf1 = 0.f, last_depth = 0.f;

		for (int top = wave_last; top > 0; top -= WAVE)
		{
			// this chunk covers list positions top-1 downto max(0, top-64); lane 0 takes the deepest one
			const int pos = top - 1 - lane;
			bool keep = false;
			uint32_t id = 0;
			float4 a, b;
			if (pos >= 0)
			{
				id = point_list[range.x + (uint32_t)pos];
				a = records[3 * (size_t)id + 0];
				b = records[3 * (size_t)id + 1];
				keep = block_reaches(a, b, rx0, rx1, ry0, ry1);
			}
			const unsigned long long mask = __ballot(keep);
			const int cnt = __popcll(mask);
			if (keep)
			{
				const int q = __popcll(mask & lt_mask); // back-to-front order is preserved
				s_a[q] = a;
				s_b[q] = b;
				s_c[q] = records[3 * (size_t)id + 2];
				s_pos[q] = (uint32_t)pos;
				s_id[q] = id;
			}
			if (lane == 0)
			{
				// inert entry behind the queue so the prefetch of entry j+1 never reads stale data
				s_a[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
				s_b[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
				s_c[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
				s_pos[cnt] = 0x7fffffffu;
				s_id[cnt] = 0u;
			}
			__syncthreads();

			float4 na = s_a[0], nb = s_b[0], nc = s_c[0];
			uint32_t npos = s_pos[0], nid = s_id[0];
			for (int j = 0; j < cnt; j++)
			{
				// software pipeline: entry j is in registers, entry j+1 is fetched now and lands while j is processed
				const float4 ea = na, eb = nb, ec = nc;
				const uint32_t epos = npos, eid = nid;
				na = s_a[j + 1]; nb = s_b[j + 1]; nc = s_c[j + 1];
				npos = s_pos[j + 1]; nid = s_id[j + 1];

				float g[16];
#pragma unroll
				for (int k = 0; k < 16; k++) g[k] = 0.f;
				const float dx = ea.x - pixfx, dy = ea.y - pixfy;
				const float power = -0.5f * (ea.z * dx * dx + eb.x * dy * dy) - ea.w * dx * dy;
				const float G = fast_exp(power);
				const float alpha = fminf(0.99f, eb.y * G);
				// one predicate instead of the reference's three nested tests (backward.cu:1040-1054)
				const bool active = ((int)epos < last_contributor) && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
				if (active)
				{
					const float inv = __builtin_amdgcn_rcpf(1.f - alpha);
					T = T * inv;
					const float dchannel_dcolor = alpha * T;
					float dL_dalpha = 0.0f;
					const float one_m_la = 1.f - last_alpha;
					acc0 = last_alpha * lc0 + one_m_la * acc0; lc0 = eb.z; dL_dalpha += (eb.z - acc0) * dLp0;
					acc1 = last_alpha * lc1 + one_m_la * acc1; lc1 = eb.w; dL_dalpha += (eb.w - acc1) * dLp1;
					acc2 = last_alpha * lc2 + one_m_la * acc2; lc2 = ec.x; dL_dalpha += (ec.x - acc2) * dLp2;
					accf0 = last_alpha * lf0 + one_m_la * accf0; lf0 = ec.z; dL_dalpha += (ec.z - accf0) * dLf0;
					accf1 = last_alpha * lf1 + one_m_la * accf1; lf1 = ec.w; dL_dalpha += (ec.w - accf1) * dLf1;
					acc_depth = last_alpha * last_depth + one_m_la * acc_depth; last_depth = ec.y;
					dL_dalpha += (ec.y - acc_depth) * dL_depth;
					acc_mask = last_alpha + one_m_la * acc_mask;
					dL_dalpha += (1.0f - acc_mask) * dL_mask;
					dL_dalpha *= T;
					last_alpha = alpha;
					dL_dalpha += (-T_final * inv) * bg_dot_dpixel;

					const float dL_dG = eb.y * dL_dalpha;
					const float gdx = G * dx, gdy = G * dy;
					const float dG_ddelx = -gdx * ea.z - gdy * ea.w;
					const float dG_ddely = -gdy * eb.x - gdx * ea.w;
					g[0] = dchannel_dcolor * dLp0;
					g[1] = dchannel_dcolor * dLp1;
					g[2] = dchannel_dcolor * dLp2;
					g[3] = dchannel_dcolor * dLf0;
					g[4] = dchannel_dcolor * dLf1;
					g[5] = dL_dG * dG_ddelx * ddelx_dx;
					g[6] = dL_dG * dG_ddely * ddely_dy;
					g[7] = dL_depth * dchannel_dcolor;
					g[8] = -0.5f * gdx * dx * dL_dG;
					g[9] = -0.5f * gdx * dy * dL_dG;
					g[10] = -0.5f * gdy * dy * dL_dG;
					g[11] = G * dL_dalpha;
				}
				if (__ballot(active) != 0ull)
				{
					const float total = transpose_reduce16(g, lane);
					if (slot_writer) atomicAdd(slot_ptr + (size_t)eid * GRAD_ACC_WORDS, total);
				}
			}
			__syncthreads();
		}
	}

	hipError_t launch_blend_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                            const float* records, const uint32_t* point_list, const uint32_t* ranges,
	                            const float* final_T, const uint32_t* n_contrib, hipStream_t stream)
	{
		const int gx = div_up(s.W, TILE_X), gy = div_up(s.H, TILE_Y);
		const int ntiles = gx * gy;
		hipLaunchKernelGGL(blend_bwd_kernel, dim3(blend_grid(ntiles)), dim3(WAVE), 0, stream,
		                   reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(records),
		                   s.W, s.H, gx, ntiles, s.bg, final_T, n_contrib,
		                   in.dL_dout_color, in.dL_dout_depth, in.dL_dout_alpha, in.dL_dout_flow,
		                   out.grad_accum);
		return hipGetLastError();
	}
}

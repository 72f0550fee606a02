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
//   * the kernel is VALU-issue bound (tools/probe/README.md), so the per-entry body is written for
//     instruction count: branch-free after one wave-uniform skip, the seven per-channel recurrences
//     collapsed into one scalar recurrence, packed-fp32 (v_pk_*) arithmetic on natural pairs;
//   * the 12 per-lane values are summed over the 64 pixels with a butterfly
//     "transpose-reduce" on DPP lane permutes: at every halving step a lane keeps one
//     half of its slots and hands the other half to its partner, so after 4 steps
//     each lane of a 16-lane row owns ONE fully row-reduced slot (29 VALU ops
//     instead of 12 x 6), two cross-row shuffles finish the sum, and lanes 0..11 then
//     issue ONE global_atomic_add_f32 instruction for 12 consecutive words of the Gaussian's
//     packed 64-byte accumulator record (one memory-side atomic request);
//   => one atomic request per surviving (block, Gaussian) instead of 12 per (pixel, Gaussian).
//   preprocess_bwd unpacks the records into the user-visible gradient tensors.
#include "blend_common.h"

namespace fdgs
{
	constexpr int NG = 12; // gradient words per Gaussian: colour 3, depth 1, flow 2, mean2D 2, conic 3, opacity 1

	// The same butterfly, hand-scheduled, with the partner order reversed (lane^8, ^4, ^2, ^1) so that the two
	// widest steps can use DPP bank masks (a bank = 4 consecutive lanes of a row, i.e. lane bits 2-3): the
	// lanes of one class are written by one v_add_f32_dpp and the other class by a second one -- no
	// v_cndmask selects -- and every step works in place.  Slots 12..15 carry junk (nobody reads them).
	// 29 VALU instead of 51; the explicit s_nops cover the VALU-write -> DPP-read hazard (2 wait states)
	// that the compiler cannot see inside inline asm.
	// On return lane L of every 16-lane row holds its ROW's sum of slot (L & 15).
	__device__ __forceinline__ float row_transpose_reduce12(float (&g)[12])
	{
		const unsigned long long m1 = 0xCCCCCCCCCCCCCCCCull, m0 = 0xAAAAAAAAAAAAAAAAull;
		asm volatile(
			"s_nop 1\n\t"
			// step 1: partner = lane ^ 8; lanes 0-7 of a row keep slots i, lanes 8-15 keep slots i + 8
			"v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
			// step 2: partner = lane ^ 4; banks 0,2 (bit 2 clear) read lane+4 and keep slots i, banks 1,3 read lane-4, keep i + 4
			"v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %3, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			// step 3: partner = lane ^ 2
			"v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_cndmask_b32_e64 %0, %0, %2, %12\n\t"
			"v_cndmask_b32_e64 %1, %1, %3, %12\n\t"
			"s_nop 1\n\t"
			// step 4: partner = lane ^ 1
			"v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_cndmask_b32_e64 %0, %0, %1, %13\n\t"
			: "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])
			: "v"(g[8]), "v"(g[9]), "v"(g[10]), "v"(g[11]), "s"(m1), "s"(m0));
		return g[0];
	}

	// The same reduction when only the colour image has an upstream gradient (AUX = false below): slots 3-5
	// (depth, flow) are identically zero, so their instructions are dropped (24 VALU); lanes 3-5 end up with
	// junk that nobody writes.  g[3..5] are not read.
	__device__ __forceinline__ float row_transpose_reduce9(float (&g)[12])
	{
		const unsigned long long m1 = 0xCCCCCCCCCCCCCCCCull, m0 = 0xAAAAAAAAAAAAAAAAull;
		float w3;
		asm volatile(
			"s_nop 1\n\t"
			"v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %2, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %3, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %2, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %3, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_cndmask_b32_e64 %0, %0, %2, %10\n\t"
			"v_cndmask_b32_e64 %1, %1, %3, %10\n\t"
			"s_nop 1\n\t"
			"v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_cndmask_b32_e64 %0, %0, %1, %11\n\t"
			: "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "=&v"(w3), "+v"(g[6]), "+v"(g[7])
			: "v"(g[8]), "v"(g[9]), "v"(g[10]), "v"(g[11]), "s"(m1), "s"(m0));
		return g[0];
	}

	// slot_off: this lane's word of the record, in bytes.  The address is the uniform base plus a 32-bit byte offset
	// (a record is 64 bytes: P < 2^26, checked by the launcher), which the atomic takes as SGPR base + VGPR offset -- one
	// shift-add per entry instead of 64-bit pointer arithmetic.
	template <bool AUX>
	__device__ __forceinline__ void reduce_and_add(float (&g)[12], float* gacc, uint32_t slot_off, bool slot_writer, uint32_t eid)
	{
		float total = AUX ? row_transpose_reduce12(g) : row_transpose_reduce9(g);
		total += __shfl_xor(total, 16);       // across the four 16-lane rows
		total += __shfl_xor(total, 32);
		if (slot_writer) atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(gacc) + (eid * (uint32_t)(GRAD_ACC_WORDS * 4) + slot_off)), total);
	}

	// AUX = false: only the colour image carries an upstream gradient (dL_dout_depth / _alpha / _flow are NULL = zero),
	// the usual case in training (photometric loss on the render only): the depth / flow / mask terms drop out.
	typedef float v2f __attribute__((ext_vector_type(2)));

	template <bool AUX>
	__global__ void __launch_bounds__(WAVE) blend_bwd_kernel(
		const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ records,
		const uint32_t* __restrict__ tile_order, int W, int H, int grid_x, int ntiles, const float* __restrict__ bg,
		const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
		const float* __restrict__ dL_dpixels, const float* __restrict__ dL_depths, const float* __restrict__ dL_masks,
		const float* __restrict__ dL_dpix_flow,
		float* __restrict__ gacc)
	{
		// Wave-private queue of the surviving entries of the current chunk, stored as PAIRS of entries interleaved word by
		// word, as in blend_fwd.hip: (x0,x1,y0,y1) (A0,A1,B0,B1) (C0,C1,o0,o1) (r0,r1,g0,g1) (b0,b1,d0,d1) (fx0,fx1,fy0,fy1):
		// the arithmetic up to alpha runs as packed fp32 on two entries per instruction.
		constexpr int QP = WAVE / 2 + 1;
		// row 6: (list position of entry 0, of entry 1, Gaussian id of entry 0, of entry 1) -- same stride as the other rows, so the
		// loop addresses the whole queue with ONE base register and immediate offsets
		__shared__ float4 s_q[7][QP];

		const BlockId blk = block_of(blockIdx.x, ntiles, tile_order);
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
			if (dL_dpixels) { dLp0 = dL_dpixels[0 * HW + pix_id]; dLp1 = dL_dpixels[1 * HW + pix_id]; dLp2 = dL_dpixels[2 * HW + pix_id]; }
			if (AUX && dL_dpix_flow) { dLf0 = dL_dpix_flow[0 * HW + pix_id]; dLf1 = dL_dpix_flow[1 * HW + pix_id]; }
			if (AUX && dL_depths) dL_depth = dL_depths[pix_id];
			if (AUX && dL_masks) dL_mask = dL_masks[pix_id];
		}
		const float nTf_bg = -T_final * (bg[0] * dLp0 + bg[1] * dLp1 + bg[2] * dLp2);

		// After row_transpose_reduce12 lane L holds slot (L & 15) = word (L & 15) of the Gaussian's packed 64-byte
		// accumulator record: colour r,g,b 0-2, depth 3, flow 4-5, then the moments of q = G dL/dalpha: q dx, q dy 6-7,
		// q dx^2, q dy^2 8-9, q dx dy 10, q 11 (preprocess_bwd turns them into dL/dmean2D, dL/dconic, dL/dopacity).  One record = one 64-B segment, so the 12-lane atomic instruction is ONE memory-side
		// request instead of five (the reference scatters into five arrays, backward.cu:1116-1133).
		const uint32_t slot_off = (uint32_t)(lane & 15) * 4u;
		const bool slot_writer = lane < NG && (AUX || lane < 3 || lane > 5);

		float S = 0.f, Lc = 0.f, last_alpha = 0.f;

		// One queue entry against this lane's pixel, after the packed head: d = mean2D - pixel, power, G = exp(power), alpha;
		// colour (cr, cg, cb), depth, flow of the entry; active = this pixel takes part; eid = Gaussian id.
		auto entry = [&](const float dx, const float dy, const float power, const float G, const float alpha, const float cr, const float cg,
		                 const float cb, const float cdepth, const float cfx, const float cfy, const lanemask active, const uint32_t eid) __attribute__((always_inline))
		{
			// Branch-free: a lane that skips this entry runs the same instructions with
			// alpha = G = 0, which leaves T and the recurrence unchanged and makes all 12 products zero.
			const float alpha_e = mask_select(active, alpha, 0.0f);
			const float G_e = mask_select(active, G, 0.0f);
			const float inv = __builtin_amdgcn_rcpf(1.f - alpha_e);
			T = T * inv;
			const float dchannel_dcolor = alpha_e * T;
			// The reference keeps seven "colour behind me" recurrences  acc_k = la * last_k + (1 - la) * acc_k
			// (rgb, flow, depth, mask; backward.cu:1063-1096) and sums (c_k - acc_k) * dL_k.  All seven share
			// the coefficients and dL_k is a per-pixel constant, so they collapse into ONE scalar recurrence on
			// S = sum_k acc_k dL_k with  last = sum_k c_k dL_k  (c_mask = 1):  same value up to fp32 rounding.
			float Cd;
			if constexpr (AUX) Cd = fmaf(cr, dLp0, fmaf(cg, dLp1, fmaf(cb, dLp2, fmaf(cfx, dLf0, fmaf(cfy, dLf1, fmaf(cdepth, dL_depth, dL_mask))))));
			else Cd = fmaf(cr, dLp0, fmaf(cg, dLp1, cb * dLp2));
			S = fmaf(last_alpha, Lc - S, S);
			Lc = Cd;
			last_alpha = alpha_e;
			const float dL_dalpha = fmaf(Cd - S, T, inv * nTf_bg);

			float g[12];
			g[0] = dchannel_dcolor * dLp0;
			g[1] = dchannel_dcolor * dLp1;
			g[2] = dchannel_dcolor * dLp2;
			if constexpr (AUX)
			{
				g[3] = dL_depth * dchannel_dcolor;
				g[4] = dchannel_dcolor * dLf0;
				g[5] = dchannel_dcolor * dLf1;
			}
			// Geometry slots: what has to be summed over the pixels are the MOMENTS of q = G dL/dalpha in the offset d from
			// the Gaussian's centre -- q, q dx, q dy, q dx^2, q dy^2, q dx dy.  dL/dmean2D, dL/dconic and dL/dopacity are
			// linear in them with per-Gaussian coefficients (conic, opacity, W/2, H/2: backward.cu:1107-1133), so that
			// conversion runs once per Gaussian in preprocess_bwd instead of once per (pixel, Gaussian) here: 6 VALU
			// instructions for these six slots instead of 16.
			const float q = G_e * dL_dalpha;
			const float qx = q * dx, qy = q * dy;
			g[6] = qx;
			g[7] = qy;
			g[8] = qx * dx;
			g[9] = qy * dy;
			g[10] = qx * dy;
			g[11] = q;
			reduce_and_add<AUX>(g, gacc, slot_off, slot_writer, eid);
		};

		// Both entries of a pair have contributing pixels (the common case): the same arithmetic as `entry` twice, but
		// everything that is not part of the sequential T / S recurrences runs packed on the two entries.
		auto entry_pair = [&](const v2f dx, const v2f dy, const v2f G, const float alpha0, const float alpha1, const v2f cr, const v2f cg,
		                      const v2f cb, const v2f cdepth, const v2f cfx, const v2f cfy, const lanemask act0, const lanemask act1,
		                      const uint32_t eid0, const uint32_t eid1) __attribute__((always_inline))
		{
			const v2f alpha_e = { mask_select(act0, alpha0, 0.0f), mask_select(act1, alpha1, 0.0f) };
			const v2f G_e = { mask_select(act0, G.x, 0.0f), mask_select(act1, G.y, 0.0f) };
			const v2f om = 1.0f - alpha_e;
			const v2f inv = { __builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y) };
			const float T0 = T * inv.x, T1 = T0 * inv.y;
			T = T1;
			const v2f Tp = { T0, T1 };
			const v2f dcd = alpha_e * Tp;
			v2f Cd;
			if constexpr (AUX) Cd = cr * dLp0 + (cg * dLp1 + (cb * dLp2 + (cfx * dLf0 + (cfy * dLf1 + (cdepth * dL_depth + dL_mask)))));
			else Cd = __builtin_elementwise_fma(cr, v2f{ dLp0, dLp0 }, __builtin_elementwise_fma(cg, v2f{ dLp1, dLp1 }, cb * dLp2));
			const float S0 = fmaf(last_alpha, Lc - S, S);
			const float S1 = fmaf(alpha_e.x, Cd.x - S0, S0);
			S = S1; Lc = Cd.y; last_alpha = alpha_e.y;
			const v2f Sp = { S0, S1 };
			const v2f dL_dalpha = __builtin_elementwise_fma(Cd - Sp, Tp, inv * nTf_bg);
			const v2f g0 = dcd * dLp0, g1 = dcd * dLp1, g2 = dcd * dLp2;
			const v2f q = G_e * dL_dalpha;
			const v2f qx = q * dx, qy = q * dy;
			const v2f qxx = qx * dx, qyy = qy * dy, qxy = qx * dy;
			float ga[12], gb[12];
			ga[0] = g0.x; ga[1] = g1.x; ga[2] = g2.x; gb[0] = g0.y; gb[1] = g1.y; gb[2] = g2.y;
			if constexpr (AUX)
			{
				const v2f g3 = dcd * dL_depth, g4 = dcd * dLf0, g5 = dcd * dLf1;
				ga[3] = g3.x; ga[4] = g4.x; ga[5] = g5.x; gb[3] = g3.y; gb[4] = g4.y; gb[5] = g5.y;
			}
			ga[6] = qx.x; ga[7] = qy.x; ga[8] = qxx.x; ga[9] = qyy.x; ga[10] = qxy.x; ga[11] = q.x;
			gb[6] = qx.y; gb[7] = qy.y; gb[8] = qxx.y; gb[9] = qyy.y; gb[10] = qxy.y; gb[11] = q.y;
			reduce_and_add<AUX>(ga, gacc, slot_off, slot_writer, eid0);
			reduce_and_add<AUX>(gb, gacc, slot_off, slot_writer, eid1);
		};

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
				a = record_word(records, id, 0);
				b = record_word(records, id, 1);
				keep = block_reaches(a, b, rx0, rx1, ry0, ry1);
			}
			const unsigned long long mask = __ballot(keep);
			const int cnt = __popcll(mask);
			if (keep)
			{
				const int slot = __popcll(mask & lt_mask); // back-to-front order is preserved
				const float4 c = record_word(records, id, 2);
				const int pr = slot >> 1, h = slot & 1;
				float* q0 = reinterpret_cast<float*>(&s_q[0][pr]) + h;
				float* q1 = reinterpret_cast<float*>(&s_q[1][pr]) + h;
				float* q2 = reinterpret_cast<float*>(&s_q[2][pr]) + h;
				float* q3 = reinterpret_cast<float*>(&s_q[3][pr]) + h;
				float* q4 = reinterpret_cast<float*>(&s_q[4][pr]) + h;
				float* q5 = reinterpret_cast<float*>(&s_q[5][pr]) + h;
				q0[0] = a.x; q0[2] = a.y; q1[0] = a.z; q1[2] = a.w; q2[0] = b.x; q2[2] = b.y;
				q3[0] = b.z; q3[2] = b.w; q4[0] = c.x; q4[2] = c.y; q5[0] = c.z; q5[2] = c.w;
				uint32_t* e = reinterpret_cast<uint32_t*>(&s_q[6][pr]) + h;
				e[0] = (uint32_t)pos; e[2] = id;
			}
			if (lane == 0 && (cnt & 1))
			{
				// inert second half of the last pair: list position beyond every last contributor -> never active
				const int pr = cnt >> 1;
#pragma unroll
				for (int k = 0; k < 6; k++) { float* q = reinterpret_cast<float*>(&s_q[k][pr]) + 1; q[0] = 0.f; q[2] = 0.f; }
				uint32_t* e = reinterpret_cast<uint32_t*>(&s_q[6][pr]) + 1;
				e[0] = 0x7fffffffu; e[2] = 0u;
			}
			__syncthreads();

			const int npairs = (cnt + 1) >> 1;
			for (int i = 0; i < npairs; i++)
			{
				const float4 Q0 = s_q[0][i], Q1 = s_q[1][i], Q2 = s_q[2][i], Q3 = s_q[3][i], Q4 = s_q[4][i], Q5 = s_q[5][i];
				const uint4 pi = *reinterpret_cast<const uint4*>(&s_q[6][i]);
				const uint2 pp = make_uint2(pi.x, pi.y), ii = make_uint2(pi.z, pi.w);
				// packed head for both entries, with the reference's association per element (forward.cu:585,
				// backward.cu:1036): keeps alpha -- and with it the alpha >= 1/255 decision -- within an ulp of the oracle's
				// (a cheaper factored form flipped cliff pairs and was dropped)
				const v2f dx = v2f{ Q0.x, Q0.y } - pixfx, dy = v2f{ Q0.z, Q0.w } - pixfy;
				const v2f cA = { Q1.x, Q1.y }, cB = { Q1.z, Q1.w }, cC = { Q2.x, Q2.y }, op = { Q2.z, Q2.w };
				const v2f s2 = __builtin_elementwise_fma(cC * dy, dy, (cA * dx) * dx);
				const v2f power = __builtin_elementwise_fma(v2f{ -0.5f, -0.5f }, s2, -((cB * dx) * dy));
				const v2f e2 = power * 1.4426950408889634f;
				const v2f G = { __builtin_amdgcn_exp2f(e2.x), __builtin_amdgcn_exp2f(e2.y) };
				const v2f al = op * G;
				const float alpha0 = fminf(0.99f, al.x), alpha1 = fminf(0.99f, al.y);
				// one predicate instead of the reference's three nested tests (backward.cu:1040-1054), as a lane mask
				const lanemask act0 = mask_of((int)pp.x < last_contributor) & mask_of(!(power.x > 0.0f)) & mask_of(!(alpha0 < 1.0f / 255.0f));
				const lanemask act1 = mask_of((int)pp.y < last_contributor) & mask_of(!(power.y > 0.0f)) & mask_of(!(alpha1 < 1.0f / 255.0f));
				const bool any0 = act0 != 0ull, any1 = act1 != 0ull;
				if (any0 && any1)
					entry_pair(dx, dy, G, alpha0, alpha1, v2f{ Q3.x, Q3.y }, v2f{ Q3.z, Q3.w }, v2f{ Q4.x, Q4.y }, v2f{ Q4.z, Q4.w },
					           v2f{ Q5.x, Q5.y }, v2f{ Q5.z, Q5.w }, act0, act1, ii.x, ii.y);
				else if (any0) entry(dx.x, dy.x, power.x, G.x, alpha0, Q3.x, Q3.z, Q4.x, Q4.z, Q5.x, Q5.z, act0, ii.x);
				else if (any1) entry(dx.y, dy.y, power.y, G.y, alpha1, Q3.y, Q3.w, Q4.y, Q4.w, Q5.y, Q5.w, act1, ii.y);
			}
			__syncthreads();
		}
	}

	hipError_t launch_blend_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                            const float* records, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* tile_order,
	                            const float* final_T, const uint32_t* n_contrib, hipStream_t stream)
	{
		const int gx = div_up(s.W, TILE_X), gy = div_up(s.H, TILE_Y);
		const int ntiles = gx * gy;
#define LAUNCH_BWD(AUX) hipLaunchKernelGGL((blend_bwd_kernel<AUX>), dim3(blend_grid(ntiles)), dim3(WAVE), 0, stream, \
		                   reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(records), \
		                   tile_order, s.W, s.H, gx, ntiles, s.bg, final_T, n_contrib, \
		                   in.dL_dout_color, in.dL_dout_depth, in.dL_dout_alpha, in.dL_dout_flow, out.grad_accum)
		if (s.P >= (1 << 26)) return hipErrorInvalidValue;   // 32-bit byte offsets into the 64-byte accumulator records
		if (in.dL_dout_depth || in.dL_dout_alpha || in.dL_dout_flow) LAUNCH_BWD(true);
		else LAUNCH_BWD(false);
#undef LAUNCH_BWD
		return hipGetLastError();
	}
}

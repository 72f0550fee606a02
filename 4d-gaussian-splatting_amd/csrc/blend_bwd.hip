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

	// Joint row reduction of the NS = 9 values of TWO entries (generated and checked by tools/gen_reduce.py: every instruction
	// simulated on symbolic lane values).  a[] = entry 0, b[] = entry 1.  Step 1 (lane ^ 8) sends entry 0 to lanes 0-7 and entry 1
	// to lanes 8-15 of every row, steps 2-4 are the halving butterfly inside the half-row.  On return lane h = lane & 7 of a half-row
	// holds, summed over the 16 lanes of its row,
	//   A0: slots [0, 2, 3, None, 5, 7, 8, None]
	//   A1: slots [1, None, 4, None, 6, None, None, None]
	// (None: a duplicate or partial that must not be used).
	__device__ __forceinline__ void pair_row_reduce9(float (&a)[12], const float (&b)[12])
	{
		const unsigned long long m1 = 0xCCCCCCCCCCCCCCCCull, m0 = 0xAAAAAAAAAAAAAAAAull;
		asm volatile(
			"s_nop 1\n\t"
			"v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %0, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %1, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %2, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %3, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %4, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %5, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %6, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %7, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %8, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %0, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %1, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %2, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %3, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_cndmask_b32_e64 %0, %0, %3, %18\n\t"
			"v_cndmask_b32_e64 %1, %1, %4, %18\n\t"
			"v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_cndmask_b32_e64 %0, %0, %2, %19"
			// Operand constraints: b[] is read after a[] has been written.  a[k] are tied in/out operands -- each holds a live input on
			// entry, so the allocator can only give a[j] and b[k] one register if they are the SAME value; the callers pass the .x and
			// .y halves of packed products of two different list entries (entry_pair), never one value twice.  The formally safe forms
			// were built and cost the both-alive path 8 ("+&v" on a[]: the early-clobber is applied to the whole 64-bit pair register
			// the packed products live in) or 6 (b[] as in/out operands) v_mov per pair: 272 -> 282 us per launch at C3.
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8])
			: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "s"(m1), "s"(m0));
	}

	// Joint row reduction of the NS = 12 values of TWO entries (generated and checked by tools/gen_reduce.py: every instruction
	// simulated on symbolic lane values).  a[] = entry 0, b[] = entry 1.  Step 1 (lane ^ 8) sends entry 0 to lanes 0-7 and entry 1
	// to lanes 8-15 of every row, steps 2-4 are the halving butterfly inside the half-row.  On return lane h = lane & 7 of a half-row
	// holds, summed over the 16 lanes of its row,
	//   A0: slots [0, 2, 3, 5, 6, 8, 9, 11]
	//   A1: slots [1, None, 4, None, 7, None, 10, None]
	// (None: a duplicate or partial that must not be used).
	__device__ __forceinline__ void pair_row_reduce12(float (&a)[12], const float (&b)[12])
	{
		const unsigned long long m1 = 0xCCCCCCCCCCCCCCCCull, m0 = 0xAAAAAAAAAAAAAAAAull;
		asm volatile(
			"s_nop 1\n\t"
			"v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %10, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %11, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
			"v_add_f32_dpp %0, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %1, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %2, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %3, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %4, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %5, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %6, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %7, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %8, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %9, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %10, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %11, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
			"v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %0, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %1, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %2, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %3, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %4, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %5, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
			"v_add_f32_dpp %5, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
			"v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
			"v_cndmask_b32_e64 %0, %0, %3, %24\n\t"
			"v_cndmask_b32_e64 %1, %1, %4, %24\n\t"
			"v_cndmask_b32_e64 %2, %2, %5, %24\n\t"
			"v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
			"v_cndmask_b32_e64 %0, %0, %2, %25"
			// (operand constraints: see pair_row_reduce9)
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
			: "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]), "v"(b[10]), "v"(b[11]), "s"(m1), "s"(m0));
	}

	// Per-wave LDS staging of the row sums: ACC_GROUP list entries x 4 rows x 16 positions.  What a lane holds after the DPP
	// reduction (the sum over the 16 pixels of its row) is STORED there -- one ds_write per pair of entries -- and every
	// ACC_GROUP / 2 queue pairs the staging area is drained: 16 lanes per entry add up the four rows of one position each and the
	// lanes that hold a record word issue ONE global_atomic_add_f32 instruction per four entries (still one memory-side request
	// per entry: a record is one 64-byte segment).  This replaces two dependent ds_bpermute round trips, two adds and an atomic per
	// entry.  (LDS float atomics instead of the plain stores -- the four rows adding into one word -- were measured: ds_add_f32 /
	// ds_wrxchg retire a few lanes per cycle, the kernel took 0.98 ms instead of 0.29.)
	// Position of gradient slot s of an entry: AUX (s = record word 0..11): (s / 3) * 4 + s % 3; colour-only (slots = words
	// 0,1,2,6..11): 0,1,2,4,5,8,9,10,12 -- where the joint reduction above leaves them (lane h of a half-row: positions 2 h and
	// 2 h + 1); the other positions hold duplicates nobody reads.
	constexpr int ACC_GROUP = 4;

	// AUX = false: only the colour image carries an upstream gradient (dL_dout_depth / _alpha / _flow are NULL = zero),
	// the usual case in training (photometric loss on the render only): the depth / flow / mask terms drop out.
	typedef float v2f __attribute__((ext_vector_type(2)));

	template <bool AUX>
	__device__ __forceinline__ void blend_bwd_body(
		const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ records,
		const uint32_t* __restrict__ tile_order, int W, int H, int grid_x, int ntiles, const float* __restrict__ bg,
		const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib,
		const float* __restrict__ dL_dpixels, const float* __restrict__ dL_depths, const float* __restrict__ dL_masks,
		const float* __restrict__ dL_dpix_flow,
		float* __restrict__ gacc, const uint32_t* __restrict__ ctl)
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
		__shared__ __attribute__((aligned(16))) float s_acc[ACC_GROUP * 64];   // [entry][row][position]
		char* const acc_b = reinterpret_cast<char*>(s_acc);
		// pair path: lanes 0-7 of a row hold entry 0, lanes 8-15 entry 1; lane h of the half-row: positions 2 h and 2 h + 1
		const uint32_t off_pair = (uint32_t)((lane >> 3) & 1) * 256u + (uint32_t)(lane >> 4) * 64u + (uint32_t)(lane & 7) * 8u;
		// single-entry path: after row_transpose_reduce* lane c = lane & 15 of every row holds record word c
		const int wl = lane & 15;
		int pos_of_word, word_of_pos;
		if (AUX)
		{
			pos_of_word = wl < NG ? (wl / 3) * 4 + wl % 3 : 3;
			word_of_pos = (wl & 3) == 3 ? -1 : (wl >> 2) * 3 + (wl & 3);
		}
		else
		{
			pos_of_word = wl < 3 ? wl : (wl == 6 ? 4 : wl == 7 ? 5 : (wl >= 8 && wl <= 10) ? wl : wl == 11 ? 12 : 3);
			word_of_pos = wl < 3 ? wl : (wl == 4 ? 6 : wl == 5 ? 7 : (wl >= 8 && wl <= 10) ? wl : wl == 12 ? 11 : -1);
		}
		const uint32_t off_single = (uint32_t)(lane >> 4) * 64u + (uint32_t)pos_of_word * 4u;
		// drain: this lane takes position lane & 15 of entry (lane >> 4) + 4 it; woff = byte offset of the record word behind the
		// position (negative: a position nobody reads)
		const int woff = word_of_pos < 0 ? -1 : word_of_pos * 4;
		const uint32_t lane_entry_bit = 1u << (lane >> 4);
		const float* const acc_lane = s_acc + (lane >> 4) * 64 + wl;   // row 0 of this lane's (entry, position); rows: + 16 floats each
		// the Gaussian id of entry e = 4 it + (lane >> 4) of the group that starts at queue pair gp0: word 2 + (e & 1) of s_q[6][gp0 + e / 2]
		const uint32_t* const id_lane = reinterpret_cast<const uint32_t*>(&s_q[6][lane >> 5]) + 2 + ((lane >> 4) & 1);
		uint32_t grp_mask = 0u;   // entries of the current group that were staged (uniform)
		auto drain = [&](const int gp0, const uint32_t mask) __attribute__((always_inline))
		{
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // compiler ordering only: the LDS executes a wave's operations in order
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int it = 0; it < ACC_GROUP / 4; it++)
			{
				if ((mask >> (4 * it)) & 0xFu)
				{
					const float* r = acc_lane + it * 256;
					const float v = (r[0] + r[16]) + (r[32] + r[48]);
					const uint32_t eid = id_lane[(gp0 + 2 * it) * 4];
					if (woff >= 0 && (mask & (lane_entry_bit << (4 * it))))
						atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(gacc) + (eid * (uint32_t)(GRAD_ACC_WORDS * 4) + (uint32_t)woff)), v);
				}
			}
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		};

		float S = 0.f, Lc = 0.f, last_alpha = 0.f;

		// One queue entry against this lane's pixel, after the packed head: d = mean2D - pixel, power, G = exp(power), alpha;
		// colour (cr, cg, cb), depth, flow of the entry; active = this pixel takes part; eid = Gaussian id.
		auto entry = [&](const float dx, const float dy, const float power, const float G, const float alpha, const float cr, const float cg,
		                 const float cb, const float cdepth, const float cfx, const float cfy, const lanemask active, const int e_in_group) __attribute__((always_inline))
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
			const float total = AUX ? row_transpose_reduce12(g) : row_transpose_reduce9(g);
			*reinterpret_cast<float*>(acc_b + (off_single + (uint32_t)e_in_group * 256u)) = total;
		};

		// Both entries of a pair have contributing pixels (the common case): the same arithmetic as `entry` twice, but
		// everything that is not part of the sequential T / S recurrences runs packed on the two entries.
		auto entry_pair = [&](const v2f dx, const v2f dy, const v2f G, const float alpha0, const float alpha1, const v2f cr, const v2f cg,
		                      const v2f cb, const v2f cdepth, const v2f cfx, const v2f cfy, const lanemask act0, const lanemask act1,
		                      const int pair_in_group) __attribute__((always_inline))
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
				ga[6] = qx.x; ga[7] = qy.x; ga[8] = qxx.x; ga[9] = qyy.x; ga[10] = qxy.x; ga[11] = q.x;
				gb[6] = qx.y; gb[7] = qy.y; gb[8] = qxx.y; gb[9] = qyy.y; gb[10] = qxy.y; gb[11] = q.y;
				pair_row_reduce12(ga, gb);
			}
			else
			{
				// colour-only: nine slots, the moments follow the colours directly
				ga[3] = qx.x; ga[4] = qy.x; ga[5] = qxx.x; ga[6] = qyy.x; ga[7] = qxy.x; ga[8] = q.x;
				gb[3] = qx.y; gb[4] = qy.y; gb[5] = qxx.y; gb[6] = qyy.y; gb[7] = qxy.y; gb[8] = q.y;
				pair_row_reduce9(ga, gb);
			}
			*reinterpret_cast<float2*>(acc_b + (off_pair + (uint32_t)pair_in_group * 512u)) = make_float2(ga[0], ga[1]);
		};

		// The forward's per-block cull is not run again: it left one bit per (list entry, block) in the binning buffer (BinLayout::
		// cull_bits; stride and offset in ctl[2], ctl[3]), one 64-bit word per 64-entry chunk of the list -- the chunks taken here are
		// the forward's (positions 64 c .. 64 c + 63), from the one that holds the block's deepest contributor downwards.  Every chunk
		// up to that one was culled by the forward before it could stop.  (The test itself was 64 VALU instructions per chunk and two
		// record words per entry, surviving or not.)  ctl[2] == 0: a forward without planes (P = 0).
		const uint32_t cull_stride = ctl ? ctl[2] : 0u;
		const unsigned long long* cull_words = nullptr;
		if (cull_stride != 0u)
			cull_words = reinterpret_cast<const unsigned long long*>(point_list) + ctl[3] + (size_t)blk.sub * cull_stride + (range.x >> 6) + (uint32_t)blk.tile;
		for (int chunk = (wave_last - 1) >> 6; chunk >= 0; chunk--)
		{
			// positions 64 chunk + 63 downto 64 chunk; lane 0 takes the deepest one
			const int pos = chunk * WAVE + (WAVE - 1) - lane;
			// (no planes: every entry is taken -- the cull only saves work, an entry it would drop contributes nothing)
			// the ids of the whole chunk travel together with the plane word (one coalesced load, no dependency between the two); only
			// the survivors' records are gathered
			const uint32_t id = pos < wave_last ? point_list[range.x + (uint32_t)pos] : 0u;
			const unsigned long long fw = cull_words != nullptr ? cull_words[chunk] : ~0ull;
			const bool keep = pos < wave_last && ((fw >> (WAVE - 1 - lane)) & 1ull) != 0ull;
			float4 a, b;
			if (keep)
			{
				a = record_word(records, id, 0);
				b = record_word(records, id, 1);
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
				const uint2 pp = make_uint2(pi.x, pi.y);   // (.z, .w: the Gaussian ids, read by the drain)
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
				const int ip = i & (ACC_GROUP / 2 - 1);
				if (any0 && any1)
				{
					entry_pair(dx, dy, G, alpha0, alpha1, v2f{ Q3.x, Q3.y }, v2f{ Q3.z, Q3.w }, v2f{ Q4.x, Q4.y }, v2f{ Q4.z, Q4.w },
					           v2f{ Q5.x, Q5.y }, v2f{ Q5.z, Q5.w }, act0, act1, ip);
					grp_mask |= 3u << (2 * ip);
				}
				else if (any0)
				{
					entry(dx.x, dy.x, power.x, G.x, alpha0, Q3.x, Q3.z, Q4.x, Q4.z, Q5.x, Q5.z, act0, 2 * ip);
					grp_mask |= 1u << (2 * ip);
				}
				else if (any1)
				{
					entry(dx.y, dy.y, power.y, G.y, alpha1, Q3.y, Q3.w, Q4.y, Q4.w, Q5.y, Q5.w, act1, 2 * ip + 1);
					grp_mask |= 2u << (2 * ip);
				}
				if (ip == ACC_GROUP / 2 - 1 || i == npairs - 1)
				{
					if (grp_mask) drain(i - ip, grp_mask);
					grp_mask = 0u;
				}
			}
			__syncthreads();
		}
	}

	// Two entry points: the colour-only kernel (the training step: photometric loss on the render only) and the general one.
#define FDGS_BWD_PARAMS \
		const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list, const float4* __restrict__ records, \
		const uint32_t* __restrict__ tile_order, int W, int H, int grid_x, int ntiles, const float* __restrict__ bg, \
		const float* __restrict__ final_Ts, const uint32_t* __restrict__ n_contrib, \
		const float* __restrict__ dL_dpixels, const float* __restrict__ dL_depths, const float* __restrict__ dL_masks, \
		const float* __restrict__ dL_dpix_flow, float* __restrict__ gacc, const uint32_t* __restrict__ ctl
#define FDGS_BWD_ARGS ranges, point_list, records, tile_order, W, H, grid_x, ntiles, bg, final_Ts, n_contrib, dL_dpixels, dL_depths, dL_masks, dL_dpix_flow, gacc, ctl
	template <bool AUX> __global__ void __launch_bounds__(WAVE) blend_bwd_kernel(FDGS_BWD_PARAMS);
	template <> __global__ void __launch_bounds__(WAVE) blend_bwd_kernel<false>(FDGS_BWD_PARAMS) { blend_bwd_body<false>(FDGS_BWD_ARGS); }
	template <> __global__ void __launch_bounds__(WAVE) blend_bwd_kernel<true>(FDGS_BWD_PARAMS) { blend_bwd_body<true>(FDGS_BWD_ARGS); }
#undef FDGS_BWD_PARAMS
#undef FDGS_BWD_ARGS

	hipError_t launch_blend_bwd(const fdgs_scene& s, const fdgs_backward_in& in, const fdgs_backward_out& out,
	                            const float* records, const uint32_t* point_list, const uint32_t* ranges, const uint32_t* tile_order,
	                            const float* final_T, const uint32_t* n_contrib, const uint32_t* ctl, hipStream_t stream)
	{
		const int gx = div_up(s.W, TILE_X), gy = div_up(s.H, TILE_Y);
		const int ntiles = gx * gy;
#define LAUNCH_BWD(AUX) hipLaunchKernelGGL((blend_bwd_kernel<AUX>), dim3(blend_grid(ntiles)), dim3(WAVE), 0, stream, \
		                   reinterpret_cast<const uint2*>(ranges), point_list, reinterpret_cast<const float4*>(records), \
		                   tile_order, s.W, s.H, gx, ntiles, s.bg, final_T, n_contrib, \
		                   in.dL_dout_color, in.dL_dout_depth, in.dL_dout_alpha, in.dL_dout_flow, out.grad_accum, ctl)
		if (s.P >= (1 << 26)) return hipErrorInvalidValue;   // 32-bit byte offsets into the 64-byte accumulator records
		if (in.dL_dout_depth || in.dL_dout_alpha || in.dL_dout_flow) LAUNCH_BWD(true);
		else LAUNCH_BWD(false);
#undef LAUNCH_BWD
		return hipGetLastError();
	}
}

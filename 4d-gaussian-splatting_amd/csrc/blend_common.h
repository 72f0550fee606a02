// blend_common.h -- pieces shared by the forward and backward blend kernels (gfx950, wave64).
//
// Work decomposition: one wave64 (= one 64-thread workgroup) owns one 8x8 pixel
// block, four blocks per 16x16 tile.  The tile list itself (point_list / ranges) is
// the reference's AABB list, bit-identical to the oracle (with fdgs_forward_out.tile_cull: that list
// without the instances that cannot reach alpha >= 1/255 in the tile, preprocess_fwd.hip); what changes is how a wave
// consumes it: each lane fetches ONE list entry, tests whether that Gaussian can
// reach alpha >= 1/255 anywhere inside the wave's pixel block (exact minimum of the
// conic quadratic over the block rectangle, conservative slack), and the survivors
// are ballot-compacted into a wave-private LDS queue.  On the C3 workload only ~43 %
// of (entry, block) pairs survive, and every survivor costs the 64 lanes the full
// per-pixel evaluation, so the test (done once per entry by one lane) removes more
// than half of the blend arithmetic without changing a single pixel: an entry is
// dropped only if the per-pixel test (alpha < 1/255, reference forward.cu:590) would
// have rejected it for every pixel of the block.  List positions are kept so
// n_contrib keeps the reference's meaning (position in the full tile list).
#pragma once
#include "fdgs_common.h"

namespace fdgs
{
	constexpr int BLK = 8;        // pixel block edge per wave
	constexpr int NUM_XCDS = 8;

	__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

	// Lane predicates as wave masks.  A predicate that is the AND of several compares costs the compiler a
	// v_cndmask(0/1) + v_cmp_ne pair whenever a ballot of it is needed; keeping the compares as 64-bit masks (one v_cmp each,
	// combined on the scalar unit) and selecting with the mask directly avoids both.  Uniform control flow only (all 64 lanes active).
	typedef unsigned long long lanemask;
	__device__ __forceinline__ lanemask mask_of(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
	// mask bit set ? a : b
	__device__ __forceinline__ float mask_select(lanemask m, float a, float b)
	{
		float r;
		asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
		return r;
	}
	__device__ __forceinline__ uint32_t mask_select(lanemask m, uint32_t a, uint32_t b)
	{
		uint32_t r;
		asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
		return r;
	}

	// float4 k of Gaussian id's 48-byte blend record: uniform base + 32-bit byte offset (P < 2^26, checked by the launchers), so
	// that the gather needs one multiply-add per lane instead of 64-bit pointer arithmetic
	__device__ __forceinline__ float4 record_word(const float4* __restrict__ records, uint32_t id, uint32_t k)
	{
		return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(records) + (id * 48u + 16u * k));
	}

	struct BlockId { int tile, sub; };
	// Workgroup id -> (tile, 8x8 sub-block).  Workgroups are dealt round-robin to the 8 XCDs (id % 8): the 4 sub-blocks of a tile stay on
	// one XCD (the shared list and records in one L2), and position p of the tile order goes to XCD p % 8.  `order` (tilebin.hip,
	// tile_order_block) lists ALL tiles longest-first, so every XCD gets the same share of the long and the short lists and ends its
	// part of the launch on short tiles (the ramp-down of a 21 760-workgroup launch in index order was ~10 % of its duration; with the
	// longest-first order inside a contiguous eighth of the tiles per XCD -- rounds 2-4 -- the XCDs' unequal shares of the scene cost
	// the blend kernels another 16 % on the bench's cameras); NULL: index order, dealt out the same way.
	__device__ __forceinline__ BlockId block_of(int wg, int ntiles, const uint32_t* __restrict__ order)
	{
		const int xcd = wg % NUM_XCDS, k = wg / NUM_XCDS;
		BlockId b;
		b.sub = k & 3;
		const int p = (k >> 2) * NUM_XCDS + xcd;   // may be >= ntiles at the end of the grid: the caller returns
		b.tile = (order != nullptr && p < ntiles) ? (int)order[p] : p;
		return b;
	}
	static inline int blend_grid(int ntiles) { return div_up(ntiles, NUM_XCDS) * NUM_XCDS * 4; }

	// Can the Gaussian (record words a = (x, y, conic.x, conic.y), b = (conic.z, opacity, ..))
	// reach alpha >= 1/255 at some point of the rectangle [rx0,rx1] x [ry0,ry1] (pixel centres)?
	// alpha = opacity * exp(-q), q(d) = 0.5 (A dx^2 + C dy^2) + B dx dy  ==>  need min q <= ln(255 opacity).
	// The minimum of the convex q over the rectangle is 0 if the mean is inside, otherwise it lies
	// on an edge that faces the mean (1-D quadratics, clamped).  Slack 0.05 in q (5 % in alpha) covers
	// every rounding difference to the per-pixel evaluation; degenerate conics are never culled.
	__device__ __forceinline__ bool block_reaches(const float4 a, const float4 b, float rx0, float rx1, float ry0, float ry1)
	{
		const float A = a.z, B = a.w, Cc = b.x, op = b.y;
		if (!(op >= 0.0039f)) return false;                 // alpha <= opacity < 1/255 (1/255 = 0.003922)
		// degenerate conics, and conics that are not positive definite with a margin (a needle of 300 : 1 whose determinant has
		// lost its bits): never culled -- what follows needs q convex
		if (!(A > 0.0f && Cc > 0.0f && A * Cc > 1.00001f * (B * B))) return true;
		const float dx0 = rx0 - a.x, dx1 = rx1 - a.x, dy0 = ry0 - a.y, dy1 = ry1 - a.y;
		const bool xin = dx0 <= 0.0f && dx1 >= 0.0f, yin = dy0 <= 0.0f && dy1 >= 0.0f;
		if (xin && yin) return true;
		// 255 op >= 0.99: normal range; v_log_f32 (log2, 1 ulp) is exact enough by five orders of magnitude for a bound with 0.05 of slack
		const float tau = __builtin_amdgcn_logf(255.0f * op) * 0.69314718056f + 0.05f;
		// A convex q takes its minimum over the rectangle on an edge that FACES the mean (the mean strictly beyond the edge's
		// line: at the minimiser p the direction towards the mean leaves the rectangle, and it can only leave it through a
		// constraint that is active at p).  That is at most one vertical and one horizontal edge.
		const float ex = dx0 > 0.0f ? dx0 : dx1, ey = dy0 > 0.0f ? dy0 : dy1;
		const float invA = __builtin_amdgcn_rcpf(A), invC = __builtin_amdgcn_rcpf(Cc);
		const float dyv = __builtin_amdgcn_fmed3f(-B * ex * invC, dy0, dy1);   // clamp (dy0 <= dy1)
		const float qv = 0.5f * (A * ex * ex + Cc * dyv * dyv) + B * ex * dyv;
		const float dxh = __builtin_amdgcn_fmed3f(-B * ey * invA, dx0, dx1);
		const float qh = 0.5f * (A * dxh * dxh + Cc * ey * ey) + B * dxh * ey;
		const float qmin = fminf(xin ? 3.0e38f : qv, yin ? 3.0e38f : qh);
		return qmin <= tau;
	}
}

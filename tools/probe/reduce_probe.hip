// dev probe: transpose_reduce16 in isolation
#include <hip/hip_runtime.h>
#include <cstdio>
namespace fdgs {
	template <int CTRL>
	__device__ __forceinline__ float dpp_mov(float v)
	{
		return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
	}

	// Sums 16 slots (v[12..15] are zero) over the wave with a butterfly "transpose-reduce": at every halving
	// step a lane keeps one half of its slots and receives its partner's copy of that half.
	//   lane ^ 32, lane ^ 16 : gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper / odd halves of a
	//                          register PAIR in one instruction, so a step is swap + add per output slot;
	//   lane ^ 8,  lane ^ 4  : DPP row_ror:8 and row_shr:4 / row_shl:4 on the selected half;
	//   lane ^ 2,  lane ^ 1  : plain DPP quad-permute adds (every lane of a quad ends with the same total).
	// 8*2 + 4*2 + 2*3 + 6 + 2 = 38 VALU ops, no LDS traffic.  On return every lane holds the wave total of slot
	//   s(L) = 8*b5 + 4*b4 + 2*b3 + b2   (b_i = bit i of the lane id).
	__device__ __forceinline__ float2 swap32(float a, float b)
	{
		// (read the two results through __uint_as_float: __builtin_bit_cast on r[1] miscompiles to r[0] with ROCm 7.2's clang)
		const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
		const unsigned x0 = r[0], x1 = r[1];
		return make_float2(__uint_as_float(x0), __uint_as_float(x1));
	}
	__device__ __forceinline__ float2 swap16(float a, float b)
	{
		const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
		const unsigned x0 = r[0], x1 = r[1];
		return make_float2(__uint_as_float(x0), __uint_as_float(x1));
	}
	__device__ __forceinline__ float transpose_reduce16(const float (&v)[16], int lane)
	{
		const bool b2 = lane & 4, b3 = lane & 8;
		float w[8], u[4], t[2];
#pragma unroll
		for (int i = 0; i < 8; i++)   // partner = lane ^ 32: lanes 0..31 keep slot i, lanes 32..63 keep slot i + 8
		{
			const float2 x = swap32(v[i], v[i + 8]);
			w[i] = x.x + x.y;
		}
#pragma unroll
		for (int i = 0; i < 4; i++)   // partner = lane ^ 16: even rows keep slot i, odd rows keep slot i + 4
		{
			const float2 x = swap16(w[i], w[i + 4]);
			u[i] = x.x + x.y;
		}
#pragma unroll
		for (int i = 0; i < 2; i++)   // partner = lane ^ 8 (row_ror:8)
		{
			const float keep = b3 ? u[i + 2] : u[i];
			const float send = b3 ? u[i] : u[i + 2];
			t[i] = keep + dpp_mov<0x128>(send);
		}
		float r;
		{                             // partner = lane ^ 4 (row_shr:4 for bit-2 lanes, row_shl:4 otherwise)
			const float keep = b2 ? t[1] : t[0];
			const float send = b2 ? t[0] : t[1];
			const float from_lo = dpp_mov<0x114>(send); // lane i <- lane i-4
			const float from_hi = dpp_mov<0x104>(send); // lane i <- lane i+4
			r = keep + (b2 ? from_lo : from_hi);
		}
		r += dpp_mov<0x4E>(r);        // lane ^ 2 (quad_perm [2,3,0,1])
		r += dpp_mov<0xB1>(r);        // lane ^ 1 (quad_perm [1,0,3,2])
		return r;
	}


}
__global__ void k(float* out, int mode)
{
	const int lane = threadIdx.x;
	float v[16];
	for (int i = 0; i < 16; i++) v[i] = (i < 12) ? (float)(lane * 16 + i) : 0.f;
	bool active = mode == 0 ? true : (lane % 3 == 0);
	if (!active) for (int i = 0; i < 16; i++) v[i] = 0.f;
	if (__ballot(active) != 0ull)
	{
		const float r = fdgs::transpose_reduce16(v, lane);
		out[lane] = r;
	}
}
int main()
{
	float* d; (void)hipMalloc(&d, 64 * 4);
	for (int mode = 0; mode < 2; mode++)
	{
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
		float h[64]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
		int bad = 0;
		for (int lane = 0; lane < 64; lane++)
		{
			const int slot = ((lane & 32) >> 2) | ((lane & 16) >> 2) | ((lane & 8) >> 2) | ((lane & 4) >> 2);
			double want = 0;
			for (int l = 0; l < 64; l++) if (mode == 0 || l % 3 == 0) want += (slot < 12) ? (double)(l * 16 + slot) : 0.0;
			if (h[lane] != (float)want) { if (bad < 6) printf("mode %d lane %d slot %d got %g want %g\n", mode, lane, slot, h[lane], want); bad++; }
		}
		printf("mode %d: %d bad lanes\n", mode, bad);
	}
	return 0;
}

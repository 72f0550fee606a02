// TEST INFRASTRUCTURE ONLY -- not part of the product.
//
// Minimal host-side stand-in for the CUDA runtime headers so that the *device
// code* of the reference rasterizer (forward.cu / backward.cu /
// rasterizer_impl.cu, read in place from /root/reference at build time and cut
// before their <<<...>>> host launchers) compiles verbatim with g++.
// Semantics mirrored here: CUDA vector types, CUDA's global non-template
// min/max overload set, __expf, atomicAdd(float*), and the thread/block
// identity + barrier hooks used by ref_emu.h's fiber block emulator.
#pragma once
#include <math.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct int2 { int x, y; };
struct dim3 {
	unsigned int x, y, z;
	constexpr dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};

// CUDA's overload set for min/max in device code (math_functions.hpp):
// non-template, with mixed signed/unsigned -> unsigned and float/double mixes.
inline int min(int a, int b) { return a < b ? a : b; }
inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
inline float min(float a, float b) { return fminf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double min(float a, double b) { return fmin((double)a, b); }
inline double min(double a, float b) { return fmin(a, (double)b); }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
inline long long max(long long a, long long b) { return a > b ? a : b; }
inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
inline double max(float a, double b) { return fmax((double)a, b); }
inline double max(double a, float b) { return fmax(a, (double)b); }

// Fast-math intrinsic: on the CPU the best available stand-in is expf().
inline float __expf(float x) { return expf(x); }
inline void __trap() { abort(); }

// atomicAdd(float*): tiles are emulated on several OS threads, so a real
// atomic RMW is required (CAS loop on the bit pattern).
inline float atomicAdd(float* addr, float val)
{
	uint32_t* ia = reinterpret_cast<uint32_t*>(addr);
	uint32_t old = __atomic_load_n(ia, __ATOMIC_RELAXED);
	for (;;)
	{
		float f;
		memcpy(&f, &old, 4);
		float nf = f + val;
		uint32_t nv;
		memcpy(&nv, &nf, 4);
		if (__atomic_compare_exchange_n(ia, &old, nv, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
			return f;
	}
}

// Identity + barrier hooks (defined in ref_emu.h / set by the drivers).
namespace refemu
{
	struct ThreadCtx
	{
		unsigned long long grid_rank;  // cg::this_grid().thread_rank()
		dim3 block_idx;                // blockIdx
		dim3 thread_idx;               // threadIdx
		unsigned int block_rank;       // linear thread index in block
	};
	extern thread_local ThreadCtx g_ctx;
	void block_barrier();              // block.sync() / __syncthreads()
	int block_barrier_count(int pred); // __syncthreads_count()
}

inline int __syncthreads_count(int pred) { return refemu::block_barrier_count(pred); }
inline void __syncthreads() { refemu::block_barrier(); }

// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (dev probe).
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed)
{
	// 8 independent accumulator chains per lane: enough ILP to hide the ALU latency with one wave per SIMD
	v2f a[8];
	for (int i = 0; i < 8; i++) a[i] = v2f{ seed + i, seed - i };
	const v2f m = { 1.0001f, 0.9999f }, c = { 1e-3f, -1e-3f };
	for (int it = 0; it < iters; it++)
	{
#pragma unroll
		for (int i = 0; i < 8; i++)
		{
			if (MODE == 0) { a[i].x = __builtin_fmaf(a[i].x, m.x, c.x); }                       // 1 scalar fma
			else if (MODE == 1) { a[i] = __builtin_elementwise_fma(a[i], m, c); }                // 1 packed fma (2 flops pairs)
			else { a[i].x = __builtin_fmaf(a[i].x, m.x, c.x); a[i].y = __builtin_fmaf(a[i].y, m.y, c.y); }  // 2 scalar fmas
		}
	}
	float s = 0.f;
	for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run(float* d, int blocks, int iters)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

int main()
{
	const int blocks = 256 * 8, iters = 20000;
	float* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
	const double t0 = run<0>(d, blocks, iters), t1 = run<1>(d, blocks, iters), t2 = run<2>(d, blocks, iters);
	const double inst = (double)blocks * 4 /*waves*/ * iters * 8;
	printf("scalar fma  : %.3f ms  (%.2f cycles/instr/SIMD at 2.4 GHz)\n", t0, t0 * 1e-3 * 2.4e9 / (inst / 1024));
	printf("packed fma  : %.3f ms  (%.2f cycles/instr/SIMD)\n", t1, t1 * 1e-3 * 2.4e9 / (inst / 1024));
	printf("2x scalar   : %.3f ms  (%.2f cycles per pair)\n", t2, t2 * 1e-3 * 2.4e9 / (inst / 1024));
	return 0;
}

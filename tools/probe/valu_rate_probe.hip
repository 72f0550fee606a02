// Issue cost of cross-lane VALU ops on gfx950 relative to v_add_f32: v_add_f32_dpp, v_permlane32_swap, v_permlane16_swap,
// ds_bpermute.  Every wave runs ITER x 16 independent instructions of one kind; 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate_probe.hip -o valu_rate_probe && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2000
#define REP16(X) X(0,8) X(1,9) X(2,10) X(3,11) X(4,12) X(5,13) X(6,14) X(7,15) X(0,8) X(1,9) X(2,10) X(3,11) X(4,12) X(5,13) X(6,14) X(7,15)
template <int KIND> __global__ void __launch_bounds__(256) k(float* out)
{
	__shared__ float4 s_lds[1024];   // 16 KB
	if (threadIdx.x == 0 && out == nullptr) s_lds[0] = make_float4(0, 0, 0, 0);
	float r[16];
	for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 0.5f + i;
	const int addr = ((threadIdx.x ^ 16) & 63) * 4;
	for (int it = 0; it < ITER; it++)
	{
#define ADD(a, b) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[a]) : "v"(r[b]));
#define DPP(a, b) asm volatile("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(r[a]) : "v"(r[b]));
#define SW32(a, b) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[a]), "+v"(r[b]));
#define SW16(a, b) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[a]), "+v"(r[b]));
#define BPERM(a, b) asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(8)" : "+v"(r[a]) : "v"(addr));
		if (KIND == 0) { REP16(ADD) }
		if (KIND == 1) { REP16(DPP) }
		if (KIND == 2) { REP16(SW32) }
		if (KIND == 3) { REP16(SW16) }
		if (KIND == 4) { REP16(BPERM) asm volatile("s_waitcnt lgkmcnt(0)"); }
		if (KIND == 5 || KIND == 6)
		{
			// 16 ds_read_b128: KIND 5 every lane the same address (the blend kernels' queue reads), KIND 6 lane-private addresses
			float4 q[4];
			const int base = (KIND == 5 ? 0 : (int)(threadIdx.x & 63) * 16) + (it & 7) * 1024;
#pragma unroll
			for (int rep = 0; rep < 4; rep++)
			{
#pragma unroll
				for (int i = 0; i < 4; i++)
					asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(base), "n"(i * 2048));
				asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
				for (int i = 0; i < 4; i++) r[i] += q[i].x + q[i].y + q[i].z + q[i].w;
			}
		}
	}
	float s = 0;
	for (int i = 0; i < 16; i++) s += r[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, float* out)
{
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out);
	hipEventRecord(a);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	const double inst = (double)blocks * 4 * ITER * 16;
	printf("%-22s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 * 1024 / inst);
}
int main()
{
	float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
	run<0>("v_add_f32", out); run<1>("v_add_f32_dpp", out); run<2>("v_permlane32_swap", out); run<3>("v_permlane16_swap", out);
	run<4>("ds_bpermute_b32", out);
	run<5>("ds_read_b128 same addr", out); run<6>("ds_read_b128 per lane", out);
	return 0;
}

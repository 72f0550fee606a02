// dev probe: what do v_permlane32_swap / v_permlane16_swap do on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
	const unsigned lane = threadIdx.x;
	unsigned a = lane, b = 100 + lane;
	auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
	out[lane] = r[0]; out[64 + lane] = r[1];
	auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
	out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main()
{
	unsigned* d; hipMalloc(&d, 256 * 4);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	const char* names[4] = { "swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1" };
	for (int j = 0; j < 4; j++) { printf("%s:", names[j]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 * j + i]); printf("\n"); }
	return 0;
}

// bin_probe.hip -- standalone timing + validation of the tile-binning kernels (csrc/tilebin.hip) on synthetic
// rectangles shaped like the C3 workload (see tools/probe/README.md).  Build: hipcc --offload-arch=gfx950 -O3
// -std=c++17 -I../../4d-gaussian-splatting_amd/csrc bin_probe.hip -o bin_probe
#include "../../4d-gaussian-splatting_amd/csrc/tilebin.hip"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace fdgs;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv)
{
	const int P = argc > 1 ? atoi(argv[1]) : 300000;
	const int W = argc > 2 ? atoi(argv[2]) : 1352, H = argc > 3 ? atoi(argv[3]) : 1014;
	const float mean_r = argc > 4 ? atof(argv[4]) : 18.0f;
	const int depth_mode = argc > 5 ? atoi(argv[5]) : 0;   // 0 uniform, 1 two surfaces + outliers, 2 all equal
	const bool quick = argc > 6;                            // only the default variant (for profiling)
	const int gx = (W + 15) / 16, gy = (H + 15) / 16, T = gx * gy;
	std::mt19937 rng(1);
	std::uniform_real_distribution<float> U(0.f, 1.f);
	std::normal_distribution<float> Nrm(0.f, 1.f);
	std::vector<ushort4> rect(P);
	std::vector<float> depth(P);
	size_t R = 0;
	for (int i = 0; i < P; i++)
	{
		const float x = (U(rng) * 1.2f - 0.1f) * W, y = (U(rng) * 1.2f - 0.1f) * H;
		const int r = (int)std::ceil(mean_r * std::exp(0.35f * Nrm(rng)));
		const int x0 = std::min(gx, std::max(0, (int)((x - r) / 16))), y0 = std::min(gy, std::max(0, (int)((y - r) / 16)));
		const int x1 = std::min(gx, std::max(0, (int)((x + r + 15) / 16))), y1 = std::min(gy, std::max(0, (int)((y + r + 15) / 16)));
		if ((x1 - x0) * (y1 - y0) == 0 || U(rng) < 0.01f) { rect[i] = make_ushort4(0, 0, 0, 0); depth[i] = 0.f; continue; }
		rect[i] = make_ushort4(x0, y0, x1, y1);
		if (depth_mode == 0) depth[i] = 2.7f + 2.6f * U(rng);
		else if (depth_mode == 1) { const float u = U(rng); depth[i] = u < 0.6f ? 3.0f + 0.02f * Nrm(rng) : (u < 0.98f ? 4.5f + 0.05f * Nrm(rng) : 0.3f + 100.f * U(rng)); }
		else depth[i] = 3.0f;
		R += (size_t)(x1 - x0) * (y1 - y0);
	}
	printf("P %d  %dx%d  T %d  R %zu (%.1f per tile)\n", P, W, H, T, R, (double)R / T);

	ushort4* d_rect; float* d_depth; uint32_t *d_cnt, *d_ctl, *d_pl; uint2 *d_pairs, *d_ranges; u64* d_big;
	CK(hipMalloc(&d_rect, P * 8)); CK(hipMalloc(&d_depth, P * 4));
	CK(hipMalloc(&d_cnt, (T + 8) * 4)); CK(hipMalloc(&d_ctl, 16)); CK(hipMalloc(&d_pl, R * 4 + 4));
	CK(hipMalloc(&d_pairs, R * 8 + 8)); CK(hipMalloc(&d_ranges, T * 8)); CK(hipMalloc(&d_big, R * 16 + 16));
	CK(hipMemcpy(d_rect, rect.data(), P * 8, hipMemcpyHostToDevice));
	CK(hipMemcpy(d_depth, depth.data(), P * 4, hipMemcpyHostToDevice));

	// CPU reference
	std::vector<std::pair<uint64_t, uint32_t>> ref; // (tile, depth<<32|id)
	std::vector<std::pair<uint32_t, uint64_t>> keys;
	keys.reserve(R);
	for (int i = 0; i < P; i++)
		for (int y = rect[i].y; y < rect[i].w; y++)
			for (int x = rect[i].x; x < rect[i].z; x++)
			{
				uint32_t db; memcpy(&db, &depth[i], 4);
				keys.push_back({ (uint32_t)(y * gx + x), ((uint64_t)db << 32) | (uint32_t)i });
			}
	std::sort(keys.begin(), keys.end());

#ifdef FDGS_TS_TIMELINE
	unsigned int* d_tl;
	CK(hipMalloc(&d_tl, (size_t)T * 64));
	CK(hipMemset(d_tl, 0, (size_t)T * 64));
	CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &d_tl, sizeof d_tl));
#endif
	hipEvent_t ev[8];
	for (auto& e : ev) CK(hipEventCreate(&e));
	const int per_thread = ((T + 1023) / 1024 + 3) / 4 * 4;
	CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_bin_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
	CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_bin_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
	// variant: rounds > 0: LDS-histogram kernels, rounds x 1024 Gaussians per workgroup; rounds == 0: one global atomic per instance
	for (int rounds : { 0, 1, 2, 4 })
		for (int caplim : { 0, 64 })
			for (int rank_max : { 0, 1 })
			{
				const int batch = rounds * 1024;
				if ((caplim == 64 || rank_max == 1) && rounds != 1) continue;
				if (quick && (rounds != 1 || caplim != 0 || rank_max != 0)) continue;
				if (caplim == 64 && rank_max == 1) continue;
				if (rounds > 0 && T > 36 * 1024) continue;
				tile_sort_debug_limits(caplim, rank_max);
				float acc[5] = { 0, 0, 0, 0, 0 };
				const int reps = 20;
				uint32_t ctl[2];
				for (int rep = 0; rep < reps + 2; rep++)
				{
					CK(hipEventRecord(ev[0]));
					CK(hipMemsetAsync(d_cnt, 0, (T + 4) * 4));
					CK(hipEventRecord(ev[1]));
					if (rounds) hipLaunchKernelGGL(tile_bin_lds_kernel<false>, dim3((P + batch - 1) / batch), dim3(1024), (size_t)std::max(T, 8 * ORDER_BUCKETS) * 4, 0, d_rect, (const float*)nullptr, P, gx, T, rounds, d_cnt, (uint2*)nullptr, (const uint32_t*)nullptr, 0u, (uint32_t*)nullptr, (T + 7) / 8);
					else hipLaunchKernelGGL(tile_bin_direct_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, 0, d_rect, (const float*)nullptr, P, gx, d_cnt, (uint2*)nullptr, (const uint32_t*)nullptr, 0u, (uint32_t*)nullptr, T, (T + 7) / 8);
					CK(hipEventRecord(ev[2]));
					hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, 0, d_cnt, T, per_thread, d_ctl, (uint32_t*)nullptr, 0u, (uint32_t*)nullptr);
					CK(hipEventRecord(ev[3]));
					CK(hipMemcpy(ctl, d_ctl, 8, hipMemcpyDeviceToHost));
					CK(hipEventRecord(ev[6]));
					if (rounds) hipLaunchKernelGGL(tile_bin_lds_kernel<true>, dim3((P + batch - 1) / batch), dim3(1024), (size_t)std::max(T, 8 * ORDER_BUCKETS) * 4, 0, d_rect, d_depth, P, gx, T, rounds, d_cnt, d_pairs, (const uint32_t*)d_ctl, 0xFFFFFFFFu, (uint32_t*)nullptr, (T + 7) / 8);
					else hipLaunchKernelGGL(tile_bin_direct_kernel<true>, dim3((P + 255) / 256), dim3(256), 0, 0, d_rect, d_depth, P, gx, d_cnt, d_pairs, (const uint32_t*)d_ctl, 0xFFFFFFFFu, (uint32_t*)nullptr, T, (T + 7) / 8);
					CK(hipEventRecord(ev[4]));
					CK(launch_tile_sort(d_cnt, T, (int)ctl[1], (const uint32_t*)d_pairs, d_pl, (uint32_t*)d_ranges, d_big, (const uint32_t*)d_ctl, 0xFFFFFFFFu, (const uint32_t*)nullptr, 0));
					CK(hipEventRecord(ev[5]));
					CK(hipDeviceSynchronize());
					CK(hipGetLastError());
					if (rep >= 2)
					{
						float ms;
						for (int k = 0; k < 5; k++)
						{
							if (k == 3) CK(hipEventElapsedTime(&ms, ev[6], ev[4])); else CK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
							acc[k] += ms;
						}
					}
				}
				std::vector<uint32_t> pl(R);
				std::vector<uint2> rg(T);
				CK(hipMemcpy(pl.data(), d_pl, R * 4, hipMemcpyDeviceToHost));
				CK(hipMemcpy(rg.data(), d_ranges, T * 8, hipMemcpyDeviceToHost));
				size_t bad = 0;
				if (ctl[0] != R) bad++;
				for (size_t s = 0; s < R; s++) if (pl[s] != (uint32_t)keys[s].second) bad++;
				size_t pos = 0;
				uint32_t maxc = 0;
				for (int t = 0; t < T; t++)
				{
					size_t e = pos;
					while (e < R && keys[e].first == (uint32_t)t) e++;
					maxc = std::max<uint32_t>(maxc, e - pos);
					if (e == pos) { if (rg[t].x != 0 || rg[t].y != 0) bad++; }
					else if (rg[t].x != pos || rg[t].y != e) bad++;
					pos = e;
				}
				if (maxc != ctl[1]) bad++;
#ifdef FDGS_TS_TIMELINE
				{
					std::vector<unsigned int> tl((size_t)T * 16);
					CK(hipMemcpy(tl.data(), d_tl, tl.size() * 4, hipMemcpyDeviceToHost));
					const char* nm[9] = { "load", "splitters", "search+hist", "scan", "scatter", "rank", "barrier", "out", "store" };
					printf("   tile_sort cycles per tile (wave 0, last launch):");
					double tot = 0;
					for (int k = 0; k < 9; k++) { double a2 = 0; for (int t = 0; t < T; t++) a2 += tl[(size_t)t * 16 + k]; printf(" %s %.0f", nm[k], a2 / T); tot += a2 / T; }
					printf(" | total %.0f\n", tot);
					// the same for the tiles of every length class
					for (int lo : { 0, 1024, 2048, 4096, 8192 })
					{
						const int hi = lo == 0 ? 1024 : 2 * lo;
						double a3[9] = { 0 }; int cnt = 0;
						for (int t = 0; t < T; t++) { const int n = (int)(rg[t].y - rg[t].x); if (n > lo && n <= hi) { cnt++; for (int k = 0; k < 9; k++) a3[k] += tl[(size_t)t * 16 + k]; } }
						if (!cnt) continue;
						printf("   lists (%d, %d]: %d tiles:", lo, hi, cnt);
						double t3 = 0; for (int k = 0; k < 9; k++) { printf(" %s %.0f", nm[k], a3[k] / cnt); t3 += a3[k] / cnt; }
						printf(" | total %.0f cycles\n", t3);
					}
				}
#endif
				printf("batch %4d lds_cap_limit %4d rank_max %2d | memset %.1f  count %.1f  scan %.1f  scatter %.1f  sort %.1f us | max list %u | %s (%zu bad)\n",
				       batch, caplim, rank_max, acc[0] / reps * 1e3, acc[1] / reps * 1e3, acc[2] / reps * 1e3, acc[3] / reps * 1e3, acc[4] / reps * 1e3,
				       ctl[1], bad ? "MISMATCH" : "ok", bad);
			}
	return 0;
}

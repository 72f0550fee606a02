// TEST INFRASTRUCTURE ONLY.
// Runs the reference's simple-knn device code (simple-knn/simple_knn.cu:28-191: prepMorton, coord2Morton, boxMinMax,
// distBoxPoint, updateKBest, boxMeanDist -- pulled in verbatim by build_ref.py as knn_trunc.inc, cut before the host
// function SimpleKNN::knn at :193) on the CPU.  The host sequence of SimpleKNN::knn (:193-220) is restated below with
// std:: in place of cub / thrust, call by call:
//   cub::DeviceReduce::Reduce(CustomMin / CustomMax, init = {0,0,0})  -> a fold from {0,0,0} with the reference's functors
//   coord2Morton<<<>>>                                                -> the kernel, one call per thread index
//   thrust::sequence + cub::DeviceRadixSort::SortPairs (stable LSD)   -> std::iota + std::stable_sort on the codes
//   boxMinMax<<<num_boxes, 1024>>> (shared memory + __syncthreads)    -> the kernel under the fiber block emulator
//   boxMeanDist<<<num_boxes, 1024>>>                                  -> the kernel, one call per thread index
#include <cfloat>
#include <algorithm>
#include <numeric>
#include <vector>
#include "cuda_runtime.h"
#include "ref_emu.h"
// the kernels read the built-in index variables directly (the rasterizer's go through cooperative groups)
#define threadIdx (refemu::g_ctx.thread_idx)
#define blockIdx (refemu::g_ctx.block_idx)
#include "knn_trunc.inc"
#undef threadIdx
#undef blockIdx

extern "C" int oracle_ref_dist2_knn3(int P, const float* points_f, float* mean_dists)
{
	if (P <= 0) return 0;
	float3* points = (float3*)points_f;
	// simple_knn.cu:199-207
	const float3 init = { 0, 0, 0 };
	float3 minn = init, maxx = init;
	for (int i = 0; i < P; i++) { minn = CustomMin()(minn, points[i]); maxx = CustomMax()(maxx, points[i]); }
	// :209-211
	std::vector<uint32_t> morton((size_t)P);
	for (int idx = 0; idx < P; idx++)
	{
		refemu::g_ctx.grid_rank = (unsigned long long)idx;
		coord2Morton(P, points, minn, maxx, morton.data());
	}
	// :213-220
	std::vector<uint32_t> indices_sorted((size_t)P);
	std::iota(indices_sorted.begin(), indices_sorted.end(), 0u);
	std::stable_sort(indices_sorted.begin(), indices_sorted.end(), [&](uint32_t a, uint32_t b) { return morton[a] < morton[b]; });
	// :222-224
	const uint32_t num_boxes = ((uint32_t)P + BOX_SIZE - 1) / BOX_SIZE;
	std::vector<MinMax> boxes(num_boxes);
	for (uint32_t b = 0; b < num_boxes; b++)
		refemu::runner().run(dim3(b), dim3(BOX_SIZE), [&]() { boxMinMax((uint32_t)P, points, indices_sorted.data(), boxes.data()); });
	// :225 (threads are independent: no barrier in this kernel)
#pragma omp parallel for schedule(dynamic, 256)
	for (int idx = 0; idx < (int)(num_boxes * BOX_SIZE); idx++)
	{
		refemu::g_ctx.grid_rank = (unsigned long long)idx;
		boxMeanDist((uint32_t)P, points, indices_sorted.data(), boxes.data(), mean_dists);
	}
	return 0;
}

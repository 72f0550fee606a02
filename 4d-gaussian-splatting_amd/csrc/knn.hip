// knn.hip -- distCUDA2: mean squared distance to the three nearest neighbours (gfx950).
//
// SURVEY.md section 8f rank 4, second half: the only other native code the reference runs by default, once at
// initialisation, to size the initial Gaussians (scene/gaussian_model.py:274; simple-knn/simple_knn.cu).
// Same plan as the reference -- Morton order, boxes of 1024 consecutive points, exact search with box pruning --
// re-cut for wave64 / LDS:
//   * bounds and Morton codes stay on the device (the reference copies min / max to the host: two syncs);
//   * the (code, index) pairs go through this library's own radix sort (binning.hip);
//   * the search runs one workgroup per 256 consecutive (Morton-sorted, hence spatially close) queries; a candidate
//     box that ANY of them still needs is staged once into LDS and scanned from there by the lanes that need it,
//     instead of every thread gathering every candidate point from global memory (simple_knn.cu:176-188).
// The result does not depend on the search order; the arithmetic that defines it is kept literally and the TU is
// compiled with FP contraction off:  d = (dx*dx + dy*dy) + dz*dz ;  mean = ((b0 + b1) + b2) / 3  -> bit-exact vs the oracle.
#include "fdgs_common.h"
#include <cfloat>

namespace fdgs
{
	constexpr int KNN_BOX = 1024;       // simple_knn.cu:12
	constexpr int KNN_THREADS = 256;

	struct KnnLayout { size_t code[2], idx[2], hist, boxes, bounds, total; };
	static inline KnnLayout knn_layout(int P)
	{
		KnnLayout L;
		size_t o = 0;
		const size_t p = (size_t)(P > 0 ? P : 1);
		for (int i = 0; i < 2; i++) { L.code[i] = o; o = align_up(o + p * 4); }
		for (int i = 0; i < 2; i++) { L.idx[i] = o; o = align_up(o + p * 4); }
		L.hist = o; o = align_up(o + (size_t)RADIX * (sort_blocks((int)p) + 1) * 4);
		L.boxes = o; o = align_up(o + (size_t)div_up((int)p, KNN_BOX) * 6 * 4);
		L.bounds = o; o = align_up(o + 6 * 4);
		L.total = o;
		return L;
	}

	// min / max over all points AND the origin (cub::DeviceReduce with init {0,0,0}, simple_knn.cu:198-205)
	__global__ void __launch_bounds__(1024) knn_bounds_kernel(int P, const float* __restrict__ pts, float* __restrict__ bounds)
	{
		__shared__ float red[6][1024 / WAVE];
		float mn[3] = { 0.f, 0.f, 0.f }, mx[3] = { 0.f, 0.f, 0.f };
		for (int i = threadIdx.x; i < P; i += 1024)
			for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
		for (int k = 0; k < 3; k++)
			for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		if (lane == 0) for (int k = 0; k < 3; k++) { red[k][wave] = mn[k]; red[3 + k][wave] = mx[k]; }
		__syncthreads();
		if (threadIdx.x < 6)
		{
			float v = red[threadIdx.x][0];
			for (int w = 1; w < 1024 / WAVE; w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
			bounds[threadIdx.x] = v;
		}
	}

	__device__ __forceinline__ uint32_t prep_morton(uint32_t x)   // simple_knn.cu:45-52
	{
		x = (x | (x << 16)) & 0x030000FF;
		x = (x | (x << 8)) & 0x0300F00F;
		x = (x | (x << 4)) & 0x030C30C3;
		x = (x | (x << 2)) & 0x09249249;
		return x;
	}

	__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float* __restrict__ pts, const float* __restrict__ bounds,
	                                                         uint32_t* __restrict__ codes, uint32_t* __restrict__ idx)
	{
		const int i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= P) return;
		uint32_t c[3];
		for (int k = 0; k < 3; k++)   // simple_knn.cu:54-61
			c[k] = prep_morton((uint32_t)(((pts[3 * (size_t)i + k] - bounds[k]) / (bounds[3 + k] - bounds[k])) * ((1 << 10) - 1)));
		codes[i] = c[0] | (c[1] << 1) | (c[2] << 2);
		idx[i] = (uint32_t)i;
	}

	// one workgroup per box of 1024 Morton-consecutive points (simple_knn.cu:77-122)
	__global__ void __launch_bounds__(KNN_THREADS) knn_box_bounds_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
	                                                                     float* __restrict__ boxes)
	{
		__shared__ float red[6][KNN_THREADS / WAVE];
		float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
		for (int j = threadIdx.x; j < KNN_BOX; j += KNN_THREADS)
		{
			const int i = blockIdx.x * KNN_BOX + j;
			if (i < P)
				for (int k = 0; k < 3; k++) { const float v = pts[3 * (size_t)order[i] + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
		}
		for (int k = 0; k < 3; k++)
			for (int o = 32; o > 0; o >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], o)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o)); }
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
		if (lane == 0) for (int k = 0; k < 3; k++) { red[k][wave] = mn[k]; red[3 + k][wave] = mx[k]; }
		__syncthreads();
		if (threadIdx.x < 6)
		{
			float v = red[threadIdx.x][0];
			for (int w = 1; w < KNN_THREADS / WAVE; w++) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
			boxes[6 * (size_t)blockIdx.x + threadIdx.x] = v;
		}
	}

	__device__ __forceinline__ void update_k_best(float px, float py, float pz, float qx, float qy, float qz, float* knn)
	{
		// simple_knn.cu:131-145, K = 3
		const float dx = qx - px, dy = qy - py, dz = qz - pz;
		float dist = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
		for (int j = 0; j < 3; j++)
			if (knn[j] > dist) { const float t = knn[j]; knn[j] = dist; dist = t; }
	}

	__device__ __forceinline__ float dist_box_point(const float* box, float px, float py, float pz)
	{
		// simple_knn.cu:124-134
		float dx = 0.f, dy = 0.f, dz = 0.f;
		if (px < box[0] || px > box[3]) dx = fminf(fabsf(px - box[0]), fabsf(px - box[3]));
		if (py < box[1] || py > box[4]) dy = fminf(fabsf(py - box[1]), fabsf(py - box[4]));
		if (pz < box[2] || pz > box[5]) dz = fminf(fabsf(pz - box[2]), fabsf(pz - box[5]));
		return dx * dx + dy * dy + dz * dz;
	}

	__global__ void __launch_bounds__(KNN_THREADS) knn_mean_dist_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
	                                                                    const float* __restrict__ boxes, float* __restrict__ dists)
	{
		__shared__ float sx[KNN_BOX], sy[KNN_BOX], sz[KNN_BOX];
		__shared__ int s_need;
		const int idx = blockIdx.x * KNN_THREADS + threadIdx.x;
		const bool valid = idx < P;
		float px = 0.f, py = 0.f, pz = 0.f;
		uint32_t me = 0;
		if (valid) { me = order[idx]; px = pts[3 * (size_t)me]; py = pts[3 * (size_t)me + 1]; pz = pts[3 * (size_t)me + 2]; }
		float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
		if (valid)
			for (int i = max(0, idx - 3); i <= min(P - 1, idx + 3); i++)   // simple_knn.cu:164-169
			{
				if (i == idx) continue;
				const uint32_t o = order[i];
				update_k_best(px, py, pz, pts[3 * (size_t)o], pts[3 * (size_t)o + 1], pts[3 * (size_t)o + 2], best);
			}
		const float reject = best[2];
		best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;

		const int nboxes = (P + KNN_BOX - 1) / KNN_BOX;
		for (int b = 0; b < nboxes; b++)
		{
			bool need = false;
			if (valid)
			{
				const float d = dist_box_point(boxes + 6 * (size_t)b, px, py, pz);
				need = !(d > reject || d > best[2]);                        // simple_knn.cu:178-180
			}
			if (threadIdx.x == 0) s_need = 0;
			__syncthreads();
			if (need) s_need = 1;
			__syncthreads();
			if (s_need == 0) continue;                                       // nobody in this workgroup needs the box
			const int first = b * KNN_BOX, cnt = min(KNN_BOX, P - first);
			for (int j = threadIdx.x; j < cnt; j += KNN_THREADS)
			{
				const uint32_t o = order[first + j];
				sx[j] = pts[3 * (size_t)o]; sy[j] = pts[3 * (size_t)o + 1]; sz[j] = pts[3 * (size_t)o + 2];
			}
			__syncthreads();
			if (need)
				for (int j = 0; j < cnt; j++)
				{
					if (first + j == idx) continue;                          // :185
					update_k_best(px, py, pz, sx[j], sy[j], sz[j], best);
				}
			__syncthreads();
		}
		if (valid) dists[me] = ((best[0] + best[1]) + best[2]) / 3.0f;       // :191
	}
}

extern "C" size_t fdgs_knn_scratch_bytes(int32_t P) { return fdgs::knn_layout(P).total; }

extern "C" int fdgs_dist2_knn3(int32_t P, const float* points, float* mean_dist2, void* scratch, void* stream_v)
{
	using namespace fdgs;
	if (P < 0) return FDGS_ERR_INVALID_ARG;
	if (P == 0) return FDGS_OK;
	if (!points || !mean_dist2 || !scratch) return FDGS_ERR_INVALID_ARG;
	hipStream_t stream = (hipStream_t)stream_v;
	const KnnLayout L = knn_layout(P);
	char* s = (char*)scratch;
	uint32_t* codes[2] = { (uint32_t*)(s + L.code[0]), (uint32_t*)(s + L.code[1]) };
	uint32_t* idx[2] = { (uint32_t*)(s + L.idx[0]), (uint32_t*)(s + L.idx[1]) };
	float* bounds = (float*)(s + L.bounds);
	float* boxes = (float*)(s + L.boxes);
	hipLaunchKernelGGL(knn_bounds_kernel, dim3(1), dim3(1024), 0, stream, P, points, bounds);
	hipLaunchKernelGGL(knn_morton_kernel, dim3(div_up(P, 256)), dim3(256), 0, stream, P, points, bounds, codes[0], idx[0]);
	int res = 0;
	if (radix_sort_pairs(codes, idx, P, 0, 32, (uint32_t*)(s + L.hist), stream, &res) != hipSuccess) return FDGS_ERR_HIP;
	const int nboxes = div_up(P, KNN_BOX);
	hipLaunchKernelGGL(knn_box_bounds_kernel, dim3(nboxes), dim3(KNN_THREADS), 0, stream, P, points, idx[res], boxes);
	hipLaunchKernelGGL(knn_mean_dist_kernel, dim3(div_up(P, KNN_THREADS)), dim3(KNN_THREADS), 0, stream, P, points, idx[res], boxes, mean_dist2);
	return hipGetLastError() == hipSuccess ? FDGS_OK : FDGS_ERR_HIP;
}

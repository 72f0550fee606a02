// fdgs_math.h -- small fixed-size vector / matrix helpers for the per-Gaussian kernels.
//
// The forward preprocess has to reproduce the reference's float results BIT-EXACTLY
// wherever they feed integers (radius, tile rectangle, tile count, depth-key bits),
// so these helpers fix one evaluation order: matrix products accumulate
// k = 0,1,2(,3) left to right, dot3 = (x+y)+z, dot4 = (x+y)+(z+w) -- the order the
// reference's GLM expressions evaluate in (glm/detail/type_mat3x3.inl:486,
// type_mat4x4.inl:630, func_geometric.inl:48-65).  Translation units that need
// bit-exactness are compiled with FP contraction off (no FMA fusion).
#pragma once
#include <hip/hip_runtime.h>

namespace fdgs
{
	struct M3 { float c[3][3]; };   // c[column][row]
	struct M4 { float c[4][4]; };

	__device__ __forceinline__ M3 mul(const M3& A, const M3& B)
	{
		M3 R;
#pragma unroll
		for (int j = 0; j < 3; j++)
#pragma unroll
			for (int i = 0; i < 3; i++)
				R.c[j][i] = A.c[0][i] * B.c[j][0] + A.c[1][i] * B.c[j][1] + A.c[2][i] * B.c[j][2];
		return R;
	}
	__device__ __forceinline__ M3 transpose(const M3& A)
	{
		M3 R;
#pragma unroll
		for (int j = 0; j < 3; j++)
#pragma unroll
			for (int i = 0; i < 3; i++) R.c[j][i] = A.c[i][j];
		return R;
	}
	__device__ __forceinline__ M4 mul(const M4& A, const M4& B)
	{
		M4 R;
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int i = 0; i < 4; i++)
				R.c[j][i] = A.c[0][i] * B.c[j][0] + A.c[1][i] * B.c[j][1] + A.c[2][i] * B.c[j][2] + A.c[3][i] * B.c[j][3];
		return R;
	}
	__device__ __forceinline__ M4 transpose(const M4& A)
	{
		M4 R;
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int i = 0; i < 4; i++) R.c[j][i] = A.c[i][j];
		return R;
	}
	__device__ __forceinline__ M4 diag4(float a, float b, float c, float d)
	{
		M4 S;
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int i = 0; i < 4; i++) S.c[j][i] = 0.0f;
		S.c[0][0] = a; S.c[1][1] = b; S.c[2][2] = c; S.c[3][3] = d;
		return S;
	}
	__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
	{
		return ax * bx + ay * by + az * bz;
	}
	__device__ __forceinline__ float dot4(const float* a, const float* b)
	{
		return (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
	}

	// Left / right isoclinic factors of the 4D rotation, column-major
	// (reference forward.cu:315-327): R4 = M_r * M_l.
	__device__ __forceinline__ void build_Ml_Mr(const float4 rot, const float4 rot_r, M4& L, M4& Rr)
	{
		const float a = rot.x, b = rot.y, c = rot.z, d = rot.w;
		const float p = rot_r.x, q = rot_r.y, r = rot_r.z, s = rot_r.w;
		const float l[4][4] = { { a, b, -c, d }, { -b, a, d, c }, { c, -d, a, b }, { -d, -c, -b, a } };
		const float m[4][4] = { { p, q, -r, -s }, { -q, p, s, -r }, { r, -s, p, -q }, { s, r, q, p } };
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int i = 0; i < 4; i++) { L.c[j][i] = l[j][i]; Rr.c[j][i] = m[j][i]; }
	}

	// quaternion (w,x,y,z) -> rotation, column-major (reference forward.cu:251-262)
	__device__ __forceinline__ M3 quat_to_R(const float4 q)
	{
		const float r = q.x, x = q.y, y = q.z, z = q.w;
		M3 R;
		R.c[0][0] = 1.f - 2.f * (y * y + z * z); R.c[0][1] = 2.f * (x * y - r * z); R.c[0][2] = 2.f * (x * z + r * y);
		R.c[1][0] = 2.f * (x * y + r * z); R.c[1][1] = 1.f - 2.f * (x * x + z * z); R.c[1][2] = 2.f * (y * z - r * x);
		R.c[2][0] = 2.f * (x * z - r * y); R.c[2][1] = 2.f * (y * z + r * x); R.c[2][2] = 1.f - 2.f * (x * x + y * y);
		return R;
	}

	// The reference model's activations (scene/gaussian_model.py:55-66, 179-219), used when fdgs_scene.raw_params != 0
	__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
	__device__ __forceinline__ float4 act_normalize(const float4 v, float* inv_norm)
	{
		const float n = sqrtf((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
		const float inv = 1.0f / fmaxf(n, 1e-12f); // F.normalize eps
		*inv_norm = inv;
		return make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
	}
	// d/dv of v / |v| applied to g:  (g - q (q . g)) / |v|
	__device__ __forceinline__ float4 act_normalize_bwd(const float4 q, float inv_norm, const float4 g)
	{
		const float d = (q.x * g.x + q.y * g.y) + (q.z * g.z + q.w * g.w);
		return make_float4((g.x - q.x * d) * inv_norm, (g.y - q.y * d) * inv_norm, (g.z - q.z * d) * inv_norm, (g.w - q.w * d) * inv_norm);
	}

	// SH constants (reference auxiliary.h:23-40)
	__device__ constexpr float SH_C0 = 0.28209479177387814f;
	__device__ constexpr float SH_C1 = 0.4886025119029199f;
	__device__ constexpr float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f };
	__device__ constexpr float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f };
	constexpr double REF_PI = 3.14159265; // the reference's truncated pi (auxiliary.h:20), double

	// view * p with the translation row (auxiliary.h:59-67), view stored transposed
	__device__ __forceinline__ float3 xform4x3(const float3 p, const float* __restrict__ m)
	{
		return make_float3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
		                   m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
		                   m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
	}
	__device__ __forceinline__ float4 xform4x4(const float3 p, const float* __restrict__ m)
	{
		return make_float4(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
		                   m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
		                   m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
		                   m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]);
	}

	// The EWA projection shared by forward (forward.cu:198-237) and backward (backward.cu:509-537).
	struct Cov2D
	{
		float3 t;          // clamped view-space mean
		float txtz, tytz;
		M3 T, Vrk, W;
		float a, b, c;     // cov2D entries before the +0.3 low-pass
	};
	__device__ __forceinline__ Cov2D project_cov(const float3 mean, float focal_x, float focal_y, float tan_fovx, float tan_fovy,
	                                              const float* cov3D, const float* __restrict__ vm)
	{
		Cov2D o;
		float3 t = xform4x3(mean, vm);
		const float limx = 1.3f * tan_fovx, limy = 1.3f * tan_fovy;
		o.txtz = t.x / t.z; o.tytz = t.y / t.z;
		t.x = fminf(limx, fmaxf(-limx, o.txtz)) * t.z;
		t.y = fminf(limy, fmaxf(-limy, o.tytz)) * t.z;
		M3 J;
		J.c[0][0] = focal_x / t.z; J.c[0][1] = 0.0f; J.c[0][2] = -(focal_x * t.x) / (t.z * t.z);
		J.c[1][0] = 0.0f; J.c[1][1] = focal_y / t.z; J.c[1][2] = -(focal_y * t.y) / (t.z * t.z);
		J.c[2][0] = 0.0f; J.c[2][1] = 0.0f; J.c[2][2] = 0.0f;
		o.W.c[0][0] = vm[0]; o.W.c[0][1] = vm[4]; o.W.c[0][2] = vm[8];
		o.W.c[1][0] = vm[1]; o.W.c[1][1] = vm[5]; o.W.c[1][2] = vm[9];
		o.W.c[2][0] = vm[2]; o.W.c[2][1] = vm[6]; o.W.c[2][2] = vm[10];
		o.T = mul(o.W, J);
		o.Vrk.c[0][0] = cov3D[0]; o.Vrk.c[0][1] = cov3D[1]; o.Vrk.c[0][2] = cov3D[2];
		o.Vrk.c[1][0] = cov3D[1]; o.Vrk.c[1][1] = cov3D[3]; o.Vrk.c[1][2] = cov3D[4];
		o.Vrk.c[2][0] = cov3D[2]; o.Vrk.c[2][1] = cov3D[4]; o.Vrk.c[2][2] = cov3D[5];
		const M3 cov = mul(mul(transpose(o.T), transpose(o.Vrk)), o.T);
		o.a = cov.c[0][0]; o.b = cov.c[0][1]; o.c = cov.c[1][1];
		o.t = t;
		return o;
	}
}

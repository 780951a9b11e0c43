// device_math.h -- fp32 vector helpers for the HIP stages.
//
// Floating-point contract: every value is what the REFERENCE's OpenCL C source computes when it
// is compiled for this GPU by ROCm's own OpenCL C compiler with `-ffp-contract=off
// -cl-fp32-correctly-rounded-divide-sqrt` and linked against ROCm's OpenCL builtin library (opencl.bc / ocml.bc /
// ockl.bc of /opt/rocm/amdgcn/bitcode) -- the code object the test infrastructure builds from the reference's sources
//  and the GPU tests run next to these kernels.  Concretely:
//   * user expressions: written out scalar-by-scalar in the evaluation order the OpenCL C source implies, this unit is
//     compiled with -ffp-contract=off, hipcc's fp32 divide / sqrt are correctly rounded;
//   * OpenCL builtins: restated below exactly as that library implements them (read from its bitcode):
//       dot(a,b)     = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))              (fmuladd chain -> v_fma_f32 on gfx950)
//       cross(a,b).x = fma(a.y,b.z, -(a.z*b.y)), .y, .z cyclic
//       min / max / fmax = llvm.minnum / maxnum (v_min_f32 / v_max_f32: a NaN operand is dropped)
//       clamp(x,lo,hi)   = v_med3_f32(x, lo, hi)
//       mix(a,b,t)       = fma(b - a, t, a)
//       normalize(v)     = v * rsqrt(dot(v,v)) with rsqrt = v_rsq_f32 (and the library's rescaling of tiny / huge
//                          inputs), NOT v / sqrt(dot)
//       sin cos acos pow sqrt: the same OCML functions hipcc links for sinf cosf acosf powf sqrtf.
//   The CPU restatement used by the tests follows the same contract with fmaf(); it cannot reproduce v_rsq_f32 and the
//   OCML transcendentals bit for bit, so CPU-vs-GPU comparisons of anything behind a normalize carry a tolerance,
//   while GPU-vs-reference-on-GPU comparisons are bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdx {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 one_minus(f3 a) { return mk3(1.0f - a.x, 1.0f - a.y, 1.0f - a.z); }

__device__ __forceinline__ float cl_min(float x, float y) { return __builtin_fminf(x, y); }     // __ocml_min_f32
__device__ __forceinline__ float cl_max(float x, float y) { return __builtin_fmaxf(x, y); }     // __ocml_max_f32 / fmax
__device__ __forceinline__ float cl_clamp(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }   // __ockl_median3_f32
__device__ __forceinline__ float cl_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return cl_fma(a.z, b.z, cl_fma(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float dot4(f4 a, f4 b) { return cl_fma(a.w, b.w, cl_fma(a.z, b.z, cl_fma(a.y, b.y, a.x * b.x))); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3(cl_fma(a.y, b.z, b.y * (-a.z)), cl_fma(a.z, b.x, b.z * (-a.x)), cl_fma(a.x, b.y, b.x * (-a.y)));
}
__device__ __forceinline__ f3 mix3(f3 a, f3 b, float t)
{
    return mk3(cl_fma(b.x - a.x, t, a.x), cl_fma(b.y - a.y, t, a.y), cl_fma(b.z - a.z, t, a.z));
}
// __ocml_rsqrt_f32 with fp32 denormals enabled (the mode of both code objects): v_rsq_f32, denormal inputs rescaled
__device__ __forceinline__ float cl_rsqrt(float x)
{
    const bool tiny = x < 0x1p-126f;
    const float r = __builtin_amdgcn_rsqf(tiny ? x * 0x1p+24f : x);
    return tiny ? r * 4096.0f : r;
}
__device__ __forceinline__ float cl_sel01(float v) { return __builtin_copysignf(__builtin_isinf(v) ? 1.0f : 0.0f, v); }
// normalize(float3) / normalize(float4) of opencl.bc: zero vector unchanged; |v|^2 below FLT_MIN -> scaled by 2^86,
// infinite -> scaled by 2^-66 (and, if still infinite, replaced by the +-1 / 0 pattern of its infinite components)
__device__ __forceinline__ f3 normalize3(f3 v)
{
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f) return v;
    float l2 = dot3(v, v);
    if (l2 < 0x1p-126f) { v = v * 0x1p+86f; l2 = dot3(v, v); }
    else if (l2 == __builtin_inff()) {
        v = v * 0x1p-66f; l2 = dot3(v, v);
        if (l2 == __builtin_inff()) { v = mk3(cl_sel01(v.x), cl_sel01(v.y), cl_sel01(v.z)); l2 = dot3(v, v); }
    }
    return v * cl_rsqrt(l2);
}
__device__ __forceinline__ f4 normalize4(f4 v)
{
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f && v.w == 0.0f) return v;
    float l2 = dot4(v, v);
    if (l2 < 0x1p-126f) { v.x *= 0x1p+86f; v.y *= 0x1p+86f; v.z *= 0x1p+86f; v.w *= 0x1p+86f; l2 = dot4(v, v); }
    else if (l2 == __builtin_inff()) {
        v.x *= 0x1p-66f; v.y *= 0x1p-66f; v.z *= 0x1p-66f; v.w *= 0x1p-66f; l2 = dot4(v, v);
        if (l2 == __builtin_inff()) { v.x = cl_sel01(v.x); v.y = cl_sel01(v.y); v.z = cl_sel01(v.z); v.w = cl_sel01(v.w); l2 = dot4(v, v); }
    }
    const float r = cl_rsqrt(l2);
    f4 o; o.x = v.x * r; o.y = v.y * r; o.z = v.z * r; o.w = v.w * r;
    return o;
}

// row-major 4x4 times (x,y,z,w): the four-term sums of math.cl:25-31, kept in full so that zero
// signs and NaN propagation are those of the reference expression
__device__ __forceinline__ f4 mat4_mul(const float* __restrict__ m, float x, float y, float z, float w)
{
    f4 o;
    o.x = m[0] * x + m[1] * y + m[2] * z + m[3] * w;
    o.y = m[4] * x + m[5] * y + m[6] * z + m[7] * w;
    o.z = m[8] * x + m[9] * y + m[10] * z + m[11] * w;
    o.w = m[12] * x + m[13] * y + m[14] * z + m[15] * w;
    return o;
}
__device__ __forceinline__ f3 mat4_mul3(const float* __restrict__ m, float x, float y, float z, float w)
{
    return mk3(m[0] * x + m[1] * y + m[2] * z + m[3] * w,
               m[4] * x + m[5] * y + m[6] * z + m[7] * w,
               m[8] * x + m[9] * y + m[10] * z + m[11] * w);
}

// math.cl:10-23 (integer PCG3D hash; exact on every device)
__device__ __forceinline__ f3 pcg3d(uint32_t vx, uint32_t vy, uint32_t vz)
{
    vx = vx * 1664525u + 1013904223u;
    vy = vy * 1664525u + 1013904223u;
    vz = vz * 1664525u + 1013904223u;
    vx += vy * vz; vy += vz * vx; vz += vx * vy;
    vx ^= vx >> 16u; vy ^= vy >> 16u; vz ^= vz >> 16u;
    vx += vy * vz; vy += vz * vx; vz += vx * vy;
    const float denom = (float)0xffffffffu;
    return mk3((float)vx / denom, (float)vy / denom, (float)vz / denom);
}

// math.cl:56-183: cofactor inverse, term order preserved.  Returns false (out untouched) if det == 0.
__host__ __device__ inline bool inverse_mat4(const float* m, float* out)
{
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) return false;
    det = 1.0f / det;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * det;
    return true;
}

} // namespace rdx

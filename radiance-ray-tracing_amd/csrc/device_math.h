// device_math.h -- fp32 vector helpers for the HIP stages.
//
// Every expression is written out scalar-by-scalar in the evaluation order the reference's OpenCL C
// implies, and the translation unit is compiled with -ffp-contract=off, so the +,-,*,/ and sqrt
// results are the IEEE-754 values (hipcc's default fp32 divide/sqrt are correctly rounded).
// OpenCL builtins are pinned to: min(x,y) = y<x?y:x, max(x,y) = x<y?y:x, clamp = min(max(x,lo),hi),
// mix(a,b,t) = a+(b-a)*t, dot accumulated x->y->z(->w), normalize(v) = v / sqrt(dot(v,v)).
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

__device__ __forceinline__ float cl_min(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float cl_max(float x, float y) { return x < y ? y : x; }
__device__ __forceinline__ float cl_clamp(float x, float lo, float hi) { return cl_min(cl_max(x, lo), hi); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalize3(f3 v) { return v / sqrtf(dot3(v, v)); }
__device__ __forceinline__ f4 normalize4(f4 v)
{
    float l = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
    f4 r; r.x = v.x / l; r.y = v.y / l; r.z = v.z / l; r.w = v.w / l;
    return r;
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

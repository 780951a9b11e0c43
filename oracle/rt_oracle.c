/*
 * rt_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See rt_oracle.h for the
 * scope, the floating-point pin and the parity-pin status ("parity unpinned" except for the
 * builtin-free functions checked against oracle/_ref and tests/golden/ref_*.npz).
 *
 * Every function cites the reference file:line it restates.  The code deliberately keeps the
 * reference's megakernel structure (one pixel at a time, recursive traceRay from the closest-hit
 * shader, exhaustive left-first stack traversal) so that it is an independent check of the
 * product's wavefront/HIP formulation rather than a copy of it.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* OpenCL vector types and the pinned builtins                                                 */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;
typedef struct { float s[16]; } m44;           /* float16, s0..s3 = row 0 (math.cl:4) */

static inline f3 F3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 f3_splat(float a) { return F3(a, a, a); }
static inline f3 f3_add(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 f3_sub(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 f3_mul(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 f3_div(f3 a, f3 b) { return F3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline f3 f3_scale(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
static inline f3 f3_divs(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
static inline f3 f3_neg(f3 a) { return F3(-a.x, -a.y, -a.z); }

/* OpenCL builtins as ROCm's OpenCL builtin library (opencl.bc / ocml.bc, ROCm 7.2) implements them for gfx950 -- the
 * floating-point contract of the project (rt_oracle.h).  Exact on the CPU: dot / cross / mix (explicit fma chains),
 * min / max (minnum / maxnum: a NaN operand is dropped), clamp (v_med3_f32).  NOT reproducible bit for bit on the CPU:
 * normalize (v * v_rsq_f32(dot), a <= 1 ulp hardware approximation; here the correctly rounded 1/sqrt) and the OCML
 * transcendentals (here glibc's) -- everything behind those carries a tolerance against the GPU. */
static inline float cl_min(float x, float y) { return (x != x) ? y : ((y != y) ? x : (y < x ? y : x)); }
static inline float cl_max(float x, float y) { return (x != x) ? y : ((y != y) ? x : (x < y ? y : x)); }
static inline float cl_clamp(float x, float lo, float hi) { return (x != x) ? cl_min(lo, hi) : cl_min(cl_max(x, lo), hi); }
static inline f3 cl_min3(f3 a, f3 b) { return F3(cl_min(a.x, b.x), cl_min(a.y, b.y), cl_min(a.z, b.z)); }
static inline f3 cl_max3(f3 a, f3 b) { return F3(cl_max(a.x, b.x), cl_max(a.y, b.y), cl_max(a.z, b.z)); }
static inline f3 cl_clamp3(f3 a, float lo, float hi) { return F3(cl_clamp(a.x, lo, hi), cl_clamp(a.y, lo, hi), cl_clamp(a.z, lo, hi)); }
static inline float cl_dot3(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
static inline float cl_dot4(f4 a, f4 b) { return __builtin_fmaf(a.w, b.w, __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x))); }
static inline f3 cl_cross(f3 a, f3 b)
{
    return F3(__builtin_fmaf(a.y, b.z, b.y * (-a.z)), __builtin_fmaf(a.z, b.x, b.z * (-a.x)), __builtin_fmaf(a.x, b.y, b.x * (-a.y)));
}
static inline float cl_rsqrt(float x) { return (float)(1.0 / sqrt((double)x)); }
static inline float cl_sel01(float v) { return copysignf(isinf(v) ? 1.0f : 0.0f, v); }
static inline f3 cl_normalize3(f3 v)
{
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f) return v;
    float l2 = cl_dot3(v, v);
    if (l2 < 0x1p-126f) { v = f3_scale(v, 0x1p+86f); l2 = cl_dot3(v, v); }
    else if (l2 == INFINITY) {
        v = f3_scale(v, 0x1p-66f); l2 = cl_dot3(v, v);
        if (l2 == INFINITY) { v = F3(cl_sel01(v.x), cl_sel01(v.y), cl_sel01(v.z)); l2 = cl_dot3(v, v); }
    }
    return f3_scale(v, cl_rsqrt(l2));
}
static inline f4 cl_normalize4(f4 v)
{
    if (v.x == 0.0f && v.y == 0.0f && v.z == 0.0f && v.w == 0.0f) return v;
    float l2 = cl_dot4(v, v);
    if (l2 < 0x1p-126f) { v.x *= 0x1p+86f; v.y *= 0x1p+86f; v.z *= 0x1p+86f; v.w *= 0x1p+86f; l2 = cl_dot4(v, v); }
    else if (l2 == INFINITY) {
        v.x *= 0x1p-66f; v.y *= 0x1p-66f; v.z *= 0x1p-66f; v.w *= 0x1p-66f; l2 = cl_dot4(v, v);
        if (l2 == INFINITY) { v.x = cl_sel01(v.x); v.y = cl_sel01(v.y); v.z = cl_sel01(v.z); v.w = cl_sel01(v.w); l2 = cl_dot4(v, v); }
    }
    const float r = cl_rsqrt(l2);
    f4 o = {v.x * r, v.y * r, v.z * r, v.w * r};
    return o;
}
static inline f3 cl_mix3(f3 a, f3 b, float t)
{
    return F3(__builtin_fmaf(b.x - a.x, t, a.x), __builtin_fmaf(b.y - a.y, t, a.y), __builtin_fmaf(b.z - a.z, t, a.z));
}

/* ------------------------------------------------------------------------------------------ */
/* math.cl                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* math.cl:10-23 random_pcg3d */
static f3 random_pcg3d(uint32_t vx, uint32_t vy, uint32_t vz)
{
    vx = vx * 1664525u + 1013904223u;
    vy = vy * 1664525u + 1013904223u;
    vz = vz * 1664525u + 1013904223u;
    vx += vy * vz; vy += vz * vx; vz += vx * vy;
    vx ^= vx >> 16u; vy ^= vy >> 16u; vz ^= vz >> 16u;
    vx += vy * vz; vy += vz * vx; vz += vx * vy;
    f3 ret;
    ret.x = (float)vx / (float)0xffffffffu;
    ret.y = (float)vy / (float)0xffffffffu;
    ret.z = (float)vz / (float)0xffffffffu;
    return ret;
}

/* math.cl:25-31 */
static void MultiplyMat4Vec4(const m44* a, const f4* b, f4* out)
{
    const float* s = a->s;
    f4 o;
    o.x = s[0] * b->x + s[1] * b->y + s[2] * b->z + s[3] * b->w;
    o.y = s[4] * b->x + s[5] * b->y + s[6] * b->z + s[7] * b->w;
    o.z = s[8] * b->x + s[9] * b->y + s[10] * b->z + s[11] * b->w;
    o.w = s[12] * b->x + s[13] * b->y + s[14] * b->z + s[15] * b->w;
    *out = o;
}

/* math.cl:56-183 InverseMat4x4 (cofactor expansion; term order kept) */
static int InverseMat4x4(const m44* mm, m44* invOut)
{
    const float* m = mm->s;
    float inv[16], det;
    int i;

    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] +
             m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] -
             m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] +
             m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] -
              m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] -
             m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] +
             m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] -
             m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] +
              m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] +
             m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] -
             m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] +
              m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] -
              m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] -
             m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] +
             m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] -
              m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] +
              m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];

    det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0)
        return 0;
    det = 1.0f / det;
    for (i = 0; i < 16; i++)
        invOut->s[i] = inv[i] * det;
    return 1;
}

/* math.cl:185-252 Euler{X,Y,Z}ToMat4x4 */
static void EulerXToMat4x4(float t, m44* o)
{
    float c = cosf(t), s = sinf(t);
    float v[16] = {1, 0, 0, 0, 0, c, -s, 0, 0, s, c, 0, 0, 0, 0, 1};
    memcpy(o->s, v, sizeof v);
}
static void EulerYToMat4x4(float t, m44* o)
{
    float c = cosf(t), s = sinf(t);
    float v[16] = {c, 0, s, 0, 0, 1, 0, 0, -s, 0, c, 0, 0, 0, 0, 1};
    memcpy(o->s, v, sizeof v);
}
static void EulerZToMat4x4(float t, m44* o)
{
    float c = cosf(t), s = sinf(t);
    float v[16] = {c, -s, 0, 0, s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(o->s, v, sizeof v);
}

/* math.cl:269-298 GetNormalSpace */
static void GetNormalSpace(f3 normal, m44* out)
{
    f3 someVec = {1.0f, 0.0f, 0.0f};
    float dd = cl_dot3(someVec, normal);
    f3 tangent = {0.0f, 1.0f, 0.0f};
    if (1.0f - fabsf(dd) > 1e-6f)
        tangent = cl_normalize3(cl_cross(someVec, normal));
    f3 bitangent = cl_cross(normal, tangent);
    float* s = out->s;
    s[0] = tangent.x;  s[4] = tangent.y;  s[8] = tangent.z;   s[12] = 0;
    s[1] = bitangent.x; s[5] = bitangent.y; s[9] = bitangent.z; s[13] = 0;
    s[2] = normal.x;   s[6] = normal.y;   s[10] = normal.z;   s[14] = 0;
    s[3] = 0; s[7] = 0; s[11] = 0; s[15] = 1;
}

/* ------------------------------------------------------------------------------------------ */
/* pbr.cl                                                                                      */
/* ------------------------------------------------------------------------------------------ */
#define PI 3.14159265359f

/* pbr.cl:6-13 */
static float D_GGX(float dotNH, float roughness)
{
    float alpha = roughness * roughness;
    float alpha2 = alpha * alpha;
    float denom = dotNH * dotNH * (alpha2 - 1.0f) + 1.0f;
    return (alpha2) / (PI * denom * denom);
}

/* pbr.cl:31-37 */
static f3 F_Schlick(float cosTheta, float metallic, f3 albedo)
{
    f3 minValue = {0.04f, 0.04f, 0.04f};
    f3 F0 = cl_mix3(minValue, albedo, metallic);
    float p = powf(1.0f - cosTheta, 5.0f);
    f3 oneMinus = F3(1.0f - F0.x, 1.0f - F0.y, 1.0f - F0.z);
    return f3_add(F0, f3_scale(oneMinus, p));
}

/* pbr.cl:41-64 */
static inline float Cos2Theta(f3 w) { return w.z * w.z; }
static inline float Sin2Theta(f3 w) { return fmaxf(0.0f, 1.0f - Cos2Theta(w)); }
static inline float SinTheta(f3 w) { return sqrtf(Sin2Theta(w)); }
static inline float CosPhi(f3 w)
{
    float sinTheta = SinTheta(w);
    return (sinTheta == 0.0f) ? 1.0f : cl_clamp(w.x / sinTheta, -1.0f, 1.0f);
}
static inline float SinPhi(f3 w)
{
    float sinTheta = SinTheta(w);
    return (sinTheta == 0.0f) ? 0.0f : cl_clamp(w.y / sinTheta, -1.0f, 1.0f);
}
static inline float Tan2Theta(f3 w) { return Sin2Theta(w) / Cos2Theta(w); }

/* pbr.cl:66-74 */
static float Lambda(f3 w, float a)
{
    float tan2Theta = Tan2Theta(w);
    if (isinf(tan2Theta))
        return 0.0f;
    float alpha2 = (CosPhi(w) * a) * (CosPhi(w) * a) + (SinPhi(w) * a) * (SinPhi(w) * a);
    return (sqrtf(1.0f + alpha2 * tan2Theta) - 1.0f) / 2.0f;
}

/* pbr.cl:77-96 */
static float G_pbrt(f3 wo, f3 wi, f3 N, float roughness)
{
    m44 mat, matInv;
    f4 localIn, localOut;
    f4 globalIn = {wi.x, wi.y, wi.z, 0.0f};
    f4 globalOut = {wo.x, wo.y, wo.z, 0.0f};
    memset(&matInv, 0, sizeof matInv); /* reference leaves it uninitialised when det == 0 */
    GetNormalSpace(N, &mat);
    InverseMat4x4(&mat, &matInv);
    MultiplyMat4Vec4(&matInv, &globalOut, &localOut);
    MultiplyMat4Vec4(&matInv, &globalIn, &localIn);
    if (localIn.z < 0 || localOut.z < 0)
        return 0.0f;
    return 1 / (1 + Lambda(F3(localIn.x, localIn.y, localIn.z), roughness) +
                Lambda(F3(localOut.x, localOut.y, localOut.z), roughness));
}

/* pbr.cl:171-174 */
static f3 reflect_(f3 in, f3 N)
{
    return f3_add(f3_neg(in), f3_scale(N, 2 * cl_dot3(in, N)));
}

/* pbr.cl:176-186 */
static f3 refract_(f3 V, f3 H, float eta)
{
    float cosTheta_i = cl_dot3(H, V);
    float sin2Theta_i = cl_max(0.0f, 1.0f - (cosTheta_i * cosTheta_i));
    float sin2Theta_t = sin2Theta_i / (eta * eta);
    if ((1.0f - sin2Theta_t) < 0.0f)
        return f3_divs(f3_sub(f3_scale(H, cosTheta_i), V), eta);
    float cosTheta_t = sqrtf(1.0f - sin2Theta_t);
    return f3_add(f3_divs(f3_neg(V), eta), f3_scale(H, cosTheta_i / eta - cosTheta_t));
}

/* pbr.cl:268-287 */
static f3 microfacetBRDF(f3 L, f3 V, f3 N, f3 albedo, float metallicness, float roughness,
                         float transmission, float ior)
{
    (void)ior;
    f3 H = cl_normalize3(f3_add(V, L));
    float NoV = cl_clamp(cl_dot3(N, V), 0.0f, 1.0f);
    float NoL = cl_clamp(cl_dot3(N, L), 0.0f, 1.0f);
    float NoH = cl_clamp(cl_dot3(N, H), 0.0f, 1.0f);
    float VoH = cl_clamp(cl_dot3(V, H), 0.0f, 1.0f);

    f3 F = F_Schlick(VoH, metallicness, albedo);
    float D = D_GGX(NoH, roughness);
    float G = G_pbrt(V, L, N, roughness);

    f3 f_specular = f3_divs(f3_scale(F, D * G), cl_max(4.0f * NoV * NoL, 0.001f));
    f3 notSpec = f3_scale(f3_scale(F3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z), (1.0f - metallicness)),
                          (1.0f - transmission));
    f3 f_diffuse = f3_mul(notSpec, f3_divs(albedo, PI));
    return f3_scale(f3_add(f_diffuse, f_specular), NoL);
}

/* shared by the three lobes of pbr.cl:289-385: GGX half-vector / cosine sample in N's frame */
static f3 sample_local_to_world(f3 frameN, float theta, float phi)
{
    f4 local = {sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta), 0.0f};
    m44 mat; f4 tmp;
    GetNormalSpace(frameN, &mat);
    MultiplyMat4Vec4(&mat, &local, &tmp);
    return F3(tmp.x, tmp.y, tmp.z);
}

/* pbr.cl:289-385 */
static f3 sampleMicrofacetBRDF_transm(f3 V, f3 N, f3 baseColor, float metallicness, float roughness,
                                      float transmission, float ior, f3 random, f3* nextFactor)
{
    if (random.z < 0.5f) {
        if ((2.0f * random.z) < transmission) {
            f3 forwardNormal = N;
            float frontFacing = cl_dot3(V, N);
            float eta = ior;
            if (frontFacing < 0.0f) {
                forwardNormal = f3_neg(N);
                eta = 1.0f / ior;
            }
            float a = roughness * roughness;
            float theta = acosf(sqrtf((1.0f - random.y) / (1.0f + (a * a - 1.0f) * random.y)));
            float phi = 2.0f * PI * random.x;
            f3 H = sample_local_to_world(forwardNormal, theta, phi);
            f3 L = refract_(V, H, eta);

            float NoV = cl_clamp(cl_dot3(forwardNormal, V), 0.0f, 1.0f);
            float NoH = cl_clamp(cl_dot3(forwardNormal, H), 0.0f, 1.0f);
            float VoH = cl_clamp(cl_dot3(V, H), 0.0f, 1.0f);

            f3 F = F_Schlick(VoH, metallicness, baseColor);
            float G = G_pbrt(V, f3_neg(L), forwardNormal, roughness);
            f3 oneMinusF = F3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z);
            f3 nf = f3_divs(f3_scale(f3_scale(f3_mul(baseColor, oneMinusF), G), VoH),
                            cl_max((NoH * NoV), 0.001f));
            *nextFactor = f3_scale(nf, 2.0f);
            return L;
        } else {
            float theta = acosf(sqrtf(random.y));
            float phi = 2.0f * PI * random.x;
            f3 L = sample_local_to_world(N, theta, phi);
            f3 H = cl_normalize3(f3_add(V, L));
            float VoH = cl_clamp(cl_dot3(V, H), 0.0f, 1.0f);
            f3 F = F_Schlick(VoH, metallicness, baseColor);
            f3 oneMinusF = F3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z);
            f3 nf = f3_mul(f3_scale(oneMinusF, (1.0f - metallicness)), baseColor);
            *nextFactor = f3_scale(nf, 2.0f);
            return L;
        }
    } else {
        float a = roughness * roughness;
        float theta = acosf(sqrtf((1.0f - random.y) / (1.0f + (a * a - 1.0f) * random.y)));
        float phi = 2.0f * PI * random.x;
        f3 H = sample_local_to_world(N, theta, phi);
        f3 L = reflect_(V, H);

        float NoV = cl_clamp(cl_dot3(N, V), 0.0f, 1.0f);
        float NoH = cl_clamp(cl_dot3(N, H), 0.0f, 1.0f);
        float VoH = cl_clamp(cl_dot3(V, H), 0.0f, 1.0f);

        float G = G_pbrt(V, L, N, roughness);
        f3 F = F_Schlick(VoH, metallicness, baseColor);
        f3 nf = f3_divs(f3_scale(f3_scale(F, G), VoH), cl_max((NoH * NoV), 0.001f));
        *nextFactor = f3_scale(nf, 2.0f);
        return L;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* radiance.cl : traversal                                                                     */
/* ------------------------------------------------------------------------------------------ */
#define TYPE_INST 1
#define TYPE_TRIG 2
#define BVH_TOP_STACK_SIZE 8
#define BVH_BOT_STACK_SIZE 100

typedef struct {
    f3 hitPoint;
    float distance;
    uint32_t primitiveIndex, instanceIndex, instanceCustomIndex, instanceSBTOffset;
    f3 barycentric;
    m44 transform;
} HitData;

typedef struct {
    f3 color;
    int hit;
    f3 nextFactor, nextRayOrigin, nextRayDirection;
} Payload;

typedef struct {
    const OrcBindings* b;
    int depth;
    uint32_t frameID;
    uint32_t debug;
    uint32_t global_id;     /* get_global_id(0) */
} SceneData;

static __thread OrcCounters* t_ctr = NULL;
static __thread int t_cls = 0;          /* 0 = radiance ray, 1 = shadow ray (sbtRecordOffset - 1) */

static void callHit(int sbtRecordOffset, Payload* payload, HitData* hitData, SceneData* sceneData);
static void callMiss(int missIndex, Payload* payload, SceneData* sceneData);
static void callAnyHit(int* cont, int sbtRecordOffset, Payload* payload, HitData* hitData, SceneData* sceneData);

/* radiance.cl:195-208 */
static int intersectAABB(f3 rayOrigin, f3 rayDir, f3 boxMin, f3 boxMax)
{
    f3 tMin = f3_div(f3_sub(boxMin, rayOrigin), rayDir);
    f3 tMax = f3_div(f3_sub(boxMax, rayOrigin), rayDir);
    f3 t1 = cl_min3(tMin, tMax);
    f3 t2 = cl_max3(tMin, tMax);
    float tNear = cl_max(cl_max(t1.x, t1.y), t1.z);
    float tFar = cl_min(cl_min(t2.x, t2.y), t2.z);
    if (tFar > cl_max(tNear, 0.0f))
        return 1;
    return 0;
}

/* radiance.cl:211-251 */
static int intersectTriangle(f3 origin, f3 direction, f3 v0, f3 v1, f3 v2,
                             f3* intersectPoint, float* distance, f3* bary)
{
    f3 edge1 = f3_sub(v1, v0);
    f3 edge2 = f3_sub(v2, v0);
    f3 ray_cross_e2 = cl_cross(direction, edge2);
    float det = cl_dot3(edge1, ray_cross_e2);
    if (det == 0)
        return 0;
    float inv_det = 1.0f / det;
    f3 s = f3_sub(origin, v0);
    float b1 = inv_det * cl_dot3(s, ray_cross_e2);
    f3 s_cross_e1 = cl_cross(s, edge1);
    float b2 = inv_det * cl_dot3(direction, s_cross_e1);
    float t = inv_det * cl_dot3(edge2, s_cross_e1);
    if (b1 < 0 || b1 > 1)
        return 0;
    if (b2 < 0 || b1 + b2 > 1)
        return 0;
    if (t > 0) {
        *distance = t;
        *intersectPoint = f3_add(origin, f3_scale(direction, t));
        bary->x = 1 - b1 - b2;
        bary->y = b1;
        bary->z = b2;
        return 1;
    }
    return 0;
}

static inline f3 ld3(const float* p) { return F3(p[0], p[1], p[2]); }

/* radiance.cl:41-108 */
static int intersectBot(const uint8_t* accelStruct, f3 origin, f3 direction, float Tmin, float Tmax,
                        HitData* hitData, int* cont, int sbtRecordOffset, Payload* payload,
                        SceneData* sceneData)
{
    const OrcAccelBot* hdr = (const OrcAccelBot*)accelStruct;
    int hasIntersected = 0;
    uint32_t stack[BVH_BOT_STACK_SIZE + 2];
    int stackIdx = 0;
    stack[stackIdx++] = 0;

    while (stackIdx) {
        uint32_t nodeIdx = stack[stackIdx - 1];
        stackIdx--;
        const OrcNode* nodeList = (const OrcNode*)(accelStruct + hdr->nodeByteOffset);
        const OrcNode* node = nodeList + nodeIdx;
        if (t_ctr) t_ctr->bot_nodes[t_cls]++;

        if (!(node->w0 & 0x80000000u)) {
            if (intersectAABB(origin, direction, ld3(node->bottom), ld3(node->top))) {
                stack[stackIdx++] = node->w1; /* right */
                stack[stackIdx++] = node->w0; /* left */
                if (stackIdx > BVH_BOT_STACK_SIZE) {
                    printf("ERROR: Bottom AS stack overflow\n");
                    return 0;
                }
            }
        } else if (node->w2 == TYPE_TRIG) {
            const float* vertexList = (const float*)(accelStruct + hdr->vertexOffset);
            const OrcTri* faceList = (const OrcTri*)(accelStruct + hdr->faceByteOffset);
            uint32_t count = node->w0 & 0x7fffffffu;
            for (uint32_t i = 0; i < count; i++) {
                const OrcTri* face = &faceList[node->w1 + i];
                f3 intersectPoint, bary;
                float distance;
                if (t_ctr) t_ctr->tri_tests[t_cls]++;
                if (intersectTriangle(origin, direction, ld3(vertexList + 4 * face->idx0),
                                      ld3(vertexList + 4 * face->idx1), ld3(vertexList + 4 * face->idx2),
                                      &intersectPoint, &distance, &bary) &&
                    distance < hitData->distance && distance > Tmin && distance < Tmax) {
                    hitData->distance = distance;
                    hitData->hitPoint = intersectPoint;
                    hitData->primitiveIndex = face->primID;
                    hitData->barycentric = bary;
                    hasIntersected = 1;
                    callAnyHit(cont, sbtRecordOffset, payload, hitData, sceneData);
                    if (*cont == 0)
                        return hasIntersected;
                }
            }
        }
    }
    return hasIntersected;
}

/* radiance.cl:110-192 */
static int intersectTop(const uint8_t* accelStruct, f3 origin, f3 direction, float Tmin, float Tmax,
                        HitData* hitData, int sbtRecordOffset, Payload* payload, SceneData* sceneData)
{
    const OrcAccelTop* hdr = (const OrcAccelTop*)accelStruct;
    int hasIntersected = 0;
    uint32_t stack[BVH_TOP_STACK_SIZE + 2];
    int stackIdx = 0;
    stack[stackIdx++] = 0;
    int cont = 1;

    while (stackIdx) {
        uint32_t nodeIdx = stack[stackIdx - 1];
        stackIdx--;
        const OrcNode* nodeList = (const OrcNode*)(accelStruct + hdr->nodeByteOffset);
        const OrcNode* node = nodeList + nodeIdx;
        if (t_ctr) t_ctr->top_nodes[t_cls]++;

        if (!(node->w0 & 0x80000000u)) {
            if (intersectAABB(origin, direction, ld3(node->bottom), ld3(node->top))) {
                stack[stackIdx++] = node->w1;
                stack[stackIdx++] = node->w0;
                if (stackIdx > BVH_TOP_STACK_SIZE) {
                    printf("ERROR: Top AS stack overflow\n");
                    return 0;
                }
            }
        } else if (node->w2 == TYPE_INST) {
            const OrcInst* instanceList = (const OrcInst*)(accelStruct + hdr->instByteOffset);
            uint32_t count = node->w0 & 0x7fffffffu;
            for (uint32_t i = 0; i < count; i++) {
                const OrcInst* instance = &instanceList[node->w1 + i];
                const uint8_t* botAccelStruct = accelStruct + instance->instanceOffset;
                if (t_ctr) t_ctr->inst_visits[t_cls]++;

                m44 transform = hitData->transform;
                uint32_t instanceIndex = hitData->instanceIndex;
                uint32_t instanceCustomIndex = hitData->instanceCustomIndex;
                uint32_t instanceSBTOffset = hitData->instanceSBTOffset;

                f4 rayPos = {origin.x, origin.y, origin.z, 1.0f};
                f4 rayDir = {direction.x, direction.y, direction.z, 0.0f};
                f4 localOrigin, localDir;
                m44 inverse;
                memset(&inverse, 0, sizeof inverse); /* reference: uninitialised when det == 0 */

                memcpy(hitData->transform.s, instance->r, sizeof(float) * 16);
                InverseMat4x4(&hitData->transform, &inverse);
                MultiplyMat4Vec4(&inverse, &rayPos, &localOrigin);
                MultiplyMat4Vec4(&inverse, &rayDir, &localDir);

                hitData->instanceIndex = instance->instanceID;
                hitData->instanceCustomIndex = instance->customInstanceID;
                hitData->instanceSBTOffset = instance->SBTOffset;

                int result = intersectBot(botAccelStruct, F3(localOrigin.x, localOrigin.y, localOrigin.z),
                                          F3(localDir.x, localDir.y, localDir.z), Tmin, Tmax, hitData, &cont,
                                          sbtRecordOffset, payload, sceneData);
                hasIntersected = hasIntersected || result;
                if (cont == 0)
                    return hasIntersected;
                if (!result) {
                    hitData->transform = transform;
                    hitData->instanceIndex = instanceIndex;
                    hitData->instanceCustomIndex = instanceCustomIndex;
                    hitData->instanceSBTOffset = instanceSBTOffset;
                }
            }
        }
    }
    return hasIntersected;
}

static void hitdata_init(HitData* h)
{
    memset(h, 0, sizeof *h); /* reference leaves everything but distance uninitialised */
    h->distance = FLT_MAX;
}

/* radiance.cl:254-275 */
static void traceRay(const void* topLevel, int sbtRecordOffset, int missIndex, f3 origin, f3 direction,
                     float Tmin, float Tmax, Payload* payload, SceneData* sceneData)
{
    HitData hitData;
    hitdata_init(&hitData);
    int saved_cls = t_cls;
    t_cls = sbtRecordOffset == 2 ? 1 : 0;
    if (t_ctr) t_ctr->rays[t_cls]++;
    int hit = intersectTop((const uint8_t*)topLevel, origin, direction, Tmin, Tmax, &hitData, sbtRecordOffset,
                           payload, sceneData);
    t_cls = saved_cls;
    if (hit)
        callHit(sbtRecordOffset, payload, &hitData, sceneData);
    else
        callMiss(missIndex, payload, sceneData);
}

/* ------------------------------------------------------------------------------------------ */
/* samples/shader.cl                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* shader.cl:47-56 */
static f3 aces_approx(f3 v)
{
    v = f3_scale(v, 0.6f);
    float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    f3 num = f3_mul(v, F3(a * v.x + b, a * v.y + b, a * v.z + b));
    f3 den = f3_mul(v, F3(c * v.x + d, c * v.y + d, c * v.z + d));
    den = F3(den.x + e, den.y + e, den.z + e);
    return cl_clamp3(f3_div(num, den), 0.0f, 1.0f);
}

/* shader.cl:89-109 */
static void sampleUniformDisk(float ux, float uy, float* ox, float* oy)
{
    float offx = 2.0f * ux - 1.0f, offy = 2.0f * uy - 1.0f;
    if (offx == 0.0f && offy == 0.0f) { *ox = 0.0f; *oy = 0.0f; return; }
    float theta, r;
    if (fabsf(offx) > fabsf(offy)) {
        r = offx;
        theta = (PI / 4.0f) * (offy / offx);
    } else {
        r = offy;
        theta = (PI / 2.0f) - (PI / 4.0f) * (offx / offy);
    }
    *ox = r * cosf(theta);
    *oy = r * sinf(theta);
}

/* shader.cl:111-173 */
static void generateRay(const OrcCamera* cam, int index, uint32_t r0, uint32_t r1, uint32_t r2,
                        f3* position, f3* direction)
{
    const int x = index % (int)cam->widthPixel;
    const int y = index / (int)cam->widthPixel;
    f3 random = random_pcg3d(r0, r1, r2);

    float fx = (((float)x + random.x) / cam->widthPixel) - 0.5f;
    float fy = 0.5f - (((float)y + random.y) / cam->heightPixel);
    float aspectRatio = cam->heightPixel / cam->widthPixel;
    f4 pinholeDirection = {fx * cam->sensorWidth, fy * cam->sensorWidth * aspectRatio, -cam->focalLength, 0.0f};
    pinholeDirection = cl_normalize4(pinholeDirection);
    f3 pinholeOrigin = {cam->x, cam->y, cam->z};
    float time = -cam->focalDistance / pinholeDirection.z;

    m44 rotX, rotY, rotZ;
    f4 tmp;
    EulerXToMat4x4(cam->wx, &rotX);
    EulerYToMat4x4(cam->wy, &rotY);
    EulerZToMat4x4(cam->wz, &rotZ);
    MultiplyMat4Vec4(&rotZ, &pinholeDirection, &tmp);
    MultiplyMat4Vec4(&rotY, &tmp, &pinholeDirection);
    MultiplyMat4Vec4(&rotX, &pinholeDirection, &tmp);
    pinholeDirection = cl_normalize4(tmp);

    if (cam->fStop == 0.0f) {
        *position = pinholeOrigin;
        *direction = F3(pinholeDirection.x, pinholeDirection.y, pinholeDirection.z);
        return;
    }

    float lensRadius = (cam->focalLength / cam->fStop) / 2.0f;
    float lx, ly;
    sampleUniformDisk(random.y, random.z, &lx, &ly);
    lx = lensRadius * lx; ly = lensRadius * ly;

    f3 pd = F3(pinholeDirection.x, pinholeDirection.y, pinholeDirection.z);
    f3 hitPoint = f3_add(pinholeOrigin, f3_scale(pd, time));

    f4 lensOrigin = {lx, ly, 0.0f, 1.0f};
    MultiplyMat4Vec4(&rotZ, &lensOrigin, &tmp);
    MultiplyMat4Vec4(&rotY, &tmp, &lensOrigin);
    MultiplyMat4Vec4(&rotX, &lensOrigin, &tmp);
    f3 lo = f3_add(pinholeOrigin, F3(tmp.x, tmp.y, tmp.z));

    *position = lo;
    *direction = cl_normalize3(f3_sub(hitPoint, lo));
}

/* shader.cl:308-321 */
static void getIndices(SceneData* sd, HitData* hd, uint32_t* i0, uint32_t* i1, uint32_t* i2)
{
    const OrcMeshInfo* meshInfo = &sd->b->meshInfoData[hd->instanceIndex];
    int io = meshInfo->indexOffset;
    const uint32_t* indexData = sd->b->indexData;
    *i0 = indexData[io + hd->primitiveIndex * 3 + 0];
    *i1 = indexData[io + hd->primitiveIndex * 3 + 1];
    *i2 = indexData[io + hd->primitiveIndex * 3 + 2];
}

/* shader.cl:323-338 (result only consumed by texture lookups, which are stubbed to 0) */
static void getUV(SceneData* sd, HitData* hd, float* u, float* v)
{
    const OrcMeshInfo* meshInfo = &sd->b->meshInfoData[hd->instanceIndex];
    int uo = meshInfo->uvOffset;
    const float* uvData = sd->b->uvData;
    uint32_t ix, iy, iz;
    getIndices(sd, hd, &ix, &iy, &iz);
    float u0 = uvData[uo + ix * 3 + 0], v0 = uvData[uo + ix * 3 + 1];
    float u1 = uvData[uo + iy * 3 + 0], v1 = uvData[uo + iy * 3 + 1];
    float u2 = uvData[uo + iz * 3 + 0], v2 = uvData[uo + iz * 3 + 1];
    *u = hd->barycentric.x * u0 + hd->barycentric.y * u1 + hd->barycentric.z * u2;
    *v = hd->barycentric.x * v0 + hd->barycentric.y * v1 + hd->barycentric.z * v2;
}

/* read_imageui(imageArray, sampler, (float4)(u, v, layer, 0)) on a CL_RGBA / CL_UNSIGNED_INT8 2D image array, normalized
 * coordinates: OpenCL 1.2 specification 8.2 (addressing) and 5.3.3 (layer = clamp(rint(layer), 0, layers-1)).  The spec leaves
 * CLK_FILTER_LINEAR undefined for integer reads; defined here (and in the product) as the bilinear weights of 8.2 on the 8-bit
 * values, rounded to nearest.  The live reference shader has these reads commented out: "parity unpinned". */
static int tex_addr(float s, int n, uint32_t mode, float* u)
{
    if (mode == 0) { *u = (s - floorf(s)) * (float)n; int i = (int)floorf(*u); return i > n - 1 ? i - n : i; }
    if (mode == 3) { float sp = 2.0f * rintf(0.5f * s); sp = fabsf(s - sp); *u = sp * (float)n; int i = (int)floorf(*u); return i > n - 1 ? n - 1 : i; }
    *u = s * (float)n;
    int i = (int)floorf(*u);
    if (mode == 1) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    return (i < 0 || i > n - 1) ? -1 : i;
}
static int tex_wrap(int i, int n, uint32_t mode)
{
    if (mode == 0) return i < 0 ? i + n : (i > n - 1 ? i - n : i);
    if (mode == 3 || mode == 1) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    return (i < 0 || i > n - 1) ? -1 : i;
}
static void tex_texel(const OrcBindings* b, int layer, int x, int y, float* out)
{
    if (x < 0 || y < 0) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
    const uint8_t* t = b->texData + (((size_t)layer * b->texH + (uint32_t)y) * b->texW + (uint32_t)x) * 4;
    out[0] = (float)t[0]; out[1] = (float)t[1]; out[2] = (float)t[2]; out[3] = (float)t[3];
}
static void read_imageui(const OrcBindings* b, float u, float v, float layerF, uint32_t* out)
{
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!(b->texFlags & 1u) || !b->texData) return;          /* live reference shader: tex = 0 */
    const uint32_t mode = (b->texFlags >> 4) & 3u;
    int layer = (int)rintf(layerF);
    layer = layer < 0 ? 0 : (layer > (int)b->texLayers - 1 ? (int)b->texLayers - 1 : layer);
    float uu, vv, c[4];
    const int ix = tex_addr(u, (int)b->texW, mode, &uu), iy = tex_addr(v, (int)b->texH, mode, &vv);
    if (!(b->texFlags & 2u)) {
        tex_texel(b, layer, ix, iy, c);
        for (int k = 0; k < 4; ++k) out[k] = (uint32_t)c[k];
        return;
    }
    const float fu = uu - 0.5f, fv = vv - 0.5f;
    const int i0 = (int)floorf(fu), j0 = (int)floorf(fv);
    const float a = fu - floorf(fu), bb = fv - floorf(fv);
    const int x0 = tex_wrap(i0, (int)b->texW, mode), x1 = tex_wrap(i0 + 1, (int)b->texW, mode);
    const int y0 = tex_wrap(j0, (int)b->texH, mode), y1 = tex_wrap(j0 + 1, (int)b->texH, mode);
    float t00[4], t10[4], t01[4], t11[4];
    tex_texel(b, layer, x0, y0, t00); tex_texel(b, layer, x1, y0, t10); tex_texel(b, layer, x0, y1, t01); tex_texel(b, layer, x1, y1, t11);
    for (int k = 0; k < 4; ++k)
        out[k] = (uint32_t)((1.0f - a) * (1.0f - bb) * t00[k] + a * (1.0f - bb) * t10[k] + (1.0f - a) * bb * t01[k] + a * bb * t11[k] + 0.5f);
}

/* shader.cl:340-368 */
static f3 getFaceNormal(SceneData* sd, HitData* hd)
{
    const OrcMeshInfo* meshInfo = &sd->b->meshInfoData[hd->instanceIndex];
    int no = meshInfo->normalOffset;
    const float* nd = sd->b->normalData;
    uint32_t ix, iy, iz;
    getIndices(sd, hd, &ix, &iy, &iz);
    f4 n0 = {nd[no + ix * 3 + 0], nd[no + ix * 3 + 1], nd[no + ix * 3 + 2], 0.0f};
    f4 n1 = {nd[no + iy * 3 + 0], nd[no + iy * 3 + 1], nd[no + iy * 3 + 2], 0.0f};
    f4 n2 = {nd[no + iz * 3 + 0], nd[no + iz * 3 + 1], nd[no + iz * 3 + 2], 0.0f};
    float bx = hd->barycentric.x, by = hd->barycentric.y, bz = hd->barycentric.z;
    f4 normal = {bx * n0.x + by * n1.x + bz * n2.x, bx * n0.y + by * n1.y + bz * n2.y,
                 bx * n0.z + by * n1.z + bz * n2.z, bx * n0.w + by * n1.w + bz * n2.w};
    f4 tmp;
    MultiplyMat4Vec4(&hd->transform, &normal, &tmp);
    return cl_normalize3(F3(tmp.x, tmp.y, tmp.z));
}

/* shader.cl:370-396 ; read_imageui is commented out in the reference => tex == 0 unless texFlags enables the read */
static f3 getMatNormal(SceneData* sd, HitData* hd, f3 faceNormal)
{
    const OrcMeshInfo* meshInfo = &sd->b->meshInfoData[hd->instanceIndex];
    const OrcMaterial* material = &sd->b->materials[meshInfo->materialIndex];
    if (material->normalTexIdx != -1) {
        float u, v;
        uint32_t tex[4];
        getUV(sd, hd, &u, &v);
        read_imageui(sd->b, u, 1.0f - v, (float)material->normalTexIdx, tex);
        float texx = (float)tex[0], texy = (float)tex[1], texz = (float)tex[2];
        f4 localNormal = {cl_clamp(texx / 255.0f, 0.0f, 1.0f), cl_clamp(texy / 255.0f, 0.0f, 1.0f),
                          cl_clamp(texz / 255.0f, 0.0f, 1.0f), 0.0f};
        f4 ln2 = {localNormal.x * 2.0f - 1.0f, localNormal.y * 2.0f - 1.0f, localNormal.z * 2.0f - 1.0f,
                  localNormal.w * 2.0f - 1.0f};
        localNormal = cl_normalize4(ln2);
        m44 transform;
        GetNormalSpace(faceNormal, &transform);
        f4 globalNormal;
        MultiplyMat4Vec4(&transform, &localNormal, &globalNormal);
        faceNormal = cl_normalize3(F3(globalNormal.x, globalNormal.y, globalNormal.z));
    }
    return faceNormal;
}

/* shader.cl:399-431 */
static f4 getMaterialProp(SceneData* sd, HitData* hd)
{
    const OrcMeshInfo* meshInfo = &sd->b->meshInfoData[hd->instanceIndex];
    const OrcMaterial* material = &sd->b->materials[meshInfo->materialIndex];
    float u, v;
    getUV(sd, hd, &u, &v);
    float metallicFrag;
    if (material->metallicTexIdx == -1)
        metallicFrag = material->metallic;
    else {
        uint32_t tex[4];
        read_imageui(sd->b, u, 1.0f - v, (float)material->metallicTexIdx, tex);
        metallicFrag = cl_clamp((float)tex[2] / 255.0f, 0.0f, 1.0f);
    }
    float roughnessFrag;
    if (material->roughnessTexIdx == -1)
        roughnessFrag = cl_clamp(material->roughness, 0.0f, 1.0f);
    else {
        uint32_t tex[4];
        read_imageui(sd->b, u, 1.0f - v, (float)material->roughnessTexIdx, tex);
        roughnessFrag = cl_clamp((float)tex[1] / 255.0f, 0.05f, 1.0f);
    }
    float transFrag = cl_clamp(material->transmission, 0.0f, 1.0f);
    float iorFrag = cl_clamp(material->ior, 0.0f, 10.0f);
    f4 r = {metallicFrag, roughnessFrag, transFrag, iorFrag};
    return r;
}

/* shader.cl:433-452 */
static f3 getAlbedo(SceneData* sd, HitData* hd)
{
    const OrcMeshInfo* meshInfo = &sd->b->meshInfoData[hd->instanceIndex];
    const OrcMaterial* material = &sd->b->materials[meshInfo->materialIndex];
    if (material->albedoTexIdx == -1)
        return F3(material->albedo[0], material->albedo[1], material->albedo[2]);
    float u, v;
    uint32_t tex[4];
    getUV(sd, hd, &u, &v);
    read_imageui(sd->b, u, 1.0f - v, (float)material->albedoTexIdx, tex);
    return F3(cl_clamp((float)tex[0] / 255.0f, 0.0f, 1.0f), cl_clamp((float)tex[1] / 255.0f, 0.0f, 1.0f), cl_clamp((float)tex[2] / 255.0f, 0.0f, 1.0f));
}

/* shader.cl:454-469 */
static f3 getHitPosition(HitData* hd, f3 N)
{
    f4 tmp0;
    f4 tmp1 = {hd->hitPoint.x, hd->hitPoint.y, hd->hitPoint.z, 1.0f};
    MultiplyMat4Vec4(&hd->transform, &tmp1, &tmp0);
    return f3_add(F3(tmp0.x, tmp0.y, tmp0.z), f3_scale(N, 0.00001f));
}

/* shader.cl:471-476 */
static f3 getLightDirection(SceneData* sd)
{
    const float* d = sd->b->scene->lights[0].direction;
    return cl_normalize3(F3(-d[0], -d[1], -d[2]));
}

/* shader.cl:482-541 */
static void material(Payload* payload, HitData* hitData, SceneData* sceneData)
{
    payload->hit = 1;
    if (t_ctr) t_ctr->hits++;

    f3 faceN = getFaceNormal(sceneData, hitData);
    f3 hitPos = getHitPosition(hitData, faceN);
    f3 N = getMatNormal(sceneData, hitData, faceN);
    f3 L = getLightDirection(sceneData);
    f3 V = cl_normalize3(f3_neg(payload->nextRayDirection));    /* getViewDirection :478-481 */

    f4 mat = getMaterialProp(sceneData, hitData);
    f3 albedo = getAlbedo(sceneData, hitData);

    Payload shadowPayload;
    memset(&shadowPayload, 0, sizeof shadowPayload);
    if (t_ctr) t_ctr->shadow++;
    traceRay(sceneData->b->topLevel, 2, 4, hitPos, L, 0.001f, 1000, &shadowPayload, sceneData);

    f3 color = {0.0f, 0.0f, 0.0f};
    if (!shadowPayload.hit) {
        const float* lc = sceneData->b->scene->lights[0].color;
        f3 radiance = F3(lc[0], lc[1], lc[2]);
        color = f3_add(color, f3_mul(microfacetBRDF(L, V, N, albedo, mat.x, mat.y, mat.z, mat.w), radiance));
    }
    color = f3_add(color, f3_scale(albedo, 0.1f));
    payload->color = color;

    f3 random = random_pcg3d(sceneData->frameID, sceneData->global_id, (uint32_t)sceneData->depth);
    f3 nextFactor = {0.0f, 0.0f, 0.0f};
    f3 nextDir = sampleMicrofacetBRDF_transm(V, N, albedo, mat.x, mat.y, mat.z, mat.w, random, &nextFactor);
    if (cl_dot3(nextDir, N) < 0)
        hitPos = getHitPosition(hitData, f3_neg(faceN));

    payload->nextRayOrigin = hitPos;
    payload->nextRayDirection = nextDir;
    payload->nextFactor = nextFactor;
}

/* shader.cl:543-572 */
static void shadowMiss(Payload* p) { p->hit = 0; p->color = f3_splat(1.0f); }
static void environment(Payload* p) { p->hit = 0; p->color = F3(0.2f, 0.2f, 0.5f); }
static void shadow(Payload* p) { p->hit = 1; p->color = f3_splat(0.0f); }
static void anyShadow(int* cont) { *cont = 0; }

/* shader.cl:574-605 : the SBT. Row index = samples/sbt.json row. */
static void callAnyHit(int* cont, int sbtRecordOffset, Payload* payload, HitData* hitData, SceneData* sceneData)
{
    (void)payload; (void)sceneData;
    int index = (int)(hitData->instanceSBTOffset + (uint32_t)sbtRecordOffset);
    switch (index) {
    case 2: anyShadow(cont); break;
    }
}
static void callHit(int sbtRecordOffset, Payload* payload, HitData* hitData, SceneData* sceneData)
{
    int index = (int)(hitData->instanceSBTOffset + (uint32_t)sbtRecordOffset);
    switch (index) {
    case 1: material(payload, hitData, sceneData); break;
    case 2: shadow(payload); break;
    }
}
static void callMiss(int missIndex, Payload* payload, SceneData* sceneData)
{
    (void)sceneData;
    switch (missIndex) {
    case 3: environment(payload); break;
    case 4: shadowMiss(payload); break;
    }
}

/* shader.cl:175-305 : one work-item of the `raygen` kernel */
static void raygen(const OrcBindings* b, int index)
{
    const OrcRTProp* RTProp = b->RTProp;
    float* imageScratch = b->imageScratch;
    uint8_t* image = b->image;
    const int CHANNEL = 4;

    int iteration = (int)RTProp->batchSize;
    uint32_t frameID = RTProp->totalSamples;
    while (iteration > 0) {
        iteration--;
        f3 rayOrigin, rayDirection;
        generateRay(b->camData, index, frameID, RTProp->totalSamples, (uint32_t)index, &rayOrigin, &rayDirection);

        Payload payload;
        memset(&payload, 0, sizeof payload);   /* payload.hit is uninitialised in the reference */
        payload.color = f3_splat(0.0f);
        payload.nextFactor = f3_splat(1.0f);
        payload.nextRayOrigin = rayOrigin;
        payload.nextRayDirection = rayDirection;

        SceneData sceneData;
        sceneData.b = b;
        sceneData.depth = 0;
        sceneData.frameID = frameID;
        sceneData.debug = RTProp->debug;
        sceneData.global_id = (uint32_t)index;

        f3 color = f3_splat(0.0f);
        f3 contribution = f3_splat(1.0f);
        while (sceneData.depth < (int)RTProp->depth) {
            if (t_ctr) { if (sceneData.depth == 0) t_ctr->primary++; else t_ctr->bounce++; }
            traceRay(b->topLevel, 1, 3, payload.nextRayOrigin, payload.nextRayDirection, 0.001f, 1000,
                     &payload, &sceneData);
            if (payload.hit) {
                color = f3_add(color, f3_mul(contribution, payload.color));
                contribution = f3_mul(contribution, payload.nextFactor);
            } else if (sceneData.depth == 0) {
                color = payload.color;
            } else {
                break;
            }
            sceneData.depth++;
            payload.hit = 0;
            if (RTProp->debug)
                break;
        }

        if (frameID == 0) {
            imageScratch[CHANNEL * index + 0] = color.x;
            imageScratch[CHANNEL * index + 1] = color.y;
            imageScratch[CHANNEL * index + 2] = color.z;
        } else {
            float pixel;
            pixel = imageScratch[CHANNEL * index + 0];
            imageScratch[CHANNEL * index + 0] = (frameID * pixel + color.x) / (frameID + 1);
            pixel = imageScratch[CHANNEL * index + 1];
            imageScratch[CHANNEL * index + 1] = (frameID * pixel + color.y) / (frameID + 1);
            pixel = imageScratch[CHANNEL * index + 2];
            imageScratch[CHANNEL * index + 2] = (frameID * pixel + color.z) / (frameID + 1);
        }
        frameID++;
    }

    f3 color = {imageScratch[CHANNEL * index + 0], imageScratch[CHANNEL * index + 1],
                imageScratch[CHANNEL * index + 2]};
    if (!RTProp->debug) {
        color = aces_approx(color);
        color = F3(powf(color.x, 0.7f), powf(color.y, 0.7f), powf(color.z, 0.7f));
    }
    image[CHANNEL * index + 0] = (uint8_t)(int)(color.x * 255);
    image[CHANNEL * index + 1] = (uint8_t)(int)(color.y * 255);
    image[CHANNEL * index + 2] = (uint8_t)(int)(color.z * 255);
    image[CHANNEL * index + 3] = 255;
}

/* ------------------------------------------------------------------------------------------ */
/* bvh.cpp : binned-SAH builder                                                                */
/* ------------------------------------------------------------------------------------------ */
#define MAX_LEAF_PRIM_SIZE 8

typedef struct { f3 bottom, top, center; uint32_t prim; } BBoxTmp;   /* bvh.cpp:24-40 */

typedef struct BNode {
    f3 bottom, top;
    int leaf;
    struct BNode *left, *right;
    uint32_t* prims; uint32_t nprims;
} BNode;

/* linalg.h:28-40 */
static inline f3 minVec3(f3 v0, f3 v1) { return F3(v0.x < v1.x ? v0.x : v1.x, v0.y < v1.y ? v0.y : v1.y, v0.z < v1.z ? v0.z : v1.z); }
static inline f3 maxVec3(f3 v0, f3 v1) { return F3(v0.x > v1.x ? v0.x : v1.x, v0.y > v1.y ? v0.y : v1.y, v0.z > v1.z ? v0.z : v1.z); }
static inline float axisv(f3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

static BNode* make_leaf(const BBoxTmp* work, size_t n)
{
    BNode* l = (BNode*)calloc(1, sizeof(BNode));
    l->leaf = 1;
    l->nprims = (uint32_t)n;
    l->prims = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) l->prims[i] = work[i].prim;
    return l;
}

/* bvh.cpp:46-285 */
static BNode* Recurse(BBoxTmp* work, size_t n, int depth)
{
    if (n < MAX_LEAF_PRIM_SIZE)
        return make_leaf(work, n);

    f3 bottom = F3(FLT_MAX, FLT_MAX, FLT_MAX), top = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (size_t i = 0; i < n; i++) {
        bottom = minVec3(bottom, work[i].bottom);
        top = maxVec3(top, work[i].top);
    }
    float side1 = top.x - bottom.x, side2 = top.y - bottom.y, side3 = top.z - bottom.z;
    float minCost = n * (side1 * side2 + side2 * side3 + side3 * side1);
    float bestSplit = FLT_MAX;
    int bestAxis = -1;

    for (int axis = 0; axis < 3; axis++) {
        float start = axisv(bottom, axis), stop = axisv(top, axis), step;
        if (fabsf(stop - start) < 1e-4)
            continue;
        step = (stop - start) / (1024. / (depth + 1.));
        for (float testSplit = start + step; testSplit < stop - step; testSplit += step) {
            f3 lbottom = F3(FLT_MAX, FLT_MAX, FLT_MAX), ltop = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
            f3 rbottom = lbottom, rtop = ltop;
            int countLeft = 0, countRight = 0;
            for (size_t i = 0; i < n; i++) {
                const BBoxTmp* v = &work[i];
                float value = axisv(v->center, axis);
                if (value < testSplit) {
                    lbottom = minVec3(lbottom, v->bottom);
                    ltop = maxVec3(ltop, v->top);
                    countLeft++;
                } else {
                    rbottom = minVec3(rbottom, v->bottom);
                    rtop = maxVec3(rtop, v->top);
                    countRight++;
                }
            }
            if (countLeft <= 1 || countRight <= 1) {
                if (testSplit + step == testSplit) { fprintf(stderr, "oracle: reference split loop would not terminate\n"); abort(); }
                continue;
            }
            float lside1 = ltop.x - lbottom.x, lside2 = ltop.y - lbottom.y, lside3 = ltop.z - lbottom.z;
            float rside1 = rtop.x - rbottom.x, rside2 = rtop.y - rbottom.y, rside3 = rtop.z - rbottom.z;
            float surfaceLeft = lside1 * lside2 + lside2 * lside3 + lside3 * lside1;
            float surfaceRight = rside1 * rside2 + rside2 * rside3 + rside3 * rside1;
            float totalCost = surfaceLeft * countLeft + surfaceRight * countRight;
            if (totalCost < minCost) {
                minCost = totalCost;
                bestSplit = testSplit;
                bestAxis = axis;
            }
            if (testSplit + step == testSplit) { fprintf(stderr, "oracle: reference split loop would not terminate\n"); abort(); }
        }
    }

    if (bestAxis == -1)
        return make_leaf(work, n);

    BBoxTmp* left = (BBoxTmp*)malloc(sizeof(BBoxTmp) * n);
    BBoxTmp* right = (BBoxTmp*)malloc(sizeof(BBoxTmp) * n);
    size_t nl = 0, nr = 0;
    f3 lbottom = F3(FLT_MAX, FLT_MAX, FLT_MAX), ltop = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    f3 rbottom = lbottom, rtop = ltop;
    for (size_t i = 0; i < n; i++) {
        const BBoxTmp* v = &work[i];
        float value = axisv(v->center, bestAxis);
        if (value < bestSplit) {
            left[nl++] = *v;
            lbottom = minVec3(lbottom, v->bottom);
            ltop = maxVec3(ltop, v->top);
        } else {
            right[nr++] = *v;
            rbottom = minVec3(rbottom, v->bottom);
            rtop = maxVec3(rtop, v->top);
        }
    }
    BNode* inner = (BNode*)calloc(1, sizeof(BNode));
    inner->left = Recurse(left, nl, depth + 1);
    inner->left->bottom = lbottom; inner->left->top = ltop;
    inner->right = Recurse(right, nr, depth + 1);
    inner->right->bottom = rbottom; inner->right->top = rtop;
    free(left); free(right);
    return inner;
}

static void free_tree(BNode* n)
{
    if (!n) return;
    if (n->leaf) free(n->prims);
    else { free_tree(n->left); free_tree(n->right); }
    free(n);
}
static uint32_t CountBoxes(const BNode* r) { return r->leaf ? 1 : 1 + CountBoxes(r->left) + CountBoxes(r->right); }   /* bvh.cpp:426-434 */
static void CountDepth(const BNode* r, int depth, int* maxDepth)                                                    /* bvh.cpp:450-459 */
{
    if (*maxDepth < depth) *maxDepth = depth;
    if (!r->leaf) { CountDepth(r->left, depth + 1, maxDepth); CountDepth(r->right, depth + 1, maxDepth); }
}

static void node_set_box(OrcNode* n, const BNode* r)
{
    n->bottom[0] = r->bottom.x; n->bottom[1] = r->bottom.y; n->bottom[2] = r->bottom.z; n->bottom[3] = 0;
    n->top[0] = r->top.x; n->top[1] = r->top.y; n->top[2] = r->top.z; n->top[3] = 0;
}

typedef struct { uint8_t* data; uint32_t size; int maxDepth; } OrcBlas;

/* bvh.cpp:463-500 */
static void PopulateTri(const uint32_t* tris, const BNode* root, uint32_t* faceIdx, uint32_t* nodeIdx,
                        OrcTri* faceList, OrcNode* nodeList)
{
    uint32_t curr = *nodeIdx;
    node_set_box(&nodeList[curr], root);
    if (!root->leaf) {
        uint32_t idxLeft = ++(*nodeIdx);
        PopulateTri(tris, root->left, faceIdx, nodeIdx, faceList, nodeList);
        uint32_t idxRight = ++(*nodeIdx);
        PopulateTri(tris, root->right, faceIdx, nodeIdx, faceList, nodeList);
        nodeList[curr].w0 = idxLeft;
        nodeList[curr].w1 = idxRight;
    } else {
        nodeList[curr].w0 = 0x80000000u | root->nprims;
        nodeList[curr].w1 = *faceIdx;
        nodeList[curr].w2 = TYPE_TRIG;
        for (uint32_t i = 0; i < root->nprims; i++) {
            uint32_t p = root->prims[i];
            faceList[*faceIdx].primID = p;
            faceList[*faceIdx].idx0 = tris[3 * p + 0];
            faceList[*faceIdx].idx1 = tris[3 * p + 1];
            faceList[*faceIdx].idx2 = tris[3 * p + 2];
            (*faceIdx)++;
        }
    }
}

/* bvh.cpp:288-341 CreateBVH(vertices, triangles) + bvh.cpp:502-522 + radiance.cpp:318-364 */
void* orc_blas_build(const float* verts, uint32_t nverts, const uint32_t* tris, uint32_t ntris)
{
    BBoxTmp* work = (BBoxTmp*)malloc(sizeof(BBoxTmp) * (ntris ? ntris : 1));
    f3 bottom = F3(FLT_MAX, FLT_MAX, FLT_MAX), top = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (uint32_t j = 0; j < ntris; j++) {
        BBoxTmp b;
        b.bottom = F3(FLT_MAX, FLT_MAX, FLT_MAX);
        b.top = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        b.prim = j;
        f3 v0 = ld3(verts + 3 * tris[3 * j + 0]), v1 = ld3(verts + 3 * tris[3 * j + 1]), v2 = ld3(verts + 3 * tris[3 * j + 2]);
        b.bottom = minVec3(b.bottom, v0); b.bottom = minVec3(b.bottom, v1); b.bottom = minVec3(b.bottom, v2);
        b.top = maxVec3(b.top, v0); b.top = maxVec3(b.top, v1); b.top = maxVec3(b.top, v2);
        bottom = minVec3(bottom, b.bottom);
        top = maxVec3(top, b.top);
        b.center = f3_scale(f3_add(b.top, b.bottom), 0.5f);
        work[j] = b;
    }
    BNode* root = Recurse(work, ntris, 0);
    root->bottom = bottom; root->top = top;
    free(work);

    uint32_t nodeCount = CountBoxes(root);
    uint32_t nodeListSize = nodeCount * (uint32_t)sizeof(OrcNode), faceListSize = ntris * (uint32_t)sizeof(OrcTri),
             vertexListSize = nverts * 16u;
    uint32_t total = 16u + nodeListSize + faceListSize + vertexListSize;
    OrcBlas* blas = (OrcBlas*)calloc(1, sizeof(OrcBlas));
    blas->data = (uint8_t*)calloc(total, 1);
    blas->size = total;
    OrcAccelBot hdr = {2u, 16u, 16u + nodeListSize, 16u + nodeListSize + faceListSize};
    memcpy(blas->data, &hdr, 16);
    uint32_t faceIdx = 0, nodeIdx = 0;
    PopulateTri(tris, root, &faceIdx, &nodeIdx, (OrcTri*)(blas->data + hdr.faceByteOffset),
                (OrcNode*)(blas->data + hdr.nodeByteOffset));
    if (nodeIdx != nodeCount - 1 || faceIdx != ntris) { fprintf(stderr, "oracle: flatten mismatch\n"); abort(); }
    float* pv = (float*)(blas->data + hdr.vertexOffset);
    for (uint32_t i = 0; i < nverts; i++) {
        pv[4 * i + 0] = verts[3 * i + 0]; pv[4 * i + 1] = verts[3 * i + 1]; pv[4 * i + 2] = verts[3 * i + 2];
    }
    CountDepth(root, 0, &blas->maxDepth);
    free_tree(root);
    return blas;
}
uint32_t orc_blas_size(const void* b) { return ((const OrcBlas*)b)->size; }
const void* orc_blas_data(const void* b) { return ((const OrcBlas*)b)->data; }
int orc_blas_max_depth(const void* b) { return ((const OrcBlas*)b)->maxDepth; }
void orc_blas_free(void* b) { if (b) { free(((OrcBlas*)b)->data); free(b); } }
void orc_free(void* p) { free(p); }

/* bvh.cpp:524-566 */
static void PopulateInst(const OrcInstanceDesc* insts, const uint32_t* offsets, const BNode* root, uint32_t* instIdx,
                         uint32_t* nodeIdx, OrcInst* instList, OrcNode* nodeList)
{
    uint32_t curr = *nodeIdx;
    node_set_box(&nodeList[curr], root);
    if (!root->leaf) {
        uint32_t idxLeft = ++(*nodeIdx);
        PopulateInst(insts, offsets, root->left, instIdx, nodeIdx, instList, nodeList);
        uint32_t idxRight = ++(*nodeIdx);
        PopulateInst(insts, offsets, root->right, instIdx, nodeIdx, instList, nodeList);
        nodeList[curr].w0 = idxLeft;
        nodeList[curr].w1 = idxRight;
    } else {
        nodeList[curr].w0 = 0x80000000u | root->nprims;
        nodeList[curr].w1 = *instIdx;
        nodeList[curr].w2 = TYPE_INST;
        for (uint32_t i = 0; i < root->nprims; i++) {
            uint32_t p = root->prims[i];
            OrcInst* d = &instList[*instIdx];
            d->instanceID = p;
            d->customInstanceID = insts[p].customInstanceID;
            d->SBTOffset = insts[p].SBTOffset;
            memcpy(d->r, insts[p].transform, 64);
            d->instanceOffset = offsets[insts[p].blas];
            (*instIdx)++;
        }
    }
}

/* bvh.cpp:343-420 CreateBVH(instances) + bvh.cpp:568-597 + radiance.cpp:366-425.
 * The 4x4 product is assimp's aiMatrix4x4t::operator* (un-vendored third party, parity unpinned):
 * result[r][c] = v[0][c]*T[r][0] + v[1][c]*T[r][1] + v[2][c]*T[r][2] + v[3][c]*T[r][3]. */
void* orc_tlas_build(const OrcInstanceDesc* inst, uint32_t ninst, void* const* blas_handles, uint32_t* out_size,
                     int* out_max_depth)
{
    BBoxTmp* work = (BBoxTmp*)malloc(sizeof(BBoxTmp) * (ninst ? ninst : 1));
    f3 bottom = F3(FLT_MAX, FLT_MAX, FLT_MAX), top = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    uint32_t maxBlas = 0;
    for (uint32_t k = 0; k < ninst; k++) {
        const OrcBlas* bl = (const OrcBlas*)blas_handles[inst[k].blas];
        if (inst[k].blas > maxBlas) maxBlas = inst[k].blas;
        const OrcAccelBot* hdr = (const OrcAccelBot*)bl->data;
        const OrcNode* root = (const OrcNode*)(bl->data + hdr->nodeByteOffset);
        f3 bt = ld3(root->top), bb = ld3(root->bottom);
        float vi0[4][4] = {{bt.x, bb.x, bt.x, bb.x}, {bt.y, bt.y, bb.y, bb.y}, {bt.z, bt.z, bt.z, bt.z}, {1, 1, 1, 1}};
        float vi1[4][4] = {{bt.x, bb.x, bt.x, bb.x}, {bt.y, bt.y, bb.y, bb.y}, {bb.z, bb.z, bb.z, bb.z}, {1, 1, 1, 1}};
        const float* T = inst[k].transform;
        float vf0[4][4], vf1[4][4];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                vf0[r][c] = vi0[0][c] * T[4 * r + 0] + vi0[1][c] * T[4 * r + 1] + vi0[2][c] * T[4 * r + 2] + vi0[3][c] * T[4 * r + 3];
                vf1[r][c] = vi1[0][c] * T[4 * r + 0] + vi1[1][c] * T[4 * r + 1] + vi1[2][c] * T[4 * r + 2] + vi1[3][c] * T[4 * r + 3];
            }
#define COL(m, c) F3(m[0][c], m[1][c], m[2][c])
        f3 tmp0, tmp1, tmp2, tmp3, tmp4, tmp5;
        BBoxTmp b;
        tmp0 = minVec3(COL(vf0, 1), COL(vf0, 0));
        tmp1 = minVec3(COL(vf0, 3), COL(vf0, 2));
        tmp2 = minVec3(COL(vf1, 1), COL(vf1, 0));
        tmp3 = minVec3(COL(vf1, 3), COL(vf1, 2));
        tmp4 = minVec3(tmp1, tmp0);
        tmp5 = minVec3(tmp3, tmp2);
        b.bottom = minVec3(tmp4, tmp5);
        tmp0 = maxVec3(COL(vf0, 1), COL(vf0, 0));
        tmp1 = maxVec3(COL(vf0, 3), COL(vf0, 2));
        tmp2 = maxVec3(COL(vf1, 1), COL(vf1, 0));
        tmp3 = maxVec3(COL(vf1, 3), COL(vf1, 2));
        tmp4 = maxVec3(tmp1, tmp0);
        tmp5 = maxVec3(tmp3, tmp2);
        b.top = maxVec3(tmp4, tmp5);
#undef COL
        b.center = f3_scale(f3_add(b.top, b.bottom), 0.5f);
        b.prim = k;
        bottom = minVec3(bottom, b.bottom);
        top = maxVec3(top, b.top);
        work[k] = b;
    }
    BNode* root = Recurse(work, ninst, 0);
    root->bottom = bottom; root->top = top;
    free(work);

    uint32_t nodeCount = CountBoxes(root);
    uint32_t topASSize = 16u + nodeCount * (uint32_t)sizeof(OrcNode) + ninst * (uint32_t)sizeof(OrcInst);
    /* bvh.cpp:575-588 : distinct BLAS blobs appended in order of first appearance */
    uint32_t* offsets = (uint32_t*)calloc(maxBlas + 1, sizeof(uint32_t));
    uint8_t* seen = (uint8_t*)calloc(maxBlas + 1, 1);
    uint32_t nextOffset = 0;
    for (uint32_t k = 0; k < ninst; k++) {
        uint32_t bi = inst[k].blas;
        if (!seen[bi]) {
            seen[bi] = 1;
            offsets[bi] = nextOffset + topASSize;
            nextOffset += ((const OrcBlas*)blas_handles[bi])->size;
        }
    }
    uint32_t total = topASSize + nextOffset;
    uint8_t* blob = (uint8_t*)calloc(total, 1);
    OrcAccelTop hdr = {1u, 16u, 16u + nodeCount * (uint32_t)sizeof(OrcNode), total};
    memcpy(blob, &hdr, 16);
    uint32_t instIdx = 0, nodeIdx = 0;
    PopulateInst(inst, offsets, root, &instIdx, &nodeIdx, (OrcInst*)(blob + hdr.instByteOffset),
                 (OrcNode*)(blob + hdr.nodeByteOffset));
    for (uint32_t bi = 0; bi <= maxBlas; bi++)
        if (seen[bi]) {
            const OrcBlas* bl = (const OrcBlas*)blas_handles[bi];
            memcpy(blob + offsets[bi], bl->data, bl->size);
        }
    int md = 0;
    CountDepth(root, 0, &md);
    if (out_max_depth) *out_max_depth = md;
    if (out_size) *out_size = total;
    free(offsets); free(seen);
    free_tree(root);
    return blob;
}

/* ------------------------------------------------------------------------------------------ */
/* exported batch / unit entry points                                                          */
/* ------------------------------------------------------------------------------------------ */
static void ctr_add(OrcCounters* dst, const OrcCounters* s)
{
    for (int k = 0; k < 2; k++) {
        dst->rays[k] += s->rays[k]; dst->top_nodes[k] += s->top_nodes[k]; dst->inst_visits[k] += s->inst_visits[k];
        dst->bot_nodes[k] += s->bot_nodes[k]; dst->tri_tests[k] += s->tri_tests[k];
    }
    dst->hits += s->hits;
    dst->primary += s->primary; dst->bounce += s->bounce; dst->shadow += s->shadow;
}

void orc_trace_batch(const void* tlas, const float* origins, const float* dirs, uint32_t n, float tmin, float tmax,
                     int sbtRecordOffset, OrcHit* out, OrcCounters* ctr)
{
    OrcCounters local;
    memset(&local, 0, sizeof local);
    t_ctr = ctr ? &local : NULL;
    for (uint32_t i = 0; i < n; i++) {
        HitData hd;
        Payload payload;
        SceneData sd;
        memset(&payload, 0, sizeof payload);
        memset(&sd, 0, sizeof sd);
        hitdata_init(&hd);
        t_cls = sbtRecordOffset == 2 ? 1 : 0;
        if (t_ctr) t_ctr->rays[t_cls]++;
        int hit = intersectTop((const uint8_t*)tlas, ld3(origins + 3 * i), ld3(dirs + 3 * i), tmin, tmax, &hd,
                               sbtRecordOffset, &payload, &sd);
        OrcHit* o = &out[i];
        o->hitPoint[0] = hd.hitPoint.x; o->hitPoint[1] = hd.hitPoint.y; o->hitPoint[2] = hd.hitPoint.z;
        o->distance = hd.distance;
        o->primitiveIndex = hd.primitiveIndex; o->instanceIndex = hd.instanceIndex;
        o->instanceCustomIndex = hd.instanceCustomIndex; o->instanceSBTOffset = hd.instanceSBTOffset;
        o->barycentric[0] = hd.barycentric.x; o->barycentric[1] = hd.barycentric.y; o->barycentric[2] = hd.barycentric.z;
        o->hit = (uint32_t)hit;
        memcpy(o->transform, hd.transform.s, 64);
    }
    if (ctr) ctr_add(ctr, &local);
    t_ctr = NULL;
}

void orc_pcg3d(const uint32_t* in3, float* out3, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        f3 r = random_pcg3d(in3[3 * i], in3[3 * i + 1], in3[3 * i + 2]);
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}
int orc_inverse_mat4(const float* m16, float* out16)
{
    m44 a, b;
    memcpy(a.s, m16, 64);
    memset(&b, 0, sizeof b);
    int ok = InverseMat4x4(&a, &b);
    memcpy(out16, b.s, 64);
    return ok;
}
void orc_mul_mat4_vec4(const float* m16, const float* v4, float* out4)
{
    m44 a; f4 v = {v4[0], v4[1], v4[2], v4[3]}, o;
    memcpy(a.s, m16, 64);
    MultiplyMat4Vec4(&a, &v, &o);
    out4[0] = o.x; out4[1] = o.y; out4[2] = o.z; out4[3] = o.w;
}
int orc_intersect_aabb(const float* o3, const float* d3, const float* bmin3, const float* bmax3)
{
    return intersectAABB(ld3(o3), ld3(d3), ld3(bmin3), ld3(bmax3));
}
int orc_intersect_triangle(const float* o3, const float* d3, const float* v0, const float* v1, const float* v2,
                           float* t, float* point3, float* bary3)
{
    f3 p = {0, 0, 0}, b = {0, 0, 0};
    float dist = 0;
    int r = intersectTriangle(ld3(o3), ld3(d3), ld3(v0), ld3(v1), ld3(v2), &p, &dist, &b);
    *t = dist;
    point3[0] = p.x; point3[1] = p.y; point3[2] = p.z;
    bary3[0] = b.x; bary3[1] = b.y; bary3[2] = b.z;
    return r;
}
void orc_generate_ray(const OrcCamera* cam, uint32_t pixel, const uint32_t* r, float* origin3, float* dir3)
{
    f3 o, d;
    generateRay(cam, (int)pixel, r[0], r[1], r[2], &o, &d);
    origin3[0] = o.x; origin3[1] = o.y; origin3[2] = o.z;
    dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z;
}
void orc_microfacet_brdf(const float* L, const float* V, const float* N, const float* albedo, float metallic,
                         float roughness, float transmission, float ior, float* out3)
{
    f3 r = microfacetBRDF(ld3(L), ld3(V), ld3(N), ld3(albedo), metallic, roughness, transmission, ior);
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
void orc_sample_brdf_transm(const float* V, const float* N, const float* baseColor, float metallic, float roughness,
                            float transmission, float ior, const float* random3, float* nextFactor3, float* L3)
{
    f3 nf = {0, 0, 0};
    f3 L = sampleMicrofacetBRDF_transm(ld3(V), ld3(N), ld3(baseColor), metallic, roughness, transmission, ior,
                                       ld3(random3), &nf);
    nextFactor3[0] = nf.x; nextFactor3[1] = nf.y; nextFactor3[2] = nf.z;
    L3[0] = L.x; L3[1] = L.y; L3[2] = L.z;
}
float orc_d_ggx(float dotNH, float roughness) { return D_GGX(dotNH, roughness); }
void orc_aces(const float* in3, float* out3)
{
    f3 r = aces_approx(ld3(in3));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

void orc_material_batch(const OrcBindings* b, const OrcHit* hits, const float* ray_dirs, const uint32_t* pixels,
                        const uint32_t* frame_ids, const int32_t* depths, uint32_t n, OrcPayload* out)
{
    t_ctr = NULL;
    for (uint32_t i = 0; i < n; i++) {
        HitData hd;
        Payload p;
        SceneData sd;
        memset(&p, 0, sizeof p);
        hd.hitPoint = ld3(hits[i].hitPoint);
        hd.distance = hits[i].distance;
        hd.primitiveIndex = hits[i].primitiveIndex;
        hd.instanceIndex = hits[i].instanceIndex;
        hd.instanceCustomIndex = hits[i].instanceCustomIndex;
        hd.instanceSBTOffset = hits[i].instanceSBTOffset;
        hd.barycentric = ld3(hits[i].barycentric);
        memcpy(hd.transform.s, hits[i].transform, 64);
        p.nextRayDirection = ld3(ray_dirs + 3 * i);
        sd.b = b; sd.depth = depths[i]; sd.frameID = frame_ids[i]; sd.debug = 0; sd.global_id = pixels[i];
        material(&p, &hd, &sd);
        OrcPayload* o = &out[i];
        o->color[0] = p.color.x; o->color[1] = p.color.y; o->color[2] = p.color.z;
        o->hit = (uint32_t)p.hit;
        o->nextFactor[0] = p.nextFactor.x; o->nextFactor[1] = p.nextFactor.y; o->nextFactor[2] = p.nextFactor.z;
        o->nextRayOrigin[0] = p.nextRayOrigin.x; o->nextRayOrigin[1] = p.nextRayOrigin.y; o->nextRayOrigin[2] = p.nextRayOrigin.z;
        o->nextRayDirection[0] = p.nextRayDirection.x; o->nextRayDirection[1] = p.nextRayDirection.y; o->nextRayDirection[2] = p.nextRayDirection.z;
    }
}

static void render_impl(const OrcBindings* b, const uint32_t* pixels, uint32_t begin, uint32_t n, int nthreads,
                        OrcCounters* ctr)
{
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
    OrcCounters total;
    memset(&total, 0, sizeof total);
#pragma omp parallel num_threads(nthreads)
    {
        OrcCounters local;
        memset(&local, 0, sizeof local);
        t_ctr = ctr ? &local : NULL;
#pragma omp for schedule(dynamic, 256)
        for (uint32_t i = 0; i < n; i++)
            raygen(b, (int)(pixels ? pixels[i] : begin + i));
#pragma omp critical
        ctr_add(&total, &local);
        t_ctr = NULL;
    }
    if (ctr) ctr_add(ctr, &total);
}

void orc_render(const OrcBindings* b, uint32_t pixel_begin, uint32_t pixel_end, int nthreads, OrcCounters* ctr)
{
    render_impl(b, NULL, pixel_begin, pixel_end - pixel_begin, nthreads, ctr);
}
void orc_render_pixels(const OrcBindings* b, const uint32_t* pixels, uint32_t n, int nthreads, OrcCounters* ctr)
{
    render_impl(b, pixels, 0, n, nthreads, ctr);
}

void orc_struct_sizes(uint32_t* out, uint32_t n)
{
    uint32_t v[] = {sizeof(OrcNode), sizeof(OrcTri), 16, sizeof(OrcInst), sizeof(OrcAccelTop), sizeof(OrcAccelBot),
                    sizeof(OrcMaterial), sizeof(OrcMeshInfo), sizeof(OrcDirLight), sizeof(OrcSceneProps),
                    sizeof(OrcCamera), sizeof(OrcRTProp)};
    for (uint32_t i = 0; i < n && i < sizeof v / sizeof v[0]; i++) out[i] = v[i];
}

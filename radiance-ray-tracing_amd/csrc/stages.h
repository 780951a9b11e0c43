// stages.h -- the SBT stage functions of the stock pipeline as HIP device code, and the SBT
// dispatch generated from samples/sbt.json (tools/genSBT.py -> sbt_generated.h).
//
// Restates the *behaviour* of the reference's user shader and PBR library
// (samples/shader.cl:308-605, radiance/shader/pbr.cl) for a wavefront tracer: the closest-hit
// shader `material` traces its shadow ray in the middle of the function (shader.cl:499-509); here
// it runs to completion and hands back both outcomes (lit / occluded) plus the shadow-ray origin,
// and the any-hit traversal stage picks one afterwards.  Arithmetic per value is unchanged.
#pragma once
#include "device_math.h"
#include "kernels.h"
#include "rdx_types.h"
#ifndef RDX_SBT_HEADER
#define RDX_SBT_HEADER "sbt_generated.h"      // tools/genSBT.py output for samples/sbt.json; a build for another table defines this
#endif
#include RDX_SBT_HEADER

namespace rdx {

#define RDX_PI 3.14159265359f   // pbr.cl:3

struct SceneView {                  // descriptor slots 4-10 of the raygen kernel (shader.cl:175-190)
    const SceneProperties* scene;
    const MeshInfo* meshInfo;
    const uint32_t* indexData;
    const float* uvData;
    const float* normalData;
    const Material* materials;
    TexView tex;                    // slots 11 + 12 (disabled unless option "textures" is on and an image array is bound)
};

struct HitInfo {                    // what struct HitData (radiance.cl:8-18) carries into a hit shader
    f3 hitPoint;                    // object space
    float bx, by, bz;               // barycentric (1-b1-b2, b1, b2)
    uint32_t primitiveIndex, instanceIndex;
    const float* fwd;               // object->world, row-major (DInst::fwd)
};

struct Payload {                    // struct Payload, shader.cl:4-13 (+ the deferred shadow query)
    f3 color;                       // if the shadow ray is NOT occluded
    f3 colorOccluded;               // if it is
    f3 nextFactor, nextRayOrigin, nextRayDirection;
    f3 shadowOrigin;
    bool hit;
    bool wantsShadowRay;
};

// ---------------------------------------------------------------------------------------------
// pbr.cl
// ---------------------------------------------------------------------------------------------
__device__ inline void normal_space(f3 n, float* m)   // math.cl:269-298 GetNormalSpace
{
    float dd = dot3(mk3(1.0f, 0.0f, 0.0f), n);
    f3 tangent = mk3(0.0f, 1.0f, 0.0f);
    if (1.0f - fabsf(dd) > 1e-6f) tangent = normalize3(cross3(mk3(1.0f, 0.0f, 0.0f), n));
    f3 bitangent = cross3(n, tangent);
    m[0] = tangent.x;   m[4] = tangent.y;   m[8] = tangent.z;    m[12] = 0.0f;
    m[1] = bitangent.x; m[5] = bitangent.y; m[9] = bitangent.z;  m[13] = 0.0f;
    m[2] = n.x;         m[6] = n.y;         m[10] = n.z;         m[14] = 0.0f;
    m[3] = 0.0f;        m[7] = 0.0f;        m[11] = 0.0f;        m[15] = 1.0f;
}

__device__ inline float d_ggx(float dotNH, float roughness)   // pbr.cl:6-13
{
    float alpha = roughness * roughness;
    float alpha2 = alpha * alpha;
    float denom = dotNH * dotNH * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (RDX_PI * denom * denom);
}

__device__ inline f3 f_schlick(float cosTheta, float metallic, f3 albedo)   // pbr.cl:31-37
{
    const f3 lo = mk3(0.04f, 0.04f, 0.04f);
    f3 F0 = mix3(lo, albedo, metallic);
    const float p = powf(1.0f - cosTheta, 5.0f);        // the same OCML pow the reference's pow(x, 5.0f) links
    return F0 + one_minus(F0) * p;
}

__device__ inline float smith_lambda(f3 w, float a)   // pbr.cl:41-74 Lambda and its trig helpers
{
    float cos2 = w.z * w.z;
    float sin2 = fmaxf(0.0f, 1.0f - cos2);
    float tan2 = sin2 / cos2;
    if (isinf(tan2)) return 0.0f;
    float sinT = sqrtf(sin2);
    float cosPhi = (sinT == 0.0f) ? 1.0f : cl_clamp(w.x / sinT, -1.0f, 1.0f);
    float sinPhi = (sinT == 0.0f) ? 0.0f : cl_clamp(w.y / sinT, -1.0f, 1.0f);
    float alpha2 = (cosPhi * a) * (cosPhi * a) + (sinPhi * a) * (sinPhi * a);
    return (sqrtf(1.0f + alpha2 * tan2) - 1.0f) / 2.0f;
}

// Tangent frame of a normal and its inverse.  The reference rebuilds both inside every G_pbrt call
// and every sampled direction (pbr.cl:87-88, 309-311, 339-341, 362-364); for one hit they are the same
// matrices, so they are built once here and reused -- same values, fewer instructions.
//
// GetNormalSpace (math.cl:269-298) returns [T B N | 0; 0 0 0 1] and every use multiplies by a direction (w = 0),
// so only the 3x3 block and the 3x3 block of the cofactor inverse (math.cl:56-183) are ever read.  With
// m3 = m7 = m11 = m12 = m13 = m14 = 0 and m15 = 1 each of those cofactors keeps two of its six products (the
// others are x*0*y = +-0, and a*1 = a), evaluated here in the reference's order: same values, the sign of an
// exact zero aside.
struct NFrame { float tbn[9]; float inv[9]; };      // row-major 3x3
__device__ inline void make_frame(f3 n, NFrame& F)
{
    float dd = dot3(mk3(1.0f, 0.0f, 0.0f), n);
    f3 t = mk3(0.0f, 1.0f, 0.0f);
    if (1.0f - fabsf(dd) > 1e-6f) t = normalize3(cross3(mk3(1.0f, 0.0f, 0.0f), n));
    const f3 b = cross3(n, t);
    const float m0 = t.x, m1 = b.x, m2 = n.x, m4 = t.y, m5 = b.y, m6 = n.y, m8 = t.z, m9 = b.z, m10 = n.z;
    F.tbn[0] = m0; F.tbn[1] = m1; F.tbn[2] = m2; F.tbn[3] = m4; F.tbn[4] = m5; F.tbn[5] = m6; F.tbn[6] = m8; F.tbn[7] = m9; F.tbn[8] = m10;
    const float i0 = m5 * m10 - m9 * m6, i4 = -m4 * m10 + m8 * m6, i8 = m4 * m9 - m8 * m5;
    const float i1 = -m1 * m10 + m9 * m2, i5 = m0 * m10 - m8 * m2, i9 = -m0 * m9 + m8 * m1;
    const float i2 = m1 * m6 - m5 * m2, i6 = -m0 * m6 + m4 * m2, i10 = m0 * m5 - m4 * m1;
    float det = m0 * i0 + m1 * i4 + m2 * i8;
    for (int i = 0; i < 9; ++i) F.inv[i] = 0.0f;       // reference: uninitialised if det == 0
    if (det == 0) return;
    det = 1.0f / det;
    F.inv[0] = i0 * det; F.inv[1] = i1 * det; F.inv[2] = i2 * det;
    F.inv[3] = i4 * det; F.inv[4] = i5 * det; F.inv[5] = i6 * det;
    F.inv[6] = i8 * det; F.inv[7] = i9 * det; F.inv[8] = i10 * det;
}
__device__ __forceinline__ f3 mat3_mul(const float* m, float x, float y, float z)
{
    return mk3(m[0] * x + m[1] * y + m[2] * z, m[3] * x + m[4] * y + m[5] * z, m[6] * x + m[7] * y + m[8] * z);
}
// the same with the `+ s3 * w` term of MultiplyMat4Vec4 for s3 = 0, w = 0 (a -0 sum becomes +0): used where the
// result leaves the shader as a direction
__device__ __forceinline__ f3 mat3_mul_w0(const float* m, float x, float y, float z)
{
    return mk3(m[0] * x + m[1] * y + m[2] * z + 0.0f, m[3] * x + m[4] * y + m[5] * z + 0.0f, m[6] * x + m[7] * y + m[8] * z + 0.0f);
}

__device__ inline float g_pbrt(const NFrame& F, f3 wo, f3 wi, float roughness)   // pbr.cl:77-96
{
    f3 lo = mat3_mul(F.inv, wo.x, wo.y, wo.z);
    f3 li = mat3_mul(F.inv, wi.x, wi.y, wi.z);
    if (li.z < 0 || lo.z < 0) return 0.0f;
    return 1 / (1 + smith_lambda(mk3(li.x, li.y, li.z), roughness) + smith_lambda(mk3(lo.x, lo.y, lo.z), roughness));
}

__device__ inline f3 reflect3(f3 in, f3 n) { return -in + n * (2 * dot3(in, n)); }   // pbr.cl:171-174

__device__ inline f3 refract3(f3 V, f3 H, float eta)   // pbr.cl:176-186
{
    float ci = dot3(H, V);
    float s2i = cl_max(0.0f, 1.0f - (ci * ci));
    float s2t = s2i / (eta * eta);
    if ((1.0f - s2t) < 0.0f) return (H * ci - V) / eta;
    float ct = sqrtf(1.0f - s2t);
    return (-V) / eta + H * (ci / eta - ct);
}

__device__ inline f3 microfacet_brdf(const NFrame& FN, f3 L, f3 V, f3 N, f3 albedo, float metallic, float roughness,
                                     float transmission)
{   // pbr.cl:268-287
    f3 H = normalize3(V + L);
    float NoV = cl_clamp(dot3(N, V), 0.0f, 1.0f);
    float NoL = cl_clamp(dot3(N, L), 0.0f, 1.0f);
    float NoH = cl_clamp(dot3(N, H), 0.0f, 1.0f);
    float VoH = cl_clamp(dot3(V, H), 0.0f, 1.0f);
    f3 F = f_schlick(VoH, metallic, albedo);
    float D = d_ggx(NoH, roughness);
    float G = g_pbrt(FN, V, L, roughness);
    f3 spec = (F * (D * G)) / cl_max(4.0f * NoV * NoL, 0.001f);
    f3 notSpec = (one_minus(F) * (1.0f - metallic)) * (1.0f - transmission);
    f3 diff = notSpec * (albedo / RDX_PI);
    return (diff + spec) * NoL;
}

// local (theta, phi) direction rotated into a frame
__device__ inline f3 frame_dir(const NFrame& F, float theta, float phi)
{
    float st = sinf(theta), ct = cosf(theta), sp = sinf(phi), cp = cosf(phi);
    return mat3_mul_w0(F.tbn, st * cp, st * sp, ct);
}

// pbr.cl:289-385 sampleMicrofacetBRDF_transm.  The diffuse and the specular lobe share the frame of N,
// the azimuth and the Fresnel term, so they run as one converged code path with selects; only the rare
// transmission lobe (glass) branches off with the frame of the forward normal.
__device__ inline f3 sample_brdf_transm(const NFrame& FN, f3 V, f3 N, f3 base, float metallic, float roughness,
                                        float transmission, float ior, f3 rnd, f3& nextFactor)
{
    const float ggxTheta = acosf(sqrtf((1.0f - rnd.y) / (1.0f + ((roughness * roughness) * (roughness * roughness) - 1.0f) * rnd.y)));
    const float phi = 2.0f * RDX_PI * rnd.x;
    const bool lower = rnd.z < 0.5f;
    if (lower && (2.0f * rnd.z) < transmission) {
        f3 fn = N;
        float eta = ior;
        NFrame FT;
        if (dot3(V, N) < 0.0f) { fn = -N; eta = 1.0f / ior; make_frame(fn, FT); } else FT = FN;
        f3 H = frame_dir(FT, ggxTheta, phi);
        f3 L = refract3(V, H, eta);
        float NoV = cl_clamp(dot3(fn, V), 0.0f, 1.0f);
        float NoH = cl_clamp(dot3(fn, H), 0.0f, 1.0f);
        float VoH = cl_clamp(dot3(V, H), 0.0f, 1.0f);
        f3 F = f_schlick(VoH, metallic, base);
        float G = g_pbrt(FT, V, -L, roughness);
        nextFactor = ((((base * one_minus(F)) * G) * VoH) / cl_max(NoH * NoV, 0.001f)) * 2.0f;
        return L;
    }
    const bool diffuse = lower;
    const f3 dirv = frame_dir(FN, diffuse ? acosf(sqrtf(rnd.y)) : ggxTheta, phi);
    const f3 L = diffuse ? dirv : reflect3(V, dirv);
    const f3 H = diffuse ? normalize3(V + L) : dirv;
    const float VoH = cl_clamp(dot3(V, H), 0.0f, 1.0f);
    const f3 F = f_schlick(VoH, metallic, base);
    if (diffuse) {
        nextFactor = ((one_minus(F) * (1.0f - metallic)) * base) * 2.0f;
    } else {
        const float NoV = cl_clamp(dot3(N, V), 0.0f, 1.0f);
        const float NoH = cl_clamp(dot3(N, H), 0.0f, 1.0f);
        const float G = g_pbrt(FN, V, L, roughness);
        nextFactor = (((F * G) * VoH) / cl_max(NoH * NoV, 0.001f)) * 2.0f;
    }
    return L;
}

// ---------------------------------------------------------------------------------------------
// stage functions named in samples/sbt.json
// ---------------------------------------------------------------------------------------------

// world-space hit position pushed off the surface along n (shader.cl:454-469)
__device__ inline f3 offset_hit_position(const HitInfo& h, f3 n)
{
    f3 p = mat4_mul3(h.fwd, h.hitPoint.x, h.hitPoint.y, h.hitPoint.z, 1.0f);
    return p + n * 0.00001f;
}

// ---- texture array reads -------------------------------------------------------------------------------------------
// read_imageui(imageArray, sampler, (float4)(u, v, layer, 0)) on a CL_RGBA / CL_UNSIGNED_INT8 2D image array with a
// normalized-coordinate sampler, per the OpenCL 1.2 specification, section 8.2 (addressing modes) and 5.3.3 (array layer =
// clamp(rint(layer), 0, layers - 1)).  The specification leaves CLK_FILTER_LINEAR undefined for integer reads; the
// reference binds a linear sampler (tools/sceneBuilder.cpp:40), so linear is DEFINED here as the spec's bilinear weights on
// the 8-bit values, rounded to nearest.  The live reference shader never reaches this code ("parity unpinned").
__device__ inline int tex_addr(float s, int n, uint32_t mode, float& u)          // -> nearest texel index (or -1: border), u = unnormalised
{
    if (mode == TEX_ADDR_REPEAT) { u = (s - floorf(s)) * (float)n; int i = (int)floorf(u); return i > n - 1 ? i - n : i; }
    if (mode == TEX_ADDR_MIRRORED) { float sp = 2.0f * rintf(0.5f * s); sp = fabsf(s - sp); u = sp * (float)n; int i = (int)floorf(u); return i > n - 1 ? n - 1 : i; }
    u = s * (float)n;
    const int i = (int)floorf(u);
    if (mode == TEX_ADDR_CLAMP_TO_EDGE) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    return (i < 0 || i > n - 1) ? -1 : i;                                          // CLAMP: border colour (0, 0, 0, 0)
}
__device__ inline int tex_wrap(int i, int n, uint32_t mode)                       // neighbour index of the linear filter
{
    if (mode == TEX_ADDR_REPEAT) return i < 0 ? i + n : (i > n - 1 ? i - n : i);
    if (mode == TEX_ADDR_MIRRORED || mode == TEX_ADDR_CLAMP_TO_EDGE) return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    return (i < 0 || i > n - 1) ? -1 : i;
}
__device__ inline void tex_texel(const TexView& T, int layer, int x, int y, float out[4])
{
    if (x < 0 || y < 0) { out[0] = out[1] = out[2] = out[3] = 0.0f; return; }
    const uchar4 t = reinterpret_cast<const uchar4*>(T.data)[((size_t)layer * T.h + (uint32_t)y) * T.w + (uint32_t)x];
    out[0] = (float)t.x; out[1] = (float)t.y; out[2] = (float)t.z; out[3] = (float)t.w;
}
__device__ inline void tex_read_ui(const TexView& T, float u, float v, float layerF, uint32_t out[4])
{
    const uint32_t mode = (T.flags >> TEX_ADDR_SHIFT) & 3u;
    int layer = (int)rintf(layerF);
    layer = layer < 0 ? 0 : (layer > (int)T.layers - 1 ? (int)T.layers - 1 : layer);
    float uu, vv;
    const int ix = tex_addr(u, (int)T.w, mode, uu), iy = tex_addr(v, (int)T.h, mode, vv);
    float c[4];
    if (!(T.flags & TEX_LINEAR)) {
        tex_texel(T, layer, ix, iy, c);
        for (int k = 0; k < 4; ++k) out[k] = (uint32_t)c[k];
        return;
    }
    const float fu = uu - 0.5f, fv = vv - 0.5f;
    const int i0 = (int)floorf(fu), j0 = (int)floorf(fv);
    const float a = fu - floorf(fu), b = fv - floorf(fv);
    const int x0 = tex_wrap(i0, (int)T.w, mode), x1 = tex_wrap(i0 + 1, (int)T.w, mode);
    const int y0 = tex_wrap(j0, (int)T.h, mode), y1 = tex_wrap(j0 + 1, (int)T.h, mode);
    float t00[4], t10[4], t01[4], t11[4];
    tex_texel(T, layer, x0, y0, t00); tex_texel(T, layer, x1, y0, t10); tex_texel(T, layer, x0, y1, t01); tex_texel(T, layer, x1, y1, t11);
    for (int k = 0; k < 4; ++k)
        out[k] = (uint32_t)((1.0f - a) * (1.0f - b) * t00[k] + a * (1.0f - b) * t10[k] + (1.0f - a) * b * t01[k] + a * b * t11[k] + 0.5f);
}

// closest-hit, SBT row 1 (shader.cl:482-541).  `sampleNext` = false on the last bounce, whose
// sampled direction the raygen loop never uses.
__device__ inline void material(Payload& p, const HitInfo& h, const SceneView& s, f3 rayDir, uint32_t frameID,
                                uint32_t pixel, uint32_t depth, bool sampleNext)
{
    p.hit = true;
    const MeshInfo mi = s.meshInfo[h.instanceIndex];
    const Material mt = s.materials[mi.materialIndex];
    const uint32_t* ip = s.indexData + mi.indexOffset + h.primitiveIndex * 3;
#ifdef RDX_EXP_SHADE_NOGATHER     // timing experiment (wrong results): no dependent index / normal gathers
    const uint32_t i0 = 0, i1 = 1, i2 = 2;
#else
    const uint32_t i0 = ip[0], i1 = ip[1], i2 = ip[2];
#endif

    // interpolated vertex normal -> world by the object->world matrix, w = 0 (shader.cl:340-368)
    const float* nb = s.normalData + mi.normalOffset;
    f3 n0 = mk3(nb[i0 * 3], nb[i0 * 3 + 1], nb[i0 * 3 + 2]);
    f3 n1 = mk3(nb[i1 * 3], nb[i1 * 3 + 1], nb[i1 * 3 + 2]);
    f3 n2 = mk3(nb[i2 * 3], nb[i2 * 3 + 1], nb[i2 * 3 + 2]);
    f3 nl = mk3(h.bx * n0.x + h.by * n1.x + h.bz * n2.x, h.bx * n0.y + h.by * n1.y + h.bz * n2.y,
                h.bx * n0.z + h.by * n1.z + h.bz * n2.z);
    float nw = h.bx * 0.0f + h.by * 0.0f + h.bz * 0.0f;
    f3 faceN = normalize3(mat4_mul3(h.fwd, nl.x, nl.y, nl.z, nw));
    f3 hitPos = offset_hit_position(h, faceN);

    // Texture fetches are stubbed to 0 in the live reference shader (`uint4 tex = 0.0f;//read_imageui(...)`,
    // shader.cl:379,411,421,445): that is what runs unless option "textures" is on and an image array is bound, in which case
    // the commented-out read happens: coord = (uv.x, 1 - uv.y, texIdx) (shader.cl:378,410,420,444; live in shader2.cl:255-265).
    const bool texOn = (s.tex.flags & TEX_ENABLED) != 0u;
    float tu = 0.0f, tv = 0.0f;
    if (texOn) {     // getUV, shader.cl:323-338
        const float* ub = s.uvData + mi.uvOffset;
        tu = h.bx * ub[i0 * 3] + h.by * ub[i1 * 3] + h.bz * ub[i2 * 3];
        tv = h.bx * ub[i0 * 3 + 1] + h.by * ub[i1 * 3 + 1] + h.bz * ub[i2 * 3 + 1];
    }
    f3 N = faceN;
    if (mt.normalTexIdx != -1) {
        uint32_t tx[4] = {0u, 0u, 0u, 0u};
        if (texOn) tex_read_ui(s.tex, tu, 1.0f - tv, (float)mt.normalTexIdx, tx);
        f4 t; t.x = cl_clamp((float)tx[0] / 255.0f, 0.0f, 1.0f) * 2.0f - 1.0f; t.y = cl_clamp((float)tx[1] / 255.0f, 0.0f, 1.0f) * 2.0f - 1.0f;
        t.z = cl_clamp((float)tx[2] / 255.0f, 0.0f, 1.0f) * 2.0f - 1.0f; t.w = 0.0f * 2.0f - 1.0f;
        t = normalize4(t);
        float tbn[16];
        normal_space(faceN, tbn);
        N = normalize3(mat4_mul3(tbn, t.x, t.y, t.z, t.w));
    }
    const float* ld = s.scene->lights[0].direction;
    f3 L = normalize3(mk3(-ld[0], -ld[1], -ld[2]));
    f3 V = normalize3(-rayDir);

    float metallic = mt.metallic;
    if (mt.metallicTexIdx != -1) {
        uint32_t tx[4] = {0u, 0u, 0u, 0u};
        if (texOn) tex_read_ui(s.tex, tu, 1.0f - tv, (float)mt.metallicTexIdx, tx);
        metallic = cl_clamp((float)tx[2] / 255.0f, 0.0f, 1.0f);                   // .z (shader.cl:412)
    }
    float roughness = cl_clamp(mt.roughness, 0.0f, 1.0f);
    if (mt.roughnessTexIdx != -1) {
        uint32_t tx[4] = {0u, 0u, 0u, 0u};
        if (texOn) tex_read_ui(s.tex, tu, 1.0f - tv, (float)mt.roughnessTexIdx, tx);
        roughness = cl_clamp((float)tx[1] / 255.0f, 0.05f, 1.0f);                 // .y (shader.cl:422)
    }
    float transmission = cl_clamp(mt.transmission, 0.0f, 1.0f);
    float ior = cl_clamp(mt.ior, 0.0f, 10.0f);
    f3 albedo = mk3(mt.albedo[0], mt.albedo[1], mt.albedo[2]);
    if (mt.albedoTexIdx != -1) {
        uint32_t tx[4] = {0u, 0u, 0u, 0u};
        if (texOn) tex_read_ui(s.tex, tu, 1.0f - tv, (float)mt.albedoTexIdx, tx);
        albedo = mk3(cl_clamp((float)tx[0] / 255.0f, 0.0f, 1.0f), cl_clamp((float)tx[1] / 255.0f, 0.0f, 1.0f), cl_clamp((float)tx[2] / 255.0f, 0.0f, 1.0f));
    }

    // deferred shadow query: traceRay(topLevel, 2, 4, hitPos, L, 0.001, 1000) (shader.cl:499-501)
    p.wantsShadowRay = true;
    p.shadowOrigin = hitPos;
#ifdef RDX_EXP_SHADE_CHEAP        // timing experiment (wrong results): all loads, none of the BRDF arithmetic
    p.color = albedo * (metallic + roughness + transmission + ior) + V + L; p.colorOccluded = albedo;
    p.nextRayOrigin = hitPos; p.nextRayDirection = faceN; p.nextFactor = albedo;
    return;
#endif
    const float* lc = s.scene->lights[0].color;
    NFrame FN;
    make_frame(N, FN);
    f3 direct = mk3(0.0f, 0.0f, 0.0f) + microfacet_brdf(FN, L, V, N, albedo, metallic, roughness, transmission) * mk3(lc[0], lc[1], lc[2]);
    f3 ambient = albedo * 0.1f;
    p.color = direct + ambient;
    p.colorOccluded = mk3(0.0f, 0.0f, 0.0f) + ambient;

    if (sampleNext) {
        f3 rnd = pcg3d(frameID, pixel, depth);
        f3 nf = mk3(0.0f, 0.0f, 0.0f);
        f3 nd = sample_brdf_transm(FN, V, N, albedo, metallic, roughness, transmission, ior, rnd, nf);
        if (dot3(nd, N) < 0) hitPos = offset_hit_position(h, -faceN);
        p.nextRayOrigin = hitPos;
        p.nextRayDirection = nd;
        p.nextFactor = nf;
    }
}

// closest-hit, SBT row 2 (shader.cl:559-565)
__device__ inline void shadow(Payload& p, const HitInfo&, const SceneView&, f3, uint32_t, uint32_t, uint32_t, bool)
{
    p.hit = true;
    p.color = mk3(0.0f, 0.0f, 0.0f);
    p.colorOccluded = p.color;
}
// any-hit, SBT row 2 (shader.cl:567-572)
__device__ inline void anyShadow(bool& cont) { cont = false; }
// miss, SBT rows 3 and 4 (shader.cl:543-557)
__device__ inline void environment(Payload& p) { p.hit = false; p.color = mk3(0.2f, 0.2f, 0.5f); p.colorOccluded = p.color; }
__device__ inline void shadowMiss(Payload& p) { p.hit = false; p.color = mk3(1.0f, 1.0f, 1.0f); p.colorOccluded = p.color; }

// ---- SBT dispatch, expanded from the generated tables (the reference keeps these switches by hand
//      in shader.cl:574-605; index = instanceSBTOffset + sbtRecordOffset, or missIndex) -----------
__device__ inline void callHit(int index, Payload& p, const HitInfo& h, const SceneView& s, f3 rayDir,
                               uint32_t frameID, uint32_t pixel, uint32_t depth, bool sampleNext)
{
    switch (index) {
#define X(row, fn) case row: fn(p, h, s, rayDir, frameID, pixel, depth, sampleNext); break;
        RDX_SBT_CLOSEST_HIT(X)
#undef X
    default: break;
    }
}
__device__ inline void callAnyHit(bool& cont, int index)
{
    switch (index) {
#define X(row, fn) case row: fn(cont); break;
        RDX_SBT_ANY_HIT(X)
#undef X
    default: break;
    }
}
__device__ inline void callMiss(int index, Payload& p)
{
    switch (index) {
#define X(row, fn) case row: fn(p); break;
        RDX_SBT_MISS(X)
#undef X
    default: break;
    }
}

} // namespace rdx

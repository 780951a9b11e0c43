/* pbr.cl -- microfacet BRDF helpers and the material / light structs of the device library (product-owned text).
 *
 * Names, signatures and arithmetic follow the subset of the reference's pbr.cl that its live shader reaches
 * (radiance/shader/pbr.cl:6-13, 31-37, 41-96, 171-186, 268-385, 387-425), so a closest-hit shader written against that
 * library shades identically.  The unreached variants of the reference file (Disney diffuse, the remapped GGX pair, the
 * non-transmissive sampler of the stale shader2.cl) are not provided.
 */
#ifndef RDX_PBR_CL
#define RDX_PBR_CL

#include "math.cl"

#define PI 3.14159265359f

/* GGX / Trowbridge-Reitz normal distribution with alpha = roughness^2 */
float D_GGX(float dotNH, float roughness)
{
    const float a2 = (roughness * roughness) * (roughness * roughness);
    const float den = dotNH * dotNH * (a2 - 1.0f) + 1.0f;
    return a2 / (PI * den * den);
}

/* Schlick's Fresnel approximation around F0 = mix(0.04, albedo, metallic) */
float3 F_Schlick(float cosTheta, float metallic, float3 albedo)
{
    const float3 F0 = mix((float3)(0.04f), albedo, metallic);
    return F0 + (1.0f - F0) * pow(1.0f - cosTheta, 5.0f);
}

/* spherical-coordinate helpers of a direction in the local shading frame (z = normal) */
inline float Cos2Theta(float3 w) { return w.z * w.z; }
inline float Sin2Theta(float3 w) { return max(0.0f, 1.0f - Cos2Theta(w)); }
inline float SinTheta(float3 w) { return sqrt(Sin2Theta(w)); }
inline float CosPhi(float3 w) { const float st = SinTheta(w); return (st == 0.0f) ? 1.0f : clamp(w.x / st, -1.0f, 1.0f); }
inline float SinPhi(float3 w) { const float st = SinTheta(w); return (st == 0.0f) ? 0.0f : clamp(w.y / st, -1.0f, 1.0f); }
inline float Tan2Theta(float3 w) { return Sin2Theta(w) / Cos2Theta(w); }

/* Smith's Lambda for an isotropic GGX lobe of width a (pbrt's form) */
float Lambda(float3 w, float a)
{
    const float t2 = Tan2Theta(w);
    if (isinf(t2)) return 0.0f;
    const float cp = CosPhi(w) * a, sp = SinPhi(w) * a;
    const float alpha2 = cp * cp + sp * sp;
    return (sqrt(1.0f + alpha2 * t2) - 1.0f) / 2.0f;
}

/* height-correlated Smith masking-shadowing, both directions taken to the frame of N by inverting its tangent frame */
float G_pbrt(float3 wo, float3 wi, float3 N, float roughness)
{
    mat4x4 frame, inv;
    GetNormalSpace(N, &frame);
    InverseMat4x4(&frame, &inv);
    vec4 o4 = (vec4)(wo, 0.0f), i4 = (vec4)(wi, 0.0f), lo, li;
    MultiplyMat4Vec4(&inv, &o4, &lo);
    MultiplyMat4Vec4(&inv, &i4, &li);
    if (li.z < 0 || lo.z < 0) return 0.0f;
    return 1 / (1 + Lambda(li.xyz, roughness) + Lambda(lo.xyz, roughness));
}

float3 reflect(float3 in, float3 N) { return -in + 2 * dot(in, N) * N; }

/* refraction of V about the half vector H with relative index eta; total internal reflection mirrors */
float3 refract(float3 V, float3 H, float eta)
{
    const float ci = dot(H, V);
    const float s2i = max(0.0f, 1.0f - ci * ci);
    const float s2t = s2i / (eta * eta);
    if (1.0f - s2t < 0.0f) return (H * ci - V) / eta;
    const float ct = sqrt(1.0f - s2t);
    return -V / eta + (ci / eta - ct) * H;
}

/* Cook-Torrance BRDF times N.L with a Lambert term scaled by (1 - F)(1 - metallic)(1 - transmission); `ior` is part of the
 * reference signature and unused (pbr.cl:268-287) */
float3 microfacetBRDF(float3 L, float3 V, float3 N, float3 albedo, float metallicness, float roughness, float transmission, float ior)
{
    const float3 H = normalize(V + L);
    const float NoV = clamp(dot(N, V), 0.0f, 1.0f);
    const float NoL = clamp(dot(N, L), 0.0f, 1.0f);
    const float NoH = clamp(dot(N, H), 0.0f, 1.0f);
    const float VoH = clamp(dot(V, H), 0.0f, 1.0f);
    const float3 F = F_Schlick(VoH, metallicness, albedo);
    const float D = D_GGX(NoH, roughness);
    const float G = G_pbrt(V, L, N, roughness);
    const float3 spec = (D * G * F) / max(4.0f * NoV * NoL, 0.001f);
    const float3 notSpec = (1.0f - F) * (1.0f - metallicness) * (1.0f - transmission);
    const float3 diff = notSpec * (albedo / PI);
    return (diff + spec) * NoL;
}

/* local (theta, phi) direction rotated into the tangent frame of n */
float3 rdx_frame_dir(float3 n, float theta, float phi)
{
    mat4x4 frame;
    GetNormalSpace(n, &frame);
    vec4 l = (vec4)(sin(theta) * cos(phi), sin(theta) * sin(phi), cos(theta), 0.0f), w;
    MultiplyMat4Vec4(&frame, &l, &w);
    return w.xyz;
}

/* importance sampling of the next direction: random.z picks the lobe -- [0.5, 1) specular (GGX half vector, mirrored V),
 * [0, 0.5) diffuse (cosine lobe) or, while 2 z < transmission, refraction through a GGX half vector -- each with the x2
 * weight that compensates the 1/2 probability; returns L and the throughput factor (pbr.cl:289-385) */
float3 sampleMicrofacetBRDF_transm(float3 V, float3 N, float3 baseColor, float metallicness, float roughness,
                                   float transmission, float ior, float3 random, float3* nextFactor)
{
    const float a = roughness * roughness;
    const float ggxTheta = acos(sqrt((1.0f - random.y) / (1.0f + (a * a - 1.0f) * random.y)));
    const float phi = 2.0f * PI * random.x;
    if (random.z < 0.5f) {
        if (2.0f * random.z < transmission) {
            float3 fn = N;
            float eta = ior;
            if (dot(V, N) < 0.0f) { fn = -N; eta = 1.0f / ior; }
            const float3 H = rdx_frame_dir(fn, ggxTheta, phi);
            const float3 L = refract(V, H, eta);
            const float NoV = clamp(dot(fn, V), 0.0f, 1.0f);
            const float NoH = clamp(dot(fn, H), 0.0f, 1.0f);
            const float VoH = clamp(dot(V, H), 0.0f, 1.0f);
            const float3 F = F_Schlick(VoH, metallicness, baseColor);
            const float G = G_pbrt(V, -L, fn, roughness);
            *nextFactor = baseColor * (1.0f - F) * G * VoH / max(NoH * NoV, 0.001f) * 2.0f;
            return L;
        }
        const float3 L = rdx_frame_dir(N, acos(sqrt(random.y)), phi);
        const float3 H = normalize(V + L);
        const float VoH = clamp(dot(V, H), 0.0f, 1.0f);
        const float3 F = F_Schlick(VoH, metallicness, baseColor);
        *nextFactor = (1.0f - F) * (1.0f - metallicness) * baseColor * 2.0f;
        return L;
    }
    const float3 H = rdx_frame_dir(N, ggxTheta, phi);
    const float3 L = reflect(V, H);
    const float NoV = clamp(dot(N, V), 0.0f, 1.0f);
    const float NoH = clamp(dot(N, H), 0.0f, 1.0f);
    const float VoH = clamp(dot(V, H), 0.0f, 1.0f);
    const float3 F = F_Schlick(VoH, metallicness, baseColor);
    const float G = G_pbrt(V, L, N, roughness);
    *nextFactor = F * G * VoH / max(NoH * NoV, 0.001f) * 2.0f;
    return L;
}

/* buffers bound to the raygen kernel (host twins: include/core.h) */
struct Material {
    float4 albedo;
    float metallic, roughness, transmission, ior;
    int albedoTexIdx, metallicTexIdx, roughnessTexIdx, normalTexIdx;      /* layer of the texture array, -1 = none */
};
struct MeshInfo { int vertexOffset, indexOffset, uvOffset, normalOffset, materialIndex, _pad0, _pad1, _pad2; };   /* offsets in floats / indices */
struct DirLight { float4 direction; float4 color; };
struct SceneProperties { uint4 lightCount; struct DirLight lights[5]; };

#endif

/*
 * ref_gpu_dev.cl -- TEST INFRASTRUCTURE (oracle side), never part of the product path.
 *
 * Translation unit 1 of the "reference on the GPU" oracle: it textually includes the REAL reference
 * device code where it lies (`#include "shader.cl"` resolves through -I/root/reference/samples and
 * -I/root/reference/radiance/shader, see oracle/Makefile) and adds thin batch wrappers around the
 * reference's own functions.  Compiled by ROCm clang's OpenCL C front end for gfx950 and linked against
 * ROCm's own OpenCL builtin library (opencl.bc / ocml.bc / ockl.bc) -- no stand-ins for anything.
 *
 * The wrappers only marshal arrays <-> the reference's argument types; every arithmetic step is the
 * reference's.  They keep the `image2d_array_t, sampler_t` tail of the reference's signatures; the
 * __kernel entry points live in ref_gpu_kern.cl (unit 2), which passes null descriptors for both (the
 * live shader never samples: samples/shader.cl:375-449 are stubs).
 *
 * Record layouts written here are the C-ABI test-seam records of include/rdx.h:
 *   rdx_hit     28 words: hitPoint[3] distance prim inst custom sbtOffset bary[3] hit transform[16]
 *   rdx_payload 13 words: color[3] hit nextFactor[3] nextRayOrigin[3] nextRayDirection[3]
 */
#include "shader.cl"

static void rdxref_scene(struct SceneData* sd,
    __global struct PhysicalCamera* cam, __global struct SceneProperties* scene, __global struct MeshInfo* meshInfo,
    __global float* vertex, __global uint* index, __global float* uv, __global float* normal,
    __global struct Material* materials, __global struct AccelStruct* tlas, int depth, uint frameID, uint debug)
{
    sd->camData = cam; sd->scene = scene; sd->meshInfoData = meshInfo; sd->vertexData = vertex;
    sd->indexData = index; sd->uvData = uv; sd->normalData = normal; sd->materials = materials;
    sd->topLevel = tlas; sd->depth = depth; sd->frameID = frameID; sd->debug = debug;
}

/* traceRay's traversal half (radiance/shader/radiance.cl:254-262 up to the dispatch): HitData of one ray per
 * work-item.  sbt = 1 closest hit (no any-hit shader on row 1), 2 = shadow row (anyShadow ends the walk). */
void rdxref_trace(__global struct AccelStruct* tlas, __global const float* org, __global const float* dir, uint n,
                  float tmin, float tmax, int sbt, __global uint* out, image2d_array_t img, sampler_t smp)
{
    uint i = get_global_id(0);
    if (i >= n) return;
    struct HitData h;
    h.hitPoint = 0.0f; h.primitiveIndex = 0; h.instanceIndex = 0; h.instanceCustomIndex = 0; h.instanceSBTOffset = 0;
    h.barycentric = 0.0f; h.transform = 0.0f;
    h.distance = FLT_MAX;                                    /* radiance.cl:265 */
    struct Payload p;
    struct SceneData sd;
    rdxref_scene(&sd, 0, 0, 0, 0, 0, 0, 0, 0, tlas, 0, 0, 0);
    float3 o = (float3)(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
    float3 d = (float3)(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    bool hit = intersectTop(tlas, o, d, tmin, tmax, &h, sbt, &p, &sd, img, smp);
    __global uint* r = out + 28 * (size_t)i;
    r[0] = as_uint(h.hitPoint.x); r[1] = as_uint(h.hitPoint.y); r[2] = as_uint(h.hitPoint.z);
    r[3] = as_uint(h.distance);
    r[4] = h.primitiveIndex; r[5] = h.instanceIndex; r[6] = h.instanceCustomIndex; r[7] = h.instanceSBTOffset;
    r[8] = as_uint(h.barycentric.x); r[9] = as_uint(h.barycentric.y); r[10] = as_uint(h.barycentric.z);
    r[11] = hit ? 1u : 0u;
    float t[16];
    vstore16(h.transform, 0, t);
    for (int k = 0; k < 16; k++) r[12 + k] = as_uint(t[k]);
}

/* closest-hit `material` (samples/shader.cl:482-541) through the reference's own SBT switch, on captured hits.
 * The reference draws its bounce RNG from get_global_id(0) (shader.cl:523): work-item i IS pixel i. */
void rdxref_material(__global const uint* hits, __global const float* raydir, __global const uint* frameIDs,
    __global const int* depths, uint n,
    __global struct PhysicalCamera* cam, __global struct SceneProperties* scene, __global struct MeshInfo* meshInfo,
    __global float* vertex, __global uint* index, __global float* uv, __global float* normal,
    __global struct Material* materials, __global struct AccelStruct* tlas,
    __global uint* out, image2d_array_t img, sampler_t smp)
{
    uint i = get_global_id(0);
    if (i >= n) return;
    __global const uint* r = hits + 28 * (size_t)i;
    struct HitData h;
    h.hitPoint = (float3)(as_float(r[0]), as_float(r[1]), as_float(r[2]));
    h.distance = as_float(r[3]);
    h.primitiveIndex = r[4]; h.instanceIndex = r[5]; h.instanceCustomIndex = r[6]; h.instanceSBTOffset = r[7];
    h.barycentric = (float3)(as_float(r[8]), as_float(r[9]), as_float(r[10]));
    float t[16];
    for (int k = 0; k < 16; k++) t[k] = as_float(r[12 + k]);
    h.transform = vload16(0, t);
    struct SceneData sd;
    rdxref_scene(&sd, cam, scene, meshInfo, vertex, index, uv, normal, materials, tlas, depths[i], frameIDs[i], 0);
    struct Payload p;
    p.color = 0.0f; p.hit = false; p.nextFactor = 1.0f; p.nextRayOrigin = 0.0f;
    p.nextRayDirection = (float3)(raydir[3 * i], raydir[3 * i + 1], raydir[3 * i + 2]);
    callHit(1, &p, &h, &sd, img, smp);
    __global uint* w = out + 13 * (size_t)i;
    w[0] = as_uint(p.color.x); w[1] = as_uint(p.color.y); w[2] = as_uint(p.color.z);
    w[3] = p.hit ? 1u : 0u;
    w[4] = as_uint(p.nextFactor.x); w[5] = as_uint(p.nextFactor.y); w[6] = as_uint(p.nextFactor.z);
    w[7] = as_uint(p.nextRayOrigin.x); w[8] = as_uint(p.nextRayOrigin.y); w[9] = as_uint(p.nextRayOrigin.z);
    w[10] = as_uint(p.nextRayDirection.x); w[11] = as_uint(p.nextRayDirection.y); w[12] = as_uint(p.nextRayDirection.z);
}

/* generateRay (samples/shader.cl:111-173): the pixel is get_global_id(0) there, so work-item i = pixel i. */
void rdxref_generate(__global struct PhysicalCamera* cam, __global const uint* rnd3, uint n,
                     __global float* org, __global float* dir)
{
    uint i = get_global_id(0);
    if (i >= n) return;
    uint3 rin = (uint3)(rnd3[3 * i], rnd3[3 * i + 1], rnd3[3 * i + 2]);
    float3 o, d;
    generateRay(cam, rin, &o, &d);
    org[3 * i] = o.x; org[3 * i + 1] = o.y; org[3 * i + 2] = o.z;
    dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
}

/* intersectAABB (radiance.cl:195-208): in = o[3] d[3] lo[3] hi[3] per item */
void rdxref_aabb(__global const float* in, uint n, __global uint* out)
{
    uint i = get_global_id(0);
    if (i >= n) return;
    __global const float* p = in + 12 * (size_t)i;
    out[i] = intersectAABB((float3)(p[0], p[1], p[2]), (float3)(p[3], p[4], p[5]),
                           (float3)(p[6], p[7], p[8]), (float3)(p[9], p[10], p[11])) ? 1u : 0u;
}

/* intersectTriangle (radiance.cl:211-251): in = o[3] d[3] per item, tris = one {0,1,2,i} record per item,
 * verts = 3 float4 per item; out = hit, t, point[3], bary[3] (8 words) */
void rdxref_triangle(__global const float* in, __global const struct Triangle* tris, __global float4* verts, uint n,
                     __global uint* out)
{
    uint i = get_global_id(0);
    if (i >= n) return;
    __global const float* p = in + 6 * (size_t)i;
    float3 pt = 0.0f, bary = 0.0f;
    float t = 0.0f;
    bool hit = intersectTriangle((float3)(p[0], p[1], p[2]), (float3)(p[3], p[4], p[5]),
                                 tris + i, verts + 3 * (size_t)i, &pt, &t, &bary);
    __global uint* w = out + 8 * (size_t)i;
    w[0] = hit ? 1u : 0u; w[1] = as_uint(t);
    w[2] = as_uint(pt.x); w[3] = as_uint(pt.y); w[4] = as_uint(pt.z);
    w[5] = as_uint(bary.x); w[6] = as_uint(bary.y); w[7] = as_uint(bary.z);
}

/* microfacetBRDF (pbr.cl:268-287) and sampleMicrofacetBRDF_transm (pbr.cl:289-385):
 * in = L[3] V[3] N[3] albedo[3] metallic roughness transmission ior random[3] (19 floats);
 * out = brdf[3] nextDir[3] nextFactor[3] */
void rdxref_brdf(__global const float* in, uint n, __global float* out)
{
    uint i = get_global_id(0);
    if (i >= n) return;
    __global const float* p = in + 19 * (size_t)i;
    float3 L = (float3)(p[0], p[1], p[2]), V = (float3)(p[3], p[4], p[5]), N = (float3)(p[6], p[7], p[8]);
    float3 albedo = (float3)(p[9], p[10], p[11]);
    float3 rnd = (float3)(p[16], p[17], p[18]);
    float3 f = microfacetBRDF(L, V, N, albedo, p[12], p[13], p[14], p[15]);
    float3 nf = 0.0f;
    float3 nd = sampleMicrofacetBRDF_transm(V, N, albedo, p[12], p[13], p[14], p[15], rnd, &nf);
    __global float* w = out + 9 * (size_t)i;
    w[0] = f.x; w[1] = f.y; w[2] = f.z; w[3] = nd.x; w[4] = nd.y; w[5] = nd.z; w[6] = nf.x; w[7] = nf.y; w[8] = nf.z;
}

/* the whole `raygen` megakernel (samples/shader.cl:175-305), called as a function so that unit 2 can supply the
 * two opaque arguments; get_global_id(0) inside it is the calling kernel's */
void rdxref_raygen(__global struct RayTraceProperties* RTProp, __global float* imageScratch, __global uchar* image,
    __global struct PhysicalCamera* camData, __global struct SceneProperties* scene, __global struct MeshInfo* meshInfoData,
    __global float* vertexData, __global uint* indexData, __global float* uvData, __global float* normalData,
    __global struct Material* materials, __global struct AccelStruct* topLevel, uint npixels,
    image2d_array_t img, sampler_t smp)
{
    if (get_global_id(0) >= npixels) return;
    raygen(RTProp, imageScratch, image, camData, scene, meshInfoData, vertexData, indexData, uvData, normalData,
           materials, img, smp, topLevel);
}

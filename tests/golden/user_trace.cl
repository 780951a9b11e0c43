/* user_trace.cl -- test program for the product's own device library (radiance-ray-tracing_amd/shader/radiance.cl): a user
 * `raygen` that traces the rays it finds in descriptor slot 6 (six floats per ray: origin, direction) through traceRay()
 * and writes the HitData its closest-hit callback received into slot 1 (28 words per ray, the layout of tests/oracle_bind.py
 * HIT_DTYPE).  Own code; `#include "radiance.cl"` resolves to the library's file, no reference source is involved. */
#include "radiance.cl"

struct Payload { struct HitData h; uint hit; };
struct SceneData { int unused; };

void callHit(int sbtRecordOffset, struct Payload* payload, struct HitData* hitData, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    payload->h = *hitData;
    payload->hit = 1u;
}
void callMiss(int missIndex, struct Payload* payload, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    payload->hit = 0u;
}
void callAnyHit(bool* cont, int sbtRecordOffset, struct Payload* payload, struct HitData* hitData, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    if (sbtRecordOffset == 2) *cont = false;        /* shadow-type rays stop at the first accepted candidate */
}

__kernel void raygen(__global uint* props, __global float* out, __global uchar* image, __global float* cam, __global float* scene,
                     __global int* meshInfo, __global float* rays, __global uint* indexData, __global float* uvData, __global float* normalData,
                     __global float* materials, image2d_array_t textures, sampler_t sampler, __global struct AccelStruct* topLevel)
{
    const int i = get_global_id(0);
    const uint n = props[1];                        /* batchSize carries the ray count, depth the sbtRecordOffset */
    if ((uint)i >= n) return;
    const float3 o = vload3(2 * i, rays), d = vload3(2 * i + 1, rays);
    struct Payload p;
    struct SceneData sd;
    p.hit = 0u;
    traceRay(topLevel, (int)props[2], 3, o, d, 0.001f, 1000.0f, &p, &sd, textures, sampler);
    __global float* w = out + 28 * i;
    __global uint* wu = (__global uint*)w;
    if (p.hit) {
        w[0] = p.h.hitPoint.x; w[1] = p.h.hitPoint.y; w[2] = p.h.hitPoint.z; w[3] = p.h.distance;
        wu[4] = p.h.primitiveIndex; wu[5] = p.h.instanceIndex; wu[6] = p.h.instanceCustomIndex; wu[7] = p.h.instanceSBTOffset;
        w[8] = p.h.barycentric.x; w[9] = p.h.barycentric.y; w[10] = p.h.barycentric.z;
        wu[11] = 1u;
        vstore16(p.h.transform, 0, w + 12);
    } else {
        for (int k = 0; k < 28; ++k) wu[k] = 0u;
        w[3] = FLT_MAX;
    }
}

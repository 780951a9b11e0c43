/* user_stages.cl -- a user program written against the reference's shader INTERFACE (the payload / scene-data structs a raygen
 * loop and its stage functions share, samples/sbt.json's rows), with its own closest-hit and miss shaders (user_material.inc, user_environment.inc).  Own text.
 * Run with rdx_set_option("user_stages", 2): the program's stage functions on the product's wavefront pipeline; the `raygen`
 * below is then not used (the pipeline's own generate / accumulate stages are the stock raygen loop). */
#include "radiance.cl"
#include "pbr.cl"

struct Payload { float3 color; bool hit; float3 nextFactor; float3 nextRayOrigin; float3 nextRayDirection; };
struct PhysicalCamera { float widthPixel, heightPixel, focalLength, sensorWidth, focalDistance, fStop, x, y, z, wx, wy, wz; };
struct SceneData {
    __global struct PhysicalCamera* camData; __global struct SceneProperties* scene; __global struct MeshInfo* meshInfoData;
    __global float* vertexData; __global uint* indexData; __global float* uvData; __global float* normalData;
    __global struct Material* materials; __global struct AccelStruct* topLevel;
    int depth; unsigned int frameID; unsigned int debug;
};

void material(struct Payload* payload, struct HitData* hitData, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
#include "user_material.inc"

void environment(struct Payload* payload, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
#include "user_environment.inc"

void callHit(int sbtRecordOffset, struct Payload* payload, struct HitData* hitData, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    const int row = (int)hitData->instanceSBTOffset + sbtRecordOffset;
    if (row == 1) material(payload, hitData, sceneData, imageArray, sampler);
    else if (row == 2) { payload->hit = true; payload->color = 0.0f; }            /* the shadow row's closest-hit */
}
void callAnyHit(bool* cont, int sbtRecordOffset, struct Payload* payload, struct HitData* hitData, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    if ((int)hitData->instanceSBTOffset + sbtRecordOffset == 2) *cont = false;   /* first candidate ends a shadow ray */
}
void callMiss(int missIndex, struct Payload* payload, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    if (missIndex == 3) environment(payload, sceneData, imageArray, sampler);
    else if (missIndex == 4) { payload->hit = false; payload->color = 1.0f; }    /* the shadow ray's miss */
}

__kernel void raygen(__global struct RayTraceProperties* RTProp, __global float* imageScratch, __global uchar* image,
                     __global struct PhysicalCamera* camData, __global struct SceneProperties* scene, __global struct MeshInfo* meshInfoData,
                     __global float* vertexData, __global uint* indexData, __global float* uvData, __global float* normalData,
                     __global struct Material* materials, image2d_array_t imageArray, sampler_t sampler, __global struct AccelStruct* topLevel)
{
}

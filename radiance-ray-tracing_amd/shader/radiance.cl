/* radiance.cl -- the ray-tracing device library a user's OpenCL C shader program includes (product-owned text).
 *
 * Interface = the reference's (radiance/shader/radiance.cl:8-28, 254-275): struct HitData, the three callbacks the user's
 * translation unit defines (callHit / callMiss / callAnyHit, normally generated from sbt.json by tools/genSBT.py), and
 * traceRay().  A program written against the reference's library compiles against this one unchanged and, compiled under
 * the pinned floating-point contract, computes the same HitData bit for bit
 * (tests/test_gpu_parity.py::test_user_program_traces_rays_with_the_product_library).
 *
 * This is the megakernel-side library: traceRay walks the acceleration-structure blob the way the reference does --
 * exhaustively, left child first, first strictly smaller t wins, every accepted candidate offered to the any-hit callback
 * (radiance.cl:41-192) -- because a user program may depend on any of it.  The stock shader program does not come
 * through here: it is served by the hand-written wavefront pipeline (csrc/kernels.hip).
 */
#ifndef RDX_RADIANCE_CL
#define RDX_RADIANCE_CL

#include "data.cl"
#include "math.cl"

struct Payload;                 /* defined by the user program */
struct SceneData;

struct HitData {
    float3 hitPoint;                    /* object space */
    float distance;                     /* ray parameter; the direction is not renormalised in object space, so this is world-parametric */
    unsigned int primitiveIndex;        /* index of the triangle in the caller's index buffer */
    unsigned int instanceIndex;         /* index of the instance in the caller's instance list */
    unsigned int instanceCustomIndex;
    unsigned int instanceSBTOffset;
    float3 barycentric;                 /* (1 - b1 - b2, b1, b2) */
    mat4x4 transform;                   /* object -> world, row-major */
};

/* defined by the user program (tools/genSBT.py emits them from sbt.json) */
void callHit(int sbtRecordOffset, struct Payload* payload, struct HitData* hitData,
             struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler);
void callMiss(int missIndex, struct Payload* payload,
              struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler);
void callAnyHit(bool* cont, int sbtRecordOffset, struct Payload* payload, struct HitData* hitData,
                struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler);

#define BVH_TOP_STACK_SIZE 8
#define BVH_BOT_STACK_SIZE 100

/* slab test by division; a hit needs tFar > max(tNear, 0) -- no upper bound, no best-t culling (radiance.cl:195-208) */
bool intersectAABB(float3 rayOrigin, float3 rayDir, float3 boxMin, float3 boxMax)
{
    const float3 ta = (boxMin - rayOrigin) / rayDir;
    const float3 tb = (boxMax - rayOrigin) / rayDir;
    const float3 lo = min(ta, tb);
    const float3 hi = max(ta, tb);
    const float tNear = max(max(lo.x, lo.y), lo.z);
    const float tFar = min(min(hi.x, hi.y), hi.z);
    return tFar > max(tNear, 0.0f);
}

/* Moeller-Trumbore without epsilon and without back-face culling: only an exactly zero determinant is "parallel"
 * (radiance.cl:211-251) */
bool intersectTriangle(float3 origin, float3 direction,
                       __global const struct Triangle* triangle, __global const Vertex* vertexList,
                       float3* intersectPoint, float* distance, float3* bary)
{
    const float3 v0 = vertexList[triangle->idx0].xyz;
    const float3 e1 = vertexList[triangle->idx1].xyz - v0;
    const float3 e2 = vertexList[triangle->idx2].xyz - v0;
    const float3 p = cross(direction, e2);
    const float det = dot(e1, p);
    if (det == 0) return false;
    const float inv = 1.0f / det;
    const float3 s = origin - v0;
    const float b1 = inv * dot(s, p);
    const float3 q = cross(s, e1);
    const float b2 = inv * dot(direction, q);
    const float t = inv * dot(e2, q);
    if (b1 < 0 || b1 > 1) return false;
    if (b2 < 0 || b1 + b2 > 1) return false;
    if (!(t > 0)) return false;
    *distance = t;
    *intersectPoint = origin + direction * t;
    *bary = (float3)(1 - b1 - b2, b1, b2);
    return true;
}

/* one bottom-level structure, ray in its object space (radiance.cl:41-108) */
bool intersectBot(__global const struct AccelStruct* accelStruct, float3 origin, float3 direction,
                  float Tmin, float Tmax, struct HitData* hitData, bool* cont, int sbtRecordOffset,
                  struct Payload* payload, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    __global const struct BVHNode* nodes = RDX_NODES(accelStruct);
    __global const struct Triangle* tris = RDX_TRIANGLES(accelStruct);
    __global const Vertex* verts = RDX_VERTICES(accelStruct);
    unsigned int pending[BVH_BOT_STACK_SIZE];
    int top = 0;
    bool any = false;
    pending[top++] = 0;
    while (top) {
        __global const struct BVHNode* n = nodes + pending[--top];
        if (!RDX_IS_LEAF(n)) {
            if (intersectAABB(origin, direction, n->lo.xyz, n->hi.xyz)) {
                pending[top++] = n->w1;         /* the right child waits, the left one is next */
                pending[top++] = n->w0;
                if (top > BVH_BOT_STACK_SIZE) { printf("ERROR: Bottom AS stack overflow\n"); return false; }
            }
            continue;
        }
        if (n->w2 != TYPE_TRIG) continue;
        const unsigned int count = RDX_COUNT(n);
        for (unsigned int i = 0; i < count; ++i) {
            __global const struct Triangle* f = tris + n->w1 + i;
            float3 where, b;
            float t;
            if (intersectTriangle(origin, direction, f, verts, &where, &t, &b) && t < hitData->distance && t > Tmin && t < Tmax) {
                hitData->distance = t;
                hitData->hitPoint = where;
                hitData->primitiveIndex = f->primID;
                hitData->barycentric = b;
                any = true;
                callAnyHit(cont, sbtRecordOffset, payload, hitData, sceneData, imageArray, sampler);
                if (!*cont) return any;
            }
        }
    }
    return any;
}

/* the top-level structure: every instance of every visited leaf is entered -- its matrix inverted, the ray taken to its
 * object space with w = 1 / w = 0 -- and leaves its identity in hitData only if it produced a candidate (radiance.cl:110-192) */
bool intersectTop(__global const struct AccelStruct* accelStruct, float3 origin, float3 direction,
                  float Tmin, float Tmax, struct HitData* hitData, int sbtRecordOffset,
                  struct Payload* payload, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    __global const struct BVHNode* nodes = RDX_NODES(accelStruct);
    __global const struct Instance* insts = RDX_INSTANCES(accelStruct);
    unsigned int pending[BVH_TOP_STACK_SIZE];
    int top = 0;
    bool any = false, cont = true;
    pending[top++] = 0;
    while (top) {
        __global const struct BVHNode* n = nodes + pending[--top];
        if (!RDX_IS_LEAF(n)) {
            if (intersectAABB(origin, direction, n->lo.xyz, n->hi.xyz)) {
                pending[top++] = n->w1;
                pending[top++] = n->w0;
                if (top > BVH_TOP_STACK_SIZE) { printf("ERROR: Top AS stack overflow\n"); return false; }
            }
            continue;
        }
        if (n->w2 != TYPE_INST) continue;
        const unsigned int count = RDX_COUNT(n);
        for (unsigned int i = 0; i < count; ++i) {
            __global const struct Instance* I = insts + n->w1 + i;
            __global const struct AccelStruct* blas = (__global const struct AccelStruct*)(((__global const char*)accelStruct) + I->instanceOffset);
            /* what a miss in this instance puts back */
            const mat4x4 keepM = hitData->transform;
            const unsigned int keepId = hitData->instanceIndex, keepCustom = hitData->instanceCustomIndex, keepSbt = hitData->instanceSBTOffset;
            vec4 o4 = (vec4)(origin, 1.0f), d4 = (vec4)(direction, 0.0f), oL, dL;
            mat4x4 inverse;
            Vec4ToMat4x4(I->r0, I->r1, I->r2, I->r3, &hitData->transform);
            InverseMat4x4(&hitData->transform, &inverse);
            MultiplyMat4Vec4(&inverse, &o4, &oL);
            MultiplyMat4Vec4(&inverse, &d4, &dL);
            hitData->instanceIndex = I->instanceID;
            hitData->instanceCustomIndex = I->customInstanceID;
            hitData->instanceSBTOffset = I->SBTOffset;
            const bool found = intersectBot(blas, oL.xyz, dL.xyz, Tmin, Tmax, hitData, &cont, sbtRecordOffset, payload, sceneData, imageArray, sampler);
            any = any || found;
            if (!cont) return any;
            if (!found) {
                hitData->transform = keepM;
                hitData->instanceIndex = keepId; hitData->instanceCustomIndex = keepCustom; hitData->instanceSBTOffset = keepSbt;
            }
        }
    }
    return any;
}

#ifdef RDX_WAVEFRONT_STAGES
/* ---- stage mode (csrc/user_stages.cpp): the user's closest-hit / miss functions run inside the wavefront pipeline ------------
 * The pipeline's own kernels trace; a hit shader that calls traceRay itself (the stock `material` asks for a shadow ray in the
 * middle of the function, shader.cl:499-509) is run TWICE around that traversal stage: pass 0 RECORDS the query -- traceRay
 * notes origin / direction / bounds / SBT indices and answers "miss" so that the shader runs to its end; its output is
 * dropped -- pass 1 REPLAYS it: traceRay answers with what the traversal stage found and dispatches the hit or the miss
 * callback as the megakernel's traceRay would.  Shaders are pure functions of their inputs (the RNG is a hash of frame, pixel,
 * depth), so pass 1 computes exactly what the megakernel computes.  The context travels in front of the SceneData object the
 * stage kernel hands to the shader -- the one pointer a shader passes on to traceRay unchanged. */
struct RdxStageCtx {
    uint mode;                  /* 0 record, 1 replay */
    uint calls;                 /* nested traceRay calls of this shader invocation */
    uint answer;                /* replay: the recorded query found a candidate */
    uint error;                 /* 1: a second nested trace; 2: a query the traversal stage cannot serve */
    float4 origin;              /* recorded query: origin | Tmin */
    float4 direction;           /* direction | Tmax */
    int sbtRecordOffset, missIndex;
    uint pixel;                 /* what get_global_id(0) is in the megakernel: the pixel this path belongs to */
    uint pad1;
};
#define RDX_STAGE_CTX(sd) ((struct RdxStageCtx*)(((char*)(sd)) - sizeof(struct RdxStageCtx)))
/* The stage kernel's work-items are compacted paths, not pixels; the user's text is compiled with
 * `#define get_global_id(d) rdx_stage_gid((d), sceneData)` (user_shader.cpp), which works wherever the stage functions' own
 * `sceneData` parameter is in scope -- as in the stock shader (shader.cl:522) -- and otherwise fails to compile, upon which the
 * runtime falls back to the megakernel. */
#define rdx_stage_gid(d, sd) ((size_t)((d) == 0 ? RDX_STAGE_CTX(sd)->pixel : 0u))

void traceRay(__global struct AccelStruct* topLevel, int sbtRecordOffset, int missIndex,
              float3 origin, float3 direction, float Tmin, float Tmax,
              struct Payload* payload, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    struct RdxStageCtx* c = RDX_STAGE_CTX(sceneData);
    c->calls++;
    if (c->calls > 1u) { c->error = 1u; callMiss(missIndex, payload, sceneData, imageArray, sampler); return; }
    if (c->mode == 0u) {
        c->origin = (float4)(origin, Tmin);
        c->direction = (float4)(direction, Tmax);
        c->sbtRecordOffset = sbtRecordOffset; c->missIndex = missIndex;
        callMiss(missIndex, payload, sceneData, imageArray, sampler);
        return;
    }
    if (c->answer) {
        /* an any-hit-terminated query: which candidate ended the walk is not recorded (the stock `shadow` does not look) */
        struct HitData hd;
        hd.hitPoint = (float3)(0.0f); hd.distance = 0.0f; hd.primitiveIndex = 0u; hd.instanceIndex = 0u; hd.instanceCustomIndex = 0u;
        hd.instanceSBTOffset = 0u; hd.barycentric = (float3)(0.0f); hd.transform = (mat4x4)(0.0f);
        callHit(sbtRecordOffset, payload, &hd, sceneData, imageArray, sampler);
    } else callMiss(missIndex, payload, sceneData, imageArray, sampler);
}
#else
/* closest hit over the whole scene, then the hit or the miss callback (radiance.cl:254-275) */
void traceRay(__global struct AccelStruct* topLevel, int sbtRecordOffset, int missIndex,
              float3 origin, float3 direction, float Tmin, float Tmax,
              struct Payload* payload, struct SceneData* sceneData, image2d_array_t imageArray, sampler_t sampler)
{
    struct HitData hitData;
    hitData.distance = FLT_MAX;
    if (intersectTop(topLevel, origin, direction, Tmin, Tmax, &hitData, sbtRecordOffset, payload, sceneData, imageArray, sampler))
        callHit(sbtRecordOffset, payload, &hitData, sceneData, imageArray, sampler);
    else
        callMiss(missIndex, payload, sceneData, imageArray, sampler);
}
#endif

#endif

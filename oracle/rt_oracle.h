/*
 * rt_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the hot path of zekailin00/Radiance-Ray-Tracing:
 *   host  : binned-SAH BVH builder + BLAS/TLAS blob packers
 *           (radiance/src/bvh.cpp:46-597, radiance/src/radiance.cpp:318-425)
 *   device: traceRay / intersectTop / intersectBot / intersectAABB /
 *           intersectTriangle (radiance/shader/radiance.cl:41-275),
 *           math + PBR helpers (radiance/shader/math.cl, pbr.cl),
 *           raygen / generateRay / material / SBT switches (samples/shader.cl).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline.  The product
 * (radiance-ray-tracing_amd/) never links, imports or calls it.
 *
 * PARITY PIN STATUS
 *   The reference has no tests, golden vectors or fixtures.
 *   Its HOST side (bvh.cpp / radiance.cpp) needs the assimp headers (empty
 *   submodule): unbuildable here, so the SAH builder and the blob packers in this
 *   file are "parity unpinned" (a restatement by reading).
 *   Its DEVICE side is pinned to the real thing: oracle/Makefile compiles the
 *   reference's own samples/shader.cl + radiance/shader/ *.cl, where they lie, with
 *   ROCm clang's OpenCL C front end for gfx950 and links ROCm's own OpenCL builtin
 *   library -- nothing stood in for -- into oracle/_ref/ref_shader_gfx950_p.co,
 *   which tests/refgpu_bind.py runs on the MI355X.  Outputs of that code object
 *   (slab test, triangle test, HitData of ray batches, `material` payloads,
 *   primary rays, BRDF values, whole imageScratch / RGBA8 frames) are committed as
 *   tests/golden/refgpu_*.npz (tests/golden/make_golden_gpu.py) and this oracle is
 *   checked against them in the CPU suite; the HIP product is checked against the
 *   same fixtures AND against the live code object in the GPU suite.
 *   Also pinned by real reference code, on the CPU: random_pcg3d, InverseMat4x4,
 *   MultiplyMat4Vec4 / Mat4, D_GGX (x86 build of shader.cl, oracle/ref_harness.c,
 *   tests/golden/ref_kat.npz) and the SBT row mapping of tools/genSBT.py.
 *
 * Floating-point contract = what that reference build computes:
 *   user code with -ffp-contract=off and -cl-fp32-correctly-rounded-divide-sqrt
 *   (IEEE fp32 + - * / sqrt, nothing fused), OpenCL builtins as ROCm's library
 *   implements them (read from its bitcode):
 *     dot = fma chain (x*x', then fma y, fma z[, fma w]); cross.x = fma(a.y,b.z,-(a.z*b.y)), cyclic;
 *     min/max/fmax = minnum/maxnum (NaN operand dropped); clamp = v_med3_f32; mix = fma(b-a,t,a);
 *     normalize(v) = v * rsqrt(dot(v,v)) with v_rsq_f32; sin/cos/acos/pow = OCML.
 *   This file reproduces dot/cross/min/max/clamp/mix exactly (fmaf); normalize uses
 *   the correctly rounded 1/sqrt and sin/cos/acos/pow are glibc's, so: slab test,
 *   triangle test and traversal of GIVEN rays are bit-exact against the reference
 *   build; anything behind a normalize or a transcendental (primary rays, shading,
 *   frames) agrees to a stated tolerance only.  The HIP product has no such limit:
 *   it is bit-identical to the reference build throughout.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- blob layouts (radiance/shader/data.cl:4-83, radiance/src/core.h:34-101) */
typedef struct { uint32_t type, nodeByteOffset, instByteOffset, totalBufferSize; } OrcAccelTop;
typedef struct { uint32_t type, nodeByteOffset, faceByteOffset, vertexOffset; }    OrcAccelBot;
typedef struct { float bottom[4]; float top[4]; uint32_t w0, w1, w2, w3; }          OrcNode;     /* 48 B */
typedef struct { uint32_t idx0, idx1, idx2, primID; }                               OrcTri;      /* 16 B */
typedef struct { float r[16]; uint32_t SBTOffset, instanceID, customInstanceID, instanceOffset; } OrcInst; /* 80 B */

/* host-side structs bound to the raygen kernel (radiance/src/core.h:103-158) */
typedef struct { uint32_t totalSamples, batchSize, depth, debug; } OrcRTProp;
typedef struct { float albedo[4]; float metallic, roughness, transmission, ior;
                 int32_t albedoTexIdx, metallicTexIdx, roughnessTexIdx, normalTexIdx; } OrcMaterial;   /* 48 B */
typedef struct { int32_t vertexOffset, indexOffset, uvOffset, normalOffset, materialIndex, _0, _1, _2; } OrcMeshInfo; /* 32 B */
typedef struct { float direction[4]; float color[4]; } OrcDirLight;                 /* 32 B */
typedef struct { uint32_t lightCount[4]; OrcDirLight lights[5]; } OrcSceneProps;    /* 176 B */
typedef struct { float widthPixel, heightPixel, focalLength, sensorWidth, focalDistance, fStop,
                 x, y, z, wx, wy, wz; } OrcCamera;                                   /* 48 B */

/* per-ray result, mirrors struct HitData (radiance/shader/radiance.cl:8-18) */
typedef struct {
    float    hitPoint[3];
    float    distance;
    uint32_t primitiveIndex, instanceIndex, instanceCustomIndex, instanceSBTOffset;
    float    barycentric[3];
    uint32_t hit;              /* return value of intersectTop */
    float    transform[16];
} OrcHit;

/* visit counters for the algorithmic-bytes model */
typedef struct {            /* index 0 = radiance rays (sbtRecordOffset 1), 1 = shadow rays (2) */
    uint64_t rays[2];        /* traceRay invocations */
    uint64_t top_nodes[2];   /* nodes popped in intersectTop */
    uint64_t inst_visits[2]; /* instances entered */
    uint64_t bot_nodes[2];   /* nodes popped in intersectBot */
    uint64_t tri_tests[2];   /* intersectTriangle calls */
    uint64_t hits;           /* closest-hit shader invocations (material) */
    uint64_t primary, bounce, shadow;   /* ray classes as issued by raygen / material */
} OrcCounters;

/* instance description handed to the TLAS builder (radiance.h:67-74) */
typedef struct {
    float    transform[16];     /* row-major object->world */
    uint32_t SBTOffset;
    uint32_t customInstanceID;
    uint32_t blas;              /* index into the blas handle array */
} OrcInstanceDesc;

/* the 14 descriptor slots of the live pipeline (samples/shader.cl:175-190) */
typedef struct {
    OrcRTProp*      RTProp;
    float*          imageScratch;
    uint8_t*        image;
    const OrcCamera*      camData;
    const OrcSceneProps*  scene;
    const OrcMeshInfo*    meshInfoData;
    const float*          vertexData;
    const uint32_t*       indexData;
    const float*          uvData;
    const float*          normalData;
    const OrcMaterial*    materials;
    const void*           topLevel;     /* TLAS blob */
    /* slots 11 + 12 (texture array + sampler).  texFlags = 0: texture reads return 0, as in the live reference shader
     * (`uint4 tex = 0.0f;//read_imageui(...)`, shader.cl:379-445).  Otherwise: bit 0 enabled, bit 1 linear filter,
     * bits 4-5 addressing (0 repeat, 1 clamp-to-edge, 2 clamp, 3 mirrored repeat) -- the product's TexView flags. */
    const uint8_t*        texData;      /* RGBA8, layer-major */
    uint32_t              texW, texH, texLayers, texFlags;
} OrcBindings;

/* ---- builder */
void*       orc_blas_build(const float* verts_xyz, uint32_t nverts, const uint32_t* tris, uint32_t ntris);
uint32_t    orc_blas_size(const void* blas);
const void* orc_blas_data(const void* blas);
int         orc_blas_max_depth(const void* blas);
void        orc_blas_free(void* blas);
/* returns malloc'ed TLAS blob (free with orc_free); size via *out_size */
void*       orc_tlas_build(const OrcInstanceDesc* inst, uint32_t ninst, void* const* blas_handles,
                           uint32_t* out_size, int* out_max_depth);
void        orc_free(void* p);

/* ---- traversal */
/* sbtRecordOffset: 1 = radiance ray (closest hit), 2 = shadow ray (any-hit terminates) */
void orc_trace_batch(const void* tlas, const float* origins, const float* dirs, uint32_t n,
                     float tmin, float tmax, int sbtRecordOffset, OrcHit* out, OrcCounters* ctr);

/* ---- unit entry points (KATs) */
void  orc_pcg3d(const uint32_t* in3, float* out3, uint32_t n);
int   orc_inverse_mat4(const float* m16, float* out16);
void  orc_mul_mat4_vec4(const float* m16, const float* v4, float* out4);
int   orc_intersect_aabb(const float* o3, const float* d3, const float* bmin3, const float* bmax3);
int   orc_intersect_triangle(const float* o3, const float* d3, const float* v0, const float* v1,
                             const float* v2, float* t, float* point3, float* bary3);
void  orc_generate_ray(const OrcCamera* cam, uint32_t pixel, const uint32_t* rand_in3,
                       float* origin3, float* dir3);
void  orc_microfacet_brdf(const float* L, const float* V, const float* N, const float* albedo,
                          float metallic, float roughness, float transmission, float ior, float* out3);
void  orc_sample_brdf_transm(const float* V, const float* N, const float* baseColor,
                             float metallic, float roughness, float transmission, float ior,
                             const float* random3, float* nextFactor3, float* L3);
float orc_d_ggx(float dotNH, float roughness);
void  orc_aces(const float* in3, float* out3);

/* closest-hit `material` on captured hits: payload in = nextRayDirection; out = color, nextFactor,
   nextRayOrigin, nextRayDirection (shader.cl:482-541). pixel/frameID/depth feed the RNG. */
typedef struct { float color[3]; uint32_t hit; float nextFactor[3]; float nextRayOrigin[3]; float nextRayDirection[3]; } OrcPayload;
void  orc_material_batch(const OrcBindings* b, const OrcHit* hits, const float* ray_dirs, const uint32_t* pixels,
                         const uint32_t* frame_ids, const int32_t* depths, uint32_t n, OrcPayload* out);

/* ---- whole-kernel: runs `raygen` (shader.cl:175-305) for pixels [begin,end) ; nthreads<=0 -> all */
void  orc_render(const OrcBindings* b, uint32_t pixel_begin, uint32_t pixel_end, int nthreads, OrcCounters* ctr);
/* same but for an explicit pixel list (tile sharding tests) */
void  orc_render_pixels(const OrcBindings* b, const uint32_t* pixels, uint32_t n, int nthreads, OrcCounters* ctr);

/* struct size table for the layout KATs */
void  orc_struct_sizes(uint32_t* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif

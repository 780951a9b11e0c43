/*
 * rdx.h -- C ABI of the MI355X-native ray-tracing core ("librdx.so").
 *
 * This is the drop-in boundary for the hot path  raygen -> BVH traversal -> triangle
 * intersection -> closest-hit / any-hit / miss -> accumulate  of zekailin00/Radiance-Ray-Tracing.
 * Every entry point replaces one function of the reference's host runtime
 * (radiance/include/radiance.h, implemented over OpenCL in radiance/src/radiance.cpp); the
 * reference interface it stands in for is cited per function.  Plain pointers, sizes and POD
 * structs only -- no C++ types, no torch types.  The C++ facade `namespace RD` in
 * include/radiance.h is a thin inline layer over these calls, so reference callers
 * (samples/sample1.cpp, tools/sceneBuilder.cpp) compile against it unchanged.
 *
 * Conventions
 *   - every function that can fail returns int: 0 = OK, <0 = error (rdx_last_error() has text);
 *     constructors return a handle or NULL.  The reference never returns errors (it prints and
 *     exit(-1)s, radiance/src/clcontext.h:27-47); the RD:: facade reproduces that policy on top.
 *   - single caller thread, blocking calls (radiance.cpp:190-223 use CL_TRUE everywhere).
 *   - handles stay valid until rdx_shutdown(); nothing needs to be destroyed
 *     (the reference has no Destroy/Release calls at all, radiance.h:88-144).
 */
#ifndef RDX_H
#define RDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rdx_buffer_s* rdx_buffer;   /* replaces RD::Buffer/Image/TopAccelStruct = cl_mem   (radiance.h:12-19) */
typedef struct rdx_blas_s*   rdx_blas;     /* replaces RD::BottomAccelStruct                     (radiance.h:52-60) */
typedef struct rdx_shader_s* rdx_shader;   /* replaces RD::ShaderModule = cl_kernel              (radiance.h:79)    */
typedef struct rdx_sampler_s* rdx_sampler; /* replaces RD::Sampler = cl_sampler                  (radiance.h:16)    */

/* Host instance record, mirrors RD::Instance (radiance.h:67-74). */
typedef struct rdx_instance {
    float    transform[16];      /* row-major object->world (aiMatrix4x4 a1..d4) */
    uint32_t SBTOffset;
    uint32_t customInstanceID;
    rdx_blas bottomAccelStruct;
} rdx_instance;

/* ---- platform: replaces RD::Platform::GetPlatform / CLContext::GetCLContext
 *      (radiance.h:146-174, radiance/src/clcontext.cpp:12-40) */
int         rdx_init(int device_ordinal);            /* idempotent; -1 = current device */
/* Single-process multi-device rendering (no reference counterpart: the reference is single-device, clcontext.cpp:17-36).  After
 * this call every buffer is replicated on `n` devices (writes go to all, reads come from logical device 0) and rdx_trace_rays
 * shards the frame by interleaved 64x64 image tiles over them -- one internal host thread per device, the caller still makes
 * one blocking call -- and copies every device's RGBA8 and imageScratch tiles into device 0's buffers before it returns.
 * ordinals[i] = HIP device of logical device i (NULL: consecutive devices starting at the one rdx_init chose).  Call before
 * creating buffers.  Results are bit-identical to one device (pixel / RNG indices stay global). */
int         rdx_init_devices(uint32_t n, const int* ordinals);
int         rdx_device_count(void);
int         rdx_shutdown(void);
const char* rdx_last_error(void);
int         rdx_device_name(char* out, size_t cap);

/* ---- resources: replaces CreateBuffer / CreateImage / ReadBuffer / WriteBuffer
 *      (radiance.h:115-128, radiance.cpp:86-93,139-146,190-204) */
rdx_buffer  rdx_buffer_create(size_t size);
rdx_buffer  rdx_buffer_wrap(void* device_ptr, size_t size);   /* adopt caller-owned device memory (e.g. a torch tensor) */
int         rdx_buffer_write(rdx_buffer b, size_t offset, size_t size, const void* src);
int         rdx_buffer_read(rdx_buffer b, size_t offset, size_t size, void* dst);
void*       rdx_buffer_device_ptr(rdx_buffer b);
size_t      rdx_buffer_size(rdx_buffer b);

/* ---- texture arrays and samplers: replaces CreateImageArray / CreateSampler / ReadImage / WriteImage
 *      (radiance.h:117-124, radiance.cpp:96-137,202-224).  An image array is `layers` RGBA8 images of width x height
 *      (CL_RGBA / CL_UNSIGNED_INT8, layer-major, rows tightly packed); it is also a buffer (rdx_buffer_read/write see the raw
 *      bytes).  write / read move the (width, height) top-left region of one layer, host rows tightly packed, like the
 *      reference's clEnqueue{Write,Read}Image(origin (0,0,layer), region (width,height,1)).  addressingMode / filterMode take
 *      the CL_ADDRESS_* / CL_FILTER_* values the reference's RD_ADDRESS_* / RD_FILTER_* macros expand to.  Bound to descriptor
 *      slots 11 and 12; sampled by the stock closest-hit shader only when option "textures" is 1 (see rdx_set_option). */
rdx_buffer  rdx_image_array_create(uint32_t width, uint32_t height, uint32_t layers);
int         rdx_image_write(rdx_buffer imageArray, uint32_t width, uint32_t height, size_t layer, const void* rgba8);
int         rdx_image_read(rdx_buffer imageArray, uint32_t width, uint32_t height, size_t layer, void* rgba8);
rdx_sampler rdx_sampler_create(uint32_t addressingMode, uint32_t filterMode);

/* ---- acceleration structures: replaces both RD::BuildAccelStruct overloads,
 *      TopAccelStructToFile and FileToTopAccelStruct
 *      (radiance.h:88-92, radiance.cpp:20-84,318-479, radiance/src/bvh.cpp:46-597).
 *      The blobs are byte-identical to the reference's (layout: radiance/shader/data.cl:237-278). */
rdx_blas    rdx_blas_build(const float* vertices_xyz, uint32_t nvertices,
                           const uint32_t* indices, uint32_t ntriangles);
/* `count` meshes at once, built on a pool of host threads inside the library (the caller stays single-threaded);
 * out[i] is what rdx_blas_build(verts[i], ...) would return.  The reference builds its meshes one after the other
 * (tools/sceneBuilder.cpp:229-258); at 10 M triangles that serial loop is the whole scene-load time. */
int         rdx_blas_build_many(uint32_t count, const float* const* verts_xyz, const uint32_t* nvertices,
                                const uint32_t* const* indices, const uint32_t* ntriangles, rdx_blas* out);
const void* rdx_blas_data(rdx_blas b, uint32_t* size_out);
int         rdx_blas_max_depth(rdx_blas b);
rdx_buffer  rdx_tlas_build(const rdx_instance* instances, uint32_t ninstances);
/* host-only variant: returns the malloc'ed TLAS blob (release with rdx_free); needs no GPU */
void*       rdx_tlas_build_blob(const rdx_instance* instances, uint32_t ninstances, uint32_t* size_out,
                                int* max_depth_out);
void        rdx_free(void* p);
/* The cache file is the raw blob, as the reference writes it; rdx_tlas_to_file also writes `<path>.meta` (magic,
 * version, byte count, FNV-1a hash) and rdx_tlas_from_file refuses a blob that contradicts an existing side-car. */
int         rdx_tlas_to_file(rdx_buffer tlas, const char* path);
rdx_buffer  rdx_tlas_from_file(const char* path);

/* ---- scene ingestion: replaces the assimp import of RD::Scene::Load (tools/sceneBuilder.cpp:27-258) for Wavefront
 *      OBJ + MTL files: the concatenated vertex / index / uv / normal streams, one MeshInfo per mesh, the Material
 *      table (core.h:122-158 layouts).  Host only, needs no GPU.  Every mesh is one instance (identity transform,
 *      customInstanceID = materialIndex, sceneBuilder.cpp:287-315).  Arrays are malloc'ed; release with rdx_obj_free. */
typedef struct rdx_material {        /* RD::Material, core.h:122-136 / pbr.cl:387-425 (48 B) */
    float   albedo[4];
    float   metallic, roughness, transmission, ior;
    int32_t albedoTexIdx, metallicTexIdx, roughnessTexIdx, normalTexIdx;
} rdx_material;
typedef struct rdx_mesh_info {       /* RD::MeshInfo, core.h:138-148 (32 B); offsets in floats / indices */
    int32_t vertexOffset, indexOffset, uvOffset, normalOffset, materialIndex, _0, _1, _2;
} rdx_mesh_info;
typedef struct rdx_obj_scene {
    uint32_t nmeshes, nvertices, ntriangles, nmaterials;
    rdx_mesh_info* meshInfo;          /* [nmeshes] */
    float*    vertex;                 /* [nvertices][3] */
    uint32_t* index;                  /* [ntriangles][3], mesh-local vertex numbers */
    float*    uv;                     /* [nvertices][3] (u, v, 0) */
    float*    normal;                 /* [nvertices][3] */
    rdx_material* materials;          /* [nmaterials] */
    uint32_t* meshVertexCount;        /* [nmeshes] */
    uint32_t* meshTriangleCount;      /* [nmeshes] */
} rdx_obj_scene;
int         rdx_obj_load(const char* path, rdx_obj_scene* out);
void        rdx_obj_free(rdx_obj_scene* scene);

/* ---- pipeline: replaces CreateShaderModule / BindPipeline / BindDescriptorSet / TraceRays
 *      (radiance.h:130-144, radiance.cpp:152-179,226-267).
 *      `code` is the user's shader text.  The reference's stock program (samples/shader.cl: the stage functions named in
 *      samples/sbt.json -- raygen, material, shadow, anyShadow, environment, shadowMiss) is recognised and served by the
 *      hand-written HIP wavefront pipeline of this library (tools/genSBT.py emits the dispatch); so is a placeholder whose
 *      `raygen` takes no parameters.  ANY OTHER program is compiled at run time by ROCm's OpenCL C compiler for the GPU in
 *      use and launched as a megakernel, one work-item per pixel, parameters bound by position (slots 11 / 12 as null
 *      descriptors) -- as the reference's clBuildProgram + clEnqueueNDRangeKernel would, csrc/user_shader.cpp.  A text that
 *      names no `raygen` kernel, or does not compile (the build log is in rdx_last_error), fails here. */
rdx_shader  rdx_shader_module_create(const char* code, uint32_t size, const char* name);
/* -I directory for `#include "radiance.cl"` etc. in user shader programs: the reference bakes SHADER_LIB_PATH into its
 * binary (radiance.h:7, radiance.cpp:165-167); here it is a run-time setting (the RD:: facade passes SHADER_LIB_PATH). */
int         rdx_shader_include_path(const char* path);
int         rdx_bind_pipeline(rdx_shader raygen_module);
/* handles[i] binds to parameter i of the raygen kernel (samples/shader.cl:175-190):
 * 0 RTProp, 1 imageScratch, 2 image, 3 camData, 4 scene, 5 meshInfo, 6 vertex, 7 index, 8 uv,
 * 9 normal, 10 material, 11 textureArray (may be NULL), 12 sampler (may be NULL), 13 TLAS. */
int         rdx_bind_descriptor_set(void* const* handles, uint32_t n);
/* the three SBT indices are accepted and ignored, exactly like radiance.cpp:242-259 */
int         rdx_trace_rays(uint32_t raygenGroupIndex, uint32_t missGroupIndex, uint32_t hitGroupIndex,
                           uint32_t width, uint32_t height);

/* ---- extensions (no reference counterpart) ------------------------------------------------ */

/* Image-tile sharding for multi-GPU: this process renders only the pixels whose tile id
 * (row-major over tile_w x tile_h tiles) is congruent to `rank` mod `world`.  Pixel/RNG indices
 * stay global, so the union over ranks is bit-identical to an unsharded frame. */
int         rdx_set_shard(uint32_t rank, uint32_t world, uint32_t tile_w, uint32_t tile_h);
/* pack this rank's tiles of a W*H image of `elem_size`-byte pixels into a contiguous buffer
 * (and the inverse on the gathering rank) */
int         rdx_pack_tiles(rdx_buffer image, rdx_buffer packed, uint32_t width, uint32_t height,
                           uint32_t elem_size, uint32_t rank, uint32_t world);
int         rdx_unpack_tiles(rdx_buffer packed, rdx_buffer image, uint32_t width, uint32_t height,
                             uint32_t elem_size, uint32_t rank, uint32_t world);
/* the same for the packed buffers of ranks first_rank .. first_rank + n - 1 in one call (one synchronisation) */
int         rdx_unpack_tiles_multi(const rdx_buffer* packed, uint32_t first_rank, uint32_t n, rdx_buffer image,
                                   uint32_t width, uint32_t height, uint32_t elem_size, uint32_t world);
uint32_t    rdx_shard_pixel_count(uint32_t width, uint32_t height, uint32_t rank, uint32_t world);

/* Statistics of the last rdx_trace_rays call. */
typedef struct rdx_trace_stats {
    uint64_t rays_primary, rays_bounce, rays_shadow;   /* rays actually traced */
    uint64_t closest_hits;                             /* `material` invocations */
    uint64_t pixels;                                   /* pixels rendered by this rank */
    /* visit counters of the reference algorithm's exhaustive walk, filled only when option
     * "count_visits" is 1: index 0 = radiance rays, 1 = shadow rays (SURVEY.md 8d byte model) */
    uint64_t visit_top_nodes[2], visit_instances[2], visit_bot_nodes[2], visit_triangles[2];
    float    ms_total;                                 /* HIP-event time of the whole call */
    float    ms_generate, ms_extend, ms_shade, ms_shadow, ms_accumulate, ms_fused;  /* ms_fused: shadow(d)+extend(d+1) launches */
    float    ms_path;                                  /* whole-path launches ("pipeline" 1) */
    uint32_t launches_extend, launches_shadow;
    uint32_t groups;                                   /* sample groups the last chunk was traced in (option "groups") */
    float    ms_sort;                                  /* per-bounce ray sort launches (option "sort") */
} rdx_trace_stats;
int         rdx_get_trace_stats(rdx_trace_stats* out);
/* per-bounce visit counters of the last frame traced with "count_visits": out[8*d + 4*c + k], c = 0
 * radiance / 1 shadow rays of bounce d, k = top nodes, instance visits, bottom nodes, triangle tests;
 * returns the number of bounces written (<= max_bounces) */
int         rdx_get_visit_profile(uint64_t* out, uint32_t max_bounces);
/* out[d] = closest-hit rays traced at bounce d of the last frame; out[d+1] is also the number of hits,
 * i.e. of shadow rays, of bounce d */
int         rdx_get_bounce_counts(uint64_t* out, uint32_t n);
/* 0 = per-stage HIP events off (default), 1 = on (adds launch gaps; for profiling only) */
int         rdx_set_profiling(int on);
/* knobs: "chunk_paths" (paths in flight per chunk), "count_visits" (0/1: also count node /
 * triangle visits; slower, for the roofline byte model), "kernel" (traversal kernel: 3 = wave-
 * cooperative with a shared node pool (default), 2 = wave-cooperative with per-lane node stacks, 1 = per-lane wide
 * nodes, 0 = reference order; all four give identical results, the option exists for A/B measurements and
 * cross-checks), "user_shader_local_size" (work-group size of a user shader program's launch, default 64; the reference
 * launches with 1, radiance.cpp:250-259 -- results do not depend on it), "sort" (-1 (default) = automatic: on for scenes of >= 1 M inner BVH nodes, and from 32 k inner nodes on in chunks of more than 1.5 M paths; 1 / 0 = on / off: per-bounce ray sort -- the survivors of a bounce are handed to the
 * traversal launch in (Morton cell of the origin, direction octant) order, by an index permutation from a counting sort; the
 * path streams are not moved and no result depends on it), "textures" (0 (default) / 1.  The live reference shader has every texture read commented out (`uint4 tex =
 * 0.0f;//read_imageui(...)`, samples/shader.cl:379,411,421,445), so a material with a texture index renders with texel 0; that
 * is what 0 reproduces, bit for bit.  1 performs the commented-out read -- coord (uv.x, 1 - uv.y, texIdx), as the reference's
 * older shader2.cl:255-265 does live -- from the image array in slot 11 through the sampler in slot 12), "cull" (pool kernel: -1 (default) = automatic, 1 / 0 = on / off: closest-hit rays skip subtrees the ray enters
 * beyond the best t found so far, every ray skips leaves whose box it misses -- only where a per-node normal cone proves the
 * reference's fp32 intersection test well conditioned for that ray, with margins that cover its error: the result is the
 * reference's exhaustive walk's, docs/CULLED_WALK.md has the proof; automatic = on for scenes with at least 1 M inner BVH
 * nodes, where it pays), "quad" (1 (default) / 0 / -1: the exhaustive walk of the pool engine pops 128-byte quad records -- two levels of the
 * reference's tree per item; 0 = the 64-byte records; -1 = quad records only for chunks of at most 3 M paths;
 * results do not depend on it), "gpu_build" (1 (default) / 0: the BVH builder bins the candidate planes of large nodes on the GPU, DESIGN.md 7.1;
 * "gpu_build_min": nodes and meshes of at least this many primitives, default 32768; blobs do not depend on either),
 * "user_stages" (1 (default) = a user program that equals the stock program outside the bodies of its closest-hit / miss
 * stage functions runs those functions on the wavefront pipeline; 0 = every user program is a megakernel; 2 = the
 * caller asserts eligibility for a program written from scratch, whose raygen is then ignored.  Read by
 * rdx_shader_module_create), "group_instances" (1 (default) / 0: instances whose inverse matrices are bit-identical share one
 * object-space ray and are walked concurrently, traverse_pool.h), "top_flat" (1 (default) / 0: the pool kernel evaluates a top-level tree of <= 64 nodes all at once per
 * ray instead of walking it), "inline_leaf_roots" (1 (default) / 0: ... and tests the triangles of single-leaf BLASes
 * right there), "pipeline" (0 = staged
 * wavefront: one launch per stage per bounce; 1 = whole paths -- camera ray to path end -- in one persistent launch per
 * sample chunk, the closest-hit shader called from the traversal waves; -1 (default) = automatic), "fuse" (1 / -1 = on (default), 0 = off:
 * trace the shadow rays of bounce d and the extend rays of bounce d+1 in one cooperative launch, which
 * halves the fixed ramp + tail cost per bounce), "groups" (0 (default) = automatic, 1..4: the samples of a chunk are
 * traced as that many independent groups on their own streams, each launching its share of the persistent grid, so
 * that one group's launches fill the ramp and drain of the others'; automatic = 2 for chunks of <= 4.7 M paths with
 * at least 2 samples (shards of a multi-GPU frame, low resolutions), else 1; results do not depend on it), "overlap"
 * (experimental: shadow rays on a second stream when "fuse" is 0; off by default) */
int         rdx_set_option(const char* name, int64_t value);

/* Test seams: run single stages on caller-supplied batches (device or host pointers are NOT
 * accepted -- plain host arrays in, host arrays out; the library stages them through HBM). */
typedef struct rdx_hit {
    float    hitPoint[3];
    float    distance;
    uint32_t primitiveIndex, instanceIndex, instanceCustomIndex, instanceSBTOffset;
    float    barycentric[3];
    uint32_t hit;
    float    transform[16];
} rdx_hit;      /* mirrors struct HitData, radiance/shader/radiance.cl:8-18 (+ the hit flag) */
/* mode 0: production kernel (wide nodes, fast slab test; for sbtRecordOffset 2 only `hit` is
 *         meaningful -- any accepted candidate ends a shadow ray);
 * mode 1: reference-order kernel (the reference's own DFS order: full HitData also for shadow rays).
 * visit4 (optional, implies mode 1): {top_nodes, instances, bot_nodes, triangles} visit counts of
 * the reference algorithm summed over the batch. */
int         rdx_trace_batch(rdx_buffer tlas, const float* origins_xyz, const float* dirs_xyz, uint32_t n,
                            float tmin, float tmax, int sbtRecordOffset, int mode, rdx_hit* out, uint64_t* visit4);
typedef struct rdx_payload {
    float color[3]; uint32_t hit; float nextFactor[3]; float nextRayOrigin[3]; float nextRayDirection[3];
} rdx_payload;  /* mirrors struct Payload, samples/shader.cl:4-13 */
/* closest-hit `material` (samples/shader.cl:482-541) on captured hits, using the bound descriptors */
int         rdx_material_batch(const rdx_hit* hits, const float* ray_dirs_xyz, const uint32_t* pixels,
                               const uint32_t* frame_ids, const int32_t* depths, uint32_t n, rdx_payload* out);
/* primary rays (samples/shader.cl:111-173) for explicit pixels / rng inputs, using the bound camera */
int         rdx_generate_batch(const uint32_t* pixels, const uint32_t* rand_in3, uint32_t n,
                               float* origins_xyz, float* dirs_xyz);
int         rdx_pcg3d_batch(const uint32_t* in3, float* out3, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RDX_H */

// kernels.h -- launch interface between the host runtime (rdx_runtime.cpp) and the HIP stages
// (kernels.hip).  All pointers are device pointers; all launches go to the given stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rdx.h"
#include "rdx_types.h"

namespace rdx {

struct AccelView {                 // derived traversal layout (see rdx_types.h)
    const DNode* tnodes;
    const DNode* ctnodes;          // the same nodes with the children ordered for the cooperative kernel (smaller stack need first)
    const DInst* insts;
    const DNode* bnodes;
    const DTri*  tris;
    const DWide* wide;             // wide BLAS nodes (production kernels)
    uint32_t stackNeed;            // worst-case stack entries of an exhaustive DFS (per-lane kernels)
    uint32_t coopNeed;             // the same for the wave-cooperative kernel (see derive_accel)
    uint32_t topNeed, blasNeed;    // its top-level / per-BLAS parts (pool engine: private top-level stacks + shared node pool)
    uint32_t leafRoots;            // pool engine: the scene has single-leaf BLASes handled in the flat top-level step (needs topFlat)
    uint32_t topFlat;              // pool engine: > 0 = number of top-level nodes, evaluated all at once per ray (<= 64 nodes)
    uint32_t numInsts;             // instances in `insts`
    uint32_t* status;              // device-visible status word (pinned host memory): bit 0 = a traversal wave hit its iteration bound
    uint32_t cull;                 // pool engine: culled walk (best-t culling of closest-hit rays, leaf-box test; kernels.hip)
    uint32_t kernel;               // 2 = wave-cooperative (default), 3 = wave-cooperative with a shared node pool, 1 = per-lane wide, 0 = reference order
    uint32_t groupCount;           // pool engine: instances in the shared-transform group (rdx_runtime.cpp derive_accel), 0 = none
    const uint32_t* groupBits;     // ... and their slots as a bitmap of 9 words in device memory (flat top level only: <= 256 instances)
    uint32_t unifiedRoot;          // pool engine: > 0 = wide index of the super-root of the unified tree (derive_accel): rays start there
    uint32_t quadWaves;            // ... its kernels' waves per SIMD: 6, or 7 for full-size frames (kernels.hip k_*_pool_q)
    const DQuad* quad;             // pool engine, exhaustive walk: two tree levels per record (rdx_types.h); null = walk the DWide records
};

// limits of the cooperative engines' packed words, shared by the kernels (traverse_coop.h) and the host's fallback rule
constexpr uint32_t RDX_COOP_MAX_TRI_SLOTS = 1u << 25;    // queue entry: lane << 26 | parity << 25 | absolute triangle slot
constexpr uint32_t RDX_COOP_MAX_WIDE = 1u << 26;         // pool item: lane << 26 | wide-node index
constexpr uint32_t RDX_COOP_MAX_INSTANCES = 1u << 10;    // key: instance slot << 22 | BLAS-local triangle slot
constexpr uint32_t RDX_COOP_MAX_BLAS_TRIS = 1u << 22;
constexpr uint32_t RDX_LDS_WORDS_PER_WAVE_MAX = 16384u;  // 64 KB per workgroup of one wave

// descriptor slots 11 + 12: the RGBA8 texture array and its sampler (radiance.cpp:96-137; read by the commented-out
// read_imageui calls of samples/shader.cl:379-445, live in shader2.cl:255-265).  flags = 0: the stock pipeline behaves
// like the LIVE reference shader, whose texture reads are stubbed to 0.
enum : uint32_t { TEX_ENABLED = 1u, TEX_LINEAR = 2u, TEX_ADDR_SHIFT = 4u,      // flags
                  TEX_ADDR_REPEAT = 0u, TEX_ADDR_CLAMP_TO_EDGE = 1u, TEX_ADDR_CLAMP = 2u, TEX_ADDR_MIRRORED = 3u };
struct TexView { const uint8_t* data; uint32_t w, h, layers, flags; };

struct SceneArgs {                 // descriptor slots 4-12 (samples/shader.cl:175-190)
    const SceneProperties* scene;
    const MeshInfo* meshInfo;
    const uint32_t* indexData;
    const float* uvData;
    const float* normalData;
    const Material* materials;
    TexView tex;
};

struct CameraArgs {                // slot 3 + per-frame constants hoisted out of generateRay
    PhysicalCamera cam;
    float rotX[16], rotY[16], rotZ[16];   // EulerX/Y/ZToMat4x4(cam.wx/wy/wz), math.cl:185-252
};

// Wavefront path state: float4 streams, one entry per path.
//   cur  (read by extend / shade of bounce d):  rayO, rayD, thr, col  + the hit record
//   next (written by shade of bounce d, compacted to the paths that hit; they become `cur` of bounce
//         d+1 -- the host swaps the pointers): nRayO, nRayD, nThr, and nCol, which the shadow stage of
//         bounce d fills in
//   shadow query of bounce d: shO, colLit, colSh
// shade(d) -> { extend(d+1), shadow(d) } -> shade(d+1): extend(d+1) and shadow(d) touch disjoint streams
// and may run concurrently on two HIP streams.
struct PathStreams {
    float4* rayO;      // origin.xyz, w = global pixel index (bits)
    float4* rayD;      // direction.xyz, w = frameID (bits)
    float4* thr;       // contribution.xyz, w = owned-pixel slot (bits)
    float4* col;       // accumulated colour.xyz
    float4* hitA;      // t, b1, b2, primID (bits)
    uint32_t* hitInst; // instance slot or 0xffffffff (miss)
    float4* nRayO;     // next bounce: origin | pixel
    float4* nRayD;     // next bounce: direction | frameID
    float4* nThr;      // next bounce: contribution after this bounce | slot
    float4* nCol;      // next bounce: colour, chosen by the shadow stage
    float4* shO;       // shadow-ray origin.xyz, w != 0: the closest-hit shader asked for a shadow query
    float4* colLit;    // colour if the light is visible
    float4* colSh;     // colour if it is occluded
    float4* sampleColor; // [samples_in_chunk][pixels] final per-sample radiance
    // user stage functions on the wavefront pipeline (user_shader.cpp "stage mode"; allocated only then): the shadow query's own
    // direction and answer, and the two payload members the stock raygen carries from bounce to bounce (color, nextFactor)
    float4* shD; uint32_t* shHit;
    float4* payC; float4* payF; float4* nPayC; float4* nPayF;
    // per-bounce ray sort (option "sort"): the order in which the persistent traversal launch HANDS OUT the rays of this
    // bounce -- work item i is path permS[i] (shadow query) / permE[i] (next-bounce ray).  Null = identity.  Nothing is moved:
    // the streams stay in compaction order, only the waves' ray assignment follows the sort.
    unsigned short* sortKey;     // per-bounce ray sort: the sort key of survivor j, written by the shade stage (which has the ray in
                                 // registers) so that the sort reads 2 bytes per path instead of the 32 of its origin and direction; null = off
    const uint32_t* permS;
    const uint32_t* permE;
};

// per-bounce ray sort (option "sort"): counting sort of the m survivors of shade(d) by (origin cell -- the high bits of the Morton
// code of the origin in a 16^3 grid over the scene box -- then direction octant) into ONE permutation that serves the shadow
// queries and the next bounce's rays alike (same origin).  No device-scope atomics: per-block LDS histograms, one scan, LDS
// positions (kernels.hip k_sortg_*).
constexpr uint32_t SORT_TILE = 4096u;                // counters per scan tile
struct SortBox { float lo[3]; float inv[3]; };      // cell = (p - lo) * inv, clamped to 0..15
// perm: nMax words; H: ray_sort_tiles_words() words of scratch
void launch_ray_sort_tiles(hipStream_t st, const PathStreams& ps, const uint32_t* mPtr, uint32_t nMax, const SortBox& box, uint32_t* H,
                           uint32_t* perm, bool largeScene = false);
uint32_t ray_sort_tiles_words();

// LDS words one wave of the cooperative / pool engine needs for the given stack needs (traverse_coop.h, traverse_pool.h)
uint32_t coop_lds_words(uint32_t coopNeed);
uint32_t pool_lds_words(uint32_t topNeed, uint32_t blasNeed);

// how many persistent traversal launches are to share the GPU from now on (sample groups on their own streams):
// each launch takes 1/groups of the resident grid.  Host state, set by rdx_trace_rays per chunk.
void set_grid_share(uint32_t groups);

// cos / sin of the three camera angles, evaluated on the device (out6 = cx sx cy sy cz sz, device memory)
void launch_euler_trig(hipStream_t st, float wx, float wy, float wz, float* out6);
void launch_generate(hipStream_t st, const CameraArgs& cam, const PathStreams& ps, const uint32_t* ownedPixels,
                     uint32_t nPixels, uint32_t sampleBegin, uint32_t sampleCount, uint32_t totalSamples);
// nPtr: device word holding the live count of this stage; nMax: upper bound used to size the grid
void launch_extend(hipStream_t st, const AccelView& av, const PathStreams& ps, const uint32_t* nPtr, uint32_t nMax,
                   float tmin, float tmax, unsigned long long* visit, uint32_t* counter);
void launch_shade(hipStream_t st, const AccelView& av, const SceneArgs& sc, const PathStreams& ps, const uint32_t* nPtr,
                  uint32_t* nOut, uint32_t nMax, uint32_t depth, uint32_t maxDepth, uint32_t nPixels, uint32_t sampleBase, const SortBox* sortBox = nullptr);
void launch_shadow(hipStream_t st, const AccelView& av, const SceneArgs& sc, const PathStreams& ps, const uint32_t* nPtr,
                   uint32_t nMax, bool lastBounce, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax,
                   unsigned long long* visit, uint32_t* counter);
// shadow(d) of `psShadow` and extend(d+1) of `psExtend` (its streams already swapped) in one cooperative launch
// any-hit walk of the shadow queries a user's closest-hit shader recorded (origin shO, direction shD per path): answer -> shHit
void launch_shadow_user(hipStream_t st, const AccelView& av, const PathStreams& ps, const uint32_t* nPtr, uint32_t nMax, float tmin, float tmax,
                        uint32_t* counter);
void launch_fused(hipStream_t st, const AccelView& av, const SceneArgs& sc, const PathStreams& psShadow, const PathStreams& psExtend,
                  const uint32_t* mPtr, uint32_t mMax, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax, uint32_t* counter);
// whole paths (camera ray to path end) on the persistent cooperative engine, one launch per sample chunk;
// tally[0] += closest-hit rays, tally[1] += shadow rays
void launch_path(hipStream_t st, const AccelView& av, const SceneArgs& sc, const CameraArgs& cam, const PathStreams& ps,
                 const uint32_t* owned, uint32_t nPixels, uint32_t sampleBegin, uint32_t sampleCount, uint32_t totalSamples,
                 uint32_t maxDepth, uint32_t sampleBase, uint32_t* counter, unsigned long long* tally, float tmin, float tmax);
void launch_accumulate(hipStream_t st, const PathStreams& ps, const uint32_t* ownedPixels, uint32_t nPixels,
                       uint32_t sampleBegin, uint32_t sampleCount, uint32_t totalSamples, bool tonemap, uint32_t debug,
                       float* imageScratch, uint8_t* image);
void launch_finalize_all(hipStream_t st, const PathStreams& ps, uint32_t n, uint32_t nPixels, uint32_t sampleBase);

// GPU-assisted BVH build (bvh_build.cpp GpuBinner): bins of one node; out = 3 axes x 7 x 1025 words (kernels.hip k_bvh_bin)
void launch_bvh_bin(hipStream_t st, const float* prims9, const uint32_t* work, uint32_t n, const float* cand, const uint32_t K[3], uint32_t* out);
void launch_pack_tiles(hipStream_t st, const uint8_t* image, uint8_t* packed, uint32_t w, uint32_t h, uint32_t elem,
                       uint32_t tileW, uint32_t tileH, uint32_t rank, uint32_t world, bool unpack);

// test seams
void launch_trace_batch(hipStream_t st, const AccelView& av, const float* o, const float* d, uint32_t n, float tmin,
                        float tmax, int rec, rdx_hit* out, unsigned long long* visit, int mode, uint32_t* counter);
void launch_material_batch(hipStream_t st, const SceneArgs& sc, const rdx_hit* hits, const float* dirs,
                           const uint32_t* pixels, const uint32_t* frames, const int32_t* depths, uint32_t n,
                           rdx_payload* out);
void launch_generate_batch(hipStream_t st, const CameraArgs& cam, const uint32_t* pixels, const uint32_t* rnd,
                           uint32_t n, float* o, float* d);
void launch_pcg3d_batch(hipStream_t st, const uint32_t* in3, float* out3, uint32_t n);

} // namespace rdx

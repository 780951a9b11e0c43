// rdx_runtime.cpp -- host runtime behind the C ABI (include/rdx.h): device/stream singleton,
// buffers, acceleration-structure upload + derived layout, pipeline binding and the per-frame
// wavefront schedule.  Replaces the OpenCL host runtime of the reference
// (radiance/src/radiance.cpp, radiance/src/clcontext.cpp) for the ray-tracing hot path.
#include "../../include/rdx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <cfloat>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#ifndef RDX_SBT_HEADER
#define RDX_SBT_HEADER "sbt_generated.h"      // tools/genSBT.py output (the table this library's stage kernels were built for)
#endif
#ifndef RDX_QUAD_AUTO_MAX_PATHS
#define RDX_QUAD_AUTO_MAX_PATHS (3u << 20)      // option "quad" -1: chunks of at most this many paths walk the quad records (1/4 of a 1080p x 4 spp frame: 2.1 M)
#endif
#ifndef RDX_STOCK_REDUCED_HASH
#define RDX_STOCK_REDUCED_HASH 0xf95635133b09cb3full         // of samples/shader.cl; tools/stock_shader_hash.py prints it
#endif
#include RDX_SBT_HEADER
#include "bvh_build.h"
#include "device_math.h"
#include "kernels.h"
#include "rdx_types.h"
#include "user_shader.h"

using namespace rdx;

#define HIP_IGN(expr) do { (void)(expr); } while (0)

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
constexpr int RDX_MAX_DEVICES = 16;
// option "sort" -1: scenes with at least this many inner BVH nodes are sorted.  Measured with the r02d engine (1080p x 4 spp; unsorted
// / sorted): 10.4 M triangles 56.9 / 54.3 ms (the sorted hand-out makes the traversal launches 9 % faster, the eight sorts cost
// 1.5 ms), Sponza-class 24.5 / 24.6 (break-even), sample1 13.3 / 15.4 (its rays are coherent as they come; the sort scrambles the
// pixel order).
constexpr uint32_t RDX_SORT_AUTO_MIN_WIDE = 1u << 20;      // option "sort" -1: scenes with at least this many inner BVH nodes sort their rays per bounce ...
constexpr uint32_t RDX_SORT_AUTO_MIN_WIDE_FULL = 1u << 15; // ... and so do scenes from this size on for chunks of more than sortMinPaths paths (the sort is four more dependent
                                                           // launches per bounce: small shards lose with it, and so does a scene that sits in L2 anyway)
constexpr uint32_t RDX_CULL_AUTO_MIN_WIDE = 1u << 20;      // option "cull" -1: scenes with at least this many inner BVH nodes take the culled walk
struct AccelCache {                // derived traversal layout of one TLAS buffer
    uint64_t version = ~0ull;
    DNode* tnodes = nullptr; DNode* ctnodes = nullptr; DInst* insts = nullptr; DNode* bnodes = nullptr; DTri* tris = nullptr;
    DWide* wide = nullptr;
    uint32_t stackNeed = 1;            // per-lane kernels (reference order: left child followed, right child pushed)
    uint32_t coopNeed = 1;             // wave-cooperative kernel (leaf children are never pushed, smaller subtree first)
    uint32_t topNeed = 1, blasNeed = 0; // its two parts: top-level entries of one ray / entries inside one BLAS (pool engine)
    bool leafRoots = false;             // some instance's BLAS is a single leaf of <= 8 triangles
    float sceneLo[3] = {0, 0, 0}, sceneHi[3] = {1, 1, 1};   // box of the top-level root (per-bounce ray sort grid)
    bool sbtOffsets = false;               // an instance has SBTOffset != 0: reference-order kernel only
    uint32_t nWide = 0;                    // inner BLAS nodes of the scene (sizes the automatic choice of the culled walk)
    uint32_t blasNeedAny = 0;              // BLAS stack need of the pool engine when the push order depends on the ray (culled walk)
    uint32_t nInst = 0;
    uint32_t topFlat = 0, topFlatNeed = 1; // pool engine: number of top-level nodes if they are few enough (<= 64) to be evaluated
                                        // all at once per ray instead of walked, and the instance-mask entries that can then pile up
    bool coopOK = true;                // scene fits the key packing of the wave-cooperative kernel
    uint32_t* groupBits = nullptr;     // pool engine: instance slots of the shared-transform group (bitmap, 9 words on the device), see derive_accel
    uint32_t groupCount = 0;
    bool groupIdentity = false;        // ... and the group's transform is the identity: root tests in the flat top-level step
    uint32_t unifiedRoot = 0, unifiedNeed = 0;   // pool engine: one tree over top level + instances + BLASes (derive_accel), 0 = not built
    DQuad* quad = nullptr;                 // pool engine, exhaustive walk: quad records (rdx_types.h), index = DWide index
    uint32_t quadNeed = 0, quadUnifiedNeed = 0;  // pool-stack need of the quad walk inside one BLAS / from the unified root
    void release()
    {
        if (groupBits) HIP_IGN(hipFree(groupBits));
        groupBits = nullptr;
        if (tnodes) HIP_IGN(hipFree(tnodes));
        if (ctnodes) HIP_IGN(hipFree(ctnodes));
        if (insts) HIP_IGN(hipFree(insts));
        if (bnodes) HIP_IGN(hipFree(bnodes));
        if (tris) HIP_IGN(hipFree(tris));
        if (wide) HIP_IGN(hipFree(wide));
        if (quad) HIP_IGN(hipFree(quad));
        quad = nullptr;
        tnodes = nullptr; ctnodes = nullptr; insts = nullptr; bnodes = nullptr; tris = nullptr; wide = nullptr;
    }
};

struct rdx_buffer_s {
    void* dptr = nullptr;
    size_t size = 0;
    bool owned = true;
    uint64_t version = 0;                  // bumped by every write
    std::vector<uint8_t> shadow;           // host copy of a TLAS blob (valid iff shadowVersion == version)
    uint64_t shadowVersion = ~0ull;
    std::unique_ptr<AccelCache> accel;
    // small parameter buffers (RTProp, camera): a host mirror kept current by the write path, so that TraceRays does
    // not read them back from the device every frame.  Valid only for library-owned buffers whose every byte has been
    // written through the API since creation (device code never writes them); wrapped memory is never mirrored.
    std::vector<uint8_t> mirror;
    bool mirrorValid = false;
    uint32_t imgW = 0, imgH = 0, imgLayers = 0;   // != 0: an RGBA8 image array created by rdx_image_array_create
    // single-process multi-device mode (rdx_init_devices): the copy of this buffer on logical device d >= 1 and the traversal
    // layout derived from it there; device 0 uses dptr / accel
    void* rep[RDX_MAX_DEVICES] = {};
    std::unique_ptr<AccelCache> accelRep[RDX_MAX_DEVICES];
};
struct rdx_sampler_s { uint32_t addressing = 0, filter = 0; };
struct rdx_blas_s { std::unique_ptr<Blas> blas; };
struct rdx_shader_s { std::string name; bool hasRaygen = false;
                      UserProgram* program = nullptr; };   // != null: a user's program, compiled at run time -- as its raygen megakernel, or
                                                           // (program->stages) as the shade stage of the wavefront pipeline around its stage functions

namespace {

struct Context {
    bool initialized = false;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t evA = nullptr, evB = nullptr, evChunk = nullptr;
    std::string err;
    std::vector<std::unique_ptr<rdx_buffer_s>> buffers;
    std::vector<std::unique_ptr<rdx_blas_s>> blases;
    std::vector<std::unique_ptr<rdx_shader_s>> shaders;
    rdx_shader_s* pipeline = nullptr;
    void* slots[14] = {};
    uint32_t nslots = 0;
    // sharding
    uint32_t rank = 0, world = 1, tileW = 64, tileH = 64;
    uint32_t* ownedPixels = nullptr; uint32_t ownedCount = 0, ownedW = 0, ownedH = 0, ownedRank = 0, ownedWorld = 0,
              ownedTileW = 0, ownedTileH = 0;
    // path streams: up to MAX_GROUPS independent sets of paths in flight (sample groups of one chunk);
    // each has its own streams (s0: generate/extend/shade, s1: shadow beside the next extend), live
    // counts and events, so launches of different groups overlap and cover each other's tails
    struct Group {
        PathStreams ps{};
        size_t cap = 0;
        uint32_t* sortBins = nullptr;       // per-bounce ray sort: histogram scratch (ray_sort_tiles_words()) and the permutation
        uint32_t* permE = nullptr; size_t permCap = 0;
        unsigned short* sortKey = nullptr;  // per-bounce ray sort: keys of the survivors, written by the shade stage
        size_t stageCap = 0;                // user stage mode: paths the extra streams (shD, shHit, payC ...) hold
        uint32_t* dCounts = nullptr;        // [0] = paths generated, [d+1] = hits of bounce d, [64+d] / [128+d] ray counters
        uint32_t* hCounts = nullptr;        // pinned
        hipStream_t s0 = nullptr, s1 = nullptr;
        hipEvent_t evShade[64] = {}, evShadow[64] = {}, evDone = nullptr;
    };
    static constexpr int MAX_GROUPS = 4;
    Group groups[MAX_GROUPS];
    size_t sampleCap = 0;
    float4* sampleColor = nullptr;
    uint32_t* dCounts = nullptr;            // = groups[0].dCounts (test seams)
    void* gatherStage[4] = {};              // multi-device gather: packed tiles (RGBA8, imageScratch) on this device and their landing buffers on device 0
    size_t gatherCap[4] = {};
    uint32_t* hStatus = nullptr;            // pinned, device-mapped: bit 0 = a traversal wave hit its iteration bound
    uint32_t* dStatus = nullptr;            // its device address
    int groupsOpt = 0;                      // sample groups in flight: 1..4, 0 = two for chunks small enough to be ramp + drain bound
    int fuse = -1;                          // shadow(d) + extend(d+1) in one launch: 1 / -1 on, 0 off
    int pathMode = 0;                       // 0 = staged wavefront (launch per stage per bounce), 1 = whole paths in one launch
    unsigned long long* dVisit = nullptr;   // 8 words
    unsigned long long* hVisit = nullptr;   // pinned
    // options
    int64_t chunkPaths = 16ll << 20;
    bool countVisits = false, profiling = false;
    int inlineLeafRoots = 1;                // pool engine: single-leaf BLASes handled in the flat top-level step (option "inline_leaf_roots")
    int cull = -1;                          // pool engine: culled walk (option "cull"): 1 on, 0 off, -1 = on for scenes of >= 1 M inner nodes
    int textures = 0;                       // option "textures": 1 = the stock shader samples the bound image array
    std::vector<std::unique_ptr<rdx_sampler_s>> samplers;
    std::string shaderInclude;              // -I for user shader programs (rdx_shader_include_path; the reference's SHADER_LIB_PATH)
    int userLocalSize = 64;                 // option "user_shader_local_size": work-group size of a user program's launch (the reference uses 1)
    int sortRays = -1;                      // option "sort": per-bounce ray sort: 1 on, 0 off, -1 automatic
    int topFlat = 1;                        // pool engine: evaluate small top-level trees all at once (option "top_flat")
    int groupInstances = 1;                 // pool engine: instances with bit-identical inverse matrices share one ray slot (option "group_instances")
    int userStages = 1;                     // user programs that differ from the stock one only inside stage functions run on the wavefront pipeline (option "user_stages")
    int64_t sortMinPaths = 3ll << 19;       // chunks of more paths than this (1.5 M) sort the rays of mid-size scenes and use the 7-wave quad kernels (option
                                            // "sort_min_paths"; 4.7 M / 3 M / 1.5 M: 1/2 frame 13.6 / 12.4 / 12.4 ms, 1/4 frame 7.57 / 7.57 / 7.35 ms, Sponza-class)
    int64_t smallChunkPaths = 9ll << 19;    // chunks of at most this many paths (4.7 M) do not fill the chip: two sample groups, 6-wave quad kernels, no ray sort
                                            // (option "small_chunk_paths")
    int gpuBuild = 1;                       // BVH builder: large nodes are binned on the GPU (option "gpu_build")
    int64_t gpuBuildMin = 32768;            // ... nodes (and meshes) of at least this many primitives (option "gpu_build_min")
    int quad = 1;                           // pool engine, exhaustive walk: quad records -- two tree levels per fetch (option "quad"): 1 on (default), 0 off,
                                            // -1 = only for chunks the chip is not filled by (<= RDX_QUAD_AUTO_MAX_PATHS paths)
    int unifiedTree = 1;                    // pool engine: large top levels of identity instances are walked by the pool (option "unified_tree")
    int kernel = 3;                         // traversal kernel: 3 cooperative + shared node pool, 2 cooperative, 1 per-lane wide, 0 reference order
    int overlap = 0;                        // extend(d+1) || shadow(d) on two streams (experimental): 1 on, 0 off
    rdx_trace_stats stats{};
    float camAngles[3] = {0, 0, 0}, camTrig[6] = {1, 0, 1, 0, 1, 0};      // camera_args: cos / sin of the camera angles, evaluated on the device
    bool camCached = false;
    uint32_t visitDepth = 0;                // bounces covered by hVisit after a count_visits frame
    uint64_t bounceCounts[65] = {};         // [d] = closest-hit rays of bounce d, [d+1] = hits = shadow rays of bounce d (last frame)
};
// The process has one Context per LOGICAL device.  g0 is device 0 and also the registry (buffers, shaders, descriptor slots,
// options).  Every function below reads `g`, which is the context of the calling thread: g0 on the caller's thread; inside
// rdx_trace_rays in multi-device mode each worker thread points it at its own device's context (streams, path buffers, counters,
// shard) after copying the registry-side fields it needs (slots, pipeline, options).
Context g0;
Context* g_dev[RDX_MAX_DEVICES] = {&g0};     // logical device -> context (entries >= 1 are heap-allocated by rdx_init_devices)
int g_phys[RDX_MAX_DEVICES] = {0};           // logical device -> HIP device ordinal
int g_ndev = 1;
thread_local Context* tl_ctx = &g0;
thread_local int tl_dev = 0;                 // logical device of the calling thread
#define g (*tl_ctx)

inline void* dp(const rdx_buffer_s* b) { return tl_dev == 0 ? b->dptr : b->rep[tl_dev]; }
inline std::unique_ptr<AccelCache>& acc(rdx_buffer_s* b) { return tl_dev == 0 ? b->accel : b->accelRep[tl_dev]; }
inline const std::unique_ptr<AccelCache>& acc(const rdx_buffer_s* b) { return tl_dev == 0 ? b->accel : b->accelRep[tl_dev]; }

int fail(const char* fmt, ...)
{
    // sized to the message (a compiler log of a user shader program runs to thousands of characters)
    va_list ap, ap2; va_start(ap, fmt); va_copy(ap2, ap);
    const int n = vsnprintf(nullptr, 0, fmt, ap); va_end(ap);
    std::string buf((size_t)(n > 0 ? n : 0) + 1, '\0');
    vsnprintf(&buf[0], buf.size(), fmt, ap2); va_end(ap2);
    buf.resize((size_t)(n > 0 ? n : 0));
    g.err = std::move(buf);
    return -1;
}
int fail_str(const std::string& text) { g.err = text; return -1; }
#define HIP_OK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return fail("HIP error: '%s' returned %d (%s)", #expr, (int)_e, hipGetErrorString(_e)); } while (0)
#define HIP_OKP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fail("HIP error: '%s' returned %d (%s)", #expr, (int)_e, hipGetErrorString(_e)); return nullptr; } } while (0)

bool known_buffer(const void* h)
{
    for (auto& b : g0.buffers) if (b.get() == h) return true;
    return false;
}

// ---- derived traversal layout ------------------------------------------------------------------
// worst-case stack occupancy of the left-first DFS in kernels.hip (right child pushed, left followed)
uint32_t blas_need(const BlobNode* nodes, uint32_t idx)
{
    // iterative post-order to survive deep trees
    struct Frame { uint32_t idx; uint32_t needL; int state; };
    std::vector<Frame> st{{idx, 0, 0}};
    uint32_t ret = 0;
    while (!st.empty()) {
        Frame& f = st.back();
        const BlobNode& n = nodes[f.idx];
        if (n.w0 & LEAF_BIT) { ret = 0; st.pop_back(); continue; }
        if (f.state == 0) { f.state = 1; st.push_back({n.w0, 0, 0}); continue; }
        if (f.state == 1) { f.needL = ret; f.state = 2; st.push_back({n.w1, 0, 0}); continue; }
        ret = std::max(1u + f.needL, ret);
        st.pop_back();
    }
    return ret;
}

// Cone of lines around the normals of a set of triangles + their worst shape; see DESIGN.md 4.1c for what the culled walk
// proves from it.  kappa0 = 2^-7: the culled walk skips a subtree / a leaf only for rays that make at least asin(kappa0 / q)
// with the plane of every triangle below the node.
struct NormalCone {
    double a[3] = {0, 0, 0};       // axis (unit) -- valid when n > 0
    double alpha = 0;              // half-angle: every normal line is within alpha of the axis line
    double q = 1;                  // min over the triangles of sin(angle(e1, e2))
    bool never = false;            // degenerate triangle, or the normals do not fit a cone of < 90 degrees
    uint32_t n = 0;
    static double ang(const double* x, const double* y)      // angle between two LINES
    {
        const double c = std::fabs(x[0] * y[0] + x[1] * y[1] + x[2] * y[2]);
        return std::acos(std::min(1.0, c));
    }
    void add_normal(const double* nn, double a1)
    {
        if (n == 0) { a[0] = nn[0]; a[1] = nn[1]; a[2] = nn[2]; alpha = a1; n = 1; return; }
        NormalCone o; o.a[0] = nn[0]; o.a[1] = nn[1]; o.a[2] = nn[2]; o.alpha = a1; o.n = 1;
        merge(o);
    }
    void add_triangle(const DTri& t)
    {
        const double e1[3] = {t.e1[0], t.e1[1], t.e1[2]}, e2[3] = {t.e2[0], t.e2[1], t.e2[2]};
        const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double lc = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        if (!(lc > 0) || !(l1 > 0) || !(l2 > 0) || !std::isfinite(lc) || !std::isfinite(l1 * l2)) { never = true; return; }
        q = std::min(q, lc / (l1 * l2));
        const double nn[3] = {c[0] / lc, c[1] / lc, c[2] / lc};
        add_normal(nn, 0.0);
    }
    void merge(const NormalCone& o)
    {
        never = never || o.never; q = std::min(q, o.q);
        if (o.n == 0) return;
        if (n == 0) { a[0] = o.a[0]; a[1] = o.a[1]; a[2] = o.a[2]; alpha = o.alpha; n = o.n; return; }
        const double sgn = (a[0] * o.a[0] + a[1] * o.a[1] + a[2] * o.a[2]) < 0 ? -1.0 : 1.0;
        const double gam = ang(a, o.a);
        n += o.n;
        if (gam + o.alpha <= alpha) return;                                        // o inside this cone
        if (gam + alpha <= o.alpha) { a[0] = o.a[0]; a[1] = o.a[1]; a[2] = o.a[2]; alpha = o.alpha; return; }
        // smallest cone around both: axis between the two, rotated from a towards o by (gam + o.alpha - alpha) / 2
        const double na = (gam + alpha + o.alpha) / 2;
        const double w = gam > 1e-12 ? (na - alpha) / gam : 0.5;
        double m[3] = {a[0] * (1 - w) + sgn * o.a[0] * w, a[1] * (1 - w) + sgn * o.a[1] * w, a[2] * (1 - w) + sgn * o.a[2] * w};
        const double lm = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        if (!(lm > 1e-9)) { never = true; return; }
        for (int k = 0; k < 3; ++k) m[k] /= lm;
        // (the linear blend is not the exact bisecting rotation: take the half-angle from the blended axis itself)
        const double sgn_o[3] = {sgn * o.a[0], sgn * o.a[1], sgn * o.a[2]};
        alpha = std::max(ang(m, a) + alpha, ang(m, sgn_o) + o.alpha);
        a[0] = m[0]; a[1] = m[1]; a[2] = m[2];
    }
    // x | y << 8 | z << 16 | T << 24 (T: 7 bits, rdx_types.h wide_desc): a ray may be culled against this leaf / subtree only if
    // |d^ . a'| >= T / 127 with a' = (b - 127.5) / 127 the quantised axis.  Derivation: every normal line is within
    // alpha + eq of a' (eq: quantisation), so the ray makes >= asin|d^ . a'^| - (alpha + eq) with every triangle's plane; that must be
    // >= asin(kappa0 / q).  |a'| is within 0.7 % of 1, which the factor 1.0075 covers, fp32 evaluation another 1e-5.
    uint32_t pack() const
    {
        const double kappa0 = 1.0 / 128.0, eq = 0.0085;
        uint32_t T = WIDE_CONE_NEVER;
        uint32_t b[3] = {128, 128, 128};
        if (!never && n > 0 && q > kappa0 && alpha + eq < 1.5) {
            const double need = std::asin(std::min(1.0, kappa0 / q)) + alpha + eq;     // angle the ray must make with the axis PLANE
            if (need < 1.55) {
                const double thr = std::sin(need) * 1.0075 * 1.00002;
                const double t8 = std::ceil(thr * 127.0) + 1.0;
                if (t8 <= 126.0) T = (uint32_t)t8;
            }
            for (int k = 0; k < 3; ++k) b[k] = (uint32_t)std::min(255.0, std::max(0.0, std::floor(127.5 + 127.0 * a[k] + 0.5)));
        }
        return b[0] | (b[1] << 8) | (b[2] << 16) | (T << 24);
    }
};

// Can the order-free engines (pool / cooperative / per-lane wide) trace instances whose SBT offset is k?  They assume that a
// radiance ray's row (1 + k) has NO any-hit shader -- so the winner is the minimum, whatever the visiting order -- and that a
// shadow ray's row (2 + k) is the stock pair: an any-hit shader that ends the walk at the first accepted candidate and a
// closest-hit shader that only flags the hit (`anyShadow` / `shadow`: neither looks at WHICH candidate it was).  Rows are those
// of the sbt.json this library was generated from (tools/genSBT.py -> sbt_generated.h); k = 0 always qualifies for the stock
// table.  Anything else keeps the reference's DFS order: the reference-order kernel.
bool sbt_offset_is_order_free(uint32_t k)
{
    struct Row { int row; const char* fn; };
    static const Row anyHit[] = {
#define X(row, fn) {row, #fn},
        RDX_SBT_ANY_HIT(X)
#undef X
        {-1, nullptr}};
    static const Row closest[] = {
#define X(row, fn) {row, #fn},
        RDX_SBT_CLOSEST_HIT(X)
#undef X
        {-1, nullptr}};
    auto find = [](const Row* t, int row) -> const char* { for (; t->fn; ++t) if (t->row == row) return t->fn; return nullptr; };
    if (k > 1000000u) return false;
    const int r1 = 1 + (int)k, r2 = 2 + (int)k;
    if (find(anyHit, r1)) return false;
    const char* a2 = find(anyHit, r2); const char* c2 = find(closest, r2);
    const char* a0 = find(anyHit, 2); const char* c0 = find(closest, 2);
    auto same = [](const char* x, const char* y) { return (x == nullptr && y == nullptr) || (x && y && !std::strcmp(x, y)); };
    return same(a2, a0) && same(c2, c0);
}

int derive_accel(rdx_buffer_s* tb)
{
    if (acc(tb) && acc(tb)->version == tb->version) return 0;
    // host copy of the blob
    if (tb->shadowVersion != tb->version) {
        tb->shadow.resize(tb->size);
        HIP_OK(hipMemcpy(tb->shadow.data(), tb->dptr, tb->size, hipMemcpyDeviceToHost));
        tb->shadowVersion = tb->version;
    }
    const uint8_t* blob = tb->shadow.data();
    const size_t bsz = tb->shadow.size();
    if (bsz < 16) return fail("TLAS buffer too small");
    const auto* th = reinterpret_cast<const BlobTopHeader*>(blob);
    if (th->type != TYPE_TOP_AS || th->nodeByteOffset != 16 || th->instByteOffset < 16 + sizeof(BlobNode) ||
        th->instByteOffset > bsz || th->totalBufferSize > bsz)
        return fail("descriptor slot 13 does not hold a top-level acceleration structure blob");
    const uint32_t nTop = (th->instByteOffset - th->nodeByteOffset) / sizeof(BlobNode);
    const auto* tnodes = reinterpret_cast<const BlobNode*>(blob + th->nodeByteOffset);
    const auto* binst = reinterpret_cast<const BlobInst*>(blob + th->instByteOffset);
    // instance count = max leaf (start+count)
    uint32_t nInst = 0;
    for (uint32_t i = 0; i < nTop; ++i)
        if (tnodes[i].w0 & LEAF_BIT) nInst = std::max(nInst, tnodes[i].w1 + (tnodes[i].w0 & 0x7fffffffu));
    if ((size_t)th->instByteOffset + (size_t)nInst * sizeof(BlobInst) > bsz) return fail("TLAS blob: instance array out of range");

    // The derived layout (stack needs computed children-first, first-in-DFS tie-break = lowest slot) relies on the numbering
    // the reference's flattener produces (bvh.cpp:475-497,551-563): DFS pre-order -- left child = parent + 1, right child
    // behind the whole left subtree -- and leaves listing their instances / triangles in leaf order.  A foreign or
    // corrupted blob (e.g. a cache file without side-car) that breaks it is refused here rather than mis-sized on the GPU.
    std::vector<DNode> dT(nTop);
    {
        uint32_t expectInst = 0;
        for (uint32_t i = 0; i < nTop; ++i) {
            std::memcpy(&dT[i], &tnodes[i], sizeof(BlobNode));
            if (!(tnodes[i].w0 & LEAF_BIT)) {
                if (tnodes[i].w0 >= nTop || tnodes[i].w1 >= nTop) return fail("TLAS blob: child index out of range");
                if (tnodes[i].w0 != i + 1 || tnodes[i].w1 <= tnodes[i].w0) return fail("TLAS blob: node %u is not in DFS pre-order (children %u, %u)", i, tnodes[i].w0, tnodes[i].w1);
            } else {
                if (tnodes[i].w1 != expectInst) return fail("TLAS blob: leaf %u does not list its instances in leaf order (start %u, expected %u)", i, tnodes[i].w1, expectInst);
                expectInst += tnodes[i].w0 & 0x7fffffffu;
            }
        }
    }
    std::vector<DNode> dB;
    std::vector<DTri> dTri;
    std::vector<DWide> dW;
    std::vector<DInst> dI(nInst);
    struct BlasInfo { uint32_t nodeBase; uint32_t need; uint32_t coopNeed; uint32_t anyNeed; uint32_t triBase; uint32_t rootDesc0, rootDesc1; float rootMin[3], rootMax[3];
                      uint32_t nTris; uint32_t users; };
    bool hugeLeaf = false;                  // a leaf of more triangles than the wide layout's count field holds
    bool coopOK = nInst <= RDX_COOP_MAX_INSTANCES;
    bool sbtOffsets = false;
    uint32_t maxLeafChunks = 0;             // extra stack entries an oversized (> 8 triangle) leaf can push
    uint32_t maxLeafTris = 0;
    std::map<uint32_t, BlasInfo> blasAt;    // byte offset -> merged-array base
    for (uint32_t k = 0; k < nInst; ++k) {
        const BlobInst& bi = binst[k];
        // Dispatch index = instanceSBTOffset + sbtRecordOffset (radiance.cl:281, shader.cl:574-605).  With a non-zero offset the
        // any-hit shader of a RADIANCE ray's row may end the walk at the first accepted candidate in the reference's DFS order --
        // an order only the reference-order kernel keeps -- so scenes with such a row are traced by that kernel; offsets whose
        // rows behave like the stock rows 1 / 2 (sbt_offset_is_order_free) stay on the production engines (the live loader always
        // writes 0, tools/sceneBuilder.cpp:302).
        if (bi.SBTOffset != 0 && !sbt_offset_is_order_free(bi.SBTOffset)) sbtOffsets = true;
        auto it = blasAt.find(bi.instanceOffset);
        if (it == blasAt.end()) {
            if ((size_t)bi.instanceOffset + 16 > bsz) return fail("TLAS blob: BLAS offset out of range");
            const uint8_t* bb = blob + bi.instanceOffset;
            const auto* bh = reinterpret_cast<const BlobBotHeader*>(bb);
            if (bh->type != TYPE_BOT_AS || bh->faceByteOffset < bh->nodeByteOffset || bh->vertexOffset < bh->faceByteOffset ||
                (size_t)bi.instanceOffset + bh->vertexOffset > bsz)
                return fail("TLAS blob: malformed bottom-level structure at byte %u", bi.instanceOffset);
            const uint32_t nNodes = (bh->faceByteOffset - bh->nodeByteOffset) / sizeof(BlobNode);
            const uint32_t nTris = (bh->vertexOffset - bh->faceByteOffset) / sizeof(BlobTri);
            const auto* bn = reinterpret_cast<const BlobNode*>(bb + bh->nodeByteOffset);
            const auto* bt = reinterpret_cast<const BlobTri*>(bb + bh->faceByteOffset);
            const auto* bv = reinterpret_cast<const float*>(bb + bh->vertexOffset);
            const size_t vertFloatsAvail = (bsz - bi.instanceOffset - bh->vertexOffset) / 4;
            const uint32_t nodeBase = (uint32_t)dB.size(), triBase = (uint32_t)dTri.size();
            if ((uint64_t)nodeBase + nNodes >= (1u << 30)) return fail("too many BVH nodes for 30-bit references");
            dB.resize(nodeBase + nNodes);
            uint32_t expectTri = 0;
            for (uint32_t i = 0; i < nNodes; ++i) {
                DNode& d = dB[nodeBase + i];
                std::memcpy(&d, &bn[i], sizeof(BlobNode));
                if (bn[i].w0 & LEAF_BIT) {
                    if ((uint64_t)bn[i].w1 + (bn[i].w0 & 0x7fffffffu) > nTris) return fail("BLAS blob: leaf range out of bounds");
                    if (bn[i].w2 == TYPE_TRIG) {
                        if (bn[i].w1 != expectTri) return fail("BLAS blob: leaf %u does not list its triangles in leaf order (start %u, expected %u)", i, bn[i].w1, expectTri);
                        expectTri += bn[i].w0 & 0x7fffffffu;
                    }
                    maxLeafChunks = std::max(maxLeafChunks, 2u * (((bn[i].w0 & 0x7fffffffu) + 7u) / 8u));
                    maxLeafTris = std::max(maxLeafTris, bn[i].w0 & 0x7fffffffu);
                    d.w1 = bn[i].w1 + triBase;
                } else {
                    if (bn[i].w0 >= nNodes || bn[i].w1 >= nNodes) return fail("BLAS blob: child index out of range");
                    if (bn[i].w0 != i + 1 || bn[i].w1 <= bn[i].w0) return fail("BLAS blob: node %u is not in DFS pre-order (children %u, %u)", i, bn[i].w0, bn[i].w1);
                    d.w0 = bn[i].w0 + nodeBase; d.w1 = bn[i].w1 + nodeBase;
                }
            }
            if ((uint64_t)triBase + nTris > LEAF_START_MASK) return fail("too many triangles for 27-bit triangle-run references");
            dTri.resize(triBase + nTris);
            for (uint32_t i = 0; i < nTris; ++i) {
                const BlobTri& t = bt[i];
                if ((size_t)std::max({t.idx0, t.idx1, t.idx2}) * 4 + 3 > vertFloatsAvail)
                    return fail("BLAS blob: vertex index out of range");
                const float* v0 = bv + 4 * (size_t)t.idx0; const float* v1 = bv + 4 * (size_t)t.idx1; const float* v2 = bv + 4 * (size_t)t.idx2;
                DTri& d = dTri[triBase + i];
                d.v0[0] = v0[0]; d.v0[1] = v0[1]; d.v0[2] = v0[2]; d.primID = t.primID;
                d.e1[0] = v1[0] - v0[0]; d.e1[1] = v1[1] - v0[1]; d.e1[2] = v1[2] - v0[2]; d._p0 = 0xffffffffu;   // radiance.cl:215; _p0: see "shared-transform group"
                d.e2[0] = v2[0] - v0[0]; d.e2[1] = v2[1] - v0[1]; d.e2[2] = v2[2] - v0[2]; d._p1 = triBase;       // radiance.cl:216; _p1: first triangle slot of this BLAS
            }
            // wide layout: one record per inner node, numbered in the same DFS pre-order
            const uint32_t wideBase = (uint32_t)dW.size();
            std::vector<uint32_t> wideIdx(nNodes, 0);
            uint32_t nInner = 0;
            for (uint32_t i = 0; i < nNodes; ++i) if (!(bn[i].w0 & LEAF_BIT)) wideIdx[i] = nInner++;
            if ((uint64_t)wideBase + nInner >= (1u << 30)) return fail("too many BVH nodes for 30-bit references");
            std::vector<NormalCone> cone(nNodes);
            auto desc = [&](uint32_t c, uint32_t& d0, uint32_t& d1) {
                if (bn[c].w0 & LEAF_BIT) {
                    const uint32_t cnt = bn[c].w2 == TYPE_TRIG ? (bn[c].w0 & 0x7fffffffu) : 0u;
                    if (cnt > WIDE_MAX_LEAF_TRIS) hugeLeaf = true;
                    wide_desc(true, bn[c].w1 + triBase, std::min(cnt, (uint32_t)WIDE_MAX_LEAF_TRIS), cone[c].pack(), d0, d1);
                } else wide_desc(false, wideBase + wideIdx[c], 0u, cone[c].pack(), d0, d1);
            };
            dW.resize(wideBase + nInner);
            // Normal cones (culled walk, kernels.hip): for every node the cone of LINES that holds the normals of all triangles
            // below it -- axis, half-angle alpha -- and the worst triangle shape q = min sin(angle(e1, e2)).  Bottom-up (children
            // have larger indices); computed in double from the fp32 edge vectors the intersection test uses.
            for (uint32_t i = nNodes; i-- > 0;) {
                NormalCone& c = cone[i];
                if (bn[i].w0 & LEAF_BIT) {
                    const uint32_t cnt = bn[i].w2 == TYPE_TRIG ? (bn[i].w0 & 0x7fffffffu) : 0u;
                    for (uint32_t t = 0; t < cnt; ++t) c.add_triangle(dTri[triBase + bn[i].w1 + t]);
                } else { c = cone[bn[i].w0]; c.merge(cone[bn[i].w1]); }
            }
            // Stack need of the wide walk: a leaf child is queued, never pushed; of two inner children one is followed
            // and the other pushed.  The visiting order is free, so the child with the SMALLER need goes
            // into the "followed" (left) half of the record: need = max(1 + smaller, larger) instead of
            // max(1 + left, right).  Children have larger indices than their parent (DFS pre-order).
            std::vector<uint32_t> aneed(nNodes, 0);                     // any push order: 1 + the deeper inner child
            std::vector<uint32_t> cneed(nNodes, 0), wneed(nNodes, 0);   // wneed: per-lane wide kernel on the same records (pushes leaves too)
            for (uint32_t i = nNodes; i-- > 0;) {
                if (bn[i].w0 & LEAF_BIT) continue;
                uint32_t a = bn[i].w0, b = bn[i].w1;
                const bool la = bn[a].w0 & LEAF_BIT, lb = bn[b].w0 & LEAF_BIT;
                if (!la && !lb) {
                    aneed[i] = 1u + std::max(aneed[a], aneed[b]);
                    if (cneed[b] < cneed[a]) std::swap(a, b);
                    cneed[i] = std::max(1u + cneed[a], cneed[b]);
                } else {
                    cneed[i] = la ? (lb ? 0u : cneed[b]) : cneed[a];
                    aneed[i] = la ? (lb ? 0u : aneed[b]) : aneed[a];
                }
                wneed[i] = std::max(1u + wneed[a], wneed[b]);
                DWide& w = dW[wideBase + wideIdx[i]];
                const BlobNode& L = bn[a]; const BlobNode& Rn = bn[b];
                for (int k = 0; k < 3; ++k) { w.lmin[k] = L.bottom[k]; w.lmax[k] = L.top[k]; w.rmin[k] = Rn.bottom[k]; w.rmax[k] = Rn.top[k]; }
                desc(a, w.ld0, w.ld1);
                desc(b, w.rd0, w.rd1);
            }
            BlasInfo info{};
            info.nodeBase = nodeBase; info.need = std::max(blas_need(bn, 0), wneed[0]); info.coopNeed = cneed[0]; info.anyNeed = aneed[0]; info.triBase = triBase;
            if (nTris > RDX_COOP_MAX_BLAS_TRIS) coopOK = false;
            desc(0, info.rootDesc0, info.rootDesc1);
            for (int k = 0; k < 3; ++k) { info.rootMin[k] = bn[0].bottom[k]; info.rootMax[k] = bn[0].top[k]; }
            info.nTris = nTris; info.users = 0;
            it = blasAt.emplace(bi.instanceOffset, info).first;
        }
        it->second.users++;
        DInst& d = dI[k];
        std::memset(&d, 0, sizeof d);
        std::memcpy(d.fwd, bi.m, 64);
        inverse_mat4(bi.m, d.inv);          // zeros stay if singular (oracle convention; reference: uninitialised)
        d.SBTOffset = bi.SBTOffset; d.instanceID = bi.instanceID; d.customInstanceID = bi.customInstanceID;
        d.blasRoot = it->second.nodeBase;
        d.rootDesc0 = it->second.rootDesc0; d.rootDesc1 = it->second.rootDesc1; d._p0 = it->second.triBase;
        for (int k = 0; k < 3; ++k) { d.rootMin[k] = it->second.rootMin[k]; d.rootMax[k] = it->second.rootMax[k]; }
        // conservative world-space box of the root OBB + margin coefficient for the instance pre-test
        {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, fa = 0, fi = 0;
            bool finite = true;
            for (int c = 0; c < 8; ++c) {
                const double p[3] = {(c & 1) ? d.rootMax[0] : d.rootMin[0], (c & 2) ? d.rootMax[1] : d.rootMin[1], (c & 4) ? d.rootMax[2] : d.rootMin[2]};
                for (int r = 0; r < 3; ++r) {
                    const double w = (double)bi.m[4 * r] * p[0] + (double)bi.m[4 * r + 1] * p[1] + (double)bi.m[4 * r + 2] * p[2] + (double)bi.m[4 * r + 3];
                    lo[r] = std::min(lo[r], w); hi[r] = std::max(hi[r], w);
                    finite = finite && std::isfinite(w);
                }
            }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { fa += (double)bi.m[4 * r + c] * bi.m[4 * r + c]; fi += (double)d.inv[4 * r + c] * d.inv[4 * r + c]; }
            const bool affine = bi.m[12] == 0.f && bi.m[13] == 0.f && bi.m[14] == 0.f && bi.m[15] == 1.f;
            const double kappa = std::sqrt(fa) * std::sqrt(fi);
            const bool usable = finite && affine && !(d.rootDesc1 & WIDE_LEAF) && kappa > 0 && kappa < 1e4 && std::isfinite(kappa);
            double ext = 0;
            for (int r = 0; r < 3; ++r) {
                d.worldMin[r] = std::nextafterf((float)lo[r], -INFINITY); d.worldMax[r] = std::nextafterf((float)hi[r], INFINITY);
                ext = std::max(ext, std::max(std::fabs(lo[r]), std::fabs(hi[r])));
            }
            // margin = c * (|o|_inf + ext): 64 x the first-order bound 4u*kappa on the displacement of the
            // object-space ray the reference builds in fp32 (DESIGN.md "instance pre-test")
            d.worldMin[3] = usable ? (float)(64.0 * 5.97e-8 * 4.0 * kappa) : -1.0f;
            d.worldMax[3] = (float)ext;
        }
    }
    // Shared-transform group (pool engine, flat top level).  The object-space ray of an instance is inverse(object->world) applied
    // to the world ray with the reference's expressions (radiance.cl:161-169) -- a function of the inverse matrix's BITS and the
    // ray alone.  Instances whose inverse matrices are bit-identical (a loader that puts every mesh of an OBJ under one node:
    // all identity, tools/sceneBuilder.cpp:287-315; every scene of this repository's bench) therefore share ONE object-space
    // ray: a wave lane writes it to its LDS ray slot once and enters all of them without waiting for one instance's subtree and
    // queued tests to drain before the next (traverse_pool.h).  The instance slot a candidate belongs to then cannot come from
    // the ray slot; it is kept in the triangle record (DTri._p0), which needs the BLAS to belong to exactly one instance.
    // The largest such set of instances (>= 2, inner-node roots only) is the group.
    uint32_t groupBits[8] = {0, 0, 0, 0, 0, 0, 0, 0}, groupCount = 0;
    bool groupIdentity = false;
    if (nInst <= 256) {
        std::map<std::array<uint32_t, 16>, std::vector<uint32_t>> byInv;
        for (uint32_t k = 0; k < nInst; ++k) {
            const BlasInfo& bi = blasAt[binst[k].instanceOffset];
            if (bi.users != 1 || (dI[k].rootDesc1 & WIDE_LEAF)) continue;
            std::array<uint32_t, 16> key;
            std::memcpy(key.data(), dI[k].inv, 64);
            byInv[key].push_back(k);
        }
        const std::vector<uint32_t>* best = nullptr;
        for (auto& kv : byInv) if (kv.second.size() >= 2 && (!best || kv.second.size() > best->size())) best = &kv.second;
        if (best) {
            // identity group: the group's object-space ray equals the world ray up to the sign of zeros, on which no slab decision
            // depends -- the flat top-level step then runs the reference's root-box test of these instances itself (world ray)
            groupIdentity = true;
            for (int e = 0; e < 16; ++e) if (!(dI[(*best)[0]].inv[e] == ((e % 5 == 0) ? 1.0f : 0.0f))) groupIdentity = false;
        }
        if (best)
            for (uint32_t k : *best) {
                groupBits[k >> 5] |= 1u << (k & 31u); ++groupCount;
                const BlasInfo& bi = blasAt[binst[k].instanceOffset];
                for (uint32_t t = 0; t < bi.nTris; ++t) dTri[bi.triBase + t]._p0 = k;
            }
    }
    // Unified tree (pool engine; scenes whose top level is too large for the flat step: > 64 nodes or > 256 instances -- a loader
    // that makes one instance per mesh, tools/sceneBuilder.cpp:287-315).  When EVERY instance has the identity transform and a
    // BLAS of its own, the object-space ray of all instances is one ray (the group's; it equals the world ray up to the sign
    // of zeros, which no slab decision depends on: (b - o) / d keeps its value, min / max of +-0 or of equally signed infinities
    // decide the same) -- so top-level nodes can be walked like BLAS nodes, by the pool, on that one ray slot:
    //   top-level inner node -> wide record (boxes of its two children; an inner child is entered iff its box is hit, as the
    //                           reference does when it pops the child; a leaf child is always entered: the reference never tests it)
    //   top-level leaf       -> a balanced fan-out of always-entered pseudo nodes over its instances
    //   instance             -> a child entry whose box is the BLAS root box and whose descriptor is the BLAS root (the root
    //                           test of radiance.cl:61-63 for an inner root; a leaf root has its triangles tested directly)
    // One item -- a super-root that holds the top-level root's box -- starts a ray; no top-level step, no instance step.
    uint32_t unifiedRoot = 0, unifiedNeed = 0;
    if ((nTop > 64 || nInst > 256) && nInst >= 2 && !(tnodes[0].w0 & LEAF_BIT) && !sbtOffsets) {
        bool ok = true;
        for (uint32_t k = 0; k < nInst && ok; ++k) {
            const BlasInfo& bi = blasAt[binst[k].instanceOffset];
            if (bi.users != 1) ok = false;
            for (int e = 0; e < 16 && ok; ++e) if (!(dI[k].inv[e] == ((e % 5 == 0) ? 1.0f : 0.0f))) ok = false;      // (zero signs do not matter)
            if (std::memcmp(dI[k].inv, dI[0].inv, 64) != 0) ok = false;                                               // ... but one ray needs one matrix
        }
        if (ok) {
            const float BIG = 1.0e30f;
            const uint32_t never = WIDE_CONE_NEVER << 24;
            auto always = [&](float* mn, float* mx) { for (int k = 0; k < 3; ++k) { mn[k] = -BIG; mx[k] = BIG; } };
            auto none = [&](float* mn, float* mx, uint32_t& d0, uint32_t& d1) { for (int k = 0; k < 3; ++k) { mn[k] = 0.f; mx[k] = 0.f; } wide_desc(true, 0u, 0u, never, d0, d1); };
            auto inst_child = [&](uint32_t k, float* mn, float* mx, uint32_t& d0, uint32_t& d1) {
                for (int c = 0; c < 3; ++c) { mn[c] = dI[k].rootMin[c]; mx[c] = dI[k].rootMax[c]; }
                d0 = dI[k].rootDesc0; d1 = dI[k].rootDesc1;
                const BlasInfo& bi = blasAt[binst[k].instanceOffset];
                for (uint32_t t = 0; t < bi.nTris; ++t) dTri[bi.triBase + t]._p0 = k;          // the candidate's instance comes from the triangle
            };
            // fan-out over instances [a, b): returns the child entry for that range
            std::function<uint32_t(uint32_t, uint32_t, float*, float*, uint32_t&, uint32_t&)> range_child =
                [&](uint32_t a, uint32_t b, float* mn, float* mx, uint32_t& d0, uint32_t& d1) -> uint32_t {
                    if (b - a == 1) { inst_child(a, mn, mx, d0, d1); return (dI[a].rootDesc1 & WIDE_LEAF) ? 0u : 1u + blasAt[binst[a].instanceOffset].anyNeed; }
                    const uint32_t mid = a + (b - a) / 2;
                    const uint32_t idx = (uint32_t)dW.size();
                    dW.emplace_back();
                    DWide w{};
                    const uint32_t hl = range_child(a, mid, w.lmin, w.lmax, w.ld0, w.ld1);
                    const uint32_t hr = range_child(mid, b, w.rmin, w.rmax, w.rd0, w.rd1);
                    dW[idx] = w;
                    always(mn, mx);
                    wide_desc(false, idx, 0u, never, d0, d1);
                    return 1u + std::max(hl, hr);
                };
            // top-level nodes, children first (DFS pre-order: children have larger indices)
            std::vector<uint32_t> uIdx(nTop, 0), uH(nTop, 0);
            for (uint32_t i = nTop; i-- > 0;) {
                const BlobNode& n = tnodes[i];
                if (n.w0 & LEAF_BIT) continue;
                const uint32_t idx = (uint32_t)dW.size();
                dW.emplace_back();
                DWide w{};
                uint32_t h[2] = {0, 0};
                for (int c = 0; c < 2; ++c) {
                    const uint32_t ch = c ? n.w1 : n.w0;
                    float* mn = c ? w.rmin : w.lmin; float* mx = c ? w.rmax : w.lmax;
                    uint32_t& d0 = c ? w.rd0 : w.ld0; uint32_t& d1 = c ? w.rd1 : w.ld1;
                    const BlobNode& cn = tnodes[ch];
                    if (!(cn.w0 & LEAF_BIT)) {
                        for (int k = 0; k < 3; ++k) { mn[k] = cn.bottom[k]; mx[k] = cn.top[k]; }
                        wide_desc(false, uIdx[ch], 0u, never, d0, d1);
                        h[c] = 1u + uH[ch];
                    } else {
                        const uint32_t cnt = cn.w2 == TYPE_INST ? (cn.w0 & 0x7fffffffu) : 0u;
                        if (cnt == 0) none(mn, mx, d0, d1);
                        else {
                            h[c] = range_child(cn.w1, cn.w1 + cnt, mn, mx, d0, d1);      // (one instance: its root test is this child's box test)
                        }
                    }
                }
                dW[idx] = w;
                uIdx[i] = idx; uH[i] = std::max(h[0], h[1]);
            }
            // super-root: the reference tests the top-level root's own box when it pops it
            DWide sr{};
            for (int k = 0; k < 3; ++k) { sr.lmin[k] = tnodes[0].bottom[k]; sr.lmax[k] = tnodes[0].top[k]; }
            wide_desc(false, uIdx[0], 0u, never, sr.ld0, sr.ld1);
            none(sr.rmin, sr.rmax, sr.rd0, sr.rd1);
            unifiedRoot = (uint32_t)dW.size();
            dW.push_back(sr);
            unifiedNeed = uH[0] + 2u;
        }
    }
    // Quad records (rdx_types.h DQuad): one per wide record, built from the finished wide array -- BLAS nodes and the unified
    // tree's records alike.  need[i] = entries the LIFO pool grows by while the subtree of a popped record i is walked alone
    // (tight mode of the pool step): the inner entries are pushed together and popped first-entry-first, so
    // need = max(k, max_j(entries below j + need[target j])); the halves and the entries inside a half are ordered to
    // minimise it (the visiting order is free in the exhaustive walk).
    // (not built where the culled walk will run -- large scenes under the automatic rule, or option "cull" 1: the culled walk keeps
    // the 64-byte records, and the quad records of a 10 M-triangle scene are 0.6 GB and a second of host time.  Option "cull"
    // changed later: the exhaustive walk then uses the 64-byte records until the structure is derived again.)
    const bool wantQuad = g0.quad != 0 && !(g0.cull > 0 || (g0.cull < 0 && dW.size() >= RDX_CULL_AUTO_MIN_WIDE)) && !unifiedRoot;
    std::vector<DQuad> dQ(wantQuad ? dW.size() : 0);
    std::vector<uint32_t> qneed(dW.size(), 0);
    if (wantQuad) {
        struct QE { float mn[3], mx[3]; uint32_t d0, d1; };
        auto empty = [](QE& e) { for (int k = 0; k < 3; ++k) { e.mn[k] = 0.f; e.mx[k] = 0.f; } e.d0 = 0u; e.d1 = WIDE_LEAF; };
        auto entry_of = [](const DWide& w, int side, QE& e) {
            for (int k = 0; k < 3; ++k) { e.mn[k] = side ? w.rmin[k] : w.lmin[k]; e.mx[k] = side ? w.rmax[k] : w.lmax[k]; }
            const uint32_t d0 = side ? w.rd0 : w.ld0, d1 = side ? w.rd1 : w.ld1;
            if (d1 & WIDE_LEAF) { e.d0 = wide_slot(d0); e.d1 = WIDE_LEAF | (wide_count(d1) << 24); }
            else { e.d0 = d0; e.d1 = 0u; }
        };
        // the half for child `side` of wide record N
        auto half_of = [&](const DWide& N, int side, QE out[2]) {
            QE c;
            entry_of(N, side, c);
            bool pair = false;
            if (!(c.d1 & WIDE_LEAF) && c.d0 < dW.size()) {
                const DWide& C = dW[c.d0];
                pair = true;
                for (int k = 0; k < 3; ++k)
                    if (!(std::min(C.lmin[k], C.rmin[k]) == c.mn[k] && std::max(C.lmax[k], C.rmax[k]) == c.mx[k])) pair = false;
                // an empty entry in C (count-0 leaf with a zero box) would have entered the union above: such a record keeps its own test
                if (pair) {
                    entry_of(C, 0, out[0]); entry_of(C, 1, out[1]);
                    for (int e = 0; e < 2; ++e) {
                        out[e].d1 |= QUAD_PAIR;
                        if (out[e].d1 & WIDE_LEAF) for (int k = 0; k < 3; ++k) { out[e].mn[k] = c.mn[k]; out[e].mx[k] = c.mx[k]; }
                    }
                }
            }
            if (!pair) { out[0] = c; empty(out[1]); }
        };
        // children first: explicit DFS over the records (BLAS records have larger-index children, unified records smaller ones)
        std::vector<uint8_t> state(dW.size(), 0);       // 0 new, 1 open, 2 done
        std::vector<uint32_t> stk;
        for (size_t r = 0; r < dW.size(); ++r) {
            if (state[r]) continue;
            stk.push_back((uint32_t)r);
            while (!stk.empty()) {
                const uint32_t i = stk.back();
                QE e[4];
                half_of(dW[i], 0, e); half_of(dW[i], 1, e + 2);
                if (state[i] == 0) {
                    state[i] = 1;
                    bool wait = false;
                    for (int k = 0; k < 4; ++k)
                        if (!(e[k].d1 & WIDE_LEAF)) {
                            if (e[k].d0 >= dW.size()) return fail("derive_accel: wide record %u refers to record %u of %zu", i, e[k].d0, dW.size());
                            if (state[e[k].d0] == 1) return fail("derive_accel: the wide records are not a tree (cycle through record %u)", e[k].d0);
                            if (state[e[k].d0] == 0) { stk.push_back(e[k].d0); wait = true; }
                        }
                    if (wait) continue;
                }
                // all targets done: order the entries and store the record
                auto nd = [&](const QE& x) -> int { return (x.d1 & WIDE_LEAF) ? -1 : (int)qneed[x.d0]; };
                uint32_t bestNeed = ~0u; int bestArr = 0;
                for (int arr = 0; arr < 8; ++arr) {
                    int ord[4];
                    const int h0 = (arr & 1) ? 2 : 0, h1 = (arr & 1) ? 0 : 2;
                    ord[0] = h0 + ((arr >> 1) & 1); ord[1] = h0 + 1 - ((arr >> 1) & 1);
                    ord[2] = h1 + ((arr >> 2) & 1); ord[3] = h1 + 1 - ((arr >> 2) & 1);
                    uint32_t inner = 0, need = 0;
                    for (int j = 3; j >= 0; --j) {          // j = pop position; `inner` = inner entries popped after j
                        const int n = nd(e[ord[j]]);
                        if (n < 0) continue;
                        need = std::max(need, inner + (uint32_t)n);
                        ++inner;
                    }
                    need = std::max(need, inner);
                    if (need < bestNeed) { bestNeed = need; bestArr = arr; }
                }
                {
                    const int arr = bestArr;
                    int ord[4];
                    const int h0 = (arr & 1) ? 2 : 0, h1 = (arr & 1) ? 0 : 2;
                    ord[0] = h0 + ((arr >> 1) & 1); ord[1] = h0 + 1 - ((arr >> 1) & 1);
                    ord[2] = h1 + ((arr >> 2) & 1); ord[3] = h1 + 1 - ((arr >> 2) & 1);
                    DQuad& q = dQ[i];
                    for (int hh = 0; hh < 2; ++hh) {
                        const QE& a = e[ord[2 * hh]]; const QE& b = e[ord[2 * hh + 1]];
                        DWide& w = q.half[hh];
                        for (int k = 0; k < 3; ++k) { w.lmin[k] = a.mn[k]; w.lmax[k] = a.mx[k]; w.rmin[k] = b.mn[k]; w.rmax[k] = b.mx[k]; }
                        w.ld0 = a.d0; w.ld1 = a.d1; w.rd0 = b.d0; w.rd1 = b.d1;
                    }
                }
                qneed[i] = bestNeed;
                state[i] = 2;
                stk.pop_back();
            }
        }
    }
    uint32_t maxBlasQuad = 0;
    for (auto& kv : blasAt) if (!(kv.second.rootDesc1 & WIDE_LEAF)) maxBlasQuad = std::max(maxBlasQuad, qneed[kv.second.rootDesc0]);
    // stack need: TLAS part
    // (cooperative kernel: the instances of a top-level leaf are pushed as 16-bit masks, one entry per 16 instances,
    //  and the entry being consumed is pushed back while one of its instances is walked)
    std::vector<uint32_t> needT(nTop, 0), needC(nTop, 0), needTopOnly(nTop, 0);
    uint32_t maxBlasCoop = 0, maxBlasAny = 0;
    std::vector<DNode> dTc(dT);
    for (uint32_t i = nTop; i-- > 0;) {
        const BlobNode& n = tnodes[i];
        if (n.w0 & LEAF_BIT) {
            const uint32_t cnt = n.w0 & 0x7fffffffu;
            uint32_t mx = 0, mxc = 0;
            for (uint32_t k = 0; k < cnt; ++k) {
                const BlasInfo& bi = blasAt[binst[n.w1 + k].instanceOffset];
                mx = std::max(mx, bi.need); mxc = std::max(mxc, bi.coopNeed); maxBlasAny = std::max(maxBlasAny, bi.anyNeed);
            }
            maxBlasCoop = std::max(maxBlasCoop, mxc);
            needTopOnly[i] = (cnt + 15u) / 16u;
            needT[i] = (cnt ? cnt - 1 : 0) + mx;
            needC[i] = (cnt + 15u) / 16u + mxc;
        } else {
            needT[i] = std::max(1u + needT[n.w0], needT[n.w1]);   // children have larger indices (DFS pre-order)
            // cooperative kernel: its own copy of the top-level nodes with the smaller-need child in the followed slot
            if (needC[n.w1] < needC[n.w0]) std::swap(dTc[i].w0, dTc[i].w1);
            needC[i] = std::max(1u + needC[dTc[i].w0], needC[dTc[i].w1]);
            needTopOnly[i] = std::max(1u + needTopOnly[dTc[i].w0], needTopOnly[dTc[i].w1]);
        }
    }
    auto ac = std::make_unique<AccelCache>();
    ac->stackNeed = std::max(1u, needT[0]) + 1u + maxLeafChunks;
    // oversized leaves are cut into 8-triangle work items: all but the first piece of each child are pushed
    ac->coopNeed = std::max(1u, needC[0]) + 1u + 2u * ((std::max(maxLeafTris, 1u) + 7u) / 8u - 1u);
    ac->topNeed = std::max(1u, needTopOnly[0]) + 1u;
    {
        uint32_t masks = 0;
        for (uint32_t i = 0; i < nTop; ++i) if (tnodes[i].w0 & LEAF_BIT) masks += ((tnodes[i].w0 & 0x7fffffffu) + 15u) / 16u;
        // flat top level: <= 64 nodes (one reach bit each); the pending instances of a ray are a bitmap of <= 8 words per lane
        ac->topFlat = (nTop <= 64 && nInst <= 256) ? nTop : 0u;
        // spare word of a top-level leaf: it holds instances whose BLAS is a single leaf of <= 8 triangles (pool engine)
        for (uint32_t i = 0; i < nTop; ++i) {
            dT[i].w3 = 0;
            if (!(tnodes[i].w0 & LEAF_BIT)) continue;
            for (uint32_t k = 0; k < (tnodes[i].w0 & 0x7fffffffu); ++k) {
                const DInst& di = dI[tnodes[i].w1 + k];
                if ((di.rootDesc1 & WIDE_LEAF) && wide_count(di.rootDesc1) <= 8u) { dT[i].w3 = 1; ac->leafRoots = true; }
            }
        }
        ac->nInst = nInst;
        ac->topFlatNeed = std::max(1u, (nInst + 31u) / 32u);       // words per lane of the pending-instance bitmap
        (void)masks;
    }
    ac->blasNeed = maxBlasCoop;
    ac->blasNeedAny = maxBlasAny;
    for (int k = 0; k < 3; ++k) { ac->sceneLo[k] = tnodes[0].bottom[k]; ac->sceneHi[k] = tnodes[0].top[k]; }
    ac->sbtOffsets = sbtOffsets || hugeLeaf;      // (either way: the reference-order kernel, which reads the blob's own node layout)
    ac->groupCount = groupCount; ac->groupIdentity = groupIdentity;
    ac->unifiedRoot = unifiedRoot; ac->unifiedNeed = unifiedNeed;
    ac->quadNeed = maxBlasQuad; ac->quadUnifiedNeed = unifiedRoot ? qneed[unifiedRoot] + 1u : 0u;
    ac->nWide = (uint32_t)dW.size();
    // per-lane kernels: [need][64 lanes] words of LDS per wave, 64 KB at most
    if (ac->stackNeed > 250) return fail("BVH too deep for the LDS traversal stack: %u entries per ray needed, 250 available", ac->stackNeed);
    auto up = [&](auto*& dptr, const auto& vec) -> hipError_t {
        using T = typename std::remove_reference<decltype(vec)>::type::value_type;
        const size_t bytes = std::max<size_t>(vec.size(), 1) * sizeof(T);
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&dptr), bytes);
        if (e != hipSuccess) return e;
        return vec.empty() ? hipSuccess : hipMemcpy(dptr, vec.data(), vec.size() * sizeof(T), hipMemcpyHostToDevice);
    };
    HIP_OK(up(ac->tnodes, dT));
    HIP_OK(up(ac->ctnodes, dTc));
    HIP_OK(up(ac->insts, dI));
    HIP_OK(up(ac->bnodes, dB));
    HIP_OK(up(ac->tris, dTri));
    HIP_OK(up(ac->wide, dW));
    if (!dQ.empty()) HIP_OK(up(ac->quad, dQ));
    {
        const std::vector<uint32_t> gb{groupBits[0], groupBits[1], groupBits[2], groupBits[3], groupBits[4], groupBits[5], groupBits[6], groupBits[7], 0u};
        HIP_OK(up(ac->groupBits, gb));
    }
    // packed-word limits of the cooperative engines (kernels.h) and their LDS footprint; beyond them the per-lane wide kernel runs
    ac->coopOK = coopOK && dTri.size() <= RDX_COOP_MAX_TRI_SLOTS - 1u && dW.size() < RDX_COOP_MAX_WIDE &&
                 coop_lds_words(ac->coopNeed) <= RDX_LDS_WORDS_PER_WAVE_MAX &&
                 pool_lds_words(std::max(ac->topNeed, ac->topFlatNeed), std::max({ac->blasNeed, ac->blasNeedAny, ac->quadNeed, ac->quadUnifiedNeed})) <= RDX_LDS_WORDS_PER_WAVE_MAX;
    if (std::getenv("RDX_VERBOSE"))
        std::fprintf(stderr, "[rdx] accel: %u top nodes, %u instances, %zu wide nodes, %zu triangle slots, stack need %u (cooperative kernel %u = top %u + BLAS %u; quad walk %u)\n",
                     nTop, nInst, dW.size(), dTri.size(), ac->stackNeed, ac->coopNeed, ac->topNeed, ac->blasNeed, ac->quadNeed);
    ac->version = tb->version;
    if (acc(tb)) acc(tb)->release();
    acc(tb) = std::move(ac);
    return 0;
}

// smallChunk: the caller's launches will not fill the chip (trace_rays_device: chunks of <= RDX_QUAD_AUTO_MAX_PATHS paths)
AccelView view_of(const rdx_buffer_s* tb, bool smallChunk = false)
{
    AccelView v{};
    v.tnodes = acc(tb)->tnodes; v.ctnodes = acc(tb)->ctnodes; v.insts = acc(tb)->insts; v.bnodes = acc(tb)->bnodes; v.tris = acc(tb)->tris;
    v.wide = acc(tb)->wide;
    v.status = g.dStatus;
    v.kernel = acc(tb)->sbtOffsets ? 0u : (g.kernel >= 2 && !acc(tb)->coopOK) ? 1u : (uint32_t)g.kernel;
    v.stackNeed = acc(tb)->stackNeed;
    v.coopNeed = acc(tb)->coopNeed;
    // culled walk (docs/CULLED_WALK.md): with the conditioning gate that makes it exact it pays on the scene whose BVH lives in HBM
    // (10.4 M triangles: 66.2 vs 72.1 ms) and not on scenes that fit the caches (Sponza-class: 25.9 vs 25.3 ms exhaustive -- the
    // gate's ~45 vector instructions per node cost what the skipped triangle tests save) -- hence the size rule
    v.cull = (v.kernel == 3 && (g.cull > 0 || (g.cull < 0 && acc(tb)->nWide >= RDX_CULL_AUTO_MIN_WIDE))) ? 1u : 0u;
    v.topNeed = acc(tb)->topNeed; v.blasNeed = v.cull ? acc(tb)->blasNeedAny : acc(tb)->blasNeed;
    v.topFlat = g.topFlat ? acc(tb)->topFlat : 0u;
    v.numInsts = acc(tb)->nInst;
    if (v.topFlat) v.topNeed = acc(tb)->topFlatNeed;          // flat top level: words per lane of the pending-instance bitmap
    v.leafRoots = (v.topFlat && g.inlineLeafRoots && acc(tb)->leafRoots) ? 1u : 0u;
    v.unifiedRoot = 0;
    if (v.kernel == 3 && !v.topFlat && g.unifiedTree && acc(tb)->unifiedRoot) {
        // unified tree: no top-level state per lane at all (one bitmap word stays allocated: the engine's flat-mode bookkeeping)
        v.unifiedRoot = acc(tb)->unifiedRoot;
        v.topFlat = 1u; v.topNeed = 1u; v.leafRoots = 0u;
        v.blasNeed = acc(tb)->unifiedNeed;
    }
    // exhaustive walk of the pool engine: quad records, two tree levels per fetch (option "quad"; the culled walk keeps the
    // wide records, whose children carry the normal cones)
    v.quad = nullptr;
    // (not for the unified tree: its always-entered fan-outs gain nothing from a second level per item -- 39.4 vs 35.4 ms on the
    // 400-instance scene)
    v.quadWaves = 6u;
    if (v.kernel == 3 && !v.cull && acc(tb)->quad && !v.unifiedRoot && (g.quad > 0 || (g.quad < 0 && smallChunk))) {
        v.quad = acc(tb)->quad;
        v.quadWaves = (!smallChunk && pool_lds_words(v.topNeed, std::max(acc(tb)->quadNeed, v.blasNeed)) <= 1462u) ? 7u : 6u;      // (7 waves: 160 KB / 28)
        // (launches without a quad variant walk the wide records on the same view)
        v.blasNeed = std::max(acc(tb)->quadNeed, v.blasNeed);
    }
    v.groupCount = (v.topFlat && !v.unifiedRoot && g.groupInstances) ? acc(tb)->groupCount : 0u;
    v.groupBits = acc(tb)->groupBits;
    return v;
}

// ---- path streams --------------------------------------------------------------------------------
int ensure_group(Context::Group& G, size_t paths)
{
    if (paths <= G.cap) return 0;
    float4** arr[] = {&G.ps.rayO, &G.ps.rayD, &G.ps.thr, &G.ps.col, &G.ps.hitA, &G.ps.nRayO, &G.ps.nRayD,
                      &G.ps.nThr, &G.ps.nCol, &G.ps.shO, &G.ps.colLit, &G.ps.colSh};
    for (auto a : arr) { if (*a) HIP_IGN(hipFree(*a)); *a = nullptr; }
    if (G.ps.hitInst) HIP_IGN(hipFree(G.ps.hitInst));
    G.ps.hitInst = nullptr;
    G.cap = 0;
    for (auto a : arr) HIP_OK(hipMalloc(reinterpret_cast<void**>(a), paths * sizeof(float4)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&G.ps.hitInst), paths * sizeof(uint32_t)));
    G.cap = paths;
    return 0;
}

int ensure_samples(size_t samplesTimesPixels)
{
    if (samplesTimesPixels <= g.sampleCap) return 0;
    if (g.sampleColor) HIP_IGN(hipFree(g.sampleColor));
    g.sampleColor = nullptr; g.sampleCap = 0;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&g.sampleColor), samplesTimesPixels * sizeof(float4)));
    g.sampleCap = samplesTimesPixels;
    return 0;
}

// pixels owned by (rank, world) for a w x h image, ascending tile id, row-major inside a tile
void owned_pixel_list(uint32_t w, uint32_t h, uint32_t tileW, uint32_t tileH, uint32_t rank, uint32_t world,
                      std::vector<uint32_t>& out)
{
    const uint32_t tilesX = (w + tileW - 1) / tileW, tilesY = (h + tileH - 1) / tileH;
    out.clear();
    for (uint32_t t = rank; t < tilesX * tilesY; t += world) {
        const uint32_t x0 = (t % tilesX) * tileW, y0 = (t / tilesX) * tileH;
        for (uint32_t y = y0; y < std::min(h, y0 + tileH); ++y)
            for (uint32_t x = x0; x < std::min(w, x0 + tileW); ++x) out.push_back(y * w + x);
    }
}

int ensure_owned(uint32_t w, uint32_t h)
{
    if (g.world <= 1) { g.ownedCount = w * h; return 0; }
    if (g.ownedPixels && g.ownedW == w && g.ownedH == h && g.ownedRank == g.rank && g.ownedWorld == g.world &&
        g.ownedTileW == g.tileW && g.ownedTileH == g.tileH)
        return 0;
    std::vector<uint32_t> px;
    owned_pixel_list(w, h, g.tileW, g.tileH, g.rank, g.world, px);
    if (g.ownedPixels) HIP_IGN(hipFree(g.ownedPixels));
    g.ownedPixels = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&g.ownedPixels), std::max<size_t>(px.size(), 1) * 4));
    if (!px.empty()) HIP_OK(hipMemcpy(g.ownedPixels, px.data(), px.size() * 4, hipMemcpyHostToDevice));
    g.ownedCount = (uint32_t)px.size();
    g.ownedW = w; g.ownedH = h; g.ownedRank = g.rank; g.ownedWorld = g.world; g.ownedTileW = g.tileW; g.ownedTileH = g.tileH;
    return 0;
}

int camera_args(const PhysicalCamera& cam, CameraArgs& C)
{
    C.cam = cam;
    // EulerX/Y/ZToMat4x4 (math.cl:185-252): per-frame constants.  cos / sin are evaluated ON THE DEVICE (k_euler_trig) so
    // that they are the OCML values the reference's own cos() / sin() give on this GPU -- libm's differ in the last bit
    // and every primary ray would with them; the six values are cached until the camera angles change.
    float (&cachedAngles)[3] = g.camAngles; float (&cachedTrig)[6] = g.camTrig; bool& cached = g.camCached;
    if (!cached || std::memcmp(cachedAngles, &cam.wx, 12) != 0) {
        float* d6 = nullptr;
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&d6), 6 * sizeof(float)));
        launch_euler_trig(g.stream, cam.wx, cam.wy, cam.wz, d6);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(cachedTrig, d6, sizeof cachedTrig, hipMemcpyDeviceToHost, g.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
        HIP_IGN(hipFree(d6));
        HIP_OK(e);
        std::memcpy(cachedAngles, &cam.wx, 12);
        cached = true;
    }
    const float cx = cachedTrig[0], sx = cachedTrig[1], cy = cachedTrig[2], sy = cachedTrig[3], cz = cachedTrig[4], sz = cachedTrig[5];
    const float rx[16] = {1, 0, 0, 0, 0, cx, -sx, 0, 0, sx, cx, 0, 0, 0, 0, 1};
    const float ry[16] = {cy, 0, sy, 0, 0, 1, 0, 0, -sy, 0, cy, 0, 0, 0, 0, 1};
    const float rz[16] = {cz, -sz, 0, 0, sz, cz, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(C.rotX, rx, 64); std::memcpy(C.rotY, ry, 64); std::memcpy(C.rotZ, rz, 64);
    return 0;
}

int scene_args(SceneArgs& sc)
{
    for (int i : {4, 5, 7, 8, 9, 10})
        if (!g.slots[i] || !known_buffer(g.slots[i])) return fail("descriptor slot %d is not a buffer", i);
    auto ptr = [&](int i) { return dp(static_cast<rdx_buffer_s*>(g.slots[i])); };
    sc.scene = static_cast<const SceneProperties*>(ptr(4));
    sc.meshInfo = static_cast<const MeshInfo*>(ptr(5));
    sc.indexData = static_cast<const uint32_t*>(ptr(7));
    sc.uvData = static_cast<const float*>(ptr(8));
    sc.normalData = static_cast<const float*>(ptr(9));
    sc.materials = static_cast<const Material*>(ptr(10));
    if (static_cast<rdx_buffer_s*>(g.slots[4])->size < sizeof(SceneProperties)) return fail("scene buffer smaller than SceneProperties");
    // slots 11 / 12: texture array + sampler.  Read only when option "textures" is on (the live reference shader has its
    // reads stubbed to 0, samples/shader.cl:379-445)
    sc.tex = TexView{nullptr, 0, 0, 0, 0};
    if (g.textures && g.slots[11] && known_buffer(g.slots[11])) {
        const auto* img = static_cast<const rdx_buffer_s*>(g.slots[11]);
        if (img->imgW && img->imgH && img->imgLayers) {
            uint32_t mode = TEX_ADDR_REPEAT, linear = 0;           // no sampler bound: repeat + nearest
            for (auto& sm : g.samplers)
                if (sm.get() == g.slots[12]) {
                    mode = sm->addressing == 0x1131 ? TEX_ADDR_CLAMP_TO_EDGE : sm->addressing == 0x1132 ? TEX_ADDR_CLAMP
                         : sm->addressing == 0x1134 ? TEX_ADDR_MIRRORED : TEX_ADDR_REPEAT;
                    linear = sm->filter == 0x1141 ? TEX_LINEAR : 0u;
                }
            sc.tex = TexView{static_cast<const uint8_t*>(dp(img)), img->imgW, img->imgH, img->imgLayers, TEX_ENABLED | linear | (mode << TEX_ADDR_SHIFT)};
        }
    }
    return 0;
}

// The status word the traversal kernels raise (bit 0: a wave hit its iteration bound) is tested and cleared after EVERY
// synchronise that follows a traversal launch -- frames and the batch seams alike -- so that a raised bit is reported to
// the call that caused it and never to the next one.
bool take_status()
{
    if (!g.hStatus || !*static_cast<volatile uint32_t*>(g.hStatus)) return false;
    *g.hStatus = 0;
    return true;
}

struct StageTimer {
    // per-stage HIP-event timing (profiling mode): events are recorded around each launch and
    // resolved after the frame so that the stream is never drained in the middle
    struct Span { hipEvent_t a, b; float* dst; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    hipEvent_t get()
    {
        if (used == pool.size()) { hipEvent_t e; HIP_IGN(hipEventCreate(&e)); pool.push_back(e); }
        return pool[used++];
    }
    void begin(float* dst, hipStream_t st = nullptr) { if (!g.profiling) return; Span s{get(), get(), dst}; HIP_IGN(hipEventRecord(s.a, st ? st : g.stream)); spans.push_back(s); }
    void end(hipStream_t st = nullptr) { if (!g.profiling) return; HIP_IGN(hipEventRecord(spans.back().b, st ? st : g.stream)); }
    void resolve()
    {
        for (auto& s : spans) { float ms = 0; HIP_IGN(hipEventElapsedTime(&ms, s.a, s.b)); *s.dst += ms; }
        spans.clear(); used = 0;
    }
    ~StageTimer() { for (hipEvent_t e : pool) HIP_IGN(hipEventDestroy(e)); }      // (worker threads of the multi-device mode end per frame)
};
thread_local StageTimer g_timer;

} // namespace

// ------------------------------------------------------------------------------------------------
// platform
// ------------------------------------------------------------------------------------------------
// streams, events, counters of the calling thread's context `g` on HIP device `device` (which must be current)
static int init_device_state(int device)
{
    g.device = device;
    HIP_OK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_OK(hipEventCreate(&g.evA));
    HIP_OK(hipEventCreate(&g.evB));
    HIP_OK(hipEventCreateWithFlags(&g.evChunk, hipEventDisableTiming));
    for (int k = 0; k < Context::MAX_GROUPS; ++k) {
        Context::Group& G = g.groups[k];
        if (k == 0) G.s0 = g.stream; else HIP_OK(hipStreamCreateWithFlags(&G.s0, hipStreamNonBlocking));
        HIP_OK(hipStreamCreateWithFlags(&G.s1, hipStreamNonBlocking));
        for (int i = 0; i < 64; ++i) {
            HIP_OK(hipEventCreateWithFlags(&G.evShade[i], hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&G.evShadow[i], hipEventDisableTiming));
        }
        HIP_OK(hipEventCreateWithFlags(&G.evDone, hipEventDisableTiming));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&G.dCounts), 256 * sizeof(uint32_t)));
        HIP_OK(hipMemset(G.dCounts, 0, 256 * sizeof(uint32_t)));
        HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&G.hCounts), 256 * sizeof(uint32_t), hipHostMallocDefault));
    }
    g.dCounts = g.groups[0].dCounts;
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&g.hStatus), 64, hipHostMallocMapped));
    *g.hStatus = 0;
    HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&g.dStatus), g.hStatus, 0));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&g.dVisit), 64 * 8 * sizeof(unsigned long long)));     // [bounce][class*4 + kind]
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&g.hVisit), 64 * 8 * sizeof(unsigned long long), hipHostMallocDefault));
    g.initialized = true;
    return 0;
}

static void release_device_state()
{
    if (!g.initialized) return;
    HIP_IGN(hipSetDevice(g.device));
    HIP_IGN(hipStreamSynchronize(g.stream));
    for (int k = 0; k < Context::MAX_GROUPS; ++k) {
        Context::Group& G = g.groups[k];
        float4** arr[] = {&G.ps.rayO, &G.ps.rayD, &G.ps.thr, &G.ps.col, &G.ps.hitA, &G.ps.nRayO, &G.ps.nRayD, &G.ps.nThr,
                          &G.ps.nCol, &G.ps.shO, &G.ps.colLit, &G.ps.colSh};
        for (auto a : arr) { if (*a) HIP_IGN(hipFree(*a)); *a = nullptr; }
        if (G.ps.hitInst) HIP_IGN(hipFree(G.ps.hitInst));
        if (G.dCounts) HIP_IGN(hipFree(G.dCounts));
        if (G.hCounts) HIP_IGN(hipHostFree(G.hCounts));
        if (G.sortBins) HIP_IGN(hipFree(G.sortBins));
        if (G.permE) HIP_IGN(hipFree(G.permE));
        if (G.sortKey) HIP_IGN(hipFree(G.sortKey));
        G.sortKey = nullptr;
        for (int i = 0; i < 64; ++i) { HIP_IGN(hipEventDestroy(G.evShade[i])); HIP_IGN(hipEventDestroy(G.evShadow[i])); }
        HIP_IGN(hipEventDestroy(G.evDone));
        HIP_IGN(hipStreamSynchronize(G.s1)); HIP_IGN(hipStreamDestroy(G.s1));
        if (k > 0) { HIP_IGN(hipStreamSynchronize(G.s0)); HIP_IGN(hipStreamDestroy(G.s0)); }
    }
    if (g.sampleColor) HIP_IGN(hipFree(g.sampleColor));
    if (g.ownedPixels) HIP_IGN(hipFree(g.ownedPixels));
    for (void* p : g.gatherStage) if (p) HIP_IGN(hipFree(p));
    if (g.hStatus) HIP_IGN(hipHostFree(g.hStatus));
    g.hStatus = nullptr; g.dStatus = nullptr;
    if (g.dVisit) HIP_IGN(hipFree(g.dVisit));
    if (g.hVisit) HIP_IGN(hipHostFree(g.hVisit));
    HIP_IGN(hipEventDestroy(g.evA)); HIP_IGN(hipEventDestroy(g.evB)); HIP_IGN(hipEventDestroy(g.evChunk));
    HIP_IGN(hipStreamDestroy(g.stream));
}

// GPU-assisted candidate evaluation of the BVH builder (bvh_build.h GpuBinner; reference: the candidate loop of
// radiance/src/bvh.cpp:90-205): the primitives of a large mesh live on the device while it is built, the binning pass of its
// large nodes is one kernel (kernels.hip k_bvh_bin).  Builder threads share one set of staging buffers behind a mutex; a call
// is a few hundred microseconds.
struct HipBinner final : GpuBinner {
    std::mutex m;
    std::map<uint64_t, float*> sets;
    uint64_t next = 1;
    std::atomic<uint64_t> calls{0};     // nodes binned on the device (rdx_debug_gpu_bin_calls)
    int device = 0;
    uint32_t* dWork = nullptr; size_t workCap = 0;
    float* dCand = nullptr; uint32_t* dOut = nullptr; uint32_t* hOut = nullptr;
    hipStream_t st = nullptr;
    static constexpr size_t kSlots = 1025, kOutWords = 3 * 7 * kSlots;
    bool ready()
    {
        if (st) return true;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = nullptr; return false; }
        if (hipMalloc(reinterpret_cast<void**>(&dCand), 3 * 1024 * sizeof(float)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&dOut), kOutWords * 4) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&hOut), kOutWords * 4, hipHostMallocDefault) != hipSuccess) { shutdown(); return false; }
        return true;
    }
    void shutdown()
    {
        std::lock_guard<std::mutex> lk(m);
        for (auto& kv : sets) HIP_IGN(hipFree(kv.second));
        sets.clear();
        if (dWork) HIP_IGN(hipFree(dWork));
        if (dCand) HIP_IGN(hipFree(dCand));
        if (dOut) HIP_IGN(hipFree(dOut));
        if (hOut) HIP_IGN(hipHostFree(hOut));
        if (st) HIP_IGN(hipStreamDestroy(st));
        dWork = nullptr; workCap = 0; dCand = nullptr; dOut = nullptr; hOut = nullptr; st = nullptr;
    }
    uint64_t upload(const float* prims9, uint32_t n) override
    {
        std::lock_guard<std::mutex> lk(m);
        if (hipSetDevice(device) != hipSuccess || !ready()) return 0;
        float* d = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&d), (size_t)n * 36) != hipSuccess) return 0;
        if (hipMemcpy(d, prims9, (size_t)n * 36, hipMemcpyHostToDevice) != hipSuccess) { HIP_IGN(hipFree(d)); return 0; }
        sets[next] = d;
        return next++;
    }
    void release(uint64_t h) override
    {
        std::lock_guard<std::mutex> lk(m);
        auto it = sets.find(h);
        if (it == sets.end()) return;
        HIP_IGN(hipSetDevice(device));
        HIP_IGN(hipFree(it->second));
        sets.erase(it);
    }
    bool bin(uint64_t h, const uint32_t* work, size_t n, const std::vector<float> cand[3], std::vector<float>& out) override
    {
        std::lock_guard<std::mutex> lk(m);
        auto it = sets.find(h);
        if (it == sets.end() || n == 0 || n > 0xffffffffull) return false;
        if (hipSetDevice(device) != hipSuccess || !ready()) return false;
        uint32_t K[3];
        for (int a = 0; a < 3; ++a) { K[a] = (uint32_t)cand[a].size(); if (K[a] > 1024u) return false; }
        if (workCap < n) {
            if (dWork) HIP_IGN(hipFree(dWork));
            dWork = nullptr; workCap = 0;
            if (hipMalloc(reinterpret_cast<void**>(&dWork), n * 4) != hipSuccess) return false;
            workCap = n;
        }
        bool ok = hipMemcpyAsync(dWork, work, n * 4, hipMemcpyHostToDevice, st) == hipSuccess;
        for (int a = 0; a < 3 && ok; ++a)
            if (K[a]) ok = hipMemcpyAsync(dCand + 1024 * a, cand[a].data(), K[a] * sizeof(float), hipMemcpyHostToDevice, st) == hipSuccess;
        if (!ok) { HIP_IGN(hipStreamSynchronize(st)); return false; }
        launch_bvh_bin(st, it->second, dWork, (uint32_t)n, dCand, K, dOut);
        ok = hipMemcpyAsync(hOut, dOut, kOutWords * 4, hipMemcpyDeviceToHost, st) == hipSuccess;
        if (hipStreamSynchronize(st) != hipSuccess || !ok || hipGetLastError() != hipSuccess) return false;
        calls.fetch_add(1);
        out.clear();
        for (int a = 0; a < 3; ++a) {
            if (!K[a]) continue;
            const uint32_t* o = hOut + (size_t)a * 7 * kSlots;
            const size_t base = out.size(), nb = K[a] + 1u;
            out.resize(base + 7 * nb);
            for (size_t b = 0; b < nb; ++b) {
                std::memcpy(&out[base + b], &o[b], 4);
                float* q = &out[base + nb + 6 * b];
                for (int k = 0; k < 6; ++k) {
                    if (o[b] == 0u) { q[k] = k < 3 ? FLT_MAX : -FLT_MAX; continue; }
                    const uint32_t key = o[(size_t)(1 + k) * kSlots + b];
                    const uint32_t bits = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key;
                    std::memcpy(&q[k], &bits, 4);
                }
            }
        }
        return true;
    }
};
static HipBinner g_hipBinner;
extern "C" unsigned long long rdx_debug_gpu_bin_calls(void) { return g_hipBinner.calls.load(); }

extern "C" int rdx_init(int device)
{
    if (g0.initialized) return 0;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail("no HIP device available (hipGetDeviceCount -> %d, count %d): the ray-tracing core needs a GPU", (int)e, count);
    if (device < 0) { HIP_OK(hipGetDevice(&device)); }
    if (device >= count) return fail("device ordinal %d out of range (%d devices)", device, count);
    HIP_OK(hipSetDevice(device));
    g_phys[0] = device;
    g_ndev = 1;
    if (init_device_state(device)) return -1;
    g_hipBinner.device = device;
    set_gpu_binner(g0.gpuBuild ? &g_hipBinner : nullptr, (size_t)g0.gpuBuildMin);
    return 0;
}

// Single-process multi-device rendering (SURVEY 8b "Threading", 8e): after this call every buffer lives on all `n` devices
// (writes are replicated, reads come from device 0) and rdx_trace_rays renders the frame sharded by interleaved 64x64 tiles --
// one internal host thread per device, the caller still makes one blocking call -- then gathers RGBA8 and imageScratch tiles
// to device 0 with peer copies.  ordinals[i] = HIP device of logical device i (NULL: 0..n-1); logical device 0 is the one
// rdx_init chose.  Must be called before any buffer is created.  RDX_ALLOW_VIRTUAL_DEVICES=1 lets several logical devices
// share one GPU (rehearsal on a 1-GPU box; results are identical by construction).
extern "C" int rdx_init_devices(uint32_t n, const int* ordinals)
{
    if (!g0.initialized && rdx_init(ordinals ? ordinals[0] : -1)) return -1;
    if (n == 0 || n > (uint32_t)RDX_MAX_DEVICES) return fail("rdx_init_devices: %u devices (1..%d supported)", n, RDX_MAX_DEVICES);
    if (g_ndev > 1) return (uint32_t)g_ndev == n ? 0 : fail("rdx_init_devices: already initialised with %d devices", g_ndev);
    if (n == 1) return 0;
    if (!g0.buffers.empty()) return fail("rdx_init_devices must be called before any buffer is created");
    for (auto& sh : g0.shaders)
        if (sh->program) return fail("rdx_init_devices must be called before a user shader program is compiled (its code object is loaded on device 0 only)");
    int count = 0;
    HIP_OK(hipGetDeviceCount(&count));
    const bool virt = std::getenv("RDX_ALLOW_VIRTUAL_DEVICES") && std::atoi(std::getenv("RDX_ALLOW_VIRTUAL_DEVICES")) != 0;
    for (uint32_t d = 1; d < n; ++d) {
        int phys = ordinals ? ordinals[d] : (int)((g_phys[0] + d) % (uint32_t)count);
        if (phys < 0 || phys >= count) return fail("rdx_init_devices: HIP device %d does not exist (%d devices)", phys, count);
        for (uint32_t e = 0; e < d && !virt; ++e)
            if (g_phys[e] == phys) return fail("rdx_init_devices: %u devices requested, %d present (set RDX_ALLOW_VIRTUAL_DEVICES=1 to let logical devices share a GPU)", n, count);
        g_phys[d] = phys;
    }
    for (uint32_t d = 1; d < n; ++d) {
        g_dev[d] = new Context();
        tl_ctx = g_dev[d]; tl_dev = (int)d;
        hipError_t e = hipSetDevice(g_phys[d]);
        int rc = e == hipSuccess ? init_device_state(g_phys[d]) : -1;
        if (rc == 0 && g_phys[d] != g_phys[0]) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, g_phys[0], g_phys[d]) == hipSuccess && can) {
                HIP_IGN(hipSetDevice(g_phys[0])); HIP_IGN(hipDeviceEnablePeerAccess(g_phys[d], 0)); (void)hipGetLastError();
            }
        }
        std::string msg = g.err;
        tl_ctx = &g0; tl_dev = 0;
        HIP_IGN(hipSetDevice(g_phys[0]));
        if (rc) {
            // give back what was set up so far: contexts 1..d (the failing one holds whatever its init got to)
            for (uint32_t e = 1; e <= d; ++e) {
                tl_ctx = g_dev[e]; tl_dev = (int)e;
                HIP_IGN(hipSetDevice(g_phys[e]));
                g.initialized = true;          // release_device_state frees the non-null members
                release_device_state();
                tl_ctx = &g0; tl_dev = 0;
                delete g_dev[e]; g_dev[e] = nullptr;
            }
            HIP_IGN(hipSetDevice(g_phys[0]));
            return fail("rdx_init_devices: device %u (HIP %d): %s", d, g_phys[d], msg.c_str());
        }
    }
    g_ndev = (int)n;
    return 0;
}

extern "C" int rdx_device_count(void) { return g_ndev; }

extern "C" int rdx_shutdown(void)
{
    if (!g0.initialized) return 0;
    for (int d = 1; d < g_ndev; ++d) {
        tl_ctx = g_dev[d]; tl_dev = d;
        HIP_IGN(hipSetDevice(g_phys[d]));
        for (auto& b : g0.buffers) { if (b->accelRep[d]) b->accelRep[d]->release(); if (b->rep[d]) HIP_IGN(hipFree(b->rep[d])); }
        release_device_state();
        tl_ctx = &g0; tl_dev = 0;
        delete g_dev[d];
        g_dev[d] = nullptr;
    }
    g_ndev = 1;
    HIP_IGN(hipSetDevice(g_phys[0]));
    set_gpu_binner(nullptr, 0);
    g_hipBinner.shutdown();
    HIP_IGN(hipStreamSynchronize(g.stream));
    for (auto& b : g.buffers) { if (b->accel) b->accel->release(); if (b->owned && b->dptr) HIP_IGN(hipFree(b->dptr)); }
    {   // (modules created from the same text share one compiled program: user_shader.cpp's cache)
        std::vector<UserProgram*> seen;
        for (auto& sh : g.shaders)
            if (sh->program && std::find(seen.begin(), seen.end(), sh->program) == seen.end()) { seen.push_back(sh->program); release_user_shader(sh->program); }
    }
    g.buffers.clear(); g.blases.clear(); g.shaders.clear();
    release_device_state();
    g = Context{};
    return 0;
}

extern "C" const char* rdx_last_error(void) { return g.err.c_str(); }

extern "C" int rdx_device_name(char* out, size_t cap)
{
    if (!g.initialized) return fail("rdx_init has not been called");
    hipDeviceProp_t p;
    HIP_OK(hipGetDeviceProperties(&p, g.device));
    snprintf(out, cap, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// buffers
// ------------------------------------------------------------------------------------------------
extern "C" rdx_buffer rdx_buffer_create(size_t size)
{
    if (!g.initialized) { fail("rdx_init has not been called"); return nullptr; }
    auto b = std::make_unique<rdx_buffer_s>();
    b->size = size;
    HIP_OKP(hipMalloc(&b->dptr, std::max<size_t>(size, 16)));
    HIP_OKP(hipMemset(b->dptr, 0, std::max<size_t>(size, 16)));
    for (int d = 1; d < g_ndev; ++d) {          // multi-device mode: one copy per device
        HIP_OKP(hipSetDevice(g_phys[d]));
        hipError_t e = hipMalloc(&b->rep[d], std::max<size_t>(size, 16));
        if (e == hipSuccess) e = hipMemset(b->rep[d], 0, std::max<size_t>(size, 16));
        HIP_IGN(hipSetDevice(g_phys[0]));
        HIP_OKP(e);
    }
    g.buffers.push_back(std::move(b));
    return g.buffers.back().get();
}

// ---- texture arrays and samplers: replaces CreateImageArray / CreateSampler / ReadImage / WriteImage
//      (radiance/src/radiance.cpp:96-137, 202-224): a 2D image array of CL_RGBA / CL_UNSIGNED_INT8 texels, layer-major
extern "C" rdx_buffer rdx_image_array_create(uint32_t width, uint32_t height, uint32_t layers)
{
    if ((uint64_t)width * height * std::max(layers, 1u) * 4ull > (1ull << 36)) { fail("CreateImageArray: %ux%ux%u is too large", width, height, layers); return nullptr; }
    rdx_buffer b = rdx_buffer_create((size_t)width * height * layers * 4);
    if (!b) return nullptr;
    b->imgW = width; b->imgH = height; b->imgLayers = layers;
    return b;
}

static int image_region(rdx_buffer img, uint32_t width, uint32_t height, size_t layer, const char* what)
{
    if (!img || !known_buffer(img) || !img->imgW) return fail("%s: not an image array", what);
    if (layer >= img->imgLayers) return fail("%s: layer %zu of %u", what, layer, img->imgLayers);
    if (width > img->imgW || height > img->imgH) return fail("%s: region %ux%u exceeds the image (%ux%u)", what, width, height, img->imgW, img->imgH);
    return 0;
}

// origin (0, 0, layer), region (width, height, 1), host rows tightly packed -- as radiance.cpp:202-224 calls clEnqueue{Write,Read}Image
extern "C" int rdx_image_write(rdx_buffer img, uint32_t width, uint32_t height, size_t layer, const void* rgba8)
{
    if (image_region(img, width, height, layer, "WriteImage")) return -1;
    if (!width || !height) return 0;
    uint8_t* dst = static_cast<uint8_t*>(img->dptr) + layer * (size_t)img->imgW * img->imgH * 4;
    HIP_OK(hipMemcpy2D(dst, (size_t)img->imgW * 4, rgba8, (size_t)width * 4, (size_t)width * 4, height, hipMemcpyHostToDevice));
    for (int d = 1; d < g_ndev; ++d) {
        HIP_OK(hipSetDevice(g_phys[d]));
        hipError_t e = hipMemcpy2D(static_cast<uint8_t*>(img->rep[d]) + layer * (size_t)img->imgW * img->imgH * 4, (size_t)img->imgW * 4, rgba8,
                                   (size_t)width * 4, (size_t)width * 4, height, hipMemcpyHostToDevice);
        HIP_IGN(hipSetDevice(g_phys[0]));
        HIP_OK(e);
    }
    ++img->version;
    return 0;
}

extern "C" int rdx_image_read(rdx_buffer img, uint32_t width, uint32_t height, size_t layer, void* rgba8)
{
    if (image_region(img, width, height, layer, "ReadImage")) return -1;
    if (!width || !height) return 0;
    const uint8_t* src = static_cast<const uint8_t*>(img->dptr) + layer * (size_t)img->imgW * img->imgH * 4;
    HIP_OK(hipMemcpy2D(rgba8, (size_t)width * 4, src, (size_t)img->imgW * 4, (size_t)width * 4, height, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" rdx_sampler rdx_sampler_create(uint32_t addressingMode, uint32_t filterMode)
{
    if (addressingMode < 0x1131 || addressingMode > 0x1134 || (filterMode != 0x1140 && filterMode != 0x1141)) {
        fail("CreateSampler: addressing mode 0x%x / filter mode 0x%x are not CL_ADDRESS_* / CL_FILTER_* values", addressingMode, filterMode);
        return nullptr;
    }
    auto sm = std::make_unique<rdx_sampler_s>();
    sm->addressing = addressingMode; sm->filter = filterMode;
    g.samplers.push_back(std::move(sm));
    return g.samplers.back().get();
}

extern "C" rdx_buffer rdx_buffer_wrap(void* device_ptr, size_t size)
{
    if (!g.initialized) { fail("rdx_init has not been called"); return nullptr; }
    if (!device_ptr) { fail("rdx_buffer_wrap: null device pointer"); return nullptr; }
    auto b = std::make_unique<rdx_buffer_s>();
    if (g_ndev > 1) { fail("rdx_buffer_wrap: caller-owned device memory cannot be replicated in multi-device mode"); return nullptr; }
    b->size = size; b->dptr = device_ptr; b->owned = false;
    g.buffers.push_back(std::move(b));
    return g.buffers.back().get();
}

extern "C" int rdx_buffer_write(rdx_buffer b, size_t offset, size_t size, const void* src)
{
    if (!b || !known_buffer(b)) return fail("WriteBuffer: invalid buffer handle");
    if (offset + size > b->size) return fail("WriteBuffer: range [%zu, %zu) exceeds buffer size %zu", offset, offset + size, b->size);
    if (size) HIP_OK(hipMemcpy(static_cast<uint8_t*>(b->dptr) + offset, src, size, hipMemcpyHostToDevice));
    for (int d = 1; d < g_ndev && size; ++d) {   // replicate
        HIP_OK(hipSetDevice(g_phys[d]));
        hipError_t e = hipMemcpy(static_cast<uint8_t*>(b->rep[d]) + offset, src, size, hipMemcpyHostToDevice);
        HIP_IGN(hipSetDevice(g_phys[0]));
        HIP_OK(e);
    }
    b->version++;
    if (b->owned && b->size <= 256) {
        if (offset == 0 && size == b->size) { b->mirror.assign(static_cast<const uint8_t*>(src), static_cast<const uint8_t*>(src) + size); b->mirrorValid = true; }
        else if (b->mirrorValid && size) std::memcpy(b->mirror.data() + offset, src, size);
    }
    return 0;
}

extern "C" int rdx_buffer_read(rdx_buffer b, size_t offset, size_t size, void* dst)
{
    if (!b || !known_buffer(b)) return fail("ReadBuffer: invalid buffer handle");
    if (offset + size > b->size) return fail("ReadBuffer: range [%zu, %zu) exceeds buffer size %zu", offset, offset + size, b->size);
    HIP_OK(hipStreamSynchronize(g.stream));
    if (size) HIP_OK(hipMemcpy(dst, static_cast<const uint8_t*>(b->dptr) + offset, size, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void* rdx_buffer_device_ptr(rdx_buffer b) { return (b && known_buffer(b)) ? b->dptr : nullptr; }
extern "C" size_t rdx_buffer_size(rdx_buffer b) { return (b && known_buffer(b)) ? b->size : 0; }

// ------------------------------------------------------------------------------------------------
// acceleration structures
// ------------------------------------------------------------------------------------------------
extern "C" rdx_blas rdx_blas_build(const float* v, uint32_t nv, const uint32_t* idx, uint32_t nt)
{
    std::string err;
    Blas* b = build_blas(v, nv, idx, nt, err);
    if (!b) { fail("%s", err.c_str()); return nullptr; }
    auto h = std::make_unique<rdx_blas_s>();
    h->blas.reset(b);
    g.blases.push_back(std::move(h));
    return g.blases.back().get();
}

// Several meshes at once: the builds are independent (one BVH per mesh), so they run on a pool of host threads inside the
// library -- the caller stays single-threaded (radiance.h has no such call; the reference builds its meshes one after the
// other, tools/sceneBuilder.cpp:229-258).  Results are identical to `count` rdx_blas_build calls in order.
extern "C" int rdx_blas_build_many(uint32_t count, const float* const* verts, const uint32_t* nverts,
                                   const uint32_t* const* indices, const uint32_t* ntris, rdx_blas* out)
{
    if (count == 0) return 0;
    if (!verts || !nverts || !indices || !ntris || !out) return fail("rdx_blas_build_many: null argument");
    std::vector<Blas*> built(count, nullptr);
    std::vector<std::string> errs(count);
    std::atomic<uint32_t> next{0};
    const uint32_t nthreads = std::max(1u, std::min(count, std::min(16u, std::thread::hardware_concurrency())));
    auto work = [&]() {
        for (;;) {
            const uint32_t i = next.fetch_add(1);
            if (i >= count) return;
            built[i] = build_blas(verts[i], nverts[i], indices[i], ntris[i], errs[i]);
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < nthreads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    for (uint32_t i = 0; i < count; ++i)
        if (!built[i]) {
            for (Blas* b : built) delete b;
            return fail("mesh %u: %s", i, errs[i].c_str());
        }
    for (uint32_t i = 0; i < count; ++i) {
        auto h = std::make_unique<rdx_blas_s>();
        h->blas.reset(built[i]);
        g.blases.push_back(std::move(h));
        out[i] = g.blases.back().get();
    }
    return 0;
}

extern "C" const void* rdx_blas_data(rdx_blas b, uint32_t* size_out)
{
    if (!b) return nullptr;
    if (size_out) *size_out = (uint32_t)b->blas->data.size();
    return b->blas->data.data();
}
extern "C" int rdx_blas_max_depth(rdx_blas b) { return b ? b->blas->maxDepth : -1; }

static rdx_buffer tlas_from_blob(std::vector<uint8_t>&& blob)
{
    rdx_buffer tb = rdx_buffer_create(blob.size());
    if (!tb) return nullptr;
    if (rdx_buffer_write(tb, 0, blob.size(), blob.data()) != 0) return nullptr;
    tb->shadow = std::move(blob);
    tb->shadowVersion = tb->version;
    return tb;
}

static bool tlas_blob(const rdx_instance* inst, uint32_t n, std::vector<uint8_t>& blob, int& depth)
{
    std::vector<InstanceDesc> d(n);
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(d[i].transform, inst[i].transform, 64);
        d[i].SBTOffset = inst[i].SBTOffset;
        d[i].customInstanceID = inst[i].customInstanceID;
        d[i].blas = inst[i].bottomAccelStruct ? inst[i].bottomAccelStruct->blas.get() : nullptr;
    }
    std::string err;
    if (!build_tlas(d.data(), n, blob, depth, err)) { fail("%s", err.c_str()); return false; }
    return true;
}

extern "C" void* rdx_tlas_build_blob(const rdx_instance* inst, uint32_t n, uint32_t* size_out, int* max_depth_out)
{
    std::vector<uint8_t> blob;
    int depth = 0;
    if (!tlas_blob(inst, n, blob, depth)) return nullptr;
    void* p = malloc(blob.size());
    if (!p) { fail("out of memory"); return nullptr; }
    std::memcpy(p, blob.data(), blob.size());
    if (size_out) *size_out = (uint32_t)blob.size();
    if (max_depth_out) *max_depth_out = depth;
    return p;
}
extern "C" void rdx_free(void* p) { free(p); }

extern "C" rdx_buffer rdx_tlas_build(const rdx_instance* inst, uint32_t n)
{
    if (!g.initialized) { fail("rdx_init has not been called"); return nullptr; }
    std::vector<uint8_t> blob;
    int depth = 0;
    if (!tlas_blob(inst, n, blob, depth)) return nullptr;
    return tlas_from_blob(std::move(blob));
}

// radiance.cpp:428-448: raw dump of the TLAS buffer, size from header word 3
// error text set from the library's other translation units (scene_obj.cpp)
namespace rdx { int fail_text(const char* text) { return fail_str(text ? text : ""); } }

// Side-car of a TLAS cache file: the cache itself stays the raw blob the reference writes
// (radiance.cpp:428-448: totalBufferSize bytes, no header of its own), so files written by either side load on the
// other; `<path>.meta` adds what the raw format cannot say -- a magic / version line, the byte count and an FNV-1a
// hash of the blob.  FileToTopAccelStruct verifies a side-car when one exists and refuses a blob that does not
// match it (a truncated or stale cache is otherwise only noticed as a wrong picture); a cache without side-car
// loads as before.
static uint64_t fnv1a64(const uint8_t* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
static int write_cache_meta(const char* path, const std::vector<uint8_t>& data)
{
    const std::string mp = std::string(path) + ".meta";
    FILE* fp = fopen(mp.c_str(), "w");
    if (!fp) return fail("TopAccelStructToFile: cannot open '%s' for writing", mp.c_str());
    const int ok = fprintf(fp, "RDXCACHE 1\nbytes %zu\nfnv1a64 %016llx\n", data.size(), (unsigned long long)fnv1a64(data.data(), data.size()));
    fclose(fp);
    return ok > 0 ? 0 : fail("TopAccelStructToFile: short write to '%s'", mp.c_str());
}
// 0 = no side-car or it matches, -1 = mismatch (error text set)
static int check_cache_meta(const char* path, const std::vector<uint8_t>& data)
{
    const std::string mp = std::string(path) + ".meta";
    FILE* fp = fopen(mp.c_str(), "r");
    if (!fp) return 0;
    char magic[16] = ""; int ver = 0; size_t bytes = 0; unsigned long long h = 0;
    const int n = fscanf(fp, "%15s %d bytes %zu fnv1a64 %llx", magic, &ver, &bytes, &h);
    fclose(fp);
    if (n != 4 || strcmp(magic, "RDXCACHE")) return fail("FileToTopAccelStruct: '%s' is not a cache side-car", mp.c_str());
    if (ver != 1) return fail("FileToTopAccelStruct: side-car '%s' has version %d, this library reads version 1", mp.c_str(), ver);
    if (bytes != data.size()) return fail("FileToTopAccelStruct: '%s' holds %zu bytes, its side-car says %zu (stale or truncated cache)", path, data.size(), bytes);
    if (h != fnv1a64(data.data(), data.size())) return fail("FileToTopAccelStruct: '%s' does not match the hash in its side-car (stale or corrupted cache)", path);
    return 0;
}

extern "C" int rdx_tlas_to_file(rdx_buffer tlas, const char* path)
{
    if (!tlas || !known_buffer(tlas)) return fail("TopAccelStructToFile: invalid handle");
    BlobTopHeader hdr;
    if (rdx_buffer_read(tlas, 0, sizeof hdr, &hdr)) return -1;
    if (hdr.totalBufferSize > tlas->size) return fail("TopAccelStructToFile: header size %u exceeds buffer size %zu", hdr.totalBufferSize, tlas->size);
    std::vector<uint8_t> data(hdr.totalBufferSize);
    if (rdx_buffer_read(tlas, 0, data.size(), data.data())) return -1;
    FILE* fp = fopen(path, "wb");
    if (!fp) return fail("TopAccelStructToFile: cannot open '%s' for writing", path);
    const size_t wr = fwrite(data.data(), 1, data.size(), fp);
    fclose(fp);
    if (wr != data.size()) return fail("TopAccelStructToFile: short write to '%s'", path);
    return write_cache_meta(path, data);
}

// radiance.cpp:450-479
extern "C" rdx_buffer rdx_tlas_from_file(const char* path)
{
    if (!g.initialized) { fail("rdx_init has not been called"); return nullptr; }
    FILE* fp = fopen(path, "rb");
    if (!fp) { fail("FileToTopAccelStruct: cannot open '%s'", path); return nullptr; }
    BlobTopHeader hdr;
    if (fread(&hdr, 1, sizeof hdr, fp) != sizeof hdr) { fclose(fp); fail("FileToTopAccelStruct: short read of header in '%s'", path); return nullptr; }
    if (hdr.type != TYPE_TOP_AS || hdr.totalBufferSize < sizeof hdr) { fclose(fp); fail("FileToTopAccelStruct: '%s' is not a TLAS cache file", path); return nullptr; }
    std::vector<uint8_t> data(hdr.totalBufferSize);
    rewind(fp);
    const size_t rd = fread(data.data(), 1, data.size(), fp);
    fclose(fp);
    if (rd != data.size()) { fail("FileToTopAccelStruct: short read of '%s' (%zu of %zu bytes)", path, rd, data.size()); return nullptr; }
    if (check_cache_meta(path, data)) return nullptr;
    return tlas_from_blob(std::move(data));
}

// ------------------------------------------------------------------------------------------------
// pipeline
// ------------------------------------------------------------------------------------------------
static bool has_identifier(const std::string& text, const char* name)
{
    const size_t n = strlen(name);
    size_t pos = 0;
    auto isid = [](char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; };
    while ((pos = text.find(name, pos)) != std::string::npos) {
        const bool l = pos == 0 || !isid(text[pos - 1]);
        const bool r = pos + n >= text.size() || !isid(text[pos + n]);
        if (l && r) return true;
        pos += n;
    }
    return false;
}

// text with comments removed (so that a commented-out parameter list or identifier does not count)
static std::string strip_comments(const std::string& t)
{
    std::string o;
    o.reserve(t.size());
    for (size_t i = 0; i < t.size();) {
        if (t.compare(i, 2, "//") == 0) { while (i < t.size() && t[i] != '\n') ++i; }
        else if (t.compare(i, 2, "/*") == 0) { const size_t e = t.find("*/", i + 2); const size_t j = e == std::string::npos ? t.size() : e + 2; o.push_back(' '); for (; i < j; ++i) if (t[i] == '\n') o.push_back('\n'); }
        else o.push_back(t[i++]);
    }
    return o;
}

// does `raygen` take parameters?  (-1: no raygen( found)
static int raygen_has_parameters(const std::string& t)
{
    for (size_t pos = t.find("raygen"); pos != std::string::npos; pos = t.find("raygen", pos + 1)) {
        size_t i = pos + 6;
        while (i < t.size() && std::isspace((unsigned char)t[i])) ++i;
        if (i >= t.size() || t[i] != '(') continue;
        const size_t e = t.find(')', i);
        if (e == std::string::npos) return -1;
        std::string inner = t.substr(i + 1, e - i - 1);
        inner.erase(std::remove_if(inner.begin(), inner.end(), [](unsigned char c) { return std::isspace(c); }), inner.end());
        return (inner.empty() || inner == "void") ? 0 : 1;
    }
    return -1;
}

static uint64_t fnv1a64_nows(const std::string& t)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char c : t) { if (std::isspace(c)) continue; h ^= c; h *= 0x100000001b3ull; }
    return h;
}

// A program text with the BODIES of its closest-hit / miss stage functions blanked out (the functions sbt.json names for rows
// without an any-hit shader, and the miss rows): two programs that agree on it differ only inside those functions -- the raygen
// loop, the payload / scene structs, the helper functions, the any-hit rows and the dispatch tables are the same.
static void blank_bodies(std::string& t, const std::vector<std::string>& names)
{
    for (const std::string& fn : names) {
        size_t pos = 0;
        while ((pos = t.find(fn, pos)) != std::string::npos) {
            const size_t end = pos + fn.size();
            const bool startOK = pos == 0 || !(std::isalnum((unsigned char)t[pos - 1]) || t[pos - 1] == '_');
            size_t q = end;
            while (q < t.size() && std::isspace((unsigned char)t[q])) ++q;
            if (!startOK || q >= t.size() || t[q] != '(') { pos = end; continue; }
            // parameter list, then a definition's '{' (a call or a prototype has something else there)
            int depth = 0;
            size_t r = q;
            for (; r < t.size(); ++r) { if (t[r] == '(') ++depth; else if (t[r] == ')') { if (--depth == 0) { ++r; break; } } }
            while (r < t.size() && std::isspace((unsigned char)t[r])) ++r;
            if (r >= t.size() || t[r] != '{') { pos = end; continue; }
            int braces = 0;
            size_t e = r;
            for (; e < t.size(); ++e) { if (t[e] == '{') ++braces; else if (t[e] == '}') { if (--braces == 0) { ++e; break; } } }
            std::string blank = "{";
            blank.append((size_t)std::count(t.begin() + r, t.begin() + e, '\n'), '\n');       // line numbers stay (compiler diagnostics)
            blank += "}";
            t.replace(r, e - r, blank);
            pos = r + blank.size();
        }
    }
}
static std::string blank_stage_bodies(const std::string& text)
{
    std::string t = strip_comments(text);
    std::vector<std::string> names;
    {
        struct Row { int row; const char* fn; };
        static const Row anyHit[] = {
#define X(row, fn) {row, #fn},
            RDX_SBT_ANY_HIT(X)
#undef X
            {-1, nullptr}};
        static const Row closest[] = {
#define X(row, fn) {row, #fn},
            RDX_SBT_CLOSEST_HIT(X)
#undef X
            {-1, nullptr}};
        static const Row miss[] = {
#define X(row, fn) {row, #fn},
            RDX_SBT_MISS(X)
#undef X
            {-1, nullptr}};
        for (const Row* c = closest; c->fn; ++c) {
            bool hasAny = false;
            for (const Row* a = anyHit; a->fn; ++a) if (a->row == c->row) hasAny = true;
            if (!hasAny) names.push_back(c->fn);
        }
        for (const Row* m = miss; m->fn; ++m) names.push_back(m->fn);
    }
    blank_bodies(t, names);
    return t;
}
extern "C" unsigned long long rdx_debug_stage_reduced_hash(const char* code, uint32_t size)
{
    return code ? fnv1a64_nows(blank_stage_bodies(std::string(code, size))) : (unsigned long long)RDX_STOCK_REDUCED_HASH;   // null: the stock program's
}

// Compile-only check of the run-time shader compiler (no device needed; the CPU suite uses it): 0 = the program compiles in
// the given mode (stages: with its raygen body blanked, as rdx_shader_module_create does), -1 = it does not, the log in the
// last-error string.  "Compiles" = every step up to loading the code object succeeded.
extern "C" int rdx_debug_jit_compiles(const char* code, uint32_t size, const char* arch, int stages)
{
    if (!code || !arch) return fail("rdx_debug_jit_compiles: null argument");
    std::string text(code, size), err;
    if (stages) { text = strip_comments(text); blank_bodies(text, {"raygen", "generateRay"}); }
    setenv("RDX_JIT_COMPILE_ONLY", "1", 1);
    UserProgram* p = compile_user_shader(text, g0.shaderInclude, arch, stages != 0, err);
    unsetenv("RDX_JIT_COMPILE_ONLY");
    (void)p;
    if (err == "compiled") return 0;
    return fail_str(err.empty() ? std::string("rdx_debug_jit_compiles: unexpected state") : err);
}

extern "C" int rdx_shader_include_path(const char* path)
{
    g0.shaderInclude = path ? path : "";
    return 0;
}

extern "C" rdx_shader rdx_shader_module_create(const char* code, uint32_t size, const char* name)
{
    if (!g.initialized) { fail("rdx_init has not been called"); return nullptr; }
    if (!code) { fail("CreateShaderModule: null shader text"); return nullptr; }
    const std::string text(code, size);
    // The reference JIT-compiles `code` and takes the kernel named "raygen" (radiance.cpp:152-179).  Three cases here:
    //  1. the text IS the reference's stock program (samples/shader.cl, recognised by a hash of its non-blank characters): its
    //     stage functions are the ones of samples/sbt.json that this library ships as hand-written HIP -> wavefront pipeline;
    //  2. `raygen` declared without parameters: a placeholder that asks for the stock pipeline (nothing could be bound to it);
    //  3. any other program: compiled at run time by ROCm's OpenCL C compiler and run as the megakernel it is (user_shader.cpp).
    const std::string bare = strip_comments(text);
    if (!has_identifier(bare, "raygen")) {
        fail("CreateShaderModule: shader text has no `raygen` kernel (clCreateKernel(\"raygen\") would fail)");
        return nullptr;
    }
    auto s = std::make_unique<rdx_shader_s>();
    s->name = name ? name : "";
    s->hasRaygen = true;
    constexpr uint64_t kStockShaderHash = 0xc0cc932e07087140ull;      // FNV-1a-64 of samples/shader.cl without white space (16 983 characters)
    const bool stock = fnv1a64_nows(text) == kStockShaderHash || raygen_has_parameters(bare) == 0;
    if (!stock) {
        if (g_ndev > 1) { fail("CreateShaderModule: user shader programs are not supported in multi-device mode"); return nullptr; }
        hipDeviceProp_t prop;
        HIP_OKP(hipGetDeviceProperties(&prop, g0.device));
        std::string arch = prop.gcnArchName;
        arch = arch.substr(0, arch.find(':'));
        std::string err;
        // 3a. the program differs from the stock one only INSIDE its closest-hit / miss stage functions (blank_stage_bodies):
        //     those functions are compiled into the shade stage of the wavefront pipeline (user_shader.cpp "stage mode");
        // 3b. anything else: the program's own raygen, as a megakernel.
        constexpr uint64_t kStockReducedHash = RDX_STOCK_REDUCED_HASH;     // of samples/shader.cl, by tools/stock_shader_hash.py
        //     "user_stages" 2: the caller asserts it for a program written from scratch (the raygen text is then not looked at).
        const bool stages = g0.userStages == 2 || (g0.userStages == 1 && fnv1a64_nows(blank_stage_bodies(text)) == kStockReducedHash);
        if (stages) {
            // the stage kernel replaces the program's raygen: its body and that of the stock skeleton's camera helper go (dead
            // code there, and their get_global_id(0) has no `sceneData` in scope for the stage-mode macro)
            std::string t = bare;
            blank_bodies(t, {"raygen", "generateRay"});
            std::string err2;
            s->program = compile_user_shader(t, g0.shaderInclude, arch, true, err2);
            if (!s->program && g0.userStages == 2) { fail_str(err2); return nullptr; }
        }
        if (!s->program) s->program = compile_user_shader(text, g0.shaderInclude, arch, false, err);
        if (!s->program) { fail_str(err); return nullptr; }
    }
    g.shaders.push_back(std::move(s));
    return g.shaders.back().get();
}

extern "C" int rdx_bind_pipeline(rdx_shader m)
{
    if (!m) return fail("BindPipeline: null shader module");
    g.pipeline = m;
    return 0;
}

extern "C" int rdx_bind_descriptor_set(void* const* handles, uint32_t n)
{
    if (!g.pipeline) return fail("BindDescriptorSet: no pipeline bound");
    if (n > 14) return fail("BindDescriptorSet: %u descriptors, the raygen stage takes 14", n);
    for (uint32_t i = 0; i < n; ++i) g.slots[i] = handles[i];
    g.nslots = std::max(g.nslots, n);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// sharding
// ------------------------------------------------------------------------------------------------
extern "C" int rdx_set_shard(uint32_t rank, uint32_t world, uint32_t tw, uint32_t th)
{
    if (world == 0 || rank >= world || tw == 0 || th == 0) return fail("rdx_set_shard: invalid rank/world/tile");
    g.rank = rank; g.world = world; g.tileW = tw; g.tileH = th;
    return 0;
}

extern "C" uint32_t rdx_shard_pixel_count(uint32_t w, uint32_t h, uint32_t rank, uint32_t world)
{
    std::vector<uint32_t> px;
    owned_pixel_list(w, h, g.tileW, g.tileH, rank, world, px);
    return (uint32_t)px.size();
}

static int pack_impl(rdx_buffer image, rdx_buffer packed, uint32_t w, uint32_t h, uint32_t elem, uint32_t rank,
                     uint32_t world, bool unpack)
{
    if (!image || !packed || !known_buffer(image) || !known_buffer(packed)) return fail("pack_tiles: invalid buffer");
    if (elem % 4 || elem == 0) return fail("pack_tiles: element size must be a multiple of 4");
    const uint32_t tilesX = (w + g.tileW - 1) / g.tileW, tilesY = (h + g.tileH - 1) / g.tileH, nT = tilesX * tilesY;
    const uint32_t owned = nT > rank ? (nT - rank + world - 1) / world : 0;
    if ((size_t)w * h * elem > image->size) return fail("pack_tiles: image buffer too small");
    if ((size_t)owned * g.tileW * g.tileH * elem > packed->size) return fail("pack_tiles: packed buffer too small");
    launch_pack_tiles(g.stream, static_cast<uint8_t*>(dp(image)), static_cast<uint8_t*>(dp(packed)), w, h, elem,
                      g.tileW, g.tileH, rank, world, unpack);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    if (unpack) image->version++; else packed->version++;
    return 0;
}
extern "C" int rdx_pack_tiles(rdx_buffer image, rdx_buffer packed, uint32_t w, uint32_t h, uint32_t elem, uint32_t rank, uint32_t world)
{ return pack_impl(image, packed, w, h, elem, rank, world, false); }
extern "C" int rdx_unpack_tiles(rdx_buffer packed, rdx_buffer image, uint32_t w, uint32_t h, uint32_t elem, uint32_t rank, uint32_t world)
{ return pack_impl(image, packed, w, h, elem, rank, world, true); }
// the gathering rank's side of a frame: the packed buffers of ranks first_rank .. first_rank + n - 1 go into the image with
// ONE synchronisation at the end (a call per rank costs a launch + a stream synchronise each: 7 of them per frame at 8 GPUs)
extern "C" int rdx_unpack_tiles_multi(const rdx_buffer* packed, uint32_t first_rank, uint32_t n, rdx_buffer image, uint32_t w, uint32_t h,
                                      uint32_t elem, uint32_t world)
{
    if (!packed || !image || !known_buffer(image)) return fail("unpack_tiles_multi: invalid buffer");
    if (elem % 4 || elem == 0) return fail("unpack_tiles_multi: element size must be a multiple of 4");
    if ((size_t)w * h * elem > image->size) return fail("unpack_tiles_multi: image buffer too small");
    const uint32_t tilesX = (w + g.tileW - 1) / g.tileW, tilesY = (h + g.tileH - 1) / g.tileH, nT = tilesX * tilesY;
    for (uint32_t k = 0; k < n; ++k) {          // validate everything before anything is launched
        const uint32_t rank = first_rank + k;
        if (rank >= world || !packed[k] || !known_buffer(packed[k])) return fail("unpack_tiles_multi: invalid packed buffer %u", k);
        const uint32_t owned = nT > rank ? (nT - rank + world - 1) / world : 0;
        if ((size_t)owned * g.tileW * g.tileH * elem > packed[k]->size) return fail("unpack_tiles_multi: packed buffer %u too small", k);
    }
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t rank = first_rank + k;
        launch_pack_tiles(g.stream, static_cast<uint8_t*>(image->dptr), static_cast<uint8_t*>(packed[k]->dptr), w, h, elem,
                          g.tileW, g.tileH, rank, world, true);
    }
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    image->version++;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// TraceRays
// ------------------------------------------------------------------------------------------------
extern "C" int rdx_set_profiling(int on) { g.profiling = on != 0; return 0; }
extern "C" int rdx_set_option(const char* name, int64_t value)
{
    if (!name) return fail("rdx_set_option: null name");
    if (!strcmp(name, "chunk_paths")) { if (value < 1) return fail("chunk_paths must be >= 1"); g.chunkPaths = value; return 0; }
    if (!strcmp(name, "count_visits")) { g.countVisits = value != 0; return 0; }
    if (!strcmp(name, "groups")) { if (value < 0 || value > Context::MAX_GROUPS) return fail("groups must be 0 (auto) or 1..4"); g.groupsOpt = (int)value; return 0; }
    if (!strcmp(name, "overlap")) { if (value < 0 || value > 1) return fail("overlap must be 0 or 1"); g.overlap = (int)value; return 0; }
    if (!strcmp(name, "pipeline")) { if (value < 0 || value > 1) return fail("pipeline must be 0 (staged) or 1 (paths)"); g.pathMode = (int)value; return 0; }
    if (!strcmp(name, "fuse")) { if (value < -1 || value > 1) return fail("fuse must be -1 (auto), 0 or 1"); g.fuse = (int)value; return 0; }
    if (!strcmp(name, "user_shader_local_size")) { if (value < 1 || value > 1024) return fail("user_shader_local_size must be 1..1024"); g.userLocalSize = (int)value; return 0; }
    if (!strcmp(name, "sort")) { g.sortRays = value < 0 ? -1 : (value != 0); return 0; }
    if (!strcmp(name, "textures")) { g.textures = value != 0; return 0; }
    if (!strcmp(name, "cull")) { g.cull = value < 0 ? -1 : (value != 0); return 0; }
    if (!strcmp(name, "top_flat")) { g.topFlat = value != 0; return 0; }
    if (!strcmp(name, "group_instances")) { g.groupInstances = value != 0; return 0; }
    if (!strcmp(name, "unified_tree")) { g.unifiedTree = value != 0; return 0; }
    if (!strcmp(name, "gpu_build") || !strcmp(name, "gpu_build_min")) {
        if (!strcmp(name, "gpu_build")) g0.gpuBuild = value != 0; else g0.gpuBuildMin = value > 0 ? value : 32768;
        set_gpu_binner((g0.initialized && g0.gpuBuild) ? &g_hipBinner : nullptr, (size_t)g0.gpuBuildMin);
        return 0;
    }
    if (!strcmp(name, "sort_min_paths")) { g.sortMinPaths = value > 0 ? value : (3ll << 19); return 0; }
    if (!strcmp(name, "small_chunk_paths")) { g.smallChunkPaths = value > 0 ? value : (9ll << 19); return 0; }
    if (!strcmp(name, "quad")) { g.quad = value < 0 ? -1 : (value != 0); return 0; }
    if (!strcmp(name, "user_stages")) { g.userStages = value > 2 ? 1 : (int)value; return 0; }
    if (!strcmp(name, "inline_leaf_roots")) { g.inlineLeafRoots = value != 0; return 0; }
    if (!strcmp(name, "kernel")) { if (value < 0 || value > 3) return fail("kernel must be 0, 1, 2 or 3"); g.kernel = (int)value; return 0; }
    return fail("rdx_set_option: unknown option '%s'", name);
}
extern "C" int rdx_get_bounce_counts(uint64_t* out, uint32_t n)
{
    if (!out) return fail("null");
    for (uint32_t d = 0; d < n && d < 65; ++d) out[d] = g.bounceCounts[d];
    return 0;
}

extern "C" int rdx_get_visit_profile(uint64_t* out, uint32_t max_bounces)
{
    if (!out) return fail("null");
    const uint32_t n = std::min(max_bounces, g.visitDepth);
    for (uint32_t d = 0; d < n; ++d)
        for (int k = 0; k < 8; ++k) out[8 * d + k] = g.hVisit[8 * d + k];
    return (int)n;
}

extern "C" int rdx_get_trace_stats(rdx_trace_stats* out) { if (!out) return fail("null"); *out = g.stats; return 0; }

// one frame on the calling thread's device (context `g`, logical device tl_dev): every pixel of its shard
static int trace_rays_device(uint32_t width, uint32_t height)
{
    if (!g.initialized) return fail("rdx_init has not been called");
    if (!g.pipeline) return fail("TraceRays: no pipeline bound");
    if (g.nslots < 14) return fail("TraceRays: %u descriptors bound, the raygen stage takes 14", g.nslots);
    for (int i : {0, 1, 2, 3, 13})
        if (!g.slots[i] || !known_buffer(g.slots[i])) return fail("descriptor slot %d is not a buffer", i);
    auto* bRT = static_cast<rdx_buffer_s*>(g.slots[0]);
    auto* bScratch = static_cast<rdx_buffer_s*>(g.slots[1]);
    auto* bImage = static_cast<rdx_buffer_s*>(g.slots[2]);
    auto* bCam = static_cast<rdx_buffer_s*>(g.slots[3]);
    auto* bTlas = static_cast<rdx_buffer_s*>(g.slots[13]);
    const uint64_t nPix = (uint64_t)width * height;
    if (nPix == 0) return 0;
    if (g.pipeline->program && !g.pipeline->program->stages) {
        // a user's own raygen program: the megakernel, one work-item per pixel, bound by position like clSetKernelArg
        // (radiance.cpp:231-259); slots 11 / 12 (texture array, sampler) are passed as null descriptors
        if (nPix > 0xffffffffull) return fail("TraceRays: too many pixels");
        void* ptrs[12];
        const int slotOf[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13};
        for (int i = 0; i < 12; ++i) {
            void* h = g.slots[slotOf[i]];
            if (!h || !known_buffer(h)) return fail("descriptor slot %d is not a buffer", slotOf[i]);
            ptrs[i] = dp(static_cast<rdx_buffer_s*>(h));
        }
        std::memset(&g.stats, 0, sizeof g.stats);
        g.stats.pixels = nPix;
        HIP_OK(hipEventRecord(g.evA, g.stream));
        std::string err;
        if (launch_user_shader(g.pipeline->program, g.stream, ptrs, (uint32_t)nPix, (uint32_t)g.userLocalSize, err)) return fail_str(err);
        HIP_OK(hipEventRecord(g.evB, g.stream));
        HIP_OK(hipStreamSynchronize(g.stream));
        HIP_OK(hipEventElapsedTime(&g.stats.ms_total, g.evA, g.evB));
        // the program wrote imageScratch / image itself: host mirrors of nothing are affected (only RTProp / camera are mirrored,
        // and a raygen that wrote them would be outside the reference's host contract, sample1.cpp:480-490)
        return 0;
    }
    if (nPix > 0x7fffffffull) return fail("TraceRays: %llu pixels exceed the 31-bit pixel index", (unsigned long long)nPix);
    if (bRT->size < sizeof(RayTraceProperties) || bCam->size < sizeof(PhysicalCamera)) return fail("TraceRays: RTProp / camera buffer too small");
    if (bScratch->size < nPix * 16) return fail("TraceRays: imageScratch holds %zu bytes, %llu needed", bScratch->size, (unsigned long long)nPix * 16);
    if (bImage->size < nPix * 4) return fail("TraceRays: image holds %zu bytes, %llu needed", bImage->size, (unsigned long long)nPix * 4);

    SceneArgs sc;
    if (scene_args(sc)) return -1;
    if (derive_accel(bTlas)) return -1;

    // per-frame constants live in device buffers the caller may have rewritten (sample1.cpp:480-490)
    RayTraceProperties rt; PhysicalCamera cam;
    if (bRT->mirrorValid && bRT->mirror.size() >= sizeof rt) std::memcpy(&rt, bRT->mirror.data(), sizeof rt);
    else HIP_OK(hipMemcpy(&rt, dp(bRT), sizeof rt, hipMemcpyDeviceToHost));
    if (bCam->mirrorValid && bCam->mirror.size() >= sizeof cam) std::memcpy(&cam, bCam->mirror.data(), sizeof cam);
    else HIP_OK(hipMemcpy(&cam, dp(bCam), sizeof cam, hipMemcpyDeviceToHost));
    CameraArgs C;
    if (camera_args(cam, C)) return -1;
    if ((uint32_t)cam.widthPixel == 0) return fail("TraceRays: camera widthPixel is 0");

    if (ensure_owned(width, height)) return -1;
    const uint32_t P = g.ownedCount;
    const uint32_t* owned = g.world > 1 ? g.ownedPixels : nullptr;

    // depth as the raygen loop sees it: `debug` breaks after the first bounce (shader.cl:256-259)
    uint32_t maxDepth = rt.depth;
    if (rt.debug && maxDepth > 1) maxDepth = 1;
    if (maxDepth > 62) return fail("TraceRays: depth %u exceeds the supported maximum of 62", maxDepth);

    std::memset(&g.stats, 0, sizeof g.stats);
    std::memset(g.bounceCounts, 0, sizeof g.bounceCounts);
    g.stats.pixels = P;
    unsigned long long* visit = g.countVisits ? g.dVisit : nullptr;
    if (visit) HIP_OK(hipMemsetAsync(g.dVisit, 0, 64 * 8 * sizeof(unsigned long long), g.stream));

    HIP_OK(hipEventRecord(g.evA, g.stream));
    const uint32_t batch = rt.batchSize;
    uint32_t samplesPerChunk = batch;
    if (P && (uint64_t)batch * P > (uint64_t)g.chunkPaths) samplesPerChunk = (uint32_t)std::max<int64_t>(1, g.chunkPaths / P);
    if (P && batch) { if (ensure_samples((size_t)samplesPerChunk * P)) return -1; }
    // quad records (two tree levels per fetch, kernels at 4 waves per SIMD) for chunks whose launches do not fill the chip --
    // shards of a multi-GPU frame, low resolutions -- where a launch lasts as long as its longest chain of dependent fetches
    const AccelView av = view_of(bTlas, (uint64_t)samplesPerChunk * P <= (uint64_t)g.sortMinPaths);

    const float tmin = 0.001f, tmax = 1000.0f;      // shader.cl:235-236, 500
    for (uint32_t s0 = 0; s0 < batch && P; s0 += samplesPerChunk) {
        const uint32_t sc_n = std::min(samplesPerChunk, batch - s0);
        const uint32_t sampleBase = rt.totalSamples + s0;
        const uint64_t chunkPaths = (uint64_t)sc_n * P;
        // Small chunks (multi-GPU shards, low resolutions): a persistent traversal launch costs ~0.19 ms of ramp +
        // drain whatever its size (profiles/r01k_timeline_eighth_before_groups.txt: 9 launches = 1.7 of the 3.7 ms of a 1/8 frame).
        // The chunk's samples are split into groups with their own streams and counts, each launching 1/nGroups of
        // the resident grid, so that one group's launches run inside the ramp and drain of the other's
        // (kernels.hip: set_grid_share).  Two groups: -5 % at 1/8, -3 % at 1/2 of a 1080p x 4 spp frame; four: slower.
        const bool small = chunkPaths <= (uint64_t)g.smallChunkPaths && sc_n >= 2 && av.kernel >= 2;
        const uint32_t wantGroups = visit ? 1u : g.groupsOpt ? (uint32_t)g.groupsOpt : small ? 2u : 1u;
        int nGroups = (int)std::min<uint32_t>(wantGroups, sc_n);
        set_grid_share((uint32_t)nGroups);
        g.stats.groups = (uint32_t)nGroups;
        // Small chunks (multi-GPU shards, low resolutions): a traversal launch costs ~0.2 ms of ramp + tail
        // whatever its size (tools/trav_scale.py), so shadow(d) and extend(d+1) -- same ray count, disjoint
        // streams -- go into ONE cooperative launch: 9 traversal launches per depth-8 frame instead of 16.
        const bool fuse = g.fuse != 0 && !visit && av.kernel >= 2;
        (void)small;
        const bool overlap = g.overlap == 1 && !fuse && !visit;
        // per-bounce ray sort (north star; kernels.h): only the cooperative engines hand rays out by index
        const bool sortOn = !visit && av.kernel >= 2 &&
                            (g.sortRays > 0 || (g.sortRays < 0 && (acc(bTlas)->nWide >= RDX_SORT_AUTO_MIN_WIDE ||
                                                                    (acc(bTlas)->nWide >= RDX_SORT_AUTO_MIN_WIDE_FULL && chunkPaths > (uint64_t)g.sortMinPaths))));
        SortBox sortBox;
        for (int k = 0; k < 3; ++k) {
            const float lo = acc(bTlas)->sceneLo[k], ext = acc(bTlas)->sceneHi[k] - lo;
            sortBox.lo[k] = lo; sortBox.inv[k] = ext > 0.0f ? 16.0f / ext : 0.0f;
        }
        HIP_OK(hipEventRecord(g.evChunk, g.stream));          // everything before this chunk (previous accumulate) is done first

        if (g.pipeline->program && g.pipeline->program->stages) {
            // ---- a user's stage functions on the wavefront pipeline (user_shader.cpp "stage mode") --------------------------------
            // generate -> extend(0) -> [ stage pass 0 (records the shader's shadow query) -> shadow walk -> stage pass 1 (the shader
            // again, the query answered; raygen bookkeeping; compaction) -> extend(d + 1) ] ... -> accumulate.  One group, no fusing:
            // extend(d + 1) needs pass 1's rays.
            if (av.kernel != 3) return fail("TraceRays: user stage functions need the pool engine (kernel 3)");
            Context::Group& G = g.groups[0];
            const uint32_t n0 = sc_n * P;
            if (ensure_group(G, n0)) return -1;
            if (G.stageCap < n0) {
                float4** arr[] = {&G.ps.shD, &G.ps.payC, &G.ps.payF, &G.ps.nPayC, &G.ps.nPayF};
                for (auto a : arr) { if (*a) HIP_IGN(hipFree(*a)); *a = nullptr; HIP_OK(hipMalloc(reinterpret_cast<void**>(a), (size_t)n0 * sizeof(float4))); }
                if (G.ps.shHit) HIP_IGN(hipFree(G.ps.shHit));
                G.ps.shHit = nullptr;
                HIP_OK(hipMalloc(reinterpret_cast<void**>(&G.ps.shHit), (size_t)n0 * sizeof(uint32_t)));
                G.stageCap = n0;
            }
            G.ps.sampleColor = g.sampleColor;
            PathStreams ps = G.ps;
            std::memset(G.hCounts, 0, 256 * sizeof(uint32_t));
            G.hCounts[0] = n0;
            HIP_OK(hipMemcpyAsync(G.dCounts, G.hCounts, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
            launch_generate(g.stream, C, ps, owned, P, s0, sc_n, rt.totalSamples);
            if (maxDepth == 0) launch_finalize_all(g.stream, ps, n0, P, sampleBase);
            else launch_extend(g.stream, av, ps, G.dCounts, n0, tmin, tmax, nullptr, G.dCounts + 64);
            void* slotPtr[14];
            for (int i = 0; i < 14; ++i) slotPtr[i] = (g.slots[i] && known_buffer(g.slots[i])) ? dp(static_cast<rdx_buffer_s*>(g.slots[i])) : nullptr;
            for (uint32_t d = 0; d < maxDepth; ++d) {
                for (uint32_t pass = 0; pass < 2; ++pass) {
                    const uint32_t scalars[6] = {pass, d, maxDepth, P, sampleBase, rt.debug};
                    void* ptrs[31] = {G.dCounts + d, G.dCounts + d + 1, G.dCounts + 200,
                                      slotPtr[3], slotPtr[4], slotPtr[5], slotPtr[6], slotPtr[7], slotPtr[8], slotPtr[9], slotPtr[10], slotPtr[13],
                                      const_cast<DInst*>(av.insts),
                                      ps.rayO, ps.rayD, ps.thr, ps.col, ps.hitA, ps.hitInst, ps.payC, ps.payF,
                                      ps.shO, ps.shD, ps.shHit,
                                      ps.nRayO, ps.nRayD, ps.nThr, ps.nCol, ps.nPayC, ps.nPayF, ps.sampleColor};
                    std::string err;
                    if (launch_user_stage(g.pipeline->program, g.stream, scalars, ptrs, n0, err)) return fail_str(err);
                    if (pass == 0) launch_shadow_user(g.stream, av, ps, G.dCounts + d, n0, tmin, tmax, G.dCounts + 128 + d);
                }
                std::swap(ps.rayO, ps.nRayO); std::swap(ps.rayD, ps.nRayD); std::swap(ps.thr, ps.nThr); std::swap(ps.col, ps.nCol);
                std::swap(ps.payC, ps.nPayC); std::swap(ps.payF, ps.nPayF);
                if (d + 1 < maxDepth) launch_extend(g.stream, av, ps, G.dCounts + d + 1, n0, tmin, tmax, nullptr, G.dCounts + 64 + d + 1);
            }
            launch_accumulate(g.stream, ps, owned, P, s0, sc_n, rt.totalSamples, s0 + sc_n >= batch, rt.debug,
                              static_cast<float*>(dp(bScratch)), static_cast<uint8_t*>(dp(bImage)));
            HIP_OK(hipMemcpyAsync(G.hCounts, G.dCounts, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
            HIP_OK(hipStreamSynchronize(g.stream));
            if (G.hCounts[200] != 0)
                return fail("TraceRays: the program's closest-hit shader cannot run on the wavefront pipeline (%s); rdx_set_option(\"user_stages\", 0) runs it as a megakernel",
                            (G.hCounts[200] & 4u) ? "it calls traceRay more than once" : "its nested traceRay is not the stock shadow query: sbtRecordOffset 2, Tmin 0.001, Tmax 1000");
            const uint32_t* hc = G.hCounts;
            if (maxDepth) g.stats.rays_primary += hc[0];
            for (uint32_t d = 1; d < maxDepth; ++d) g.stats.rays_bounce += hc[d];
            for (uint32_t d = 0; d <= maxDepth && maxDepth; ++d) g.bounceCounts[d] += hc[d];
            g.stats.launches_extend += maxDepth; g.stats.launches_shadow += maxDepth;
            continue;
        }
        if (g.pathMode == 1 && !visit && av.kernel == 3 && maxDepth > 0) {
            // ---- whole paths in one persistent launch (k_path_pool) + accumulate ----
            Context::Group& G = g.groups[0];
            const uint32_t n0 = sc_n * P;
            if (ensure_group(G, n0)) return -1;
            G.ps.sampleColor = g.sampleColor;
            std::memset(G.hCounts, 0, 256 * sizeof(uint32_t));
            HIP_OK(hipMemcpyAsync(G.dCounts, G.hCounts, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
            unsigned long long* tally = reinterpret_cast<unsigned long long*>(G.dCounts + 192);     // 8-byte aligned words 192..195
            g_timer.begin(&g.stats.ms_path);
            launch_path(g.stream, av, sc, C, G.ps, owned, P, s0, sc_n, rt.totalSamples, maxDepth, sampleBase, G.dCounts + 64, tally, tmin, tmax);
            g_timer.end();
            g_timer.begin(&g.stats.ms_accumulate);
            launch_accumulate(g.stream, G.ps, owned, P, s0, sc_n, rt.totalSamples, s0 + sc_n >= batch, rt.debug,
                              static_cast<float*>(dp(bScratch)), static_cast<uint8_t*>(dp(bImage)));
            g_timer.end();
            HIP_OK(hipMemcpyAsync(G.hCounts, G.dCounts, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, g.stream));
            HIP_OK(hipStreamSynchronize(g.stream));
            const unsigned long long* ht = reinterpret_cast<const unsigned long long*>(G.hCounts + 192);
            g.stats.rays_primary += n0;
            g.stats.rays_bounce += ht[0] - n0;
            g.stats.rays_shadow += ht[1];
            g.stats.closest_hits += ht[1];
            g.stats.launches_extend++;
            continue;
        }

        uint32_t gBegin[Context::MAX_GROUPS + 1];
        for (int k = 0; k <= nGroups; ++k) gBegin[k] = (uint32_t)((uint64_t)sc_n * k / nGroups);
        PathStreams gps[Context::MAX_GROUPS];
        for (int k = 0; k < nGroups; ++k) {
            Context::Group& G = g.groups[k];
            const uint32_t ns = gBegin[k + 1] - gBegin[k], n0 = ns * P;
            if (ensure_group(G, n0)) return -1;
            G.ps.sampleColor = g.sampleColor;
            gps[k] = G.ps;
            if (k > 0) HIP_OK(hipStreamWaitEvent(G.s0, g.evChunk, 0));
            std::memset(G.hCounts, 0, 256 * sizeof(uint32_t));
            G.hCounts[0] = n0;
            HIP_OK(hipMemcpyAsync(G.dCounts, G.hCounts, 256 * sizeof(uint32_t), hipMemcpyHostToDevice, G.s0));
            g_timer.begin(&g.stats.ms_generate, G.s0);
            launch_generate(G.s0, C, gps[k], owned, P, s0 + gBegin[k], ns, rt.totalSamples);
            g_timer.end(G.s0);
            if (maxDepth == 0) launch_finalize_all(G.s0, gps[k], n0, P, sampleBase);
            if (maxDepth > 0) {
                g_timer.begin(&g.stats.ms_extend, G.s0);
                launch_extend(G.s0, av, gps[k], G.dCounts, n0, tmin, tmax, visit, G.dCounts + 64);     // visit row 0
                g_timer.end(G.s0);
                g.stats.launches_extend++;
            }
        }
        // Per bounce and group:  shade(d) -> { shadow(d), extend(d+1) } -> shade(d+1) ...
        // shadow(d) only fills nCol (or the final sample colour) and reads streams nobody writes meanwhile;
        // extend(d+1) reads the next-bounce rays shade(d) wrote.  The two are traced by one fused launch,
        // by two launches back to back, or (overlap) on two streams.
        for (uint32_t d = 0; d < maxDepth; ++d) {
            for (int k = 0; k < nGroups; ++k) {
                Context::Group& G = g.groups[k];
                PathStreams& ps = gps[k];
                const uint32_t n0 = (gBegin[k + 1] - gBegin[k]) * P;
                const bool last = d + 1 == maxDepth;
                if (overlap && d > 0) HIP_OK(hipStreamWaitEvent(G.s0, G.evShadow[d - 1], 0));   // shade(d) reads col written by shadow(d-1)
                g_timer.begin(&g.stats.ms_shade, G.s0);
                if (sortOn && G.permCap < n0) {
                    if (G.permE) HIP_IGN(hipFree(G.permE));
                    if (G.sortKey) HIP_IGN(hipFree(G.sortKey));
                    G.permE = nullptr; G.sortKey = nullptr; G.permCap = 0;
                    HIP_OK(hipMalloc(reinterpret_cast<void**>(&G.permE), (size_t)n0 * 4));
                    HIP_OK(hipMalloc(reinterpret_cast<void**>(&G.sortKey), (size_t)n0 * 2));
                    G.permCap = n0;
                }
                ps.sortKey = sortOn ? G.sortKey : nullptr;        // (the shade stage writes the survivors' sort keys)
                launch_shade(G.s0, av, sc, ps, G.dCounts + d, G.dCounts + d + 1, n0, d, maxDepth, P, sampleBase, sortOn ? &sortBox : nullptr);
                g_timer.end(G.s0);
                ps.permS = nullptr; ps.permE = nullptr;
                if (sortOn) {
                    // per-bounce ray sort: the traversal launch below hands its rays out in (octant, Morton cell) order
                    if (!G.sortBins) HIP_OK(hipMalloc(reinterpret_cast<void**>(&G.sortBins), (size_t)ray_sort_tiles_words() * 4));
                    g_timer.begin(&g.stats.ms_sort, G.s0);
                    launch_ray_sort_tiles(G.s0, ps, G.dCounts + d + 1, n0, sortBox, G.sortBins, G.permE, acc(bTlas)->nWide >= RDX_SORT_AUTO_MIN_WIDE);
                    g_timer.end(G.s0);
                    ps.permS = G.permE; ps.permE = G.permE;
                }
                const PathStreams psShadow = ps;
                // the compacted survivors become the live paths of the next bounce
                std::swap(ps.rayO, ps.nRayO); std::swap(ps.rayD, ps.nRayD); std::swap(ps.thr, ps.nThr); std::swap(ps.col, ps.nCol);
                if (fuse && !last) {
                    g_timer.begin(&g.stats.ms_fused, G.s0);
                    launch_fused(G.s0, av, sc, psShadow, ps, G.dCounts + d + 1, n0, P, sampleBase, tmin, tmax, G.dCounts + 128 + d);
                    g_timer.end(G.s0);
                    g.stats.launches_shadow++; g.stats.launches_extend++;
                    continue;
                }
                hipStream_t ss = G.s0;
                if (overlap) {
                    HIP_OK(hipEventRecord(G.evShade[d], G.s0));
                    HIP_OK(hipStreamWaitEvent(G.s1, G.evShade[d], 0));
                    ss = G.s1;
                }
                g_timer.begin(&g.stats.ms_shadow, ss);
                launch_shadow(ss, av, sc, psShadow, G.dCounts + d + 1, n0, last, P, sampleBase, tmin, tmax, visit ? visit + 8 * d : nullptr,
                              G.dCounts + 128 + d);
                g_timer.end(ss);
                if (overlap) HIP_OK(hipEventRecord(G.evShadow[d], G.s1));
                g.stats.launches_shadow++;
                if (!last) {
                    g_timer.begin(&g.stats.ms_extend, G.s0);
                    launch_extend(G.s0, av, ps, G.dCounts + d + 1, n0, tmin, tmax, visit ? visit + 8 * (d + 1) : nullptr, G.dCounts + 64 + d + 1);
                    g_timer.end(G.s0);
                    g.stats.launches_extend++;
                }
            }
        }
        for (int k = 0; k < nGroups; ++k) {
            Context::Group& G = g.groups[k];
            if (overlap && maxDepth > 0) HIP_OK(hipStreamWaitEvent(G.s0, G.evShadow[maxDepth - 1], 0));
            HIP_OK(hipMemcpyAsync(G.hCounts, G.dCounts, 256 * sizeof(uint32_t), hipMemcpyDeviceToHost, G.s0));
            if (k > 0) { HIP_OK(hipEventRecord(G.evDone, G.s0)); HIP_OK(hipStreamWaitEvent(g.stream, G.evDone, 0)); }
        }
        g_timer.begin(&g.stats.ms_accumulate);
        launch_accumulate(g.stream, gps[0], owned, P, s0, sc_n, rt.totalSamples, s0 + sc_n >= batch, rt.debug,
                          static_cast<float*>(dp(bScratch)), static_cast<uint8_t*>(dp(bImage)));
        g_timer.end();
        HIP_OK(hipStreamSynchronize(g.stream));
        for (int k = 0; k < nGroups; ++k) {
            const uint32_t* hc = g.groups[k].hCounts;
            for (uint32_t d = 0; d <= maxDepth && maxDepth; ++d) g.bounceCounts[d] += hc[d];
            if (maxDepth) g.stats.rays_primary += hc[0];
            for (uint32_t d = 1; d < maxDepth; ++d) g.stats.rays_bounce += hc[d];
            for (uint32_t d = 0; d < maxDepth; ++d) { g.stats.rays_shadow += hc[d + 1]; g.stats.closest_hits += hc[d + 1]; }
        }
    }
    if (batch == 0 && P) {
        // no samples: only the tonemap of the existing accumulator runs (shader.cl:283-304)
        PathStreams none{};
        launch_accumulate(g.stream, none, owned, P, 0, 0, rt.totalSamples, true, rt.debug,
                          static_cast<float*>(dp(bScratch)), static_cast<uint8_t*>(dp(bImage)));
    }
    HIP_OK(hipEventRecord(g.evB, g.stream));
    if (visit) HIP_OK(hipMemcpyAsync(g.hVisit, g.dVisit, 64 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, g.stream));
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));          // clFinish (radiance.cpp:261)
    if (take_status()) return fail("TraceRays: a traversal wave exceeded its iteration bound and gave up (internal error in the step selection); the frame is incomplete");
    HIP_OK(hipEventElapsedTime(&g.stats.ms_total, g.evA, g.evB));
    g_timer.resolve();
    g.visitDepth = 0;
    if (visit) {
        g.visitDepth = maxDepth;
        for (uint32_t d = 0; d < maxDepth; ++d)
            for (int c = 0; c < 2; ++c) {
                const unsigned long long* v = g.hVisit + 8 * d + 4 * c;
                g.stats.visit_top_nodes[c] += v[0]; g.stats.visit_instances[c] += v[1];
                g.stats.visit_bot_nodes[c] += v[2]; g.stats.visit_triangles[c] += v[3];
            }
    }
    bScratch->version++; bImage->version++;
    return 0;
}

static int ensure_stage(int slot, size_t bytes)
{
    if (g.gatherCap[slot] >= bytes) return 0;
    if (g.gatherStage[slot]) HIP_IGN(hipFree(g.gatherStage[slot]));
    g.gatherStage[slot] = nullptr; g.gatherCap[slot] = 0;
    HIP_OK(hipMalloc(&g.gatherStage[slot], std::max<size_t>(bytes, 16)));
    g.gatherCap[slot] = bytes;
    return 0;
}

// RD::TraceRays (radiance.cpp:242-267).  One device: the frame.  n devices (rdx_init_devices): the frame sharded by interleaved
// 64x64 tiles -- one internal host thread per device drives that device's streams, the caller's single blocking call returns
// when every shard is done and its RGBA8 + imageScratch tiles have been copied into device 0's buffers (peer copies over xGMI;
// no collective library is needed inside one process), where ReadBuffer reads them.
extern "C" int rdx_trace_rays(uint32_t, uint32_t, uint32_t, uint32_t width, uint32_t height)
{
    if (g_ndev <= 1) return trace_rays_device(width, height);
    if (!g0.initialized) return fail("rdx_init has not been called");
    if (g0.world > 1) return fail("TraceRays: rdx_set_shard and rdx_init_devices cannot be combined");
    if (g0.nslots < 14 || !g0.slots[13] || !known_buffer(g0.slots[13]) || !g0.slots[1] || !known_buffer(g0.slots[1]) ||
        !g0.slots[2] || !known_buffer(g0.slots[2]))
        return trace_rays_device(width, height);       // reports the binding error
    auto* bScratch = static_cast<rdx_buffer_s*>(g0.slots[1]);
    auto* bImage = static_cast<rdx_buffer_s*>(g0.slots[2]);
    auto* bTlas = static_cast<rdx_buffer_s*>(g0.slots[13]);
    const int n = g_ndev;
    // registry-side state each device context needs, and the derived traversal layout on every device (sequentially: the host
    // copy of the blob is shared)
    for (int d = 1; d < n; ++d) {
        Context& c = *g_dev[d];
        std::memcpy(c.slots, g0.slots, sizeof c.slots);
        c.nslots = g0.nslots; c.pipeline = g0.pipeline;
        c.groupsOpt = g0.groupsOpt; c.fuse = g0.fuse; c.pathMode = g0.pathMode; c.chunkPaths = g0.chunkPaths;
        c.countVisits = g0.countVisits; c.profiling = g0.profiling; c.inlineLeafRoots = g0.inlineLeafRoots; c.cull = g0.cull;
        c.textures = g0.textures; c.topFlat = g0.topFlat; c.kernel = g0.kernel; c.overlap = g0.overlap; c.groupInstances = g0.groupInstances; c.unifiedTree = g0.unifiedTree;
        c.sortRays = g0.sortRays; c.quad = g0.quad; c.smallChunkPaths = g0.smallChunkPaths; c.sortMinPaths = g0.sortMinPaths;
    }
    for (int d = 0; d < n; ++d) {
        tl_ctx = g_dev[d]; tl_dev = d;
        hipError_t e = hipSetDevice(g_phys[d]);
        const int rc = e == hipSuccess ? derive_accel(bTlas) : -1;
        const std::string msg = g.err;
        tl_ctx = &g0; tl_dev = 0;
        HIP_IGN(hipSetDevice(g_phys[0]));
        if (rc) return fail("device %d: %s", d, e == hipSuccess ? msg.c_str() : hipGetErrorString(e));
    }
    std::vector<int> rcs(n, 0);
    std::vector<std::string> msgs(n);
    auto work = [&](int d) {
        tl_ctx = g_dev[d]; tl_dev = d;
        if (hipSetDevice(g_phys[d]) != hipSuccess) { rcs[d] = -1; msgs[d] = "hipSetDevice failed"; return; }
        g.rank = (uint32_t)d; g.world = (uint32_t)n; g.tileW = 64; g.tileH = 64;
        rcs[d] = trace_rays_device(width, height);
        if (rcs[d]) msgs[d] = g.err;
    };
    {
        std::vector<std::thread> pool;
        for (int d = 1; d < n; ++d) pool.emplace_back(work, d);
        work(0);
        for (auto& t : pool) t.join();
    }
    tl_ctx = &g0; tl_dev = 0;
    g0.rank = 0; g0.world = 1;
    HIP_OK(hipSetDevice(g_phys[0]));
    for (int d = 0; d < n; ++d) if (rcs[d]) return fail("device %d: %s", d, msgs[d].c_str());
    // gather: device d's tiles -> device 0 (image and the running-mean accumulator)
    const uint32_t tilesX = (width + 63) / 64, tilesY = (height + 63) / 64, nTiles = tilesX * tilesY;
    for (int d = 1; d < n; ++d) {
        const uint32_t owned = nTiles > (uint32_t)d ? (nTiles - d + n - 1) / n : 0;
        if (!owned) continue;
        const size_t bytes[2] = {(size_t)owned * 64 * 64 * 4, (size_t)owned * 64 * 64 * 16};
        const uint32_t elem[2] = {4, 16};
        rdx_buffer_s* src[2] = {bImage, bScratch};
        for (int k = 0; k < 2; ++k) {
            if (ensure_stage(2 + k, bytes[k])) return -1;                      // landing buffer on device 0
            void* land = g0.gatherStage[2 + k];
            tl_ctx = g_dev[d]; tl_dev = d;
            hipError_t e = hipSetDevice(g_phys[d]);
            int rc = e == hipSuccess ? ensure_stage(k, bytes[k]) : -1;
            void* stage = g.gatherStage[k];
            if (!rc) {
                launch_pack_tiles(g.stream, static_cast<uint8_t*>(src[k]->rep[d]), static_cast<uint8_t*>(stage), width, height, elem[k], 64, 64,
                                  (uint32_t)d, (uint32_t)n, false);
                e = hipGetLastError();
                if (e == hipSuccess) e = g_phys[d] == g_phys[0] ? hipMemcpyAsync(land, stage, bytes[k], hipMemcpyDeviceToDevice, g.stream)
                                                                : hipMemcpyPeerAsync(land, g_phys[0], stage, g_phys[d], bytes[k], g.stream);
                if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
                if (e != hipSuccess) rc = -1;
            }
            const std::string msg = rc ? (e != hipSuccess ? std::string(hipGetErrorString(e)) : g.err) : std::string();
            tl_ctx = &g0; tl_dev = 0;
            HIP_IGN(hipSetDevice(g_phys[0]));
            if (rc) return fail("gather from device %d: %s", d, msg.c_str());
            launch_pack_tiles(g0.stream, static_cast<uint8_t*>(src[k]->dptr), static_cast<uint8_t*>(land), width, height, elem[k], 64, 64,
                              (uint32_t)d, (uint32_t)n, true);
            HIP_OK(hipGetLastError());
            HIP_OK(hipStreamSynchronize(g0.stream));
        }
    }
    // statistics of the whole frame: rays summed over the devices, times = the slowest device
    for (int d = 1; d < n; ++d) {
        const rdx_trace_stats& t = g_dev[d]->stats;
        rdx_trace_stats& a = g0.stats;
        a.rays_primary += t.rays_primary; a.rays_bounce += t.rays_bounce; a.rays_shadow += t.rays_shadow;
        a.closest_hits += t.closest_hits; a.pixels += t.pixels;
        a.ms_total = std::max(a.ms_total, t.ms_total);
        for (int i = 0; i < 65; ++i) g0.bounceCounts[i] += g_dev[d]->bounceCounts[i];
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// test seams
// ------------------------------------------------------------------------------------------------
namespace {
template <class T> struct DevArray {
    T* p = nullptr;
    ~DevArray() { if (p) HIP_IGN(hipFree(p)); }
    hipError_t alloc(size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T)); }
    hipError_t upload(const T* src, size_t n) { hipError_t e = alloc(n); if (e != hipSuccess || !n) return e; return hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice); }
};
}

extern "C" int rdx_trace_batch(rdx_buffer tlas, const float* o, const float* d, uint32_t n, float tmin, float tmax,
                               int rec, int mode, rdx_hit* out, uint64_t* visit4)
{
    if (mode != 0 && mode != 1) return fail("rdx_trace_batch: mode must be 0 (production) or 1 (reference order)");
    if (!g.initialized) return fail("rdx_init has not been called");
    if (!tlas || !known_buffer(tlas)) return fail("rdx_trace_batch: invalid TLAS handle");
    if (rec != 1 && rec != 2) return fail("rdx_trace_batch: sbtRecordOffset must be 1 or 2");
    if (derive_accel(tlas)) return -1;
    DevArray<float> dO, dD; DevArray<rdx_hit> dH;
    HIP_OK(dO.upload(o, 3 * (size_t)n)); HIP_OK(dD.upload(d, 3 * (size_t)n)); HIP_OK(dH.alloc(n));
    if (visit4) HIP_OK(hipMemsetAsync(g.dVisit, 0, 8 * sizeof(unsigned long long), g.stream));
    HIP_OK(hipMemsetAsync(g.dCounts + 64, 0, sizeof(uint32_t), g.stream));
    HIP_OK(hipEventRecord(g.evA, g.stream));
    launch_trace_batch(g.stream, view_of(tlas), dO.p, dD.p, n, tmin, tmax, rec, dH.p, visit4 ? g.dVisit : nullptr, mode, g.dCounts + 64);
    HIP_OK(hipEventRecord(g.evB, g.stream));
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    if (take_status()) return fail("rdx_trace_batch: a traversal wave exceeded its iteration bound and gave up; the batch is incomplete");
    std::memset(&g.stats, 0, sizeof g.stats);
    HIP_OK(hipEventElapsedTime(&g.stats.ms_extend, g.evA, g.evB));      // kernel time of this batch
    if (n) HIP_OK(hipMemcpy(out, dH.p, (size_t)n * sizeof(rdx_hit), hipMemcpyDeviceToHost));
    if (visit4) {
        unsigned long long v[8];
        HIP_OK(hipMemcpy(v, g.dVisit, sizeof v, hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; ++k) visit4[k] = v[(rec - 1) * 4 + k];
    }
    return 0;
}

extern "C" int rdx_material_batch(const rdx_hit* hits, const float* dirs, const uint32_t* pixels, const uint32_t* frames,
                                  const int32_t* depths, uint32_t n, rdx_payload* out)
{
    if (!g.initialized) return fail("rdx_init has not been called");
    SceneArgs sc;
    if (scene_args(sc)) return -1;
    DevArray<rdx_hit> dH; DevArray<float> dD; DevArray<uint32_t> dP, dF; DevArray<int32_t> dDep; DevArray<rdx_payload> dO;
    HIP_OK(dH.upload(hits, n)); HIP_OK(dD.upload(dirs, 3 * (size_t)n)); HIP_OK(dP.upload(pixels, n));
    HIP_OK(dF.upload(frames, n)); HIP_OK(dDep.upload(depths, n)); HIP_OK(dO.alloc(n));
    launch_material_batch(g.stream, sc, dH.p, dD.p, dP.p, dF.p, dDep.p, n, dO.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    if (n) HIP_OK(hipMemcpy(out, dO.p, (size_t)n * sizeof(rdx_payload), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int rdx_generate_batch(const uint32_t* pixels, const uint32_t* rnd, uint32_t n, float* o, float* d)
{
    if (!g.initialized) return fail("rdx_init has not been called");
    if (!g.slots[3] || !known_buffer(g.slots[3])) return fail("descriptor slot 3 (camera) is not a buffer");
    PhysicalCamera cam;
    HIP_OK(hipMemcpy(&cam, static_cast<rdx_buffer_s*>(g.slots[3])->dptr, sizeof cam, hipMemcpyDeviceToHost));
    CameraArgs C;
    if (camera_args(cam, C)) return -1;
    DevArray<uint32_t> dP, dR; DevArray<float> dO, dD;
    HIP_OK(dP.upload(pixels, n)); HIP_OK(dR.upload(rnd, 3 * (size_t)n)); HIP_OK(dO.alloc(3 * (size_t)n)); HIP_OK(dD.alloc(3 * (size_t)n));
    launch_generate_batch(g.stream, C, dP.p, dR.p, n, dO.p, dD.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    if (n) { HIP_OK(hipMemcpy(o, dO.p, 12 * (size_t)n, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(d, dD.p, 12 * (size_t)n, hipMemcpyDeviceToHost)); }
    return 0;
}

extern "C" int rdx_pcg3d_batch(const uint32_t* in3, float* out3, uint32_t n)
{
    if (!g.initialized) return fail("rdx_init has not been called");
    DevArray<uint32_t> dI; DevArray<float> dO;
    HIP_OK(dI.upload(in3, 3 * (size_t)n)); HIP_OK(dO.alloc(3 * (size_t)n));
    launch_pcg3d_batch(g.stream, dI.p, dO.p, n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(g.stream));
    if (n) HIP_OK(hipMemcpy(out3, dO.p, 12 * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

// bvh_build.cpp -- bit-exact, fast restatement of the reference's binned-SAH BVH builder.
//
// Reference behaviour being reproduced (radiance/src/bvh.cpp:46-285 `Recurse`):
//   * leaf if fewer than 8 primitives (:55) or if no candidate plane beats N * halfArea (:83,:212);
//   * per axis x,y,z (skipped when the extent is < 1e-4, :116) the candidate planes are
//     t0 = start + step, t(k+1) = t(k) + step (float accumulation) while t < stop - step, with
//     step = (stop - start) / (1024. / (depth + 1.)) evaluated in double and rounded to float (:123);
//   * a primitive goes left iff centroid[axis] < t; candidates leaving <= 1 primitive on a side
//     are skipped (:180); cost = SAl*nl + SAr*nr with SA = xy + yz + zx (:192-196); the first
//     candidate with the strictly smallest cost wins, axis-major (:199-203);
//   * the winning partition is stable and children carry the exact boxes of their sides (:224-282);
//   * flatten: DFS pre-order, left child = parent + 1, primitives re-emitted in leaf order (:463-597).
//
// How this implementation differs (same results, different algorithm): the reference re-scans all
// N primitives for each of up to 1024/(depth+1) candidates.  Here each node sorts its primitives by
// centroid once per axis and keeps prefix / suffix boxes, so a candidate costs one pointer advance
// (candidates are monotonically increasing) and O(1) arithmetic.  min/max of finite floats do not
// depend on evaluation order except for the sign of a zero, and a zero's sign cannot change a
// `cost < minCost` decision, so the chosen (axis, plane) is identical; the boxes that are WRITTEN to
// the blob are then recomputed in the reference's own sequential order, which makes them
// bit-identical including zero signs.
//
// Compiled with -ffp-contract=off: every float expression below is evaluated exactly as written.
#include "bvh_build.h"
#include "rdx_types.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <atomic>
#include <cstdlib>
#include <system_error>
#include <thread>

namespace rdx {
namespace {

struct V3 { float x, y, z; };
inline float comp(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
// linalg.h:28-40 -- note the tie rule (equal -> second operand), which fixes the sign of zeros
inline V3 vmin(const V3& a, const V3& b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline V3 vmax(const V3& a, const V3& b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
const V3 kBig{FLT_MAX, FLT_MAX, FLT_MAX};
const V3 kSmall{-FLT_MAX, -FLT_MAX, -FLT_MAX};

struct Prim { V3 lo, hi, c; uint32_t id; };
struct Box { V3 lo, hi; };
inline float half_area(const Box& b)
{
    float s1 = b.hi.x - b.lo.x, s2 = b.hi.y - b.lo.y, s3 = b.hi.z - b.lo.z;
    return s1 * s2 + s2 * s3 + s3 * s1;
}

struct Node {
    V3 lo{}, hi{};
    std::unique_ptr<Node> left, right;
    std::vector<uint32_t> prims;    // leaf payload (primitive ids, in work order)
    bool leaf = false;
};

// ---- candidate evaluation by bins ---------------------------------------------------------------------------------------
// The candidate planes of one axis are a short increasing sequence t_0 < t_1 < ... (at most 1024 / (depth + 1) of them).  A
// primitive with centroid c is on the left of candidate k iff c < t_k, i.e. iff k >= b(c), b(c) = index of the first candidate
// greater than c (K if there is none): one pass drops every primitive into bin b(c) -- a count and a box per bin -- and the
// left / right sets of candidate k are the bins <= k / > k.  O(n + K) per axis instead of a sort; min / max do not depend on
// the order of evaluation except for the sign of a zero, which no `cost < minCost` decision can see.  The same bins are what
// the GPU fills for large nodes (GpuBinner below).
struct Bins {
    std::vector<uint32_t> cnt;      // [K + 1]
    std::vector<Box> box;           // [K + 1]
    std::vector<float> t;           // [K] candidate planes
    void reset(size_t K) { cnt.assign(K + 1, 0u); box.assign(K + 1, Box{kBig, kSmall}); }
};

// index of the first candidate greater than c (t increasing, step > 0): an estimate from the spacing, corrected by comparison
inline uint32_t bin_of(const std::vector<float>& t, float start, float invStep, float c)
{
    const size_t K = t.size();
    float e = (c - start) * invStep - 1.0f;           // t_k ~ start + (k + 1) step
    size_t b = e <= 0.0f ? 0 : (e >= (float)K ? K : (size_t)e);
    while (b < K && !(c < t[b])) ++b;
    while (b > 0 && c < t[b - 1]) --b;
    return (uint32_t)b;
}

struct SplitChoice { float minCost; float bestSplit = FLT_MAX; int bestAxis = -1; };

// candidates of one axis + the sweep over filled bins (bvh.cpp:116-205); false = the reference would loop forever
inline bool make_candidates(float start, float stop, int depth, std::vector<float>& t, float& step, std::string& err)
{
    t.clear();
    step = (float)((double)(stop - start) / (1024. / (depth + 1.)));
    for (float testSplit = start + step; testSplit < stop - step; testSplit += step) {
        t.push_back(testSplit);
        if (testSplit + step == testSplit) {
            err = "BVH build: split step underflows at this coordinate magnitude "
                  "(the reference builder would loop forever here)";
            return false;
        }
    }
    return true;
}
inline void sweep_bins(const Bins& B, size_t n, int axis, std::vector<Box>& suf, SplitChoice& best)
{
    const size_t K = B.t.size();
    if (K == 0) return;
    suf.resize(K + 2);
    Box acc{kBig, kSmall};
    suf[K + 1] = acc;
    for (size_t b = K + 1; b-- > 0;) { acc.lo = vmin(acc.lo, B.box[b].lo); acc.hi = vmax(acc.hi, B.box[b].hi); suf[b] = acc; }
    Box pre{kBig, kSmall};
    size_t cnt = 0;
    for (size_t k = 0; k < K; ++k) {
        pre.lo = vmin(pre.lo, B.box[k].lo); pre.hi = vmax(pre.hi, B.box[k].hi);
        cnt += B.cnt[k];
        const int countLeft = (int)cnt, countRight = (int)(n - cnt);
        if (!(countLeft <= 1 || countRight <= 1)) {
            float surfaceLeft = half_area(pre);
            float surfaceRight = half_area(suf[k + 1]);
            float totalCost = surfaceLeft * countLeft + surfaceRight * countRight;
            if (totalCost < best.minCost) { best.minCost = totalCost; best.bestSplit = B.t[k]; best.bestAxis = axis; }
        }
    }
}

} // namespace

// ---- GPU-assisted candidate evaluation (bvh.cpp:90-205) -------------------------------------------------------------------
// The binning pass of a large node on the GPU (csrc/kernels.hip k_bvh_bin): the primitives of the BLAS are uploaded once, a
// node sends its work list and its candidate planes and gets the filled bins back; the sweep, the choice and the stable
// partition stay on the host.  Installed by the runtime when a device is up (rdx_runtime.cpp); absent, the host bins.
static GpuBinner* g_gpuBinner = nullptr;
static size_t g_gpuMin = 32768;
void set_gpu_binner(GpuBinner* b, size_t minPrims) { g_gpuBinner = b; g_gpuMin = minPrims ? minPrims : 32768; }

namespace {

// subtrees of one BLAS are built concurrently: a bounded number of helper threads over the whole process
std::atomic<int> g_helpers{0};
int g_helperLimit = -1;
inline int helper_limit()
{
    if (g_helperLimit < 0) {
        const char* e = std::getenv("RDX_BUILD_THREADS");
        const unsigned hw = std::thread::hardware_concurrency();
        g_helperLimit = e ? std::max(0, std::atoi(e) - 1) : (int)std::min(31u, hw > 1 ? hw - 1 : 0u);
    }
    return g_helperLimit;
}

struct Builder {
    const std::vector<Prim>& P;
    std::string& err;
    uint64_t gpuHandle;             // the BLAS's primitives on the GPU (0 = none)
    // scratch reused across nodes
    Bins bins;
    std::vector<Box> suf;
    std::vector<float> gpuOut;

    Builder(const std::vector<Prim>& p, std::string& e, uint64_t gh) : P(p), err(e), gpuHandle(gh) {}

    std::unique_ptr<Node> make_leaf(const std::vector<uint32_t>& work)
    {
        auto n = std::make_unique<Node>();
        n->leaf = true;
        n->prims.reserve(work.size());
        for (uint32_t w : work) n->prims.push_back(P[w].id);
        return n;
    }

    std::unique_ptr<Node> recurse(const std::vector<uint32_t>& work, int depth)
    {
        const size_t n = work.size();
        if (n < 8) return make_leaf(work);

        V3 bottom = kBig, top = kSmall;
        for (uint32_t w : work) { bottom = vmin(bottom, P[w].lo); top = vmax(top, P[w].hi); }
        float side1 = top.x - bottom.x, side2 = top.y - bottom.y, side3 = top.z - bottom.z;
        SplitChoice best;
        best.minCost = (float)n * (side1 * side2 + side2 * side3 + side3 * side1);

        // candidate planes of the three axes
        std::vector<float> cand[3];
        float steps[3] = {0.f, 0.f, 0.f};
        bool axisOn[3] = {false, false, false};
        for (int axis = 0; axis < 3; ++axis) {
            const float start = comp(bottom, axis), stop = comp(top, axis);
            if ((double)fabsf(stop - start) < 1e-4) continue;
            if (!make_candidates(start, stop, depth, cand[axis], steps[axis], err)) return nullptr;
            axisOn[axis] = true;
        }
        bool onGpu = false;
        if (gpuHandle && g_gpuBinner && n >= g_gpuMin) {
            // layout of the answer: per axis [K + 1] counts, then [K + 1] x {lo.xyz, hi.xyz}
            onGpu = g_gpuBinner->bin(gpuHandle, work.data(), n, cand, gpuOut);
        }
        size_t off = 0;
        for (int axis = 0; axis < 3; ++axis) {
            if (!axisOn[axis]) continue;
            const size_t K = cand[axis].size();
            bins.t.swap(cand[axis]);
            if (onGpu) {
                bins.cnt.resize(K + 1); bins.box.resize(K + 1);
                const float* o = gpuOut.data() + off;
                for (size_t b = 0; b <= K; ++b) {
                    uint32_t c; std::memcpy(&c, o + b, 4); bins.cnt[b] = c;
                    const float* q = o + (K + 1) + 6 * b;
                    bins.box[b] = Box{V3{q[0], q[1], q[2]}, V3{q[3], q[4], q[5]}};
                }
                off += 7 * (K + 1);
            } else if (K) {
                bins.reset(K);
                const float start = comp(bottom, axis), invStep = 1.0f / steps[axis];
                for (uint32_t w : work) {
                    const Prim& p = P[w];
                    const uint32_t b = bin_of(bins.t, start, invStep, comp(p.c, axis));
                    bins.cnt[b]++;
                    Box& bx = bins.box[b];
                    bx.lo = vmin(bx.lo, p.lo); bx.hi = vmax(bx.hi, p.hi);
                }
            }
            sweep_bins(bins, n, axis, suf, best);
        }
        const int bestAxis = best.bestAxis;
        const float bestSplit = best.bestSplit;
        if (bestAxis == -1) return make_leaf(work);

        // stable partition + child boxes, in the reference's sequential order (exact zero signs)
        std::vector<uint32_t> left, right;
        left.reserve(n); right.reserve(n);
        Box lb{kBig, kSmall}, rb{kBig, kSmall};
        for (uint32_t w : work) {
            const Prim& p = P[w];
            if (comp(p.c, bestAxis) < bestSplit) { left.push_back(w); lb.lo = vmin(lb.lo, p.lo); lb.hi = vmax(lb.hi, p.hi); }
            else { right.push_back(w); rb.lo = vmin(rb.lo, p.lo); rb.hi = vmax(rb.hi, p.hi); }
        }
        auto inner = std::make_unique<Node>();
        // the two subtrees are independent: the right one goes to a helper thread when one is free and it is worth a thread
        std::unique_ptr<Node> rightNode;
        std::string rightErr;
        std::thread helper;
        bool spawned = false;
        if (right.size() >= 8192 && left.size() >= 8192) {
            if (g_helpers.fetch_add(1) < helper_limit()) {
                try {
                    helper = std::thread([&]() {
                        Builder B2(P, rightErr, gpuHandle);
                        rightNode = B2.recurse(right, depth + 1);
                    });
                    spawned = true;
                } catch (const std::system_error&) { spawned = false; }      // (no thread to be had: this one does both)
            }
            if (!spawned) g_helpers.fetch_sub(1);
        }
        inner->left = recurse(left, depth + 1);
        if (spawned) {
            helper.join();
            g_helpers.fetch_sub(1);
            if (!rightNode && err.empty()) err = rightErr;
        } else if (inner->left) {
            rightNode = recurse(right, depth + 1);
        }
        if (!inner->left || !rightNode) return nullptr;
        inner->left->lo = lb.lo; inner->left->hi = lb.hi;
        inner->right = std::move(rightNode);
        inner->right->lo = rb.lo; inner->right->hi = rb.hi;
        return inner;
    }
};

uint32_t count_nodes(const Node* n) { return n->leaf ? 1u : 1u + count_nodes(n->left.get()) + count_nodes(n->right.get()); }
void max_depth(const Node* n, int d, int& m) { if (m < d) m = d; if (!n->leaf) { max_depth(n->left.get(), d + 1, m); max_depth(n->right.get(), d + 1, m); } }

// DFS pre-order flatten (bvh.cpp:463-500 / :524-566).  emit(primId) appends one leaf record.
template <class Emit>
void flatten(const Node* n, BlobNode* nodes, uint32_t& nodeIdx, uint32_t& primIdx, uint32_t leafType, Emit&& emit)
{
    const uint32_t cur = nodeIdx;
    BlobNode& o = nodes[cur];
    o.bottom[0] = n->lo.x; o.bottom[1] = n->lo.y; o.bottom[2] = n->lo.z; o.bottom[3] = 0.f;
    o.top[0] = n->hi.x; o.top[1] = n->hi.y; o.top[2] = n->hi.z; o.top[3] = 0.f;
    if (!n->leaf) {
        const uint32_t l = ++nodeIdx;
        flatten(n->left.get(), nodes, nodeIdx, primIdx, leafType, emit);
        const uint32_t r = ++nodeIdx;
        flatten(n->right.get(), nodes, nodeIdx, primIdx, leafType, emit);
        nodes[cur].w0 = l; nodes[cur].w1 = r; nodes[cur].w2 = 0; nodes[cur].w3 = 0;
    } else {
        o.w0 = LEAF_BIT | (uint32_t)n->prims.size();
        o.w1 = primIdx;
        o.w2 = leafType;
        o.w3 = 0;
        for (uint32_t id : n->prims) { emit(primIdx, id); ++primIdx; }
    }
}

} // namespace

// bvh.cpp:288-341 + :502-522 + radiance.cpp:318-364
Blas* build_blas(const float* v, uint32_t nverts, const uint32_t* idx, uint32_t ntris, std::string& err)
{
    if (ntris == 0 || nverts == 0) { err = "BuildAccelStruct(Mesh): empty mesh"; return nullptr; }
    for (uint32_t i = 0; i < 3 * ntris; ++i)
        if (idx[i] >= nverts) { err = "BuildAccelStruct(Mesh): vertex index out of range"; return nullptr; }

    std::vector<Prim> prims(ntris);
    V3 bottom = kBig, top = kSmall;
    for (uint32_t j = 0; j < ntris; ++j) {
        Prim& b = prims[j];
        b.lo = kBig; b.hi = kSmall; b.id = j;
        const V3 p0{v[3 * idx[3 * j]], v[3 * idx[3 * j] + 1], v[3 * idx[3 * j] + 2]};
        const V3 p1{v[3 * idx[3 * j + 1]], v[3 * idx[3 * j + 1] + 1], v[3 * idx[3 * j + 1] + 2]};
        const V3 p2{v[3 * idx[3 * j + 2]], v[3 * idx[3 * j + 2] + 1], v[3 * idx[3 * j + 2] + 2]};
        b.lo = vmin(b.lo, p0); b.lo = vmin(b.lo, p1); b.lo = vmin(b.lo, p2);
        b.hi = vmax(b.hi, p0); b.hi = vmax(b.hi, p1); b.hi = vmax(b.hi, p2);
        bottom = vmin(bottom, b.lo); top = vmax(top, b.hi);
        b.c = V3{(b.hi.x + b.lo.x) * 0.5f, (b.hi.y + b.lo.y) * 0.5f, (b.hi.z + b.lo.z) * 0.5f};
    }
    std::vector<uint32_t> work(ntris);
    std::iota(work.begin(), work.end(), 0u);
    // large meshes: the primitives go to the GPU once, large nodes are binned there (GpuBinner)
    uint64_t gpuHandle = 0;
    if (g_gpuBinner && ntris >= g_gpuMin) {
        std::vector<float> flat((size_t)ntris * 9u);
        for (uint32_t j = 0; j < ntris; ++j) {
            const Prim& q = prims[j];
            float* f = flat.data() + 9u * j;
            f[0] = q.lo.x; f[1] = q.lo.y; f[2] = q.lo.z; f[3] = q.hi.x; f[4] = q.hi.y; f[5] = q.hi.z; f[6] = q.c.x; f[7] = q.c.y; f[8] = q.c.z;
        }
        gpuHandle = g_gpuBinner->upload(flat.data(), ntris);
    }
    Builder B(prims, err, gpuHandle);
    std::unique_ptr<Node> root = B.recurse(work, 0);
    if (gpuHandle) g_gpuBinner->release(gpuHandle);
    if (!root) return nullptr;
    root->lo = bottom; root->hi = top;

    const uint32_t nodeCount = count_nodes(root.get());
    const uint64_t nodeBytes = (uint64_t)nodeCount * sizeof(BlobNode), faceBytes = (uint64_t)ntris * sizeof(BlobTri),
                   vertBytes = (uint64_t)nverts * 16u;
    const uint64_t total = 16u + nodeBytes + faceBytes + vertBytes;
    if (total > 0xffffffffull) { err = "BLAS exceeds the 4 GiB blob limit (32-bit byte offsets)"; return nullptr; }

    auto* blas = new Blas();
    blas->data.assign((size_t)total, 0);
    BlobBotHeader hdr{TYPE_BOT_AS, 16u, (uint32_t)(16u + nodeBytes), (uint32_t)(16u + nodeBytes + faceBytes)};
    std::memcpy(blas->data.data(), &hdr, sizeof hdr);
    auto* nodes = reinterpret_cast<BlobNode*>(blas->data.data() + hdr.nodeByteOffset);
    auto* faces = reinterpret_cast<BlobTri*>(blas->data.data() + hdr.faceByteOffset);
    uint32_t nodeIdx = 0, faceIdx = 0;
    flatten(root.get(), nodes, nodeIdx, faceIdx, TYPE_TRIG, [&](uint32_t slot, uint32_t id) {
        faces[slot] = BlobTri{idx[3 * id], idx[3 * id + 1], idx[3 * id + 2], id};
    });
    auto* pv = reinterpret_cast<float*>(blas->data.data() + hdr.vertexOffset);
    for (uint32_t i = 0; i < nverts; ++i) { pv[4 * i] = v[3 * i]; pv[4 * i + 1] = v[3 * i + 1]; pv[4 * i + 2] = v[3 * i + 2]; }
    max_depth(root.get(), 0, blas->maxDepth);
    return blas;
}

// bvh.cpp:343-420 + :568-597 + radiance.cpp:366-425.
// The corner transform is assimp's aiMatrix4x4t::operator* (un-vendored third party):
// out[r][c] = V[0][c]*T[r][0] + V[1][c]*T[r][1] + V[2][c]*T[r][2] + V[3][c]*T[r][3], left to right.
bool build_tlas(const InstanceDesc* inst, uint32_t ninst, std::vector<uint8_t>& blob, int& maxDepthOut, std::string& err)
{
    if (ninst == 0) { err = "BuildAccelStruct(instances): empty instance list"; return false; }
    std::vector<Prim> prims(ninst);
    V3 bottom = kBig, top = kSmall;
    for (uint32_t k = 0; k < ninst; ++k) {
        const Blas* bl = inst[k].blas;
        if (!bl || bl->data.size() < 16 + sizeof(BlobNode)) { err = "instance without a bottom-level structure"; return false; }
        const auto* bh = reinterpret_cast<const BlobBotHeader*>(bl->data.data());
        const auto* rootNode = reinterpret_cast<const BlobNode*>(bl->data.data() + bh->nodeByteOffset);
        const float tx = rootNode->top[0], ty = rootNode->top[1], tz = rootNode->top[2];
        const float bx = rootNode->bottom[0], by = rootNode->bottom[1], bz = rootNode->bottom[2];
        const float cornersHi[4][4] = {{tx, bx, tx, bx}, {ty, ty, by, by}, {tz, tz, tz, tz}, {1, 1, 1, 1}};
        const float cornersLo[4][4] = {{tx, bx, tx, bx}, {ty, ty, by, by}, {bz, bz, bz, bz}, {1, 1, 1, 1}};
        const float* T = inst[k].transform;
        V3 hiCol[4], loCol[4];
        for (int c = 0; c < 4; ++c) {
            float h[3], l[3];
            for (int r = 0; r < 3; ++r) {
                h[r] = cornersHi[0][c] * T[4 * r] + cornersHi[1][c] * T[4 * r + 1] + cornersHi[2][c] * T[4 * r + 2] + cornersHi[3][c] * T[4 * r + 3];
                l[r] = cornersLo[0][c] * T[4 * r] + cornersLo[1][c] * T[4 * r + 1] + cornersLo[2][c] * T[4 * r + 2] + cornersLo[3][c] * T[4 * r + 3];
            }
            hiCol[c] = V3{h[0], h[1], h[2]};
            loCol[c] = V3{l[0], l[1], l[2]};
        }
        Prim& b = prims[k];
        b.id = k;
        // reduction tree of bvh.cpp:386-401 (operand order matters for zero signs)
        b.lo = vmin(vmin(vmin(hiCol[3], hiCol[2]), vmin(hiCol[1], hiCol[0])),
                    vmin(vmin(loCol[3], loCol[2]), vmin(loCol[1], loCol[0])));
        b.hi = vmax(vmax(vmax(hiCol[3], hiCol[2]), vmax(hiCol[1], hiCol[0])),
                    vmax(vmax(loCol[3], loCol[2]), vmax(loCol[1], loCol[0])));
        b.c = V3{(b.hi.x + b.lo.x) * 0.5f, (b.hi.y + b.lo.y) * 0.5f, (b.hi.z + b.lo.z) * 0.5f};
        bottom = vmin(bottom, b.lo); top = vmax(top, b.hi);
    }
    std::vector<uint32_t> work(ninst);
    std::iota(work.begin(), work.end(), 0u);
    Builder B(prims, err, 0);
    std::unique_ptr<Node> root = B.recurse(work, 0);
    if (!root) return false;
    root->lo = bottom; root->hi = top;

    const uint32_t nodeCount = count_nodes(root.get());
    const uint64_t topSize = 16u + (uint64_t)nodeCount * sizeof(BlobNode) + (uint64_t)ninst * sizeof(BlobInst);
    // distinct BLAS blobs appended once each, in order of first appearance (bvh.cpp:575-588)
    std::map<const Blas*, uint32_t> offsetOf;
    uint64_t next = 0;
    for (uint32_t k = 0; k < ninst; ++k)
        if (!offsetOf.count(inst[k].blas)) {
            if (next + topSize > 0xffffffffull) { err = "TLAS exceeds the 4 GiB blob limit (32-bit byte offsets)"; return false; }
            offsetOf[inst[k].blas] = (uint32_t)(next + topSize);
            next += inst[k].blas->data.size();
        }
    const uint64_t total = topSize + next;
    if (total > 0xffffffffull) { err = "TLAS exceeds the 4 GiB blob limit (32-bit byte offsets)"; return false; }

    blob.assign((size_t)total, 0);
    BlobTopHeader hdr{TYPE_TOP_AS, 16u, (uint32_t)(16u + (uint64_t)nodeCount * sizeof(BlobNode)), (uint32_t)total};
    std::memcpy(blob.data(), &hdr, sizeof hdr);
    auto* nodes = reinterpret_cast<BlobNode*>(blob.data() + hdr.nodeByteOffset);
    auto* out = reinterpret_cast<BlobInst*>(blob.data() + hdr.instByteOffset);
    uint32_t nodeIdx = 0, instIdx = 0;
    flatten(root.get(), nodes, nodeIdx, instIdx, TYPE_INST, [&](uint32_t slot, uint32_t id) {
        BlobInst& d = out[slot];
        std::memcpy(d.m, inst[id].transform, 64);
        d.SBTOffset = inst[id].SBTOffset;
        d.instanceID = id;
        d.customInstanceID = inst[id].customInstanceID;
        d.instanceOffset = offsetOf[inst[id].blas];
    });
    for (auto& kv : offsetOf) std::memcpy(blob.data() + kv.second, kv.first->data.data(), kv.first->data.size());
    maxDepthOut = 0;
    max_depth(root.get(), 0, maxDepthOut);
    return true;
}

} // namespace rdx

// kernels.hip -- wavefront path-tracing stages for MI355X (gfx950, wave64).
//
// Stage graph of one bounce (all launches on one HIP stream):
//
//   generate ──► [A streams] ──► extend (closest hit, exhaustive BVH walk)
//                                   │ hit records
//                                   ▼
//                                shade  (SBT dispatch: material / environment; wave64 ballot
//                                   │    compaction of the paths that hit into the B streams)
//                                   ▼
//                                shadow (any-hit walk for the deferred shadow query, picks the
//                                   │    lit / occluded colour, writes the next bounce's A streams)
//                                   ▼
//                       ... next bounce ...  ──►  accumulate (running mean, ACES + gamma, RGBA8)
//
// Exactness contract of the traversal (see DESIGN.md "Traversal"): the set of nodes a ray visits,
// the slab test, the Möller–Trumbore arithmetic, the accept test and the visiting order are those of
// radiance/shader/radiance.cl:41-251; only the data layout and the execution model differ.
#include "kernels.h"
#include "stages.h"

#include <algorithm>
#include <cfloat>

namespace rdx {

#define RDX_BLOCK 256
#define RDX_MISS 0xffffffffu

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// ---------------------------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------------------------

// radiance.cl:195-208 -- slab test by division, tFar > max(tNear, 0); no best-t culling
__device__ __forceinline__ bool slab_hit(f3 o, f3 d, const float4& bmin, const float4& bmax)
{
    f3 tA = (mk3(bmin.x, bmin.y, bmin.z) - o) / d;
    f3 tB = (mk3(bmax.x, bmax.y, bmax.z) - o) / d;
    f3 t1 = mk3(cl_min(tA.x, tB.x), cl_min(tA.y, tB.y), cl_min(tA.z, tB.z));
    f3 t2 = mk3(cl_max(tA.x, tB.x), cl_max(tA.y, tB.y), cl_max(tA.z, tB.z));
    float tNear = cl_max(cl_max(t1.x, t1.y), t1.z);
    float tFar = cl_min(cl_min(t2.x, t2.y), t2.z);
    return tFar > cl_max(tNear, 0.0f);
}

struct TraceResult {
    float t, b1, b2;
    uint32_t prim, inst;
    bool hit;
    uint32_t nTop, nInst, nBot, nTri;   // visit counts (COUNT builds only; SURVEY.md 8d byte model)
};

// One ray, one lane.  Left-first DFS with the lane's stack in LDS ([level][lane], conflict-free).
// REC is the sbtRecordOffset of traceRay(): 1 = radiance ray, 2 = shadow ray.  Whether an accepted
// candidate terminates the walk is decided by the generated any-hit table (row 2 -> anyShadow).
template <int REC, bool COUNT>
__device__ __forceinline__ void traverse(const AccelView& A, f3 o, f3 d, float tmin, float tmax,
                                         uint32_t* __restrict__ stack, uint32_t stride, TraceResult& r)
{
    uint32_t sp = 0;
    uint32_t cur = TAG_TLAS | 0u;
    f3 lo = o, ld = d;
    uint32_t curInst = 0;
    r.t = FLT_MAX; r.b1 = 0.f; r.b2 = 0.f; r.prim = 0; r.inst = RDX_MISS; r.hit = false;
    r.nTop = r.nInst = r.nBot = r.nTri = 0;

    for (;;) {
        const uint32_t tag = cur & TAG_MASK, idx = cur & IDX_MASK;
        if (tag == TAG_INST) {
            if (COUNT) ++r.nInst;
            // enter an instance: object-space ray = inverse(object->world) * (o,1), (d,0)   (radiance.cl:161-169)
            const float4* ip = reinterpret_cast<const float4*>(A.insts[idx].inv);
            float m[16];
            *reinterpret_cast<float4*>(m + 0) = ip[0];
            *reinterpret_cast<float4*>(m + 4) = ip[1];
            *reinterpret_cast<float4*>(m + 8) = ip[2];
            *reinterpret_cast<float4*>(m + 12) = ip[3];
            lo = mat4_mul3(m, o.x, o.y, o.z, 1.0f);
            ld = mat4_mul3(m, d.x, d.y, d.z, 0.0f);
            curInst = idx;
            cur = A.insts[idx].blasRoot;
            continue;
        }
        const bool top = (tag == TAG_TLAS);
        if (COUNT) { if (top) ++r.nTop; else ++r.nBot; }
        const float4* np = reinterpret_cast<const float4*>((top ? A.tnodes : A.bnodes) + idx);
        const float4 bmin = np[0], bmax = np[1];
        const uint4 w = *reinterpret_cast<const uint4*>(np + 2);
        if (!(w.x & LEAF_BIT)) {
            if (slab_hit(top ? o : lo, top ? d : ld, bmin, bmax)) {
                stack[sp * stride] = tag | w.y;      // right child waits
                ++sp;
                cur = tag | w.x;                     // left child next
                continue;
            }
        } else if (top) {
            const uint32_t count = w.x & 0x7fffffffu;
            if (w.z == TYPE_INST && count > 0) {
                for (uint32_t i = count - 1; i >= 1; --i) { stack[sp * stride] = TAG_INST | (w.y + i); ++sp; }
                cur = TAG_INST | w.y;
                continue;
            }
        } else if (w.z == TYPE_TRIG) {
            const uint32_t count = w.x & 0x7fffffffu;
            for (uint32_t i = 0; i < count; ++i) {
                if (COUNT) ++r.nTri;
                const float4* tp = reinterpret_cast<const float4*>(A.tris + (w.y + i));
                const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                // Möller–Trumbore, radiance.cl:211-251
                const f3 e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
                const f3 rce2 = cross3(ld, e2);
                const float det = dot3(e1, rce2);
                if (det == 0) continue;
                const float inv_det = 1.0f / det;
                const f3 s = lo - mk3(q0.x, q0.y, q0.z);
                const float b1 = inv_det * dot3(s, rce2);
                const f3 sce1 = cross3(s, e1);
                const float b2 = inv_det * dot3(ld, sce1);
                const float t = inv_det * dot3(e2, sce1);
                if (b1 < 0 || b1 > 1) continue;
                if (b2 < 0 || b1 + b2 > 1) continue;
                if (!(t > 0)) continue;
                if (t < r.t && t > tmin && t < tmax) {           // radiance.cl:90-91
                    r.t = t; r.b1 = b1; r.b2 = b2; r.prim = f2u(q0.w); r.inst = curInst; r.hit = true;
                    bool cont = true;
                    callAnyHit(cont, (int)A.insts[curInst].SBTOffset + REC);
                    if (!cont) return;
                }
            }
        }
        if (sp == 0) break;
        --sp;
        cur = stack[sp * stride];
    }
}


// ---------------------------------------------------------------------------------------------
// production traversal ("wide" layout)
// ---------------------------------------------------------------------------------------------
// Same visit set and same arithmetic per test as the reference-order walk above, reorganised for the
// machine:
//   * DWide nodes: one 64-byte fetch carries both children's boxes; leaf children are embedded in
//     the parent, so a leaf costs no node fetch and the dependent-load chain per ray halves.
//   * slab test: (b - o) * (1/d) with a conservative error band; only a test that lands inside the
//     band (|tFar - max(tNear,0)| within 8 ulp-ish) is redone with the reference's IEEE divisions, so
//     the DECISION is always the reference's (see DESIGN.md "fast slab test" for the bound).
//   * visiting order inside a BLAS is free: a closest-hit candidate wins on (t, triangle slot), which
//     is exactly "first strictly-smaller t in DFS order" because triangle slots are numbered in DFS
//     leaf order (bvh.cpp:487-497); any-hit rays only report whether a candidate exists.
struct RayInst {      // per ray, per entered instance
    f3 o, d, rcp;
    bool exactOnly;   // a direction component is (nearly) zero: keep to the division form
};

__device__ __forceinline__ bool slab_fast(const RayInst& R, f3 bmin, f3 bmax)
{
#ifdef RDX_V_EXACTSLAB
    { float4 a = make_float4(bmin.x, bmin.y, bmin.z, 0.f), b = make_float4(bmax.x, bmax.y, bmax.z, 0.f); return slab_hit(R.o, R.d, a, b); }
#endif
#ifdef RDX_V_NOBAND
    { const f3 tA = (bmin - R.o) * R.rcp, tB = (bmax - R.o) * R.rcp;
      const float tn = fmaxf(fmaxf(fminf(tA.x, tB.x), fminf(tA.y, tB.y)), fminf(tA.z, tB.z));
      const float tf = fminf(fminf(fmaxf(tA.x, tB.x), fmaxf(tA.y, tB.y)), fmaxf(tA.z, tB.z));
      return tf > fmaxf(tn, 0.0f); }
#endif
    if (R.exactOnly) {
        float4 a = make_float4(bmin.x, bmin.y, bmin.z, 0.f), b = make_float4(bmax.x, bmax.y, bmax.z, 0.f);
        return slab_hit(R.o, R.d, a, b);
    }
    const f3 tA = (bmin - R.o) * R.rcp, tB = (bmax - R.o) * R.rcp;
    const float tNear = fmaxf(fmaxf(fminf(tA.x, tB.x), fminf(tA.y, tB.y)), fminf(tA.z, tB.z));
    const float tFar = fminf(fminf(fmaxf(tA.x, tB.x), fmaxf(tA.y, tB.y)), fmaxf(tA.z, tB.z));
    const float n0 = fmaxf(tNear, 0.0f);
    const float band = 4.8e-7f * (fabsf(tFar) + n0) + 1e-30f;
    const float diff = tFar - n0;
    if (diff > band) return true;
    if (diff < -band) return false;
    float4 a = make_float4(bmin.x, bmin.y, bmin.z, 0.f), b = make_float4(bmax.x, bmax.y, bmax.z, 0.f);
    return slab_hit(R.o, R.d, a, b);        // inside the band (or non-finite): the reference's own form decides
}

// ---- culled walk (pool engine, option "cull") --------------------------------------------------------------------------
// The reference's walk is exhaustive: it never compares a box with the best t found so far and never tests the box of a
// leaf (radiance.cl:41-108,195-208).  Its RESULT, though, is the minimum of (t, instance slot, triangle slot) over the
// triangles it tests and ACCEPTS (traverse_coop.h), and a triangle the reference's Moeller-Trumbore cannot accept -- or can
// accept only with a t that does not beat the current best -- never changes that minimum.  So a closest-hit ray may skip a
// subtree whose box it enters beyond best_t, and any ray a leaf whose box it misses -- IF the fp32 intersection test is known
// not to accept a triangle the exact ray stays clear of.  That is a statement about conditioning (DESIGN.md 4.1c has the proof):
//
//   if the reference accepts triangle T = (v0, e1, e2) for the ray (o, d) with the computed t~, then the point o + t~ d lies within
//       R = 24 u |o - v0| / kappa + 3 u (t~ |d| + |e1| + |e2|),     kappa = |det~| / (|d| |e1| |e2|),  u = 2^-24
//   of T.  kappa is, up to 5 u, sin(angle(e1, e2)) * |cos(angle(d, normal))|: small for slivers and for rays (nearly) IN the
//   triangle's plane, where det is rounding noise and the reference accepts or rejects at random (tests/test_gpu_cull.py has
//   such rays: a margin alone, however wide, is wrong for them).
//
// Hence the gate: every child descriptor of a wide node carries the NORMAL CONE of the triangles of that leaf / below that inner
// child (rdx_types.h, derive_accel), and the child may be skipped only by rays that provably have kappa >= 2^-7 against all of
// them (`safe`); then R <= 1.86e-4 * S with S >= |o - v| for every vertex in the child's box, and the skips keep a margin that covers R:
//   leaf      skipped iff the slabs miss each other by more than max(2^-8 (|tFar| + n0), mabs),  mabs = 4e-4 * S1 * max|1/d_k|
//   subtree   skipped iff it is entered beyond max(best_t (1 + 2^-8), best_t + mabs)              (S1 = L1 bound of S from the boxes)
// (2^-8: what round 2 used alone; it still covers the slab arithmetic's own 2^-22.)  A ray that fails a child's gate treats that
// child as the exhaustive walk does.  Verified on top: tests/test_gpu_cull.py, tests/test_gpu_reference.py (frames of configs 1-4 bit-identical).
#define RDX_CULL_K 1.00390625f            // 1 + 2^-8
#define RDX_CULL_M 0.00390625f            // 2^-8
#define RDX_CULL_ABS 4.0e-4f              // >= 2.01 * (24 / kappa0 + 15.1) * 1.01 * u * 1.05 with kappa0 = 2^-7 (3.93e-4)

// per-item constants of the gate (the ray's, not the child's)
struct CullRay {
    float sd;         // 127.5 (dx + dy + dz): the bias of the cone's axis bytes, times the direction
    float dlen;       // |d|
    float cabs;       // RDX_CULL_ABS * max |1 / d_k|
};
__device__ __forceinline__ CullRay cull_ray(const RayInst& R)
{
    CullRay C;
    C.sd = (R.d.x + R.d.y + R.d.z) * 127.5f;
    C.dlen = sqrtf(fmaf(R.d.x, R.d.x, fmaf(R.d.y, R.d.y, R.d.z * R.d.z)));
    C.cabs = RDX_CULL_ABS * fmaxf(fmaxf(fabsf(R.rcp.x), fabsf(R.rcp.y)), fabsf(R.rcp.z));
    return C;
}
struct CullGate {
    bool safe;        // the ray is well conditioned against every triangle of the leaf / below the inner child
    float mabs;       // absolute margin (in t) that covers the intersection test's error there
};
// (d0, d1): the child's descriptor with its normal cone (rdx_types.h); bmin / bmax: the child's box, which holds every vertex below it
__device__ __forceinline__ CullGate cull_gate(const RayInst& R, const CullRay& C, uint32_t d0, uint32_t d1, bool isLeaf, f3 bmin, f3 bmax)
{
    CullGate G;
#ifdef RDX_CULL_UNGATED          // timing experiment only (round 2's walk: the margin alone, wrong for rays in a triangle's plane)
    G.safe = true; G.mabs = 0.0f; return G;
#endif
    const float bz = (float)(d1 & 0xffu), bx = (float)((d1 >> 8) & 0xffu), by = (float)((d1 >> 16) & 0xffu);
    const float T = (float)(isLeaf ? (d0 >> WIDE_SLOT_BITS) : ((d1 >> 24) & 0x7fu));
    const float dp = fmaf(R.d.x, bx, fmaf(R.d.y, by, fmaf(R.d.z, bz, -C.sd)));
    G.safe = (T < 127.0f) & (fabsf(dp) >= T * C.dlen);
    const float s1 = fmaxf(fabsf(bmin.x - R.o.x), fabsf(bmax.x - R.o.x)) + fmaxf(fabsf(bmin.y - R.o.y), fabsf(bmax.y - R.o.y)) +
                     fmaxf(fabsf(bmin.z - R.o.z), fabsf(bmax.z - R.o.z));
    G.mabs = C.cabs * s1;
    return G;
}
// entry distance beyond which a child with gate G cannot matter, given the ray's bound `tlim` (best t so far, or tmax)
__device__ __forceinline__ float cull_limit(const CullGate& G, float tlim)
{
    return G.safe ? fmaxf(tlim * RDX_CULL_K, tlim + G.mabs) : __builtin_inff();
}

// slab_fast that also reports the entry distance max(tNear, 0) its decision was made with
__device__ __forceinline__ bool slab_fast_t(const RayInst& R, f3 bmin, f3 bmax, float& tn)
{
    if (!R.exactOnly) {
        const f3 tA = (bmin - R.o) * R.rcp, tB = (bmax - R.o) * R.rcp;
        const float tNear = fmaxf(fmaxf(fminf(tA.x, tB.x), fminf(tA.y, tB.y)), fminf(tA.z, tB.z));
        const float tFar = fminf(fminf(fmaxf(tA.x, tB.x), fmaxf(tA.y, tB.y)), fmaxf(tA.z, tB.z));
        const float n0 = fmaxf(tNear, 0.0f);
        const float band = 4.8e-7f * (fabsf(tFar) + n0) + 1e-30f;
        const float diff = tFar - n0;
        tn = n0;
        if (diff > band) return true;
        if (diff < -band) return false;
    }
    // inside the band, or a (nearly) zero direction component: the reference's own form decides (radiance.cl:195-208)
    const f3 tA = (bmin - R.o) / R.d, tB = (bmax - R.o) / R.d;
    const f3 t1 = mk3(cl_min(tA.x, tB.x), cl_min(tA.y, tB.y), cl_min(tA.z, tB.z));
    const f3 t2 = mk3(cl_max(tA.x, tB.x), cl_max(tA.y, tB.y), cl_max(tA.z, tB.z));
    const float tNear = cl_max(cl_max(t1.x, t1.y), t1.z);
    const float tFar = cl_min(cl_min(t2.x, t2.y), t2.z);
    tn = cl_max(tNear, 0.0f);
    return tFar > tn;
}

// One child of a wide node in the culled walk, leaf or inner, in one straight-line evaluation of the slabs (the two kinds
// differ in the margin and in what "undecided" means, not in the arithmetic -- and a wave's 64 items are a mix of both):
//   leaf  -> may the ray touch a triangle inside this box at a distance that still matters?  `false` only when the gate is
//            open and the slabs miss each other by more than the margin, or the box is entered beyond cullT
//   inner -> the reference's own decision (slab_fast_t) && !(tn > cullT)
// tlim = best t so far or tmax; tn is the entry distance (inner children only).  Rays with a (nearly) zero direction component
// never skip a leaf.
__device__ __forceinline__ bool cull_child(const RayInst& R, f3 bmin, f3 bmax, bool isLeaf, const CullGate& G, float tlim, float& tn)
{
    const f3 tA = (bmin - R.o) * R.rcp, tB = (bmax - R.o) * R.rcp;
    const float tNear = fmaxf(fmaxf(fminf(tA.x, tB.x), fminf(tA.y, tB.y)), fminf(tA.z, tB.z));
    const float tFar = fminf(fminf(fmaxf(tA.x, tB.x), fmaxf(tA.y, tB.y)), fmaxf(tA.z, tB.z));
    const float n0 = fmaxf(tNear, 0.0f);
    const float rel = (isLeaf ? RDX_CULL_M : 4.8e-7f) * (fabsf(tFar) + n0) + 1e-30f;
    const float m = isLeaf ? fmaxf(rel, G.mabs) : rel;
    const float diff = tFar - n0;
    const bool hit = diff > m, miss = (-diff > m) & (G.safe | !isLeaf);
    bool res = isLeaf ? !miss : hit;
    tn = n0;
    if (!isLeaf && (R.exactOnly || !(hit || miss))) {
        // inside the band, or a (nearly) zero direction component: the reference's own form decides (radiance.cl:195-208)
        const f3 dA = (bmin - R.o) / R.d, dB = (bmax - R.o) / R.d;
        const f3 t1 = mk3(cl_min(dA.x, dB.x), cl_min(dA.y, dB.y), cl_min(dA.z, dB.z));
        const f3 t2 = mk3(cl_max(dA.x, dB.x), cl_max(dA.y, dB.y), cl_max(dA.z, dB.z));
        const float eNear = cl_max(cl_max(t1.x, t1.y), t1.z);
        const float eFar = cl_min(cl_min(t2.x, t2.y), t2.z);
        tn = cl_max(eNear, 0.0f);
        res = eFar > tn;
    }
    if (isLeaf && R.exactOnly) return true;
    return res && !(tn > cull_limit(G, tlim));
}

struct Best {
    float t, b1, b2;
    uint32_t slot, inst;
    bool hit;
};

// Möller–Trumbore on triangle slots [start, start+count); returns true if the walk must stop (any-hit)
template <int REC>
__device__ __forceinline__ bool leaf_tris(const AccelView& A, const RayInst& R, uint32_t start, uint32_t count,
                                          uint32_t curInst, float tmin, float tmax, Best& B)
{
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t slot = start + i;
        const float4* tp = reinterpret_cast<const float4*>(A.tris + slot);
        const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
        const f3 e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
        const f3 rce2 = cross3(R.d, e2);
        const float det = dot3(e1, rce2);
        if (det == 0) continue;
        const float inv_det = 1.0f / det;
        const f3 s = R.o - mk3(q0.x, q0.y, q0.z);
        const float b1 = inv_det * dot3(s, rce2);
        const f3 sce1 = cross3(s, e1);
        const float b2 = inv_det * dot3(R.d, sce1);
        const float t = inv_det * dot3(e2, sce1);
        if (b1 < 0 || b1 > 1) continue;
        if (b2 < 0 || b1 + b2 > 1) continue;
        if (!(t > 0)) continue;
        if (!(t > tmin && t < tmax)) continue;
        // reference: accept iff t < best in DFS order  <=>  (t, slot) lexicographically smaller
        const bool better = (t < B.t) || (t == B.t && B.inst == curInst && slot < B.slot);
        if (better) {
            B.t = t; B.b1 = b1; B.b2 = b2; B.slot = slot; B.inst = curInst; B.hit = true;
            bool cont = true;
            callAnyHit(cont, (int)A.insts[curInst].SBTOffset + REC);
            if (!cont) return true;
        }
    }
    return false;
}

// stack / work-item encoding of a run of triangle slots (a leaf, or an 8-triangle piece of a big one)
__device__ __forceinline__ uint32_t leaf_item(uint32_t start, uint32_t count) { return TAG_LEAF | ((count - 1u) << LEAF_START_BITS) | start; }

template <int REC>
__device__ __forceinline__ void traverse_wide(const AccelView& A, f3 o, f3 d, float tmin, float tmax,
                                              uint32_t* __restrict__ stack, uint32_t stride, Best& B)
{
    B.t = FLT_MAX; B.b1 = 0.f; B.b2 = 0.f; B.slot = 0; B.inst = RDX_MISS; B.hit = false;
    uint32_t sp = 0;
    uint32_t cur = TAG_TLAS | 0u;            // work item: TLAS node, instance entry, wide BLAS node or triangle run
    RayInst R;                               // object-space ray of the instance being walked
    R.o = o; R.d = d; R.rcp = mk3(0.f, 0.f, 0.f); R.exactOnly = true;
    uint32_t curInst = 0;
    // One flat loop, one homogeneous work item per lane per iteration (a node: two box tests; a
    // triangle run: up to 8 Möller–Trumbore tests), so lanes of a wave stay in the same code.
    for (;;) {
        const uint32_t tag = cur & TAG_MASK;
        if (tag == TAG_LEAF) {
            const uint32_t start = cur & LEAF_START_MASK, count = ((cur >> LEAF_START_BITS) & 7u) + 1u;
            if (leaf_tris<REC>(A, R, start, count, curInst, tmin, tmax, B)) return;
        } else if (tag == TAG_BLAS) {
            const float4* wp = reinterpret_cast<const float4*>(A.wide + (cur & IDX_MASK));
            const float4 l0 = wp[0], l1 = wp[1], r0 = wp[2], r1 = wp[3];
            uint32_t item0 = RDX_MISS, item1 = RDX_MISS;        // RDX_MISS = 0xffffffff is never a valid item
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float4 bmin = c ? r0 : l0, bmax = c ? r1 : l1;
                const uint32_t d0 = f2u(bmin.w), d1 = f2u(bmax.w);
                uint32_t item = RDX_MISS;
                if (d1 & WIDE_LEAF) {
                    uint32_t cnt = wide_count(d1), st = wide_slot(d0);
                    while (cnt > 8u) { stack[sp * stride] = leaf_item(st, 8u); ++sp; st += 8u; cnt -= 8u; }   // rare: oversized leaf
                    if (cnt) item = leaf_item(st, cnt);
                } else if (slab_fast(R, mk3(bmin.x, bmin.y, bmin.z), mk3(bmax.x, bmax.y, bmax.z))) {
                    item = d0;
                }
                if (c == 0) item0 = item; else item1 = item;
            }
            if (item0 != RDX_MISS) {
                if (item1 != RDX_MISS) { stack[sp * stride] = item1; ++sp; }
                cur = item0;
                continue;
            }
            if (item1 != RDX_MISS) { cur = item1; continue; }
        } else if (tag == TAG_TLAS) {
            const float4* np = reinterpret_cast<const float4*>(A.tnodes + (cur & IDX_MASK));
            const float4 bmin = np[0], bmax = np[1];
            const uint4 w = *reinterpret_cast<const uint4*>(np + 2);
            if (!(w.x & LEAF_BIT)) {
                if (slab_hit(o, d, bmin, bmax)) {
                    stack[sp * stride] = TAG_TLAS | w.y; ++sp;
                    cur = TAG_TLAS | w.x;
                    continue;
                }
            } else {
                const uint32_t count = w.x & 0x7fffffffu;
                if (w.z == TYPE_INST && count > 0) {
                    for (uint32_t i = count - 1; i >= 1; --i) { stack[sp * stride] = TAG_INST | (w.y + i); ++sp; }
                    cur = TAG_INST | w.y;
                    continue;
                }
            }
        } else {   // TAG_INST: enter the instance (radiance.cl:161-169)
            curInst = cur & IDX_MASK;
            const float4* ip = reinterpret_cast<const float4*>(A.insts + curInst);
            float m[16];
            *reinterpret_cast<float4*>(m + 0) = ip[0];
            *reinterpret_cast<float4*>(m + 4) = ip[1];
            *reinterpret_cast<float4*>(m + 8) = ip[2];
            *reinterpret_cast<float4*>(m + 12) = ip[3];
            R.o = mat4_mul3(m, o.x, o.y, o.z, 1.0f);
            R.d = mat4_mul3(m, d.x, d.y, d.z, 0.0f);
            R.rcp = mk3(1.0f / R.d.x, 1.0f / R.d.y, 1.0f / R.d.z);
            const float amin = fminf(fminf(fabsf(R.d.x), fabsf(R.d.y)), fabsf(R.d.z));
            const float amax = fmaxf(fmaxf(fabsf(R.rcp.x), fabsf(R.rcp.y)), fabsf(R.rcp.z));
            R.exactOnly = !(amin > 1e-20f) || !(amax < 1e20f);
            const uint4 rd = *reinterpret_cast<const uint4*>(ip + 9);
            if (rd.y & WIDE_LEAF) {
                uint32_t cnt = wide_count(rd.y), st = wide_slot(rd.x);
                while (cnt > 8u) { stack[sp * stride] = leaf_item(st, 8u); ++sp; st += 8u; cnt -= 8u; }
                if (cnt) { cur = leaf_item(st, cnt); continue; }
            } else {
                const float4 rmin = ip[10], rmax = ip[11];
                if (slab_fast(R, mk3(rmin.x, rmin.y, rmin.z), mk3(rmax.x, rmax.y, rmax.z))) { cur = rd.x; continue; }
            }
        }
        if (sp == 0) return;
        --sp;
        cur = stack[sp * stride];
    }
}

} // namespace rdx
#include "traverse_coop.h"
#include "traverse_pool.h"
namespace rdx {

extern __shared__ uint32_t s_stack[];

// COUNT builds: add this lane's visit counts to visit[cls*4 + {top, inst, bot, tri}]
__device__ __forceinline__ void flush_visits(unsigned long long* visit, int cls, const TraceResult& r)
{
    atomicAdd(visit + cls * 4 + 0, (unsigned long long)r.nTop);
    atomicAdd(visit + cls * 4 + 1, (unsigned long long)r.nInst);
    atomicAdd(visit + cls * 4 + 2, (unsigned long long)r.nBot);
    atomicAdd(visit + cls * 4 + 3, (unsigned long long)r.nTri);
}

// ---------------------------------------------------------------------------------------------
// generate: primary rays (samples/shader.cl:111-173, 196-231)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void camera_ray(const CameraArgs& C, uint32_t pixel, f3 rnd, f3& org, f3& dir)
{
    const PhysicalCamera& cam = C.cam;
    const int index = (int)pixel;
    const int x = index % (int)cam.widthPixel;
    const int y = index / (int)cam.widthPixel;
    const float fx = (((float)x + rnd.x) / cam.widthPixel) - 0.5f;
    const float fy = 0.5f - (((float)y + rnd.y) / cam.heightPixel);
    const float aspect = cam.heightPixel / cam.widthPixel;
    f4 pd; pd.x = fx * cam.sensorWidth; pd.y = fy * cam.sensorWidth * aspect; pd.z = -cam.focalLength; pd.w = 0.0f;
    pd = normalize4(pd);
    const f3 eye = mk3(cam.x, cam.y, cam.z);
    const float time = -cam.focalDistance / pd.z;
    f4 t = mat4_mul(C.rotZ, pd.x, pd.y, pd.z, pd.w);
    pd = mat4_mul(C.rotY, t.x, t.y, t.z, t.w);
    t = mat4_mul(C.rotX, pd.x, pd.y, pd.z, pd.w);
    pd = normalize4(t);
    if (cam.fStop == 0.0f) { org = eye; dir = mk3(pd.x, pd.y, pd.z); return; }

    // thin lens: concentric disk sample from rnd.yz (shader.cl:89-109, 155-172)
    const float lensRadius = (cam.focalLength / cam.fStop) / 2.0f;
    float ux = 2.0f * rnd.y - 1.0f, uy = 2.0f * rnd.z - 1.0f;
    float lx = 0.0f, ly = 0.0f;
    if (!(ux == 0.0f && uy == 0.0f)) {
        float theta, rr;
        if (fabsf(ux) > fabsf(uy)) { rr = ux; theta = (RDX_PI / 4.0f) * (uy / ux); }
        else { rr = uy; theta = (RDX_PI / 2.0f) - (RDX_PI / 4.0f) * (ux / uy); }
        lx = rr * cosf(theta); ly = rr * sinf(theta);
    }
    lx = lensRadius * lx; ly = lensRadius * ly;
    const f3 focus = eye + mk3(pd.x, pd.y, pd.z) * time;
    f4 l = mat4_mul(C.rotZ, lx, ly, 0.0f, 1.0f);
    f4 l2 = mat4_mul(C.rotY, l.x, l.y, l.z, l.w);
    l = mat4_mul(C.rotX, l2.x, l2.y, l2.z, l2.w);
    org = eye + mk3(l.x, l.y, l.z);
    dir = normalize3(focus - org);
}

// EulerX/Y/ZToMat4x4 (math.cl:185-252) take cos / sin of the camera angles per ray; they are per-frame constants, so
// one thread evaluates them once -- on the device, with the OCML functions the reference's cos() / sin() link
__global__ void k_euler_trig(float wx, float wy, float wz, float* __restrict__ out6)
{
    out6[0] = cosf(wx); out6[1] = sinf(wx); out6[2] = cosf(wy); out6[3] = sinf(wy); out6[4] = cosf(wz); out6[5] = sinf(wz);
}
void launch_euler_trig(hipStream_t st, float wx, float wy, float wz, float* out6)
{
    hipLaunchKernelGGL(k_euler_trig, dim3(1), dim3(1), 0, st, wx, wy, wz, out6);
}

__global__ void __launch_bounds__(RDX_BLOCK)
k_generate(CameraArgs C, PathStreams ps, const uint32_t* __restrict__ owned, uint32_t nPixels, uint32_t sampleBegin,
           uint32_t sampleCount, uint32_t totalSamples)
{
    const uint32_t i = blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (i >= nPixels * sampleCount) return;
    const uint32_t sLocal = i / nPixels, slot = i - sLocal * nPixels;
    const uint32_t pixel = owned ? owned[slot] : slot;
    const uint32_t frameID = totalSamples + sampleBegin + sLocal;
    const f3 rnd = pcg3d(frameID, totalSamples, pixel);      // shader.cl:205
    f3 o, d;
    camera_ray(C, pixel, rnd, o, d);
    ps.rayO[i] = make_float4(o.x, o.y, o.z, u2f(pixel));
    ps.rayD[i] = make_float4(d.x, d.y, d.z, u2f(frameID));
    ps.thr[i] = make_float4(1.0f, 1.0f, 1.0f, u2f(slot));
    ps.col[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---------------------------------------------------------------------------------------------
// extend: closest hit for every live path
// ---------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(RDX_BLOCK)
k_extend(AccelView A, PathStreams ps, const uint32_t* __restrict__ nPtr, float tmin, float tmax,
         unsigned long long* __restrict__ visit)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nPtr) return;
    const float4 ro = ps.rayO[i], rd = ps.rayD[i];
    if (COUNT) {       // reference-order walk: visit counters of the reference algorithm
        TraceResult r;
        traverse<1, true>(A, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), tmin, tmax, s_stack + threadIdx.x, blockDim.x, r);
        ps.hitA[i] = make_float4(r.t, r.b1, r.b2, u2f(r.prim));
        ps.hitInst[i] = r.hit ? r.inst : RDX_MISS;
        if (visit) flush_visits(visit, 0, r);
    } else {
        Best b;
        traverse_wide<1>(A, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), tmin, tmax, s_stack + threadIdx.x, blockDim.x, b);
        ps.hitA[i] = make_float4(b.t, b.b1, b.b2, b.hit ? u2f(A.tris[b.slot].primID) : 0.0f);
        ps.hitInst[i] = b.hit ? b.inst : RDX_MISS;
    }
}

// ---------------------------------------------------------------------------------------------
// shade: SBT dispatch + ballot compaction
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_sample(const PathStreams& ps, uint32_t nPixels, uint32_t sampleBase,
                                             uint32_t frameID, uint32_t slot, f3 c)
{
    ps.sampleColor[(size_t)(frameID - sampleBase) * nPixels + slot] = make_float4(c.x, c.y, c.z, 0.0f);
}

__device__ __forceinline__ void fill_hit_info(const AccelView& A, uint32_t instSlot, f3 o, f3 d, float t, float b1,
                                              float b2, uint32_t prim, HitInfo& h)
{
    const DInst& I = A.insts[instSlot];
    // hitPoint = localOrigin + localDir * t, as computed at accept time (radiance.cl:243)
    const f3 lo = mat4_mul3(I.inv, o.x, o.y, o.z, 1.0f);
    const f3 ld = mat4_mul3(I.inv, d.x, d.y, d.z, 0.0f);
    h.hitPoint = lo + ld * t;
    h.bx = 1 - b1 - b2; h.by = b1; h.bz = b2;
    h.primitiveIndex = prim;
    h.instanceIndex = I.instanceID;
    h.fwd = I.fwd;
}

__device__ __forceinline__ uint32_t sort_key_of(const SortBox& B, f3 o, f3 d);      // (per-bounce ray sort, below)
#ifndef RDX_SHADE_WAVES
#define RDX_SHADE_WAVES 1
#endif
#ifndef RDX_SHADE_BLOCK
#define RDX_SHADE_BLOCK 256
#endif
__global__ void __launch_bounds__(RDX_SHADE_BLOCK, RDX_SHADE_WAVES)
k_shade(AccelView A, SceneArgs sc, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t* __restrict__ nOut,
        uint32_t depth, uint32_t maxDepth, uint32_t nPixels, uint32_t sampleBase, SortBox sortBox)
{
    const uint32_t i = blockIdx.x * RDX_SHADE_BLOCK + threadIdx.x;
    const bool active = i < *nPtr;
    bool alive = false;
    Payload p;
    float4 ro, rd, thr, col;
    if (active) {
        ro = ps.rayO[i]; rd = ps.rayD[i]; thr = ps.thr[i]; col = ps.col[i];
        const uint32_t instSlot = ps.hitInst[i];
        const uint32_t pixel = f2u(ro.w), frameID = f2u(rd.w), slot = f2u(thr.w);
        p.hit = false; p.wantsShadowRay = false;
        p.color = mk3(0.f, 0.f, 0.f); p.colorOccluded = p.color;
        p.nextFactor = mk3(1.f, 1.f, 1.f);
        p.nextRayOrigin = mk3(ro.x, ro.y, ro.z); p.nextRayDirection = mk3(rd.x, rd.y, rd.z);
        p.shadowOrigin = p.nextRayOrigin;
        const SceneView sv{sc.scene, sc.meshInfo, sc.indexData, sc.uvData, sc.normalData, sc.materials, sc.tex};
        if (instSlot != RDX_MISS) {
            const float4 ha = ps.hitA[i];
            HitInfo h;
            fill_hit_info(A, instSlot, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ha.x, ha.y, ha.z, f2u(ha.w), h);
            callHit((int)A.insts[instSlot].SBTOffset + 1, p, h, sv, mk3(rd.x, rd.y, rd.z), frameID, pixel, depth,
                    depth + 1 < maxDepth);
        } else {
            callMiss(3, p);
        }
        if (p.hit) {
            alive = true;
        } else {
            // shader.cl:243-252: a primary miss shows the miss colour, a later miss ends the path
            f3 c = (depth == 0) ? p.color : mk3(col.x, col.y, col.z);
            store_sample(ps, nPixels, sampleBase, frameID, slot, c);
        }
    }
    // wave64 ballot compaction of the surviving paths into the B streams.  The output cursor is ONE device word:
    // an atomic per wave (130 k of them on one address for a 1080p x 4 spp bounce) serialises in the L2 and was
    // the whole cost of this kernel (1.5 ms whatever the shader did); the block's waves are therefore summed in
    // LDS first and one lane per block moves the cursor.
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long ltMask = (1ull << lane) - 1ull;
    __shared__ uint32_t s_cnt[RDX_SHADE_BLOCK / 64], s_base;
    const unsigned long long m = __ballot(alive);
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < RDX_SHADE_BLOCK / 64; ++w) tot += s_cnt[w];
        s_base = tot ? atomicAdd(nOut, tot) : 0u;
    }
    __syncthreads();
    if (!alive) return;
    uint32_t base = s_base;
    for (uint32_t w = 0; w < wave; ++w) base += s_cnt[w];
    const uint32_t j = base + (uint32_t)__popcll(m & ltMask);
    const f3 T = mk3(thr.x, thr.y, thr.z), Cc = mk3(col.x, col.y, col.z);
    const f3 lit = Cc + T * p.color;               // color += contribution * payload.color (shader.cl:240)
    const f3 occ = Cc + T * p.colorOccluded;
    const f3 tn = T * p.nextFactor;                // contribution *= payload.nextFactor    (shader.cl:241)
    ps.shO[j] = make_float4(p.shadowOrigin.x, p.shadowOrigin.y, p.shadowOrigin.z, p.wantsShadowRay ? 1.0f : 0.0f);
    ps.nRayO[j] = make_float4(p.nextRayOrigin.x, p.nextRayOrigin.y, p.nextRayOrigin.z, ro.w);
    ps.nRayD[j] = make_float4(p.nextRayDirection.x, p.nextRayDirection.y, p.nextRayDirection.z, rd.w);
    ps.nThr[j] = make_float4(tn.x, tn.y, tn.z, thr.w);
    ps.colLit[j] = make_float4(lit.x, lit.y, lit.z, 0.0f);
    ps.colSh[j] = make_float4(occ.x, occ.y, occ.z, 0.0f);
    if (ps.sortKey) ps.sortKey[j] = (unsigned short)sort_key_of(sortBox, p.nextRayOrigin, p.nextRayDirection);
}

// ---------------------------------------------------------------------------------------------
// per-bounce ray sort (option "sort"): see kernels.h
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread4(uint32_t v) { return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6); }
__device__ __forceinline__ uint32_t sort_cell(const SortBox& B, float x, float y, float z)
{
    const uint32_t cx = (uint32_t)fminf(fmaxf((x - B.lo[0]) * B.inv[0], 0.0f), 15.0f);
    const uint32_t cy = (uint32_t)fminf(fmaxf((y - B.lo[1]) * B.inv[1], 0.0f), 15.0f);
    const uint32_t cz = (uint32_t)fminf(fmaxf((z - B.lo[2]) * B.inv[2], 0.0f), 15.0f);
    return spread4(cx) | (spread4(cy) << 1) | (spread4(cz) << 2);            // 12-bit Morton code
}
// exclusive scan inside tiles of SORT_TILE counters (256 threads x 16 consecutive counters), tile totals to sums[tile]
__global__ void __launch_bounds__(256)
k_sort_scan_tiles(uint32_t* __restrict__ bins, uint32_t* __restrict__ sums)
{
    static_assert(SORT_TILE == 256u * 16u, "tile shape");
    __shared__ uint32_t part[256];
    uint4* p = reinterpret_cast<uint4*>(bins + (size_t)blockIdx.x * SORT_TILE + threadIdx.x * 16u);
    uint4 v[4];
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = p[k]; sum += v[k].x + v[k].y + v[k].z + v[k].w; }
    const uint32_t t = threadIdx.x;
    part[t] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 256u; off <<= 1) {
        const uint32_t x = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint4 o;
        o.x = run; run += v[k].x; o.y = run; run += v[k].y; o.z = run; run += v[k].z; o.w = run; run += v[k].w;
        p[k] = o;
    }
    if (t == 255u) sums[blockIdx.x] = part[255];
}
// Per-bounce ray sort (option "sort" 1), without device-scope atomics.  Counting sort over SORT_GBINS = 4096 keys (the 9 high bits of
// the origin's Morton cell, direction octant; 10 / 11 / 12 key bits: 55.2 / 54.6 / 54.3 ms on the 10.4 M-triangle scene) in the
// classic three steps of a radix pass: (A) every block histograms ITS
// tiles of 4096 consecutive paths in LDS and writes its column of the (key, block) count matrix, (B) one exclusive scan of the
// matrix in key-major order, (C) every block reloads its scanned column into LDS and hands out positions with LDS atomics
// while it walks its tiles again.  Two passes over the rays, 4 MB of counters, no global atomic.  (A tile-LOCAL
// sort -- 0.4 ms per frame -- was tried first and gains nothing: what the sort buys is that the ~400 k rays in flight on the
// chip at any moment come from one region of the scene and meet in L2, not coherence inside a wave.)
#ifndef SORT_GBITS
#define SORT_GBITS 13u                 // key bits: 3 of the octant + the high bits - 3 bits of the 12-bit Morton cell (12 / 13 / 14: 22.15 / 22.10 / 22.31 ms Sponza-class, 64.1 / 63.1 / 62.6 ms on the 10.4 M-triangle scene)
#endif
#ifndef SORT_GCOLS_LOG2
#define SORT_GCOLS_LOG2 8u             // columns of the (key, block) count matrix = blocks of the two passes (1024 / 512 / 256 / 128: 22.38 / 22.13 / 22.06 / 22.48 ms, Sponza-class)
#endif
// The number of key bits is a launch parameter: SORT_GBITS (13) by default, SORT_GBITS_LARGE (14) for the scenes whose
// records live in HBM (62.6 vs 63.1 ms on the 10.4 M-triangle scene; the Sponza-class scene loses with 14: 22.3 vs 22.1).
#ifndef SORT_GBITS_LARGE
#define SORT_GBITS_LARGE 14u
#endif
constexpr uint32_t SORT_GBITS_MAX = SORT_GBITS_LARGE > SORT_GBITS ? SORT_GBITS_LARGE : SORT_GBITS;
constexpr uint32_t SORT_GBINS = 1u << SORT_GBITS_MAX, SORT_GCOLS = 1u << SORT_GCOLS_LOG2, SORT_GTILE = 4096u;
constexpr uint32_t SORT_GSCAN_TILES = SORT_GBINS * SORT_GCOLS / SORT_TILE;              // scan tiles of 4096 counters (at most)
static_assert(SORT_GSCAN_TILES <= 1024u, "one thread per scan tile in k_sortg_scan_sums");
static_assert(SORT_GBITS_MAX <= 14u, "the histogram of a block lives in 64 KB of LDS");
// full-resolution key, 15 bits: 12-bit Morton cell << 3 | direction octant (what the shade stage stores)
__device__ __forceinline__ uint32_t sort_key_of(const SortBox& B, f3 o, f3 d)
{
    return (sort_cell(B, o.x, o.y, o.z) << 3) | ((d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u));
}
// ... reduced to `bits` key bits: the high bits - 3 bits of the cell, the octant
__device__ __forceinline__ uint32_t sortg_key(const PathStreams& ps, const SortBox& B, uint32_t j, uint32_t bits)
{
    uint32_t k;
    if (ps.sortKey) k = ps.sortKey[j];                 // written by the shade stage
    else { const float4 ro = ps.nRayO[j], rd = ps.nRayD[j]; k = sort_key_of(B, mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z)); }
    return (((k >> 3) >> (15u - bits)) << 3) | (k & 7u);
}
__global__ void __launch_bounds__(256)
k_sortg_hist(PathStreams ps, const uint32_t* __restrict__ mPtr, SortBox B, uint32_t* __restrict__ H, uint32_t bits)
{
    __shared__ uint32_t cnt[SORT_GBINS];
    const uint32_t m = *mPtr, t = threadIdx.x, cols = gridDim.x, bins = 1u << bits;
    for (uint32_t k = t; k < bins; k += 256u) cnt[k] = 0u;
    __syncthreads();
    for (uint32_t base = blockIdx.x * SORT_GTILE; base < m; base += cols * SORT_GTILE)
        for (uint32_t i = 0; i < 16u; ++i) {
            const uint32_t j = base + i * 256u + t;
            if (j < m) atomicAdd(&cnt[sortg_key(ps, B, j, bits)], 1u);
        }
    __syncthreads();
    for (uint32_t k = t; k < bins; k += 256u) H[k * SORT_GCOLS + blockIdx.x] = cnt[k];       // (columns >= gridDim.x stay zero)
}
__global__ void __launch_bounds__(1024)
k_sortg_scan_sums(uint32_t* __restrict__ sums, uint32_t nTiles)
{
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t v = t < nTiles ? sums[t] : 0u;
    part[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {
        const uint32_t x = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += x;
        __syncthreads();
    }
    if (t < nTiles) sums[t] = part[t] - v;
}
__global__ void __launch_bounds__(256)
k_sortg_scatter(PathStreams ps, const uint32_t* __restrict__ mPtr, SortBox B, const uint32_t* __restrict__ H, const uint32_t* __restrict__ sums,
                uint32_t* __restrict__ perm, uint32_t bits)
{
    __shared__ uint32_t cnt[SORT_GBINS];
    const uint32_t m = *mPtr, t = threadIdx.x, cols = gridDim.x, bins = 1u << bits;
    for (uint32_t k = t; k < bins; k += 256u) {
        const uint32_t at = k * SORT_GCOLS + blockIdx.x;
        cnt[k] = H[at] + sums[at / SORT_TILE];
    }
    __syncthreads();
    for (uint32_t base = blockIdx.x * SORT_GTILE; base < m; base += cols * SORT_GTILE)
        for (uint32_t i = 0; i < 16u; ++i) {
            const uint32_t j = base + i * 256u + t;
            if (j < m) perm[atomicAdd(&cnt[sortg_key(ps, B, j, bits)], 1u)] = j;      // (the order inside a key is arbitrary: no result depends on it)
        }
}

// ---------------------------------------------------------------------------------------------
// shadow: any-hit walk of the deferred shadow query, then hand over to the next bounce
// ---------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ void __launch_bounds__(RDX_BLOCK)
k_shadow(AccelView A, SceneArgs sc, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t lastBounce,
         uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax, unsigned long long* __restrict__ visit)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *nPtr) return;
    const float4 so = ps.shO[i];
    bool occluded = false;
    if (so.w != 0.0f) {
        const float* ld = sc.scene->lights[0].direction;
        const f3 L = normalize3(mk3(-ld[0], -ld[1], -ld[2]));     // shader.cl:471-476
        bool anyHit; uint32_t hitInst;
        if (COUNT) {
            TraceResult r;
            traverse<2, true>(A, mk3(so.x, so.y, so.z), L, tmin, tmax, s_stack + threadIdx.x, blockDim.x, r);
            if (visit) flush_visits(visit, 1, r);
            anyHit = r.hit; hitInst = r.inst;
        } else {
            Best b;
            traverse_wide<2>(A, mk3(so.x, so.y, so.z), L, tmin, tmax, s_stack + threadIdx.x, blockDim.x, b);
            anyHit = b.hit; hitInst = b.inst;
        }
        // hit -> closest-hit row 2 `shadow` sets payload.hit; miss -> row 4 `shadowMiss` clears it
        Payload sp; sp.hit = false;
        if (anyHit) { HitInfo hh{}; SceneView sv{}; callHit((int)A.insts[hitInst].SBTOffset + 2, sp, hh, sv, L, 0, 0, 0, false); }
        else callMiss(4, sp);
        occluded = sp.hit;
    }
    const float4 c = occluded ? ps.colSh[i] : ps.colLit[i];
    if (lastBounce) {
        store_sample(ps, nPixels, sampleBase, f2u(ps.nRayD[i].w), f2u(ps.nThr[i].w), mk3(c.x, c.y, c.z));
        return;
    }
    ps.nCol[i] = make_float4(c.x, c.y, c.z, 0.0f);
}

// depth == 0 frames: every path ends with colour 0 without tracing
__global__ void __launch_bounds__(RDX_BLOCK)
k_finalize_all(PathStreams ps, uint32_t n, uint32_t nPixels, uint32_t sampleBase)
{
    const uint32_t i = blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 c = ps.col[i];
    store_sample(ps, nPixels, sampleBase, f2u(ps.rayD[i].w), f2u(ps.thr[i].w), mk3(c.x, c.y, c.z));
}

// ---------------------------------------------------------------------------------------------
// accumulate: running mean in sample order + ACES/gamma + RGBA8 (samples/shader.cl:262-304)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float aces1(float v)
{
    v = v * 0.6f;
    return cl_clamp((v * (2.51f * v + 0.03f)) / (v * (2.43f * v + 0.59f) + 0.14f), 0.0f, 1.0f);
}

__global__ void __launch_bounds__(RDX_BLOCK)
k_accumulate(PathStreams ps, const uint32_t* __restrict__ owned, uint32_t nPixels, uint32_t sampleBegin,
             uint32_t sampleCount, uint32_t totalSamples, uint32_t tonemap, uint32_t debug,
             float* __restrict__ scratch, uint8_t* __restrict__ image)
{
    const uint32_t slot = blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (slot >= nPixels) return;
    const uint32_t pixel = owned ? owned[slot] : slot;
    float4* px = reinterpret_cast<float4*>(scratch) + pixel;
    float4 acc = *px;
    for (uint32_t s = 0; s < sampleCount; ++s) {
        const uint32_t frameID = totalSamples + sampleBegin + s;
        const float4 c = ps.sampleColor[(size_t)s * nPixels + slot];
        if (frameID == 0) { acc.x = c.x; acc.y = c.y; acc.z = c.z; }
        else {
            acc.x = (frameID * acc.x + c.x) / (frameID + 1);
            acc.y = (frameID * acc.y + c.y) / (frameID + 1);
            acc.z = (frameID * acc.z + c.z) / (frameID + 1);
        }
    }
    if (sampleCount) *px = acc;
    if (!tonemap) return;
    f3 c = mk3(acc.x, acc.y, acc.z);
    if (!debug) {
        c = mk3(aces1(c.x), aces1(c.y), aces1(c.z));
        c = mk3(powf(c.x, 0.7f), powf(c.y, 0.7f), powf(c.z, 0.7f));
    }
    uchar4 o;
    o.x = (unsigned char)(int)(c.x * 255);
    o.y = (unsigned char)(int)(c.y * 255);
    o.z = (unsigned char)(int)(c.z * 255);
    o.w = 255;
    reinterpret_cast<uchar4*>(image)[pixel] = o;
}

// ---------------------------------------------------------------------------------------------
// tile pack / unpack for the multi-GPU framebuffer gather
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RDX_BLOCK)
k_pack_tiles(const uint8_t* __restrict__ image, uint8_t* __restrict__ packed, uint32_t w, uint32_t h, uint32_t elem,
             uint32_t tileW, uint32_t tileH, uint32_t rank, uint32_t world, uint32_t unpack)
{
    // one thread per (owned pixel, 4-byte word); owned tiles are enumerated in ascending tile id
    const uint32_t tilesX = (w + tileW - 1) / tileW, tilesY = (h + tileH - 1) / tileH;
    const uint32_t nTiles = tilesX * tilesY;
    const uint32_t ownedTiles = nTiles > rank ? (nTiles - rank + world - 1) / world : 0;
    const uint32_t wordsPerPixel = elem / 4;
    const uint64_t total = (uint64_t)ownedTiles * tileW * tileH * wordsPerPixel;
    uint64_t g = (uint64_t)blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (g >= total) return;
    const uint32_t word = (uint32_t)(g % wordsPerPixel);
    const uint64_t pp = g / wordsPerPixel;
    const uint32_t inTile = (uint32_t)(pp % (tileW * tileH));
    const uint32_t k = (uint32_t)(pp / (tileW * tileH));
    const uint32_t tile = rank + k * world;
    const uint32_t x = (tile % tilesX) * tileW + inTile % tileW, y = (tile / tilesX) * tileH + inTile / tileW;
    if (x >= w || y >= h) return;
    const uint32_t* img = reinterpret_cast<const uint32_t*>(image);
    uint32_t* pk = reinterpret_cast<uint32_t*>(packed);
    const uint64_t ii = ((uint64_t)y * w + x) * wordsPerPixel + word;
    if (unpack) const_cast<uint32_t*>(img)[ii] = pk[g];
    else pk[g] = img[ii];
}

__device__ __forceinline__ void write_hit_record(const AccelView& A, const TraceResult& r, f3 ro, f3 rd, rdx_hit& dst)
{
    rdx_hit h;
    h.hit = r.hit ? 1u : 0u;
    h.distance = r.t;
    if (r.hit) {
        HitInfo hi;
        fill_hit_info(A, r.inst, ro, rd, r.t, r.b1, r.b2, r.prim, hi);
        const DInst& I = A.insts[r.inst];
        h.hitPoint[0] = hi.hitPoint.x; h.hitPoint[1] = hi.hitPoint.y; h.hitPoint[2] = hi.hitPoint.z;
        h.primitiveIndex = r.prim; h.instanceIndex = I.instanceID; h.instanceCustomIndex = I.customInstanceID;
        h.instanceSBTOffset = I.SBTOffset;
        h.barycentric[0] = hi.bx; h.barycentric[1] = hi.by; h.barycentric[2] = hi.bz;
        for (int k = 0; k < 16; ++k) h.transform[k] = I.fwd[k];
    } else {
        h.hitPoint[0] = h.hitPoint[1] = h.hitPoint[2] = 0.f;
        h.primitiveIndex = h.instanceIndex = h.instanceCustomIndex = h.instanceSBTOffset = 0;
        h.barycentric[0] = h.barycentric[1] = h.barycentric[2] = 0.f;
        for (int k = 0; k < 16; ++k) h.transform[k] = 0.f;
    }
    dst = h;
}


// ---------------------------------------------------------------------------------------------
// persistent wave-cooperative extend / shadow (production): see traverse_coop.h
// ---------------------------------------------------------------------------------------------
// glue shared by the policies that trace one ray per work item and need no shade step
// 5 waves per SIMD: the register allocator is held to 96 VGPRs (no spills; left alone it takes ~107 once the steal
// step is compiled in and residency drops to 4)
#ifndef COOP_WPE
#define COOP_WPE 5
#endif
#define COOP_BOUNDS __launch_bounds__(RDX_BLOCK, COOP_WPE)
// the pool engine keeps its rays in LDS (traverse_pool.h) and fits 80 VGPRs: 6 waves per SIMD
#ifndef POOL_WPE
#define POOL_WPE 6
#endif
#define POOL_BOUNDS __launch_bounds__(RDX_BLOCK, POOL_WPE)
#define RDX_SINGLE_RAY_POLICY                                                                                   \
    struct State {};                                                                                             \
    static constexpr bool kShades = false;                                                                       \
    __device__ __forceinline__ bool load(uint32_t i, f3& o, f3& d, bool& anyHit, State&) const { return load(i, o, d, anyHit); } \
    __device__ __forceinline__ int finish(uint32_t i, const Best& b, f3& o, f3& d, bool&, State&) const { store(i, b, o, d); return COOP_RELEASE; } \
    __device__ __forceinline__ int shade(uint32_t, f3&, f3&, bool&, State&) const { return COOP_RELEASE; }            \
    __device__ __forceinline__ void retire(State&) const {}

struct ExtendPolicy {
    AccelView A; PathStreams ps;
    RDX_SINGLE_RAY_POLICY
    __device__ __forceinline__ bool load(uint32_t w, f3& o, f3& d, bool& anyHit) const
    {
        const uint32_t i = ps.permE ? ps.permE[w] : w;          // work item w = path i (per-bounce ray sort)
        const float4 ro = ps.rayO[i], rd = ps.rayD[i];
        o = mk3(ro.x, ro.y, ro.z); d = mk3(rd.x, rd.y, rd.z);
        anyHit = false;
        return true;
    }
    __device__ __forceinline__ void store(uint32_t w, const Best& b, f3, f3) const
    {
        const uint32_t i = ps.permE ? ps.permE[w] : w;
        ps.hitA[i] = make_float4(b.t, b.b1, b.b2, b.hit ? u2f(A.tris[b.slot].primID) : 0.0f);
        ps.hitInst[i] = b.hit ? b.inst : RDX_MISS;
    }
};

struct ShadowPolicy {
    AccelView A; PathStreams ps; f3 Ldir; uint32_t lastBounce, nPixels, sampleBase;
    RDX_SINGLE_RAY_POLICY
    __device__ __forceinline__ bool load(uint32_t w, f3& o, f3& d, bool& anyHit) const
    {
        const uint32_t i = ps.permS ? ps.permS[w] : w;
        const float4 so = ps.shO[i];
        o = mk3(so.x, so.y, so.z); d = Ldir;
        anyHit = true;
        return so.w != 0.0f;                    // the closest-hit shader asked for a shadow query
    }
    __device__ __forceinline__ void store(uint32_t w, const Best& b, f3, f3) const
    {
        const uint32_t i = ps.permS ? ps.permS[w] : w;
        // hit -> closest-hit row 2 `shadow` sets payload.hit; miss -> row 4 `shadowMiss` clears it
        bool occluded = false;
        if (ps.shO[i].w != 0.0f) {
            Payload sp; sp.hit = false;
            // (the cooperative engines only run scenes whose instances all have SBTOffset 0 -- others go to the reference-order
            //  kernel, rdx_runtime.cpp view_of -- so the row is the constant 2 and the switch folds to `shadow`; with the
            //  instance's offset read at run time the whole `material` shader was compiled into the traversal kernels)
            if (b.hit) { HitInfo hh{}; SceneView sv{}; callHit(2, sp, hh, sv, Ldir, 0, 0, 0, false); }
            else callMiss(4, sp);
            occluded = sp.hit;
        }
        const float4 c = occluded ? ps.colSh[i] : ps.colLit[i];
        if (lastBounce) { store_sample(ps, nPixels, sampleBase, f2u(ps.nRayD[i].w), f2u(ps.nThr[i].w), mk3(c.x, c.y, c.z)); return; }
        ps.nCol[i] = make_float4(c.x, c.y, c.z, 0.0f);
    }
};

// shadow queries recorded by a user's closest-hit shader (user_shader.cpp "stage mode"): per-path origin AND direction, the
// answer is only the flag (the row's any-hit shader ends the walk at the first accepted candidate: order-free)
struct UserShadowPolicy {
    PathStreams ps;
    RDX_SINGLE_RAY_POLICY
    __device__ __forceinline__ bool load(uint32_t i, f3& o, f3& d, bool& anyHit) const
    {
        const float4 so = ps.shO[i], sd = ps.shD[i];
        o = mk3(so.x, so.y, so.z); d = mk3(sd.x, sd.y, sd.z);
        anyHit = true;
        return so.w != 0.0f;
    }
    __device__ __forceinline__ void store(uint32_t i, const Best& b, f3, f3) const { ps.shHit[i] = b.hit ? 1u : 0u; }
};
template <bool INL, bool CULL>
__global__ void POOL_BOUNDS
k_shadow_user_pool(AccelView A, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t* __restrict__ counter, float tmin, float tmax)
{
    UserShadowPolicy pol{ps};
    traverse_pool<2, INL, CULL>(A, pol, *nPtr, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}

__global__ void COOP_BOUNDS
k_extend_coop(AccelView A, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t* __restrict__ counter, float tmin, float tmax)
{
    ExtendPolicy pol{A, ps};
    traverse_coop<1>(A, pol, *nPtr, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * coop_words_per_wave(A.coopNeed), A.coopNeed);
}

__global__ void COOP_BOUNDS
k_shadow_coop(AccelView A, SceneArgs sc, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t* __restrict__ counter,
              uint32_t lastBounce, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax)
{
    const float* ld = sc.scene->lights[0].direction;
    ShadowPolicy pol{A, ps, normalize3(mk3(-ld[0], -ld[1], -ld[2])), lastBounce, nPixels, sampleBase};   // shader.cl:471-476
    traverse_coop<2>(A, pol, *nPtr, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * coop_words_per_wave(A.coopNeed), A.coopNeed);
}

// shadow(d) and extend(d+1) in ONE launch: both have counts[d+1] rays and touch disjoint streams
// (ShadowPolicy: shO/colLit/colSh -> nCol; ExtendPolicy on the next bounce's streams: rayO/rayD -> hit).
// Ray index i < m is shadow ray i of bounce d, i >= m is extend ray i - m of bounce d+1.  One launch
// instead of two halves the fixed ramp + tail cost per bounce, which is what limits small frames
// (multi-GPU shards): see tools/trav_scale.py.
struct FusedPolicy {
    ShadowPolicy sh; ExtendPolicy ex; uint32_t m;
    RDX_SINGLE_RAY_POLICY
    __device__ __forceinline__ bool load(uint32_t i, f3& o, f3& d, bool& anyHit) const
    {
        return i < m ? sh.load(i, o, d, anyHit) : ex.load(i - m, o, d, anyHit);
    }
    __device__ __forceinline__ void store(uint32_t i, const Best& b, f3 o, f3 d) const
    {
        if (i < m) sh.store(i, b, o, d); else ex.store(i - m, b, o, d);
    }
};

__global__ void COOP_BOUNDS
k_fused_coop(AccelView A, SceneArgs sc, PathStreams psShadow, PathStreams psExtend, const uint32_t* __restrict__ mPtr,
             uint32_t* __restrict__ counter, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax)
{
    const float* ld = sc.scene->lights[0].direction;
    const uint32_t m = *mPtr;
    FusedPolicy pol{ShadowPolicy{A, psShadow, normalize3(mk3(-ld[0], -ld[1], -ld[2])), 0u, nPixels, sampleBase},
                    ExtendPolicy{A, psExtend}, m};
    traverse_coop<3>(A, pol, 2u * m, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * coop_words_per_wave(A.coopNeed), A.coopNeed);
}

// ---- the same three launches on the shared-node-pool engine (traverse_pool.h, `kernel` option 3) ------------------------
template <bool INL, bool CULL>
__global__ void POOL_BOUNDS
k_extend_pool(AccelView A, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t* __restrict__ counter, float tmin, float tmax)
{
    ExtendPolicy pol{A, ps};
    traverse_pool<1, INL, CULL>(A, pol, *nPtr, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}

template <bool INL, bool CULL>
__global__ void POOL_BOUNDS
k_shadow_pool(AccelView A, SceneArgs sc, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t* __restrict__ counter,
              uint32_t lastBounce, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax)
{
    const float* ld = sc.scene->lights[0].direction;
    ShadowPolicy pol{A, ps, normalize3(mk3(-ld[0], -ld[1], -ld[2])), lastBounce, nPixels, sampleBase};
    traverse_pool<2, INL, CULL>(A, pol, *nPtr, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}

template <bool INL, bool CULL>
__global__ void POOL_BOUNDS
k_fused_pool(AccelView A, SceneArgs sc, PathStreams psShadow, PathStreams psExtend, const uint32_t* __restrict__ mPtr,
             uint32_t* __restrict__ counter, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax)
{
    const float* ld = sc.scene->lights[0].direction;
    const uint32_t m = *mPtr;
    FusedPolicy pol{ShadowPolicy{A, psShadow, normalize3(mk3(-ld[0], -ld[1], -ld[2])), 0u, nPixels, sampleBase},
                    ExtendPolicy{A, psExtend}, m};
    traverse_pool<3, INL, CULL>(A, pol, 2u * m, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}

// ---- the same three launches over quad records (traverse_pool.h QUAD) -------------------------------------------------------
// W = waves per SIMD the kernel is built for: 6 (80 VGPRs), or 7 (72 VGPRs, 24-56 B of scratch) for full-size frames of scenes
// without inline leaf roots, where the seventh wave is worth 2 % (Sponza-class 23.7 -> 23.2 ms) -- shards and scenes with leaf
// roots lose with it (launch_*: av.quadWaves).
#ifndef POOL_WPE_QUAD
#define POOL_WPE_QUAD 6
#endif
template <bool INL, int W>
__global__ void __launch_bounds__(RDX_BLOCK, W)
k_extend_pool_q(AccelView A, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t* __restrict__ counter, float tmin, float tmax)
{
    ExtendPolicy pol{A, ps};
    traverse_pool<1, INL, false, ExtendPolicy, true>(A, pol, *nPtr, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}
template <bool INL, int W>
__global__ void __launch_bounds__(RDX_BLOCK, W)
k_shadow_pool_q(AccelView A, SceneArgs sc, PathStreams ps, const uint32_t* __restrict__ nPtr, uint32_t* __restrict__ counter,
                uint32_t lastBounce, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax)
{
    const float* ld = sc.scene->lights[0].direction;
    ShadowPolicy pol{A, ps, normalize3(mk3(-ld[0], -ld[1], -ld[2])), lastBounce, nPixels, sampleBase};
    traverse_pool<2, INL, false, ShadowPolicy, true>(A, pol, *nPtr, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}
template <bool INL, int W>
__global__ void __launch_bounds__(RDX_BLOCK, W)
k_fused_pool_q(AccelView A, SceneArgs sc, PathStreams psShadow, PathStreams psExtend, const uint32_t* __restrict__ mPtr,
               uint32_t* __restrict__ counter, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax)
{
    const float* ld = sc.scene->lights[0].direction;
    const uint32_t m = *mPtr;
    FusedPolicy pol{ShadowPolicy{A, psShadow, normalize3(mk3(-ld[0], -ld[1], -ld[2])), 0u, nPixels, sampleBase},
                    ExtendPolicy{A, psExtend}, m};
    traverse_pool<3, INL, false, FusedPolicy, true>(A, pol, 2u * m, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}

// ---------------------------------------------------------------------------------------------
// whole paths on the cooperative engine: one launch per chunk of samples
// ---------------------------------------------------------------------------------------------
// A lane owns a (pixel, sample) path from the camera ray to its end: closest-hit walk -> shade step
// (SBT dispatch, converged over the lanes waiting for it) -> any-hit walk of the shadow query -> next
// bounce ..., all inside the persistent wave-cooperative engine.  No bounce ever synchronises the grid:
// the per-launch ramp + tail of the staged pipeline (17 launches per depth-8 frame) is paid once, and the
// path streams shrink to the few values that must survive a shadow walk (indexed by path, no compaction).
// Arithmetic per value is that of the staged kernels (k_generate / k_shade / k_shadow), so frames are
// bit-identical to the staged pipeline.
// The closest-hit shader as a real function call: inlined into the persistent traversal loop its ~100 live registers add to
// the loop's own (round 2: 199-211 VGPRs, 2 waves per SIMD, 2.3x slower than the staged pipeline); called, the loop's state is
// saved around the call and the kernel needs max(loop, shader) registers.
struct ShadeCall { f3 o, d; float t, b1, b2; uint32_t triSlot, inst, frameID, pixel, depth, more; };
__device__ __attribute__((noinline)) void shade_call(const AccelView* A, const SceneArgs* sc, const ShadeCall* in, Payload* out)
{
    Payload p;
    p.hit = false; p.wantsShadowRay = false;
    p.color = mk3(0.f, 0.f, 0.f); p.colorOccluded = p.color;
    p.nextFactor = mk3(1.f, 1.f, 1.f);
    p.nextRayOrigin = in->o; p.nextRayDirection = in->d; p.shadowOrigin = in->o;
    const SceneView sv{sc->scene, sc->meshInfo, sc->indexData, sc->uvData, sc->normalData, sc->materials, sc->tex};
    HitInfo h;
    fill_hit_info(*A, in->inst, in->o, in->d, in->t, in->b1, in->b2, A->tris[in->triSlot].primID, h);
    callHit((int)A->insts[in->inst].SBTOffset + 1, p, h, sv, in->d, in->frameID, in->pixel, in->depth, in->more != 0u);
    *out = p;
}

// A wave owns its paths (traverse_pool.h "whole-path policies").  Work item = camera ray index (= path id) or a path id from
// the wave's ray queue | POOL_QUEUED: first the shadow query the closest-hit shader asked for, then the next bounce's ray.
// Path state lives in the path streams, indexed by path id and touched by the owning wave only:
//   rayO = current ray origin | pixel      rayD = direction | frameID      thr = contribution | owned-pixel slot
//   col = colour so far | depth            hitA = t, b1, b2, triangle slot  hitInst = instance slot
//   shO = shadow origin | query wanted     colLit = colour if lit | depth   colSh = colour if occluded
// Arithmetic per value is that of the staged kernels (k_generate / k_shade / ShadowPolicy): frames are bit-identical.
struct PathPolicy {
    AccelView A; SceneArgs sc; CameraArgs C; PathStreams ps; const uint32_t* owned;
    uint32_t nPixels, sampleBegin, totalSamples, maxDepth, sampleBase;
    f3 Ldir;
    unsigned long long* tally;        // [0] closest-hit rays, [1] shadow rays
    struct State { uint32_t phase;    // 0 closest-hit ray, 1 shadow query, 2 nothing to trace (the path ended in load)
                   uint32_t nClosest, nShadow; };      // rays started by this lane (never reset): summed in retire()
    static constexpr bool kShades = true;

    __device__ __forceinline__ void end_path(uint32_t p, f3 c) const
    {
        store_sample(ps, nPixels, sampleBase, f2u(ps.rayD[p].w), f2u(ps.thr[p].w), c);
    }
    // colour after bounce `depth` is `c`: the path ends, or its next ray (written by the shade step) starts
    __device__ __forceinline__ bool advance(uint32_t p, f3 c, uint32_t depth, f3& o, f3& d, bool& anyHit, State& st) const
    {
        if (depth + 1 >= maxDepth) { end_path(p, c); st.phase = 2; return false; }
        ps.col[p] = make_float4(c.x, c.y, c.z, u2f(depth + 1));
        const float4 ro = ps.rayO[p], rd = ps.rayD[p];
        o = mk3(ro.x, ro.y, ro.z); d = mk3(rd.x, rd.y, rd.z); anyHit = false;
        st.phase = 0; st.nClosest++;
        return true;
    }
    __device__ __forceinline__ bool load(uint32_t idx, f3& o, f3& d, bool& anyHit, State& st) const
    {
        if (!(idx & POOL_QUEUED)) {          // camera ray of path idx (shader.cl:196-231)
            const uint32_t p = idx, sLocal = p / nPixels, slot = p - sLocal * nPixels;
            const uint32_t pixel = owned ? owned[slot] : slot, frameID = totalSamples + sampleBegin + sLocal;
            camera_ray(C, pixel, pcg3d(frameID, totalSamples, pixel), o, d);     // shader.cl:205
            ps.rayO[p] = make_float4(o.x, o.y, o.z, u2f(pixel));
            ps.rayD[p] = make_float4(d.x, d.y, d.z, u2f(frameID));
            ps.thr[p] = make_float4(1.0f, 1.0f, 1.0f, u2f(slot));
            ps.col[p] = make_float4(0.0f, 0.0f, 0.0f, u2f(0u));
            anyHit = false; st.phase = 0; st.nClosest++;
            return true;
        }
        const uint32_t p = idx & ~POOL_QUEUED;
        const float4 so = ps.shO[p];
        if (so.w != 0.0f) {                  // traceRay(topLevel, 2, 4, hitPos, L, ...) (shader.cl:499-501)
            o = mk3(so.x, so.y, so.z); d = Ldir; anyHit = true;
            st.phase = 1; st.nShadow++;
            return true;
        }
        const float4 cl = ps.colLit[p];      // no shadow query: the colour is the lit one
        return advance(p, mk3(cl.x, cl.y, cl.z), f2u(cl.w), o, d, anyHit, st);
    }
    __device__ __forceinline__ int finish(uint32_t idx, const Best& b, f3& o, f3& d, bool& anyHit, State& st) const
    {
        const uint32_t p = idx & ~POOL_QUEUED;
        if (st.phase == 1) {
            // the shadow query is answered: hit -> closest-hit row 2 `shadow`, miss -> row 4 `shadowMiss`
            Payload sp; sp.hit = false;
            if (b.hit) { HitInfo hh{}; SceneView sv{}; callHit(2, sp, hh, sv, Ldir, 0, 0, 0, false); }      // SBTOffset is 0 on this engine, see ShadowPolicy
            else callMiss(4, sp);
            const float4 cl = ps.colLit[p];
            const float4 c = sp.hit ? ps.colSh[p] : cl;
            return advance(p, mk3(c.x, c.y, c.z), f2u(cl.w), o, d, anyHit, st) ? COOP_NEWRAY : COOP_RELEASE;
        }
        if (st.phase == 2) return COOP_RELEASE;
        if (!b.hit) {            // miss shader, row 3; a primary miss shows its colour, a later one ends the path (shader.cl:243-252)
            Payload pm; pm.hit = false; pm.color = mk3(0.f, 0.f, 0.f); pm.colorOccluded = pm.color;
            callMiss(3, pm);
            const float4 c = ps.col[p];
            end_path(p, f2u(c.w) == 0u ? pm.color : mk3(c.x, c.y, c.z));
            return COOP_RELEASE;
        }
        ps.hitA[p] = make_float4(b.t, b.b1, b.b2, u2f(b.slot));
        ps.hitInst[p] = b.inst;
        return COOP_SHADE;       // the engine queues the path for a shade step and frees the lane
    }
    // closest-hit shader of path p (any lane of the owning wave); true: the path goes on (ray queue)
    __device__ __forceinline__ bool shade_path(uint32_t p, State&) const
    {
        const float4 ro = ps.rayO[p], rd = ps.rayD[p], thr = ps.thr[p], col = ps.col[p], ha = ps.hitA[p];
        const uint32_t depth = f2u(col.w);
        Payload pl;
        const ShadeCall in{mk3(ro.x, ro.y, ro.z), mk3(rd.x, rd.y, rd.z), ha.x, ha.y, ha.z, f2u(ha.w), ps.hitInst[p], f2u(rd.w), f2u(ro.w), depth,
                           depth + 1 < maxDepth ? 1u : 0u};
        shade_call(&A, &sc, &in, &pl);
        const f3 T = mk3(thr.x, thr.y, thr.z), Cc = mk3(col.x, col.y, col.z);
        if (!pl.hit) {               // no closest-hit shader bound for this row: the raygen loop sees a miss
            end_path(p, depth == 0 ? pl.color : Cc);
            return false;
        }
        const f3 lit = Cc + T * pl.color;              // color += contribution * payload.color (shader.cl:240)
        const f3 occ = Cc + T * pl.colorOccluded;
        const f3 tn = T * pl.nextFactor;               // contribution *= payload.nextFactor    (shader.cl:241)
        ps.thr[p] = make_float4(tn.x, tn.y, tn.z, thr.w);
        ps.rayO[p] = make_float4(pl.nextRayOrigin.x, pl.nextRayOrigin.y, pl.nextRayOrigin.z, ro.w);
        ps.rayD[p] = make_float4(pl.nextRayDirection.x, pl.nextRayDirection.y, pl.nextRayDirection.z, rd.w);
        ps.shO[p] = make_float4(pl.shadowOrigin.x, pl.shadowOrigin.y, pl.shadowOrigin.z, pl.wantsShadowRay ? 1.0f : 0.0f);
        ps.colLit[p] = make_float4(lit.x, lit.y, lit.z, u2f(depth));
        ps.colSh[p] = make_float4(occ.x, occ.y, occ.z, 0.0f);
        return true;
    }
    __device__ __forceinline__ void retire(State& st) const      // one atomic per wave per tally, not one per ray
    {
        uint32_t a = st.nClosest, b = st.nShadow;
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
        if (__lane_id() == 0) { atomicAdd(tally + 0, (unsigned long long)a); atomicAdd(tally + 1, (unsigned long long)b); }
    }
};

// whole paths on the shared-node-pool engine (traverse_pool.h): option "pipeline" 1
template <bool INL, bool CULL>
__global__ void __launch_bounds__(RDX_BLOCK, 5)
k_path_pool(AccelView A, SceneArgs sc, CameraArgs C, PathStreams ps, const uint32_t* __restrict__ owned, uint32_t nPixels,
            uint32_t sampleBegin, uint32_t sampleCount, uint32_t totalSamples, uint32_t maxDepth, uint32_t sampleBase,
            uint32_t* __restrict__ counter, unsigned long long* __restrict__ tally, float tmin, float tmax)
{
    const float* ld = sc.scene->lights[0].direction;
    PathPolicy pol{A, sc, C, ps, owned, nPixels, sampleBegin, totalSamples, maxDepth, sampleBase,
                   normalize3(mk3(-ld[0], -ld[1], -ld[2])), tally};
    traverse_pool<3, INL, CULL>(A, pol, nPixels * sampleCount, counter, tmin, tmax,
                                s_stack + (threadIdx.x >> 6) * (pool_words_per_wave(A.topNeed, A.blasNeed) + POOL_PATH_LDS_WORDS));
}

struct BatchPolicy {
    AccelView A; const float* o; const float* d; rdx_hit* out;
    RDX_SINGLE_RAY_POLICY
    __device__ __forceinline__ bool load(uint32_t i, f3& ro, f3& rd, bool& anyHit) const
    {
        ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]); rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
        anyHit = false;
        return true;
    }
    __device__ __forceinline__ void store(uint32_t i, const Best& b, f3 ro, f3 rd) const
    {
        TraceResult r;
        r.t = b.t; r.b1 = b.b1; r.b2 = b.b2; r.inst = b.inst; r.hit = b.hit;
        r.prim = b.hit ? A.tris[b.slot].primID : 0;
        write_hit_record(A, r, ro, rd, out[i]);
    }
};

template <int REC>
__global__ void COOP_BOUNDS
k_trace_batch_coop(AccelView A, const float* __restrict__ o, const float* __restrict__ d, uint32_t n, uint32_t* __restrict__ counter,
                   float tmin, float tmax, rdx_hit* __restrict__ out)
{
    BatchPolicy pol{A, o, d, out};
    traverse_coop<REC>(A, pol, n, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * coop_words_per_wave(A.coopNeed), A.coopNeed);
}

template <bool INL, bool CULL>
__global__ void POOL_BOUNDS
k_trace_batch_pool1(AccelView A, const float* __restrict__ o, const float* __restrict__ d, uint32_t n, uint32_t* __restrict__ counter,
                    float tmin, float tmax, rdx_hit* __restrict__ out)
{
    BatchPolicy pol{A, o, d, out};
    traverse_pool<1, INL, CULL>(A, pol, n, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}
template <bool INL, bool CULL>
__global__ void POOL_BOUNDS
k_trace_batch_pool2(AccelView A, const float* __restrict__ o, const float* __restrict__ d, uint32_t n, uint32_t* __restrict__ counter,
                    float tmin, float tmax, rdx_hit* __restrict__ out)
{
    BatchPolicy pol{A, o, d, out};
    traverse_pool<2, INL, CULL>(A, pol, n, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}

// (the same over quad records: the test seam walks what the frames walk)
template <bool INL, int W>
__global__ void __launch_bounds__(RDX_BLOCK, W)
k_trace_batch_pool1_q(AccelView A, const float* __restrict__ o, const float* __restrict__ d, uint32_t n, uint32_t* __restrict__ counter,
                      float tmin, float tmax, rdx_hit* __restrict__ out)
{
    BatchPolicy pol{A, o, d, out};
    traverse_pool<1, INL, false, BatchPolicy, true>(A, pol, n, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}
template <bool INL, int W>
__global__ void __launch_bounds__(RDX_BLOCK, W)
k_trace_batch_pool2_q(AccelView A, const float* __restrict__ o, const float* __restrict__ d, uint32_t n, uint32_t* __restrict__ counter,
                      float tmin, float tmax, rdx_hit* __restrict__ out)
{
    BatchPolicy pol{A, o, d, out};
    traverse_pool<2, INL, false, BatchPolicy, true>(A, pol, n, counter, tmin, tmax, s_stack + (threadIdx.x >> 6) * pool_words_per_wave(A.topNeed, A.blasNeed));
}

// ---------------------------------------------------------------------------------------------
// test seams
// ---------------------------------------------------------------------------------------------
template <int REC, bool COUNT, bool WIDE>
__global__ void __launch_bounds__(RDX_BLOCK)
k_trace_batch(AccelView A, const float* __restrict__ o, const float* __restrict__ d, uint32_t n, float tmin, float tmax,
              rdx_hit* __restrict__ out, unsigned long long* __restrict__ visit)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 ro = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), rd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    TraceResult r;
    if (WIDE) {
        Best b;
        traverse_wide<REC>(A, ro, rd, tmin, tmax, s_stack + threadIdx.x, blockDim.x, b);
        r.t = b.t; r.b1 = b.b1; r.b2 = b.b2; r.inst = b.inst; r.hit = b.hit;
        r.prim = b.hit ? A.tris[b.slot].primID : 0;
    } else {
        traverse<REC, COUNT>(A, ro, rd, tmin, tmax, s_stack + threadIdx.x, blockDim.x, r);
        if (COUNT) flush_visits(visit, REC - 1, r);
    }
    write_hit_record(A, r, ro, rd, out[i]);
}

__global__ void __launch_bounds__(RDX_BLOCK)
k_material_batch(SceneArgs sc, const rdx_hit* __restrict__ hits, const float* __restrict__ dirs,
                 const uint32_t* __restrict__ pixels, const uint32_t* __restrict__ frames,
                 const int32_t* __restrict__ depths, uint32_t n, rdx_payload* __restrict__ out)
{
    const uint32_t i = blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const rdx_hit& hh = hits[i];
    HitInfo h;
    h.hitPoint = mk3(hh.hitPoint[0], hh.hitPoint[1], hh.hitPoint[2]);
    h.bx = hh.barycentric[0]; h.by = hh.barycentric[1]; h.bz = hh.barycentric[2];
    h.primitiveIndex = hh.primitiveIndex; h.instanceIndex = hh.instanceIndex;
    h.fwd = hh.transform;
    const SceneView sv{sc.scene, sc.meshInfo, sc.indexData, sc.uvData, sc.normalData, sc.materials, sc.tex};
    Payload p;
    p.hit = false; p.wantsShadowRay = false;
    p.color = p.colorOccluded = p.nextFactor = p.nextRayOrigin = p.nextRayDirection = p.shadowOrigin = mk3(0.f, 0.f, 0.f);
    material(p, h, sv, mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), frames[i], pixels[i], (uint32_t)depths[i], true);
    rdx_payload o;
    // colour reported for the "light visible" outcome; colorOccluded is recoverable as albedo*0.1
    o.color[0] = p.color.x; o.color[1] = p.color.y; o.color[2] = p.color.z;
    o.hit = p.hit ? 1u : 0u;
    o.nextFactor[0] = p.nextFactor.x; o.nextFactor[1] = p.nextFactor.y; o.nextFactor[2] = p.nextFactor.z;
    o.nextRayOrigin[0] = p.nextRayOrigin.x; o.nextRayOrigin[1] = p.nextRayOrigin.y; o.nextRayOrigin[2] = p.nextRayOrigin.z;
    o.nextRayDirection[0] = p.nextRayDirection.x; o.nextRayDirection[1] = p.nextRayDirection.y; o.nextRayDirection[2] = p.nextRayDirection.z;
    out[i] = o;
}

__global__ void __launch_bounds__(RDX_BLOCK)
k_generate_batch(CameraArgs C, const uint32_t* __restrict__ pixels, const uint32_t* __restrict__ rnd, uint32_t n,
                 float* __restrict__ o, float* __restrict__ d)
{
    const uint32_t i = blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (i >= n) return;
    f3 org, dir;
    camera_ray(C, pixels[i], pcg3d(rnd[3 * i], rnd[3 * i + 1], rnd[3 * i + 2]), org, dir);
    o[3 * i] = org.x; o[3 * i + 1] = org.y; o[3 * i + 2] = org.z;
    d[3 * i] = dir.x; d[3 * i + 1] = dir.y; d[3 * i + 2] = dir.z;
}

__global__ void __launch_bounds__(RDX_BLOCK)
k_pcg3d_batch(const uint32_t* __restrict__ in3, float* __restrict__ out3, uint32_t n)
{
    const uint32_t i = blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const f3 r = pcg3d(in3[3 * i], in3[3 * i + 1], in3[3 * i + 2]);
    out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static inline uint32_t blocks_for(uint64_t n, uint32_t b) { return (uint32_t)((n + b - 1) / b); }

// threads per block for the traversal kernels: the per-lane stack lives in LDS (4 B * need * threads)
static inline uint32_t trav_threads(uint32_t need, size_t& ldsBytes)
{
    uint32_t threads = RDX_BLOCK;
    if (need < 1) need = 1;
    while (threads > 64 && (size_t)need * threads * 4 > 40 * 1024) threads >>= 1;
    ldsBytes = (size_t)need * threads * 4;
    return threads;
}

void launch_generate(hipStream_t st, const CameraArgs& cam, const PathStreams& ps, const uint32_t* owned, uint32_t nPixels,
                     uint32_t sampleBegin, uint32_t sampleCount, uint32_t totalSamples)
{
    const uint64_t n = (uint64_t)nPixels * sampleCount;
    if (!n) return;
    hipLaunchKernelGGL(k_generate, dim3(blocks_for(n, RDX_BLOCK)), dim3(RDX_BLOCK), 0, st, cam, ps, owned, nPixels,
                       sampleBegin, sampleCount, totalSamples);
}

// threads per block for the cooperative kernels: per-wave LDS = stack + queue + ray table
static inline uint32_t coop_threads_words(uint32_t wordsPerWave, size_t& ldsBytes, uint32_t wpe);
static inline uint32_t coop_threads(uint32_t need, size_t& ldsBytes) { return coop_threads_words(coop_words_per_wave(need), ldsBytes, COOP_WPE); }
static inline uint32_t pool_threads(const AccelView& av, size_t& ldsBytes) { return coop_threads_words(pool_words_per_wave(av.topNeed, av.blasNeed), ldsBytes, POOL_WPE); }
static inline uint32_t coop_threads_words(uint32_t wordsPerWave, size_t& ldsBytes, uint32_t wpe)
{
    // experiment knobs (tools/occupancy_probe.sh): RDX_COOP_THREADS = block size, RDX_COOP_LDS_PAD = extra LDS bytes per
    // wave (lowers the residency the LDS allows)
    static const int envThreads = std::getenv("RDX_COOP_THREADS") ? std::atoi(std::getenv("RDX_COOP_THREADS")) : 0;
    static const int envPad = std::getenv("RDX_COOP_LDS_PAD") ? std::atoi(std::getenv("RDX_COOP_LDS_PAD")) : 0;
    const size_t perWave = (size_t)wordsPerWave * 4 + (size_t)envPad;
    // Residency is LDS-bound (tools/occupancy_probe.sh: frame time falls steadily up to the `wpe` waves / SIMD the
    // registers allow), so take the block size -- 4 or 2 waves -- that packs the most waves into a CU's 160 KB.
    uint32_t threads = 64;
    if (envThreads >= 64) {
        threads = (uint32_t)envThreads;
        while (threads > 64 && perWave * (threads / 64) > 64 * 1024) threads >>= 1;
    } else {
        uint32_t best = 0;
        for (uint32_t t : {128u, 256u}) {      // on a tie the smaller block: measured faster (it also packs tighter)
            const size_t blk = perWave * (t / 64);
            if (blk > 64 * 1024) continue;
            const uint32_t waves = std::min<uint32_t>((uint32_t)((160 * 1024) / blk) * (t / 64), 4u * wpe);
            if (waves > best) { best = waves; threads = t; }
        }
    }
    ldsBytes = perWave * (threads / 64);
    return threads;
}
uint32_t coop_lds_words(uint32_t coopNeed) { return coop_words_per_wave(coopNeed); }
uint32_t pool_lds_words(uint32_t topNeed, uint32_t blasNeed) { return pool_words_per_wave(topNeed, blasNeed); }
static thread_local uint32_t g_gridShare = 1;       // host side: how many concurrent persistent launches share the GPU (set per chunk)
void set_grid_share(uint32_t groups) { g_gridShare = groups ? groups : 1; }

// persistent grid: enough blocks to fill every CU at the LDS-limited residency, never more than the work
static inline uint32_t coop_blocks(uint32_t nMax, uint32_t threads, size_t ldsBytes, uint32_t wpe = COOP_WPE)
{
    const uint32_t perCU = (uint32_t)std::max<size_t>(1, std::min<size_t>((160 * 1024) / std::max<size_t>(ldsBytes, 1), (64u * 4u * wpe) / threads));
    // small launches: enough waves that each is handed COOP_MIN_QUOTA rays (its other lanes help, traverse_coop.h)
    const uint64_t spread = COOP_STEAL ? (uint64_t)nMax * (64u / COOP_MIN_QUOTA) : nMax;
    // launches of different sample groups share the GPU: each takes its part of the resident grid
    // (experiment knob RDX_COOP_GRID_DIV overrides the divisor)
    static const int envDiv = std::getenv("RDX_COOP_GRID_DIV") ? std::max(1, std::atoi(std::getenv("RDX_COOP_GRID_DIV"))) : 0;
    const uint32_t div = envDiv ? (uint32_t)envDiv : g_gridShare;
    return (uint32_t)std::min<uint64_t>((spread + threads - 1) / threads, std::max(1u, 256u * perCU / div));
}

void launch_extend(hipStream_t st, const AccelView& av, const PathStreams& ps, const uint32_t* nPtr, uint32_t nMax,
                   float tmin, float tmax, unsigned long long* visit, uint32_t* counter)
{
    if (!nMax) return;
    if (!visit && av.kernel == 3) {
        size_t lds; const uint32_t th = pool_threads(av, lds);
#define RDX_POOL_LAUNCH(K, GRID, ...)                                                                                         \
        do {                                                                                                              \
            if (av.leafRoots) { if (av.cull) hipLaunchKernelGGL((K<true, true>), GRID, dim3(th), lds, st, __VA_ARGS__);      \
                                else hipLaunchKernelGGL((K<true, false>), GRID, dim3(th), lds, st, __VA_ARGS__); }          \
            else { if (av.cull) hipLaunchKernelGGL((K<false, true>), GRID, dim3(th), lds, st, __VA_ARGS__);                 \
                   else hipLaunchKernelGGL((K<false, false>), GRID, dim3(th), lds, st, __VA_ARGS__); }                      \
        } while (0)
#define RDX_POOL_LAUNCH_Q(K, N, ...)                                                                                           \
        do {                                                                                                              \
            const uint32_t wq = (av.quadWaves == 7u && !av.leafRoots) ? 7u : (uint32_t)POOL_WPE_QUAD;                    \
            size_t ldsq; const uint32_t thq = coop_threads_words(pool_words_per_wave(av.topNeed, av.blasNeed), ldsq, wq);  \
            const dim3 gq(coop_blocks(N, thq, ldsq, wq));                                                                \
            if (av.leafRoots) hipLaunchKernelGGL((K<true, POOL_WPE_QUAD>), gq, dim3(thq), ldsq, st, __VA_ARGS__);           \
            else if (wq == 7u) hipLaunchKernelGGL((K<false, 7>), gq, dim3(thq), ldsq, st, __VA_ARGS__);                     \
            else hipLaunchKernelGGL((K<false, POOL_WPE_QUAD>), gq, dim3(thq), ldsq, st, __VA_ARGS__);                       \
        } while (0)
        if (av.quad && !av.cull) { RDX_POOL_LAUNCH_Q(k_extend_pool_q, nMax, av, ps, nPtr, counter, tmin, tmax); return; }
        RDX_POOL_LAUNCH(k_extend_pool, dim3(coop_blocks(nMax, th, lds, POOL_WPE)), av, ps, nPtr, counter, tmin, tmax);
        return;
    }
    if (!visit && av.kernel == 2) {
        size_t lds; const uint32_t th = coop_threads(av.coopNeed, lds);
        hipLaunchKernelGGL(k_extend_coop, dim3(coop_blocks(nMax, th, lds)), dim3(th), lds, st, av, ps, nPtr, counter, tmin, tmax);
        return;
    }
    size_t lds; const uint32_t th = trav_threads(av.stackNeed, lds);
    if (visit || av.kernel == 0)
        hipLaunchKernelGGL(k_extend<true>, dim3(blocks_for(nMax, th)), dim3(th), lds, st, av, ps, nPtr, tmin, tmax, visit);
    else
        hipLaunchKernelGGL(k_extend<false>, dim3(blocks_for(nMax, th)), dim3(th), lds, st, av, ps, nPtr, tmin, tmax, visit);
}

void launch_shade(hipStream_t st, const AccelView& av, const SceneArgs& sc, const PathStreams& ps, const uint32_t* nPtr,
                  uint32_t* nOut, uint32_t nMax, uint32_t depth, uint32_t maxDepth, uint32_t nPixels, uint32_t sampleBase, const SortBox* sortBox)
{
    if (!nMax) return;
    PathStreams q = ps;
    SortBox sb{};
    if (sortBox) sb = *sortBox; else q.sortKey = nullptr;
    hipLaunchKernelGGL(k_shade, dim3(blocks_for(nMax, RDX_SHADE_BLOCK)), dim3(RDX_SHADE_BLOCK), 0, st, av, sc, q, nPtr, nOut, depth,
                       maxDepth, nPixels, sampleBase, sb);
}

void launch_ray_sort_tiles(hipStream_t st, const PathStreams& ps, const uint32_t* mPtr, uint32_t nMax, const SortBox& box, uint32_t* H,
                           uint32_t* perm, bool largeScene)
{
    if (!nMax) return;
    const uint32_t bits = largeScene ? SORT_GBITS_LARGE : SORT_GBITS, bins = 1u << bits, nTiles = bins * SORT_GCOLS / SORT_TILE;
    uint32_t* sums = H + SORT_GBINS * SORT_GCOLS;
    const uint32_t cols = std::min<uint32_t>((nMax + SORT_GTILE - 1u) / SORT_GTILE, SORT_GCOLS);
    if (cols < SORT_GCOLS) (void)hipMemsetAsync(H, 0, (size_t)bins * SORT_GCOLS * sizeof(uint32_t), st);     // columns no block writes
    hipLaunchKernelGGL(k_sortg_hist, dim3(cols), dim3(256), 0, st, ps, mPtr, box, H, bits);
    hipLaunchKernelGGL(k_sort_scan_tiles, dim3(nTiles), dim3(256), 0, st, H, sums);
    hipLaunchKernelGGL(k_sortg_scan_sums, dim3(1), dim3(1024), 0, st, sums, nTiles);
    hipLaunchKernelGGL(k_sortg_scatter, dim3(cols), dim3(256), 0, st, ps, mPtr, box, H, sums, perm, bits);
}
uint32_t ray_sort_tiles_words() { return SORT_GBINS * SORT_GCOLS + SORT_GSCAN_TILES; }

void launch_shadow(hipStream_t st, const AccelView& av, const SceneArgs& sc, const PathStreams& ps, const uint32_t* nPtr,
                   uint32_t nMax, bool lastBounce, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax,
                   unsigned long long* visit, uint32_t* counter)
{
    if (!nMax) return;
    if (!visit && av.kernel == 3) {
        size_t lds; const uint32_t th = pool_threads(av, lds);
        if (av.quad && !av.cull) {
            RDX_POOL_LAUNCH_Q(k_shadow_pool_q, nMax, av, sc, ps, nPtr, counter, lastBounce ? 1u : 0u, nPixels, sampleBase, tmin, tmax);
            return;
        }
        RDX_POOL_LAUNCH(k_shadow_pool, dim3(coop_blocks(nMax, th, lds, POOL_WPE)), av, sc, ps, nPtr, counter, lastBounce ? 1u : 0u, nPixels, sampleBase,
                        tmin, tmax);
        return;
    }
    if (!visit && av.kernel == 2) {
        size_t ldsc; const uint32_t thc = coop_threads(av.coopNeed, ldsc);
        hipLaunchKernelGGL(k_shadow_coop, dim3(coop_blocks(nMax, thc, ldsc)), dim3(thc), ldsc, st, av, sc, ps, nPtr, counter,
                           lastBounce ? 1u : 0u, nPixels, sampleBase, tmin, tmax);
        return;
    }
    size_t lds; const uint32_t th = trav_threads(av.stackNeed, lds);
    if (visit || av.kernel == 0)
        hipLaunchKernelGGL(k_shadow<true>, dim3(blocks_for(nMax, th)), dim3(th), lds, st, av, sc, ps, nPtr,
                           lastBounce ? 1u : 0u, nPixels, sampleBase, tmin, tmax, visit);
    else
        hipLaunchKernelGGL(k_shadow<false>, dim3(blocks_for(nMax, th)), dim3(th), lds, st, av, sc, ps, nPtr,
                           lastBounce ? 1u : 0u, nPixels, sampleBase, tmin, tmax, visit);
}

void launch_shadow_user(hipStream_t st, const AccelView& av, const PathStreams& ps, const uint32_t* nPtr, uint32_t nMax, float tmin, float tmax,
                        uint32_t* counter)
{
    if (!nMax) return;
    size_t lds; const uint32_t th = pool_threads(av, lds);
    RDX_POOL_LAUNCH(k_shadow_user_pool, dim3(coop_blocks(nMax, th, lds, POOL_WPE)), av, ps, nPtr, counter, tmin, tmax);
}

void launch_fused(hipStream_t st, const AccelView& av, const SceneArgs& sc, const PathStreams& psShadow, const PathStreams& psExtend,
                  const uint32_t* mPtr, uint32_t mMax, uint32_t nPixels, uint32_t sampleBase, float tmin, float tmax, uint32_t* counter)
{
    if (!mMax) return;
    if (av.kernel == 3) {
        size_t lds; const uint32_t th = pool_threads(av, lds);
        if (av.quad && !av.cull) {
            RDX_POOL_LAUNCH_Q(k_fused_pool_q, 2u * mMax, av, sc, psShadow, psExtend, mPtr, counter, nPixels, sampleBase, tmin, tmax);
            return;
        }
        RDX_POOL_LAUNCH(k_fused_pool, dim3(coop_blocks(2u * mMax, th, lds, POOL_WPE)), av, sc, psShadow, psExtend, mPtr, counter, nPixels, sampleBase, tmin, tmax);
        return;
    }
    size_t lds; const uint32_t th = coop_threads(av.coopNeed, lds);
    hipLaunchKernelGGL(k_fused_coop, dim3(coop_blocks(2u * mMax, th, lds)), dim3(th), lds, st, av, sc, psShadow, psExtend, mPtr,
                       counter, nPixels, sampleBase, tmin, tmax);
}

void launch_path(hipStream_t st, const AccelView& av, const SceneArgs& sc, const CameraArgs& cam, const PathStreams& ps,
                 const uint32_t* owned, uint32_t nPixels, uint32_t sampleBegin, uint32_t sampleCount, uint32_t totalSamples,
                 uint32_t maxDepth, uint32_t sampleBase, uint32_t* counter, unsigned long long* tally, float tmin, float tmax)
{
    const uint64_t n = (uint64_t)nPixels * sampleCount;
    if (!n) return;
    size_t lds; const uint32_t th = coop_threads_words(pool_words_per_wave(av.topNeed, av.blasNeed) + POOL_PATH_LDS_WORDS, lds, 5);      // (the caller checks av.kernel == 3)
    RDX_POOL_LAUNCH(k_path_pool, dim3(coop_blocks((uint32_t)n, th, lds, 5)), av, sc, cam, ps, owned, nPixels, sampleBegin, sampleCount,
                    totalSamples, maxDepth, sampleBase, counter, tally, tmin, tmax);
}

void launch_accumulate(hipStream_t st, const PathStreams& ps, const uint32_t* owned, uint32_t nPixels, uint32_t sampleBegin,
                       uint32_t sampleCount, uint32_t totalSamples, bool tonemap, uint32_t debug, float* scratch,
                       uint8_t* image)
{
    if (!nPixels) return;
    hipLaunchKernelGGL(k_accumulate, dim3(blocks_for(nPixels, RDX_BLOCK)), dim3(RDX_BLOCK), 0, st, ps, owned, nPixels,
                       sampleBegin, sampleCount, totalSamples, tonemap ? 1u : 0u, debug, scratch, image);
}

void launch_finalize_all(hipStream_t st, const PathStreams& ps, uint32_t n, uint32_t nPixels, uint32_t sampleBase)
{
    if (!n) return;
    hipLaunchKernelGGL(k_finalize_all, dim3(blocks_for(n, RDX_BLOCK)), dim3(RDX_BLOCK), 0, st, ps, n, nPixels, sampleBase);
}

void launch_pack_tiles(hipStream_t st, const uint8_t* image, uint8_t* packed, uint32_t w, uint32_t h, uint32_t elem,
                       uint32_t tileW, uint32_t tileH, uint32_t rank, uint32_t world, bool unpack)
{
    const uint32_t tilesX = (w + tileW - 1) / tileW, tilesY = (h + tileH - 1) / tileH, nTiles = tilesX * tilesY;
    const uint32_t ownedTiles = nTiles > rank ? (nTiles - rank + world - 1) / world : 0;
    const uint64_t total = (uint64_t)ownedTiles * tileW * tileH * (elem / 4);
    if (!total) return;
    hipLaunchKernelGGL(k_pack_tiles, dim3(blocks_for(total, RDX_BLOCK)), dim3(RDX_BLOCK), 0, st, image, packed, w, h, elem,
                       tileW, tileH, rank, world, unpack ? 1u : 0u);
}

void launch_trace_batch(hipStream_t st, const AccelView& av, const float* o, const float* d, uint32_t n, float tmin,
                        float tmax, int rec, rdx_hit* out, unsigned long long* visit, int mode, uint32_t* counter)
{
    if (!n) return;
    if (!visit && mode == 0 && av.kernel == 3) {
        size_t lds; const uint32_t th = pool_threads(av, lds);
        const dim3 gp(coop_blocks(n, th, lds, POOL_WPE));
        if (av.quad && !av.cull) {
            if (rec == 2) RDX_POOL_LAUNCH_Q(k_trace_batch_pool2_q, n, av, o, d, n, counter, tmin, tmax, out);
            else RDX_POOL_LAUNCH_Q(k_trace_batch_pool1_q, n, av, o, d, n, counter, tmin, tmax, out);
            return;
        }
        if (rec == 2) RDX_POOL_LAUNCH(k_trace_batch_pool2, gp, av, o, d, n, counter, tmin, tmax, out);
        else RDX_POOL_LAUNCH(k_trace_batch_pool1, gp, av, o, d, n, counter, tmin, tmax, out);
        return;
    }
    if (!visit && mode == 0 && av.kernel == 2) {
        size_t ldsc; const uint32_t thc = coop_threads(av.coopNeed, ldsc);
        const dim3 gc(coop_blocks(n, thc, ldsc));
        if (rec == 2) hipLaunchKernelGGL(k_trace_batch_coop<2>, gc, dim3(thc), ldsc, st, av, o, d, n, counter, tmin, tmax, out);
        else hipLaunchKernelGGL(k_trace_batch_coop<1>, gc, dim3(thc), ldsc, st, av, o, d, n, counter, tmin, tmax, out);
        return;
    }
    size_t lds; const uint32_t th = trav_threads(av.stackNeed, lds);
    const dim3 g(blocks_for(n, th)), b(th);
#define RDX_TB(REC, COUNT, WIDE) hipLaunchKernelGGL((k_trace_batch<REC, COUNT, WIDE>), g, b, lds, st, av, o, d, n, tmin, tmax, out, visit)
    if (visit) { if (rec == 2) RDX_TB(2, true, false); else RDX_TB(1, true, false); }
    else if (mode == 1 || av.kernel == 0) { if (rec == 2) RDX_TB(2, false, false); else RDX_TB(1, false, false); }
    else { if (rec == 2) RDX_TB(2, false, true); else RDX_TB(1, false, true); }
#undef RDX_TB
}

void launch_material_batch(hipStream_t st, const SceneArgs& sc, const rdx_hit* hits, const float* dirs,
                           const uint32_t* pixels, const uint32_t* frames, const int32_t* depths, uint32_t n,
                           rdx_payload* out)
{
    if (!n) return;
    hipLaunchKernelGGL(k_material_batch, dim3(blocks_for(n, RDX_BLOCK)), dim3(RDX_BLOCK), 0, st, sc, hits, dirs, pixels,
                       frames, depths, n, out);
}

void launch_generate_batch(hipStream_t st, const CameraArgs& cam, const uint32_t* pixels, const uint32_t* rnd, uint32_t n,
                           float* o, float* d)
{
    if (!n) return;
    hipLaunchKernelGGL(k_generate_batch, dim3(blocks_for(n, RDX_BLOCK)), dim3(RDX_BLOCK), 0, st, cam, pixels, rnd, n, o, d);
}

void launch_pcg3d_batch(hipStream_t st, const uint32_t* in3, float* out3, uint32_t n)
{
    if (!n) return;
    hipLaunchKernelGGL(k_pcg3d_batch, dim3(blocks_for(n, RDX_BLOCK)), dim3(RDX_BLOCK), 0, st, in3, out3, n);
}

// ---------------------------------------------------------------------------------------------
// FETCH_SIZE calibration probe (tools/fetch_calibrate.py; bench.py's roofline)
// ---------------------------------------------------------------------------------------------
// rocprofv3's FETCH_SIZE is calibrated in /opt/skills/guides/MI355X_MICROARCH.md for wide coalesced reads only (it reports half
// their bytes on gfx950).  The traversal kernels read differently: every lane gathers ONE record -- a 64-byte wide node (four
// dwordx4) or a 48-byte triangle (three) -- at an address of its own.  This kernel does exactly that on a table of known size
// with a known number of reads, so that the counter can be read against a known byte count for this access shape.
// (STREAM: lane i reads 16 consecutive bytes at i -- the guide's own case, for reference.)
template <int QUADS, bool STREAM>
__global__ void __launch_bounds__(RDX_BLOCK)
k_probe_gather(const uint4* __restrict__ table, uint32_t nRecords, uint32_t nReads, uint32_t seed, uint32_t* __restrict__ sink)
{
    const uint32_t i = blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (i >= nReads) return;
    uint32_t h = i * 2654435761u + seed;             // integer hash -> record index (uniform over the table)
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const uint32_t rec = STREAM ? i % nRecords : (uint32_t)(((unsigned long long)h * nRecords) >> 32);
    const uint4* p = table + (size_t)rec * QUADS;
    uint4 acc = p[0];
#pragma unroll
    for (int k = 1; k < QUADS; ++k) { const uint4 v = p[k]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = i;      // (keeps the loads alive; the table holds other values)
}
// ---------------------------------------------------------------------------------------------
// GPU-assisted BVH build: the binning pass of one node (bvh_build.cpp "candidate evaluation by bins", GpuBinner)
// ---------------------------------------------------------------------------------------------
// prims9: the BLAS's primitives, 9 floats each (box min, box max, centroid); work[n]: the node's primitives; cand[axis * 1024 +
// k], K[axis]: candidate planes.  Bin of a primitive = index of the first candidate greater than its centroid.  Per axis the
// output holds BVH_BIN_SLOTS counts, then 3 x BVH_BIN_SLOTS order-preserving keys of the box minima (x, y, z), then the maxima.
// blockIdx.y = axis; a block bins its slice into LDS and flushes the bins it touched with device atomics.
constexpr uint32_t BVH_BIN_SLOTS = 1025u;
__device__ __forceinline__ uint32_t bvh_key(float f) { const uint32_t b = __float_as_uint(f); return b ^ (((uint32_t)((int32_t)b >> 31)) | 0x80000000u); }
__global__ void __launch_bounds__(256)
k_bvh_bin_init(uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 3u * 7u * BVH_BIN_SLOTS) return;
    const uint32_t seg = (i % (7u * BVH_BIN_SLOTS)) / BVH_BIN_SLOTS;      // 0 count, 1-3 minima, 4-6 maxima
    out[i] = (seg >= 1u && seg <= 3u) ? 0xffffffffu : 0u;
}
__global__ void __launch_bounds__(256)
k_bvh_bin(const float* __restrict__ prims9, const uint32_t* __restrict__ work, uint32_t n, const float* __restrict__ cand,
          uint3 K3, uint32_t* __restrict__ out)
{
    __shared__ uint32_t s_bin[7u * BVH_BIN_SLOTS];
    const uint32_t axis = blockIdx.y;
    const uint32_t K = axis == 0 ? K3.x : (axis == 1 ? K3.y : K3.z);
    if (K == 0u) return;
    for (uint32_t i = threadIdx.x; i < 7u * (K + 1u); i += 256u) {
        const uint32_t seg = i / (K + 1u);
        s_bin[seg * BVH_BIN_SLOTS + (i - seg * (K + 1u))] = (seg >= 1u && seg <= 3u) ? 0xffffffffu : 0u;
    }
    __syncthreads();
    const float* t = cand + axis * 1024u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float* q = prims9 + 9u * (size_t)work[i];
        const float c = q[6u + axis];
        uint32_t lo = 0u, hi = K;                  // first k with c < t[k]
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (c < t[mid]) hi = mid; else lo = mid + 1u; }
        atomicAdd(&s_bin[lo], 1u);
        atomicMin(&s_bin[1u * BVH_BIN_SLOTS + lo], bvh_key(q[0])); atomicMin(&s_bin[2u * BVH_BIN_SLOTS + lo], bvh_key(q[1]));
        atomicMin(&s_bin[3u * BVH_BIN_SLOTS + lo], bvh_key(q[2]));
        atomicMax(&s_bin[4u * BVH_BIN_SLOTS + lo], bvh_key(q[3])); atomicMax(&s_bin[5u * BVH_BIN_SLOTS + lo], bvh_key(q[4]));
        atomicMax(&s_bin[6u * BVH_BIN_SLOTS + lo], bvh_key(q[5]));
    }
    __syncthreads();
    uint32_t* o = out + axis * 7u * BVH_BIN_SLOTS;
    for (uint32_t b = threadIdx.x; b <= K; b += 256u) {
        const uint32_t c = s_bin[b];
        if (c == 0u) continue;
        atomicAdd(&o[b], c);
        for (uint32_t k = 1u; k <= 3u; ++k) atomicMin(&o[k * BVH_BIN_SLOTS + b], s_bin[k * BVH_BIN_SLOTS + b]);
        for (uint32_t k = 4u; k <= 6u; ++k) atomicMax(&o[k * BVH_BIN_SLOTS + b], s_bin[k * BVH_BIN_SLOTS + b]);
    }
}
void launch_bvh_bin(hipStream_t st, const float* prims9, const uint32_t* work, uint32_t n, const float* cand, const uint32_t K[3], uint32_t* out)
{
    hipLaunchKernelGGL(k_bvh_bin_init, dim3(blocks_for(3u * 7u * BVH_BIN_SLOTS, 256)), dim3(256), 0, st, out);
    const uint32_t blocks = std::max(1u, std::min(1024u, (n + 2047u) / 2048u));
    hipLaunchKernelGGL(k_bvh_bin, dim3(blocks, 3), dim3(256), 0, st, prims9, work, n, cand, make_uint3(K[0], K[1], K[2]), out);
}

__global__ void __launch_bounds__(RDX_BLOCK)
k_probe_gather_quad(const uint4* __restrict__ table, uint32_t nRecords, uint32_t nReads, uint32_t seed, uint32_t* __restrict__ sink)
{
    const uint32_t i = blockIdx.x * RDX_BLOCK + threadIdx.x;       // (nReads is a multiple of the block size: all lanes take part)
    uint32_t h = i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const uint32_t rec = (uint32_t)(((unsigned long long)h * nRecords) >> 32);
    uint4 o[4];
    quad_gather64(table, rec, o);
    // check: every piece must be the one a per-lane gather would have read (the table holds record * 4 + piece in .x when
    // seed & 0x80000000 asks for the check)
    uint32_t acc = o[0].x ^ o[0].y ^ o[0].z ^ o[0].w ^ o[1].x ^ o[1].y ^ o[1].z ^ o[1].w ^ o[2].x ^ o[2].y ^ o[2].z ^ o[2].w ^ o[3].x ^ o[3].y ^ o[3].z ^ o[3].w;
    if (seed & 0x80000000u) {
        bool ok = true;
        for (int c = 0; c < 4; ++c) ok = ok && o[c].x == rec * 4u + (uint32_t)c && o[c].y == ~(rec * 4u + (uint32_t)c);
        if (!ok) atomicAdd(sink + 1, 1u);
    }
    if (acc == 0x9e3779b9u) sink[0] = i;
}
__global__ void __launch_bounds__(RDX_BLOCK) k_probe_fill(uint4* __restrict__ table, uint32_t nQuads)
{
    const uint32_t i = blockIdx.x * RDX_BLOCK + threadIdx.x;
    if (i < nQuads) table[i] = make_uint4(i, ~i, 0x5a5a5a5au, 0xa5a5a5a5u);
}
// which XCD does a workgroup run on?  out[b] = HW_REG_XCC_ID of block b (guide: blocks are dealt round-robin over the 8 XCDs)
__global__ void k_probe_xcc(uint32_t* __restrict__ out)
{
    if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
}
extern "C" int rdx_debug_xcc_probe(uint32_t* out, uint32_t nBlocks)
{
    uint32_t* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), nBlocks * 4) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_probe_xcc, dim3(nBlocks), dim3(64), 0, nullptr, d);
    const bool ok = hipMemcpy(out, d, nBlocks * 4, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? 0 : -1;
}
// recBytes: 64 / 48 (gathers) or 16 (streaming); returns the kernel time of the last repetition in ms, < 0 on error
extern "C" float rdx_debug_gather_probe(uint32_t recBytes, unsigned long long tableBytes, uint32_t nReads, uint32_t reps)
{
    const bool coop = recBytes == 65u;            // 65: 64-byte records fetched by the quad-cooperative gather (checked against the table)
    if (coop) recBytes = 64u;
    const uint32_t quads = recBytes / 16u;
    if (quads != 4u && quads != 3u && quads != 1u) return -1.0f;
    if (coop) nReads &= ~(uint32_t)(RDX_BLOCK - 1);
    const uint32_t nRecords = (uint32_t)std::min<unsigned long long>(tableBytes / recBytes, 0xffffffffull);
    if (!nRecords || !nReads) return -1.0f;
    uint4* table = nullptr; uint32_t* sink = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&table), (size_t)nRecords * recBytes) != hipSuccess) return -1.0f;
    if (hipMalloc(reinterpret_cast<void**>(&sink), 64) != hipSuccess) { (void)hipFree(table); return -1.0f; }
    (void)hipMemset(table, 0x5a, (size_t)nRecords * recBytes);
    (void)hipMemset(sink, 0, 64);
    if (coop) hipLaunchKernelGGL(k_probe_fill, dim3(blocks_for(nRecords * 4u, RDX_BLOCK)), dim3(RDX_BLOCK), 0, nullptr, table, nRecords * 4u);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms = 0.0f;
    const dim3 grid(blocks_for(nReads, RDX_BLOCK)), block(RDX_BLOCK);
    for (uint32_t r = 0; r < reps; ++r) {
        (void)hipEventRecord(a, nullptr);
        if (coop) hipLaunchKernelGGL(k_probe_gather_quad, grid, block, 0, nullptr, table, nRecords, nReads, (17u + r) | 0x80000000u, sink);
        else if (quads == 4u) hipLaunchKernelGGL((k_probe_gather<4, false>), grid, block, 0, nullptr, table, nRecords, nReads, 17u + r, sink);
        else if (quads == 3u) hipLaunchKernelGGL((k_probe_gather<3, false>), grid, block, 0, nullptr, table, nRecords, nReads, 17u + r, sink);
        else hipLaunchKernelGGL((k_probe_gather<1, true>), grid, block, 0, nullptr, table, nRecords, nReads, 17u + r, sink);
        (void)hipEventRecord(b, nullptr);
        if (hipEventSynchronize(b) != hipSuccess) { ms = -1.0f; break; }
        (void)hipEventElapsedTime(&ms, a, b);
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    if (coop) {
        uint32_t bad[2] = {0, 0};
        if (hipMemcpy(bad, sink, 8, hipMemcpyDeviceToHost) != hipSuccess || bad[1] != 0u) ms = -2.0f;      // a lane got a wrong piece
    }
    (void)hipFree(table); (void)hipFree(sink);
    return ms;
}

#ifdef COOP_STATS
// experiment builds only (RDX_DEFINES=-DCOOP_STATS): read and clear the step statistics of traverse_coop.h
extern "C" int rdx_debug_coop_stats(unsigned long long* out24)     // [0,16) step statistics, [16,24) lane states, [24,32) cycles per step kind
{
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_coop_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out24 + 16, HIP_SYMBOL(g_coop_state), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out24 + 24, HIP_SYMBOL(g_coop_cycles), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_coop_state), z, 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_coop_cycles), z, 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_coop_stats), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
} // namespace rdx

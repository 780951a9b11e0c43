// traverse_pool.h -- the wave-cooperative traversal with a SHARED node pool (the default engine, `kernel` option 3).
//
// tools/coop_stats.py shows where traverse_coop.h leaves lanes idle: a test step serves 62-64 lanes because any lane
// may test any queued (ray, triangle) pair, but a node step only serves the lanes whose OWN ray holds a BLAS node at
// that moment -- 29-33 of 64.  Here BLAS nodes get the treatment the triangles already have.  A pending (ray, node)
// pair is an independent work item whose only outputs are more such items, queued triangle tests and, through them,
// the owner's best key (the exhaustive walk, CULL = false, never looks at the best t; the culled walk, CULL = true,
// reads the owner's best t to drop items that cannot improve it -- a stale value only makes it drop less):
// all of a wave's pending BLAS nodes live in ONE LIFO pool in LDS and every pool step pops up
// to 64 of them, whichever rays they belong to.  The lane that processes an item reads the object-space ray of the
// item from the owner's ray slot in LDS ("Rays in LDS" below), runs the two slab tests of the wide node with the
// reference's decision rule, pushes inner children back, queues leaf triangles for the owner and adjusts the owner's
// count of outstanding items (LDS atomic).  A lane is the top-level driver of its ray: it walks the TLAS and enters
// instances itself (private stack of top-level entries), pushes the BLAS root into the pool and waits until its
// count is back to zero before it moves on, so a ray has one instance in the pool at a time and the parity logic of
// the test queue (traverse_coop.h) carries over unchanged.
//
// Pool capacity.  Popping k items can push at most 2k, so the pool grows by at most k per step.  With
// RESERVE = BLAS stack need + 3 free entries kept back, a step pops min(64, items, free - RESERVE) items; when that is
// < 1 it pops exactly one (depth-first mode), whose subtree never needs more than need + 1 entries above it.  The pool
// therefore never overflows whatever its size; 64 * (need + 2) entries keep it in wide mode in practice.
#pragma once

namespace rdx {

// Pool capacity: 64 * BLAS stack need entries, but no more than lets the wave's LDS stay within 160 KB / 24 -- 6 waves per
// SIMD, what the registers allow (the engine lives on residency, tools/occupancy_probe.sh) -- and no less than an instance
// step needs to push 64 roots.
#ifndef POOL_LDS_WORDS
#define POOL_LDS_WORDS 1706u
#endif
// ... and no more than POOL_CAP_MAX entries even when LDS would allow it: a smaller pool throttles the breadth of the walk
// (fewer items popped per step once it fills), which keeps a ray's subtree closer to depth-first order -- better for the culled
// walk and for cache locality.  Measured with the bitmap top level (tools/gpu_pool_var.sh, ms per 1080p frame, sample1 /
// Sponza-class / 10.4 M triangles): 256 -> 15.5 / 32.4 / 84.4, 384 -> 14.5 / 32.3 / 83.5, 512 -> 14.6 / 32.8 / 83.7,
// LDS-bound 832 -> 14.4 / 34.6 / 88.6.
#ifndef POOL_CAP_MAX
#define POOL_CAP_MAX 384u
#endif
// Rays in LDS.  A lane's object-space ray lives in its LDS slot of 8 words -- rays[lane] = o.xyz | instance slot,
// rays[64 + lane] = d.xyz | flags -- written when the lane enters an instance, and a pool / test step reads the ray of its item
// with two 128-bit LDS loads.  (Until round 2 the ray stayed in the owning lane's registers and was fetched with 7-8 lane
// shuffles per step: ds_bpermute sits on the critical path of every step, and the 10 registers are better spent on a sixth
// wave per SIMD.)  There is ONE slot per lane: a lane enters its next instance only after the queued tests of the one it left
// are done (`markPrev`).  Two slots (no such wait) measured 1.5 % faster at equal residency on the Sponza-class scene, but
// cost the 512 words of LDS that, with the registers, stand between 5 and 6 waves per SIMD.  1080p frame, sample1 /
// Sponza-class / 10.4 M triangles: shuffles 14.5 / 32.3 / 83.0 ms, two slots 14.0 / 30.0 / 82.7, one slot and 6 waves
// 13.8 / 28.0 / 72.8 (7 waves: 14.5 / 28.0 / 73.9 -- spills).
#ifndef POOL_PIECE
#define POOL_PIECE 4u                  // triangles one enqueue call takes per lane (a leaf of more goes in pieces) ...
#define POOL_PIECE_BITS 3u
#endif
#define POOL_QCAP (64u * POOL_PIECE)   // ... and the queue ring holds exactly one such call: 256 entries
#define POOL_RA(RAYS, LANE) ((RAYS)[(LANE)])
#define POOL_RB(RAYS, LANE) ((RAYS)[64u + (LANE)])
#define POOL_RAY_LDS_WORDS (64u * 8u)
static_assert(POOL_QCAP >= 64u * POOL_PIECE && (POOL_QCAP & (POOL_QCAP - 1u)) == 0u, "queue ring too small for one enqueue");
__host__ __device__ inline uint32_t pool_fixed_words() { return 64u + POOL_QCAP + POOL_RAY_LDS_WORDS + 128u; }
__host__ __device__ inline uint32_t pool_cap(uint32_t topNeed, uint32_t blasNeed)
{
    const uint32_t used = topNeed * 64u + pool_fixed_words();
    const uint32_t budget = used < POOL_LDS_WORDS ? ((POOL_LDS_WORDS - used) & ~63u) : 0u;
    uint32_t cap = 64u * (blasNeed ? blasNeed : 1u);
    if (cap > budget) cap = budget;
    if (cap > POOL_CAP_MAX) cap = POOL_CAP_MAX;
    const uint32_t least = (64u + blasNeed + 3u + 64u + 63u) & ~63u;       // 64 roots + RESERVE + one wide step
    return cap < least ? least : cap;
}
__host__ __device__ inline uint32_t pool_words_per_wave(uint32_t topNeed, uint32_t blasNeed)
{
    return topNeed * 64u + pool_cap(topNeed, blasNeed) + pool_fixed_words();
}

#define POOL_LANE_SHIFT 26u            // pool item = owning lane << 26 | wide-node index
#define POOL_NODE_MASK ((1u << POOL_LANE_SHIFT) - 1u)
static_assert(POOL_NODE_MASK + 1u >= RDX_COOP_MAX_WIDE, "the host's fallback rule (derive_accel) must match the pool item layout");
#define POOL_INBLAS 0xfffffffeu        // top-level cursor of a lane whose instance is in the pool
#ifndef POOL_TEST_MIN
#define POOL_TEST_MIN 16u              // queued triangle tests a test step waits for (final engine, with POOL_REPEAT 16: 16 / 24 / 32 -> 24.09 / 24.11 / 24.31 ms, Sponza-class)
#endif
// POOL_REPEAT: pool steps run back to back (while the pool holds a full batch) before the lane states are looked at again --
// the step selection at the top of the loop is a good part of a step's scalar instructions.  1080p frame, sample1 /
// Sponza-class / 10.4 M triangles: 1 -> 13.62 / 27.64 / 72.06 ms, 2 -> 13.52 / 27.34 / 70.92, 3 -> 13.51 / 27.23 / 70.21,
// 4 -> 13.48 / 27.15 / 70.36.  Re-swept on the final engine (r02e; test threshold 24): 3 -> 13.36 / 24.58 / 54.18, 6 -> 13.32 / 24.23 /
// 53.43, 12 -> 13.30 / 24.10 / 53.26, 32 -> 13.24 / 24.11 / 53.63; 16 with a test threshold of 16: 13.24 / 24.09 / 53.21.
#ifndef POOL_REPEAT
#define POOL_REPEAT 16u
#endif
#ifndef POOL_W_TOP
#define POOL_W_TOP 8                   // weights (in quarters) of a lane waiting for a top-level / instance step against a pooled node
#endif
#ifndef POOL_W_INST
#define POOL_W_INST 12
#endif
#ifndef POOL_MAX_ITER
#define POOL_MAX_ITER (1u << 22)       // iterations of one wave before it gives up (a full 1080p frame needs ~1e5 per wave)
#endif
// Whole-path policies (Policy::kShades; kernels.hip "whole paths"): a wave OWNS its paths from the camera ray to the end.  A
// closest hit is written to the path's state and the path id goes into the wave's SHADE queue; the lane is free for another
// ray at once.  When 64 paths wait there, a shade step runs the closest-hit shader on them -- all lanes busy, whatever the
// lanes' own rays are doing (it touches path state only) -- and the survivors go into the wave's RAY queue, from which free
// lanes are refilled before new camera rays are taken.  Two small rings in LDS behind the engine's own words; nothing crosses
// a wave, so no launch ever waits for another and a frame is one ramp and one drain.
#define POOL_RQ_CAP 256u               // ray queue (entries: path ids).  Bound: camera rays are taken only while it is empty, so a
#define POOL_SQ_CAP 128u               // wave owns < 64 + 128 paths; shade queue: a hand-over adds <= 64 to < 64 entries
#define POOL_PATH_LDS_WORDS (POOL_RQ_CAP + POOL_SQ_CAP)
#define POOL_QUEUED 0x80000000u        // work item = path id | POOL_QUEUED: a path from the ray queue (not a camera ray)
#ifndef POOL_QUAD_FETCH
#define POOL_QUAD_FETCH 0              // node records fetched by quads of lanes (quad_gather64) instead of per lane
#endif
#ifndef POOL_IDLE_MIN
#define POOL_IDLE_MIN 40               // finished / free lanes a hand-over step waits for (at 6 waves / SIMD: 24: +5 %, 32: +1-2 % frame time)
#endif


// Quad-cooperative gather of 64-byte records: the four lanes of a quad fetch, with ONE 16-byte load each, the record that one
// of them wants -- four loads serve the four lanes -- and a 4 x 4 transpose inside the quad (DPP quad_perm, no LDS) gives every
// lane its own record.  A per-lane gather makes the texture addresser look up 64 different cache lines per load instruction
// (256 per 64 records); here a load instruction touches 16 lines (64 per 64 records).
template <int C> __device__ __forceinline__ uint32_t quad_from(uint32_t v)      // value of lane (own + C) & 3 of the quad
{
    constexpr int ctrl = ((0 + C) & 3) | (((1 + C) & 3) << 2) | (((2 + C) & 3) << 4) | (((3 + C) & 3) << 6);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false);
}
template <int J> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v)     // value of lane J of the quad
{
    constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false);
}
__device__ __forceinline__ uint4 sel4(uint32_t k, const uint4& a, const uint4& b, const uint4& c, const uint4& d)   // k = 0..3
{
    const bool b0 = (k & 1u) != 0u, b1 = (k & 2u) != 0u;
    uint4 r;
    r.x = b1 ? (b0 ? d.x : c.x) : (b0 ? b.x : a.x); r.y = b1 ? (b0 ? d.y : c.y) : (b0 ? b.y : a.y);
    r.z = b1 ? (b0 ? d.z : c.z) : (b0 ? b.z : a.z); r.w = b1 ? (b0 ? d.w : c.w) : (b0 ? b.w : a.w);
    return r;
}
// out[c] = 16-byte piece c of the 64-byte record at index `rec` (per lane) of `table` (records of 4 uint4)
__device__ __forceinline__ void quad_gather64(const uint4* __restrict__ table, uint32_t rec, uint4 out[4])
{
    const uint32_t q = __lane_id() & 3u;
    // load j: the quad fetches the record of its lane j; lane q takes piece (q - j) & 3
    const uint4 v0 = table[(size_t)quad_bcast<0>(rec) * 4u + ((q - 0u) & 3u)];
    const uint4 v1 = table[(size_t)quad_bcast<1>(rec) * 4u + ((q - 1u) & 3u)];
    const uint4 v2 = table[(size_t)quad_bcast<2>(rec) * 4u + ((q - 2u) & 3u)];
    const uint4 v3 = table[(size_t)quad_bcast<3>(rec) * 4u + ((q - 3u) & 3u)];
    // piece c of lane j's record sits in lane (j + c) & 3, register v[j]: that lane presents v[(own - c) & 3]
    const uint4 t0 = sel4((q - 0u) & 3u, v0, v1, v2, v3), t1 = sel4((q - 1u) & 3u, v0, v1, v2, v3);
    const uint4 t2 = sel4((q - 2u) & 3u, v0, v1, v2, v3), t3 = sel4((q - 3u) & 3u, v0, v1, v2, v3);
    out[0] = t0;
    out[1] = make_uint4(quad_from<1>(t1.x), quad_from<1>(t1.y), quad_from<1>(t1.z), quad_from<1>(t1.w));
    out[2] = make_uint4(quad_from<2>(t2.x), quad_from<2>(t2.y), quad_from<2>(t2.z), quad_from<2>(t2.w));
    out[3] = make_uint4(quad_from<3>(t3.x), quad_from<3>(t3.y), quad_from<3>(t3.z), quad_from<3>(t3.w));
}

// one test step: up to 64 queued (ray slot, triangle) pairs, one per lane; the ray comes from the entry's LDS slot
__device__ __forceinline__ void pool_test_step(const AccelView& A, const uint32_t* queue, const float4* rays, unsigned long long* best,
                                               uint32_t lane, uint32_t& qHead, uint32_t qTail, float tmin, float tmax)
{
    const uint32_t n = min(64u, qTail - qHead);
#ifdef COOP_STATS
    if (lane == 0) { atomicAdd(&g_coop_stats[7], 1ull); atomicAdd(&g_coop_stats[15], (unsigned long long)n); }
#endif
    const uint32_t e = lane < n ? queue[(qHead + lane) & (POOL_QCAP - 1u)] : (lane << COOP_LANE_SHIFT);
    const float4* tp = reinterpret_cast<const float4*>(A.tris + (e & COOP_SLOT_MASK));      // requested first (idle lanes read slot 0)
    const float4 tq0 = tp[0], tq1 = tp[1], tq2 = tp[2];
    const float4 ra = POOL_RA(rays, e >> COOP_LANE_SHIFT), rb = POOL_RB(rays, e >> COOP_LANE_SHIFT);
    if (lane < n) {
        float t, b1, b2;
        if (coop_triangle_regs(tq0, tq1, tq2, mk3(ra.x, ra.y, ra.z), mk3(rb.x, rb.y, rb.z), tmin, tmax, t, b1, b2)) {
            // the instance slot comes from the triangle record when its BLAS belongs to one instance of the shared-transform group
            // (the ray slot then serves several instances at once), else from the ray slot; tq2.w = first triangle slot of the BLAS
            const uint32_t tinst = __float_as_uint(tq1.w);
            const uint32_t inst = tinst != 0xffffffffu ? tinst : __float_as_uint(ra.w);
            const uint32_t low = (inst << COOP_INST_SHIFT) | ((e & COOP_SLOT_MASK) - __float_as_uint(tq2.w));      // BLAS-local triangle slot
            atomicMin(&best[e >> COOP_LANE_SHIFT], ((unsigned long long)__float_as_uint(t) << 32) | low);
        }
    }
    qHead += n;
}
// append `cnt` (0..POOL_PIECE) consecutive triangle slots starting at `start` for every lane; wave-uniform control
__device__ __forceinline__ void pool_enqueue(const AccelView& A, uint32_t* queue, const float4* rays, unsigned long long* best, uint32_t lane,
                                             uint32_t tagBits, uint32_t cnt, uint32_t start, uint32_t& qHead, uint32_t& qTail, float tmin, float tmax)
{
    uint32_t pre = 0, total = 0;
#pragma unroll
    for (uint32_t b = 0; b < POOL_PIECE_BITS; ++b) {
        const unsigned long long m = __ballot((cnt >> b) & 1u);
        pre += lanes_below(m) << b;
        total += (uint32_t)__popcll(m) << b;
    }
    if (total == 0) return;
    while (qTail - qHead + total > POOL_QCAP) pool_test_step(A, queue, rays, best, lane, qHead, qTail, tmin, tmax);
    const uint32_t at = qTail + pre;
    for (uint32_t k = 0; k < cnt; ++k) queue[(at + k) & (POOL_QCAP - 1u)] = tagBits | (start + k);
    qTail += total;
}

// the same for the two leaf children of a pool item at once: `ca` slots from `sa`, then `cb` slots from `sb` (ca, cb <= POOL_PIECE) --
// one prefix sum over ca + cb instead of two
__device__ __forceinline__ void pool_enqueue2(const AccelView& A, uint32_t* queue, const float4* rays, unsigned long long* best, uint32_t lane,
                                              uint32_t tagBits, uint32_t ca, uint32_t sa, uint32_t cb, uint32_t sb, uint32_t& qHead,
                                              uint32_t& qTail, float tmin, float tmax)
{
    const uint32_t cnt = ca + cb;
    uint32_t pre = 0, total = 0;
#pragma unroll
    for (uint32_t b = 0; b < POOL_PIECE_BITS + 1u; ++b) {
        const unsigned long long m = __ballot((cnt >> b) & 1u);
        pre += lanes_below(m) << b;
        total += (uint32_t)__popcll(m) << b;
    }
    if (total == 0) return;
    if (total > POOL_QCAP) {       // more than the ring holds (over half the lanes with two full leaves): one side at a time
        pool_enqueue(A, queue, rays, best, lane, tagBits, ca, sa, qHead, qTail, tmin, tmax);
        pool_enqueue(A, queue, rays, best, lane, tagBits, cb, sb, qHead, qTail, tmin, tmax);
        return;
    }
    while (qTail - qHead + total > POOL_QCAP) pool_test_step(A, queue, rays, best, lane, qHead, qTail, tmin, tmax);
    const uint32_t at = qTail + pre;
    for (uint32_t k = 0; k < ca; ++k) queue[(at + k) & (POOL_QCAP - 1u)] = tagBits | (sa + k);
    for (uint32_t k = 0; k < cb; ++k) queue[(at + ca + k) & (POOL_QCAP - 1u)] = tagBits | (sb + k);
    qTail += total;
}

// One half of a quad record (rdx_types.h DQuad) against the item's ray: entries A and B.  Inner entry: entered iff its box is
// hit.  Leaf entry of a QUAD_PAIR half: its box is the box of the skipped inner node -- its triangles are tested iff that box
// is hit; leaf entry otherwise: tested unconditionally, as the reference does with a leaf it pops.  Rays with a (nearly) zero
// direction component (exactOnly: the slab decision is not monotone under box inclusion there) also test the skipped node's
// own box -- a leaf entry's box, or the union of the two inner entries'.
__device__ __forceinline__ void quad_half(const RayInst& Q, const float4 a0, const float4 a1, const float4 b0, const float4 b1,
                                          uint32_t& pushA, uint32_t& pushB, uint32_t& runA, uint32_t& runB)
{
    // run = count << 25 | first triangle slot (WIDE_SLOT_BITS), 0 = nothing to test
    const uint32_t ad0 = __float_as_uint(a0.w), ad1 = __float_as_uint(a1.w), bd0 = __float_as_uint(b0.w), bd1 = __float_as_uint(b1.w);
    const bool leafA = (ad1 & WIDE_LEAF) != 0u, leafB = (bd1 & WIDE_LEAF) != 0u, pair = (ad1 & QUAD_PAIR) != 0u;
    bool sA = slab_fast(Q, mk3(a0.x, a0.y, a0.z), mk3(a1.x, a1.y, a1.z));
    bool sB = slab_fast(Q, mk3(b0.x, b0.y, b0.z), mk3(b1.x, b1.y, b1.z));
    // (Belt and braces.  The only box that can pass while the skipped node's box fails is one that is FLAT in the plane the ray
    // travels in -- origin on the face, direction component exactly zero: 0 / 0 on both sides, which minnum / maxnum ignore --
    // and every triangle under such a box is coplanar with the ray: det == 0, rejected.  With or without this block the results
    // are the reference's (-DPOOL_EXP_NO_EXACT_UNION passes the whole suite); with it the visit set is, too.)
#ifndef POOL_EXP_NO_EXACT_UNION
    if (Q.exactOnly && pair) {
        bool u;
        if (!leafA && !leafB)
            u = slab_fast(Q, mk3(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z)), mk3(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z)));
        else u = leafA ? sA : sB;                  // a leaf entry of the pair carries the skipped node's box
        sA = sA && u; sB = sB && u;
    }
#endif
    if (leafA) { if (sA || !pair) runA = (wide_count(ad1) << WIDE_SLOT_BITS) | ad0; } else if (sA) pushA = ad0;
    if (leafB) { if (sB || !pair) runB = (wide_count(bd1) << WIDE_SLOT_BITS) | bd0; } else if (sB) pushB = bd0;
}

// INL: the scene has instances whose BLAS is a single leaf of <= 8 triangles; they are handled inside the top-level step
// (below).  A separate instantiation, chosen by the host per scene: the kernel sits at its register budget, and the extra
// code costs scenes without such instances 10-15 % through spills even when it never runs.
// CULL: the culled walk (kernels.hip "culled walk"): closest-hit rays skip subtrees entered beyond the best t so far and
// push the nearer child on top; every ray skips leaves whose box it misses.  The push order then depends on the ray, so
// A.blasNeed must be the any-order stack need (the host passes that one when the option is on).
// QUAD: the exhaustive walk over quad records (rdx_types.h DQuad; never with CULL): separate kernels (kernels.hip k_*_pool_q), so
// that neither walk carries the other's registers.
template <int REC, bool INL, bool CULL, class Policy, bool QUAD = false>
__device__ __forceinline__ void traverse_pool(const AccelView& A, const Policy& pol, uint32_t n, uint32_t* __restrict__ counter,
                                              float tmin, float tmax, uint32_t* __restrict__ lds)
{
    const uint32_t lane = __lane_id();
    const uint32_t PCAP = pool_cap(A.topNeed, A.blasNeed), RESERVE = A.blasNeed + 3u;
    uint32_t* tstack = lds + lane;                         // [level * 64]: top-level entries of this lane's ray
    uint32_t* pool = lds + A.topNeed * 64u;
    uint32_t* pendN = pool + PCAP;                         // [64] outstanding pool items of the lane's current instance
    CoopLds L;
    L.stack = nullptr;
    L.queue = pendN + 64;
    L.ray = reinterpret_cast<float*>(L.queue + POOL_QCAP);
    L.best = reinterpret_cast<unsigned long long*>(L.ray + POOL_RAY_LDS_WORDS);
    L.pend = nullptr;
    float4* rays = reinterpret_cast<float4*>(L.ray);       // [lane * 2 + parity][2]
    uint32_t* rayQ = reinterpret_cast<uint32_t*>(L.best + 64);      // whole-path policies only (the kernel allocates POOL_PATH_LDS_WORDS more)
    uint32_t* shadeQ = rayQ + POOL_RQ_CAP;
    uint32_t rqHead = 0, rqTail = 0, sqHead = 0, sqTail = 0;        // wave-uniform
#define POOL_TEST() pool_test_step(A, L.queue, rays, L.best, lane, qHead, qTail, tmin, tmax)
#define POOL_ENQ(TAG, CNT, START) pool_enqueue(A, L.queue, rays, L.best, lane, TAG, CNT, START, qHead, qTail, tmin, tmax)
    pendN[lane] = 0u;

    const uint32_t nWavesGrid = gridDim.x * (blockDim.x >> 6);
    const uint32_t chunkMax = (REC == 1) ? 64u : (uint32_t)COOP_CHUNK_MAX;
    uint32_t chunk = min(chunkMax, max(64u, (n / (4u * nWavesGrid)) & ~63u));

    uint32_t qHead = 0, qTail = 0, poolTop = 0;            // wave-uniform
    bool exhausted = false;
    bool lastOfAll = false;
    uint32_t resBase = 0, resEnd = 0;
    uint32_t rayIdx = COOP_NONE;
    uint32_t tcur = COOP_NONE, tsp = 0;                    // top-level cursor / stack pointer (flat top level: `tsp` counts the instances
                                                           // still pending in the lane's bitmap instead)
    uint32_t markPrev = 0;                                 // queue position the lane waits for: the tests of the instance it left before it
                                                           // enters the next one -- and, once its walk is over, all of its tests (`finishing`)
    bool finishing = false, anyHit = (REC == 2);
    typename Policy::State st{};
    f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);

    // Flat top level (A.topFlat != 0): the instances a ray still has to enter are a per-lane BITMAP over the instance slots in
    // LDS (A.topNeed = ceil(instances / 32) words per lane) plus their count in a register, instead of a stack of mask
    // entries -- a quarter to a tenth of the LDS (25 instances: 1 word instead of 9 per lane), which is residency.
    const bool flatTop = A.topFlat != 0u;
    // flat mode: tsp = pending instances (bits 0-9) | of which in the shared-transform group (bits 10-19) | bit 31: the lane's ray
    // slot holds the group's object-space ray.  Shared-transform group (rdx_runtime.cpp derive_accel): instances whose inverse
    // matrices are bit-identical have the same object-space ray, so a lane enters ALL of them on one ray slot -- without waiting,
    // between two of them, for the subtree and the queued tests of the first to drain -- before it turns to the others.
#define POOL_ILEFT (tsp & 0x3ffu)
#define POOL_GLEFT ((tsp >> 10) & 0x3ffu)
#define POOL_SLOTGROUP (1u << 31)
    const bool grouping = flatTop && A.groupCount != 0u;
    if (flatTop) for (uint32_t w = 0; w < A.topNeed; ++w) tstack[w * 64u] = 0u;
#define POOL_INSTBITS (TAG_INST | IDX_MASK)            // top-level cursor in flat mode: "take the next instance from the bitmap"
#define POOL_TPOP() do {                                                                               \
        if (flatTop) tcur = POOL_ILEFT ? POOL_INSTBITS : COOP_NONE;                                    \
        else if (tsp == 0) tcur = COOP_NONE; else { --tsp; tcur = tstack[tsp * 64u]; }                 \
    } while (0)
#define POOL_DROP() do {       /* the ray needs nothing more from the top level (shadow ray answered); items of the group's */ \
                               /* instances still in the pool are discarded when they are popped, the lane waits for that    */ \
        if (flatTop && POOL_ILEFT) { for (uint32_t w_ = 0; w_ < A.topNeed; ++w_) tstack[w_ * 64u] = 0u; }  \
        tsp = 0;                                                                                       \
        tcur = (grouping && pendN[lane] != 0u) ? POOL_INBLAS : COOP_NONE;                               \
    } while (0)
#define POOL_FILE_INSTANCES(M16, FIRST) do {            /* up to 16 instances FIRST.. (wave-uniform) with lane mask M16 */ \
        if (flatTop) {                                                                                 \
            const uint32_t wi_ = (FIRST) >> 5, sh_ = (FIRST) & 31u;                                    \
            atomicOr(&tstack[wi_ * 64u], (M16) << sh_);                                                \
            if (sh_ > 16u && ((M16) >> (32u - sh_)) != 0u) atomicOr(&tstack[(wi_ + 1u) * 64u], (M16) >> (32u - sh_)); \
            tsp += (uint32_t)__popc(M16);                                                              \
            if (grouping) {      /* (A.groupBits holds one word more than the bitmap) */                \
                const uint32_t gs_ = (uint32_t)(((((unsigned long long)A.groupBits[wi_ + 1u]) << 32) | A.groupBits[wi_]) >> sh_) & 0xffffu; \
                tsp += (uint32_t)__popc((M16) & gs_) << 10;                                            \
            }                                                                                          \
        } else { tstack[tsp * 64u] = TAG_INST | ((M16) << COOP_IMASK_SHIFT) | (FIRST); ++tsp; }        \
    } while (0)
#define POOL_START_RAY(WALK) do {                                                                      \
        L.best[lane] = ~0ull;                                                                          \
        tsp = 0; markPrev = qHead; finishing = false;                                         \
        tcur = (WALK) ? (TAG_TLAS | 0u) : COOP_NONE;                                                   \
    } while (0)

    // Bounded loop: the step selection below is meant to make progress on every iteration; should a logic error ever break
    // that, the wave leaves after POOL_MAX_ITER iterations and raises the status word the host checks after the frame
    // (rdx_trace_rays then returns an error) instead of hanging the GPU.  One scalar add + compare per iteration.
    bool finished = false;
#ifdef COOP_STATS
    uint32_t statN[8] = {0, 0, 0, 0, 0, 0, 0, 0}, statL[8] = {0, 0, 0, 0, 0, 0, 0, 0}, statKind = 7u;      // as traverse_coop.h; kind 6 = pool step
    unsigned long long statC[8] = {0, 0, 0, 0, 0, 0, 0, 0}, statT = __builtin_readcyclecounter();
    uint32_t stState[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (uint32_t iter = 0; iter < POOL_MAX_ITER; ++iter) {
#ifdef COOP_STATS
        { const unsigned long long now_ = __builtin_readcyclecounter(); statC[statKind] += now_ - statT; statT = now_; statKind = 7u; }
#endif
        // a lane whose instance has left the pool moves on along its top-level stack
        if (tcur == POOL_INBLAS && pendN[lane] == 0u) { POOL_TPOP(); markPrev = qTail; }       // the tests of the instance just left
        if (rayIdx != COOP_NONE && tcur == COOP_NONE && !finishing) { finishing = true; markPrev = qTail; }
        const bool done = finishing && (int32_t)(qHead - markPrev) >= 0;
        const bool isFree = (rayIdx == COOP_NONE);
        const unsigned long long doneMask = __ballot(done), freeMask = __ballot(isFree);
        const bool has = (tcur != COOP_NONE) && (tcur != POOL_INBLAS);
        const uint32_t tag = tcur & TAG_MASK;
        const bool isTop = has && tag == TAG_TLAS;
        bool isInst = has && tag == TAG_INST;
        const int nTop = __popcll(__ballot(isTop));
        int nInst = __popcll(__ballot(isInst));
        bool chainInst = false;          // the top-level step hands its rays straight to an instance step (below)
        const int nPool = (int)min(64u, poolTop);
        const bool workAny = (nTop | nInst) != 0 || poolTop != 0u;
        // whole-path policies: free lanes can also be refilled from the wave's own ray queue
        const uint32_t rqN = Policy::kShades ? rqTail - rqHead : 0u, sqN = Policy::kShades ? sqTail - sqHead : 0u;
        const int nIdle = __popcll(doneMask) + ((exhausted && rqN == 0u) ? 0 : __popcll(freeMask));
#ifdef COOP_STATS
        stState[0] += __popcll(__ballot(tcur == POOL_INBLAS)); stState[1] += nTop; stState[2] += nInst;
        stState[4] += __popcll(__ballot(finishing && !done)); stState[5] += __popcll(doneMask); stState[6] += __popcll(freeMask); stState[7] += 64;
#endif

        if (Policy::kShades && (rqN > POOL_RQ_CAP || sqN > POOL_SQ_CAP)) break;       // (cannot happen: see the bounds above; reported like the iteration bound)
        // ---- shade step (whole-path policies): 64 queued closest hits, or what there is when the wave has nothing else -----
        if constexpr (Policy::kShades) if (sqN != 0u && (sqN >= 64u || (!workAny && qTail == qHead && nIdle == 0))) {
            const uint32_t ns = min(64u, sqN);
            COOP_STAT(1, ns);
            bool cont = false;
            uint32_t sp = 0;
            if (lane < ns) { sp = shadeQ[(sqHead + lane) & (POOL_SQ_CAP - 1u)]; cont = pol.shade_path(sp, st); }
            sqHead += ns;
            const unsigned long long cm = __ballot(cont);
            if (cont) rayQ[(rqTail + lanes_below(cm)) & (POOL_RQ_CAP - 1u)] = sp;
            rqTail += (uint32_t)__popcll(cm);
            continue;
        }
        // ---- hand-over: finished rays are written out, free lanes take new rays -----------------------------------
        // (the pool usually holds 64+ items, so -- unlike traverse_coop.h -- the trigger is an absolute number of idle lanes)
        if (nIdle > 0 && (nIdle >= POOL_IDLE_MIN || !workAny)) {
            COOP_STAT(0, nIdle);
            bool pushS = false;
            uint32_t pushP = 0;
            if (done) {
                pushP = rayIdx & ~POOL_QUEUED;
                Best B;
                B.t = FLT_MAX; B.b1 = 0.f; B.b2 = 0.f; B.slot = 0; B.inst = RDX_MISS; B.hit = false;
                const unsigned long long key = L.best[lane];
                if (key != ~0ull) {
                    const uint32_t low = (uint32_t)key;
                    const uint32_t inst = low >> COOP_INST_SHIFT;
                    const DInst& I = A.insts[inst];
                    B.slot = I._p0 + (low & COOP_LOCAL_MASK);
                    B.hit = true; B.inst = inst;
                    B.t = __uint_as_float((uint32_t)(key >> 32));
                    if (!anyHit) {       // b1 / b2 recomputed with the arithmetic of the accepting test
                        const f3 ro = mat4_mul3(I.inv, o.x, o.y, o.z, 1.0f);
                        const f3 rd = mat4_mul3(I.inv, d.x, d.y, d.z, 0.0f);
                        float t, b1, b2;
                        coop_triangle(A, B.slot, ro, rd, tmin, tmax, t, b1, b2);
                        B.t = t; B.b1 = b1; B.b2 = b2;
                    }
                }
                bool ah = anyHit;
                const int act = pol.finish(rayIdx, B, o, d, ah, st);
                finishing = false;
                if (act == COOP_RELEASE) rayIdx = COOP_NONE;
                else if (Policy::kShades && act == COOP_SHADE) { pushS = true; rayIdx = COOP_NONE; }       // the path goes to the wave's shade queue, the lane is free
                else { anyHit = (REC == 2) || (REC == 3 && ah); POOL_START_RAY(true); }
            }
            uint32_t fromQueue = 0;
            if constexpr (Policy::kShades) {
                {   // closest hits of this hand-over go to the shade queue (< 64 entries before: the shade step above comes first)
                    const unsigned long long sm = __ballot(pushS);
                    if (pushS) shadeQ[(sqTail + lanes_below(sm)) & (POOL_SQ_CAP - 1u)] = pushP;
                    sqTail += (uint32_t)__popcll(sm);
                }
                // free lanes are refilled from the wave's ray queue first
                const bool want = (rayIdx == COOP_NONE);
                const unsigned long long wm = __ballot(want);
                fromQueue = min((uint32_t)__popcll(wm), rqTail - rqHead);
                const uint32_t rank = lanes_below(wm);
                if (want && rank < fromQueue) {
                    const uint32_t idx = rayQ[(rqHead + rank) & (POOL_RQ_CAP - 1u)] | POOL_QUEUED;
                    rayIdx = idx;
                    bool ah = false;
                    const bool walk = pol.load(idx, o, d, ah, st);
                    anyHit = (REC == 2) || (REC == 3 && ah);
                    POOL_START_RAY(walk);
                }
                rqHead += fromQueue;
            }
            // ... and with new work items (camera rays for whole paths: only while the ray queue is empty, which bounds the paths a wave owns)
            if (!exhausted && (!Policy::kShades || rqHead == rqTail)) {
                const bool want = (rayIdx == COOP_NONE);
                const unsigned long long wm = __ballot(want);
                uint32_t cnt = (uint32_t)__popcll(wm);
                if (cnt) {
                    if (resBase == resEnd) {                // reserve a chunk of ray indices (guided self-scheduling, traverse_coop.h)
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(counter, chunk);
                        b = __builtin_amdgcn_readfirstlane(b);
                        resBase = min(b, n); resEnd = min(b + chunk, n);
                        lastOfAll = (b + chunk >= n);
                        const uint32_t left = n - resEnd;
                        chunk = min(chunkMax, max(64u, (left / (4u * nWavesGrid)) & ~63u));
                    }
                    cnt = min(cnt, resEnd - resBase);
                    const uint32_t base = resBase;
                    resBase += cnt;
                    if (resBase == resEnd && lastOfAll) exhausted = true;
                    const uint32_t rank = lanes_below(wm);
                    if (want && rank < cnt) {
                        const uint32_t idx = base + rank;
                        rayIdx = idx;
                        bool ah = false;
                        const bool walk = pol.load(idx, o, d, ah, st);
                        anyHit = (REC == 2) || (REC == 3 && ah);
                        POOL_START_RAY(walk);
                    }
                }
            }
            continue;
        }
        if (!workAny) {
            if (qTail != qHead) { POOL_TEST(); continue; }
            if (__ballot(rayIdx != COOP_NONE) == 0ull && rqN == 0u && sqN == 0u) {
                finished = true; break;              // exhausted, every lane free, queue and pool empty
            }
            continue;                                              // lanes still finishing: next round hands them over
        }
        if (qTail - qHead >= POOL_TEST_MIN) { POOL_TEST(); continue; }

        // ---- top-level node (radiance.cl:110-150) --------------------------------------------------------------------
        if (nTop > 0 && nTop * POOL_W_TOP >= nPool * 4 && nTop * POOL_W_TOP >= nInst * POOL_W_INST &&
            (A.unifiedRoot == 0u || PCAP - poolTop >= 64u + RESERVE)) {
            COOP_STAT(4, nTop);
            if (REC != 1) { if (anyHit && isTop && L.best[lane] != ~0ull) POOL_DROP(); }
            if (A.unifiedRoot != 0u) {
                // Unified tree (rdx_runtime.cpp derive_accel): top level, instances and BLASes are ONE tree of wide records on ONE
                // object-space ray -- every instance has the identity transform.  The ray is taken through the (identity) inverse
                // with the reference's expressions, goes into the lane's slot, and one item -- the super-root -- enters the pool.
                bool push = false;
                if (isTop && tcur != COOP_NONE) {
                    const float4* ip = reinterpret_cast<const float4*>(A.insts);          // (the same matrix for every instance)
                    float m[16];
                    *reinterpret_cast<float4*>(m + 0) = ip[0];
                    *reinterpret_cast<float4*>(m + 4) = ip[1];
                    *reinterpret_cast<float4*>(m + 8) = ip[2];
                    *reinterpret_cast<float4*>(m + 12) = ip[3];
                    const f3 ro = mat4_mul3(m, o.x, o.y, o.z, 1.0f), rdv = mat4_mul3(m, d.x, d.y, d.z, 0.0f);
                    const f3 rc = mk3(__builtin_amdgcn_rcpf(rdv.x), __builtin_amdgcn_rcpf(rdv.y), __builtin_amdgcn_rcpf(rdv.z));
                    const float amin = fminf(fminf(fabsf(rdv.x), fabsf(rdv.y)), fabsf(rdv.z));
                    const float amax = fmaxf(fmaxf(fabsf(rc.x), fabsf(rc.y)), fabsf(rc.z));
                    const bool exactOnly = !(amin > 1e-20f) || !(amax < 1e20f);
                    POOL_RA(rays, lane) = make_float4(ro.x, ro.y, ro.z, __uint_as_float(0u));
                    POOL_RB(rays, lane) = make_float4(rdv.x, rdv.y, rdv.z, __uint_as_float(((exactOnly ? 1u : 0u) | (anyHit ? 4u : 0u)) << 29));
                    pendN[lane] = 1u; tcur = POOL_INBLAS; push = true;
                }
                const unsigned long long pm = __ballot(push);
                if (push) pool[poolTop + lanes_below(pm)] = (lane << POOL_LANE_SHIFT) | (A.unifiedRoot & POOL_NODE_MASK);
                poolTop += (uint32_t)__popcll(pm);
                continue;
            }
            if (!INL && A.topFlat != 0u) {
                // Small top-level tree (<= 64 nodes): no walk.  Every node is looked at once, in index order (parents come
                // before their children, DFS pre-order), by all lanes of the step together.  `reach` says which nodes the
                // reference's walk arrives at: the root, and the children of every reached inner node whose box the ray
                // hits (same decision rule); reached leaves file their instances (after the pre-test) in the lane's bitmap.
                // Same visit set, 1 step instead of one per visited node.
                // The node records come in with ONE vector load -- lane i holds node i -- and likewise the pre-test boxes of
                // 64 instances at a time; the loop broadcasts them with v_readlane.  (Until round 2 every node and instance
                // was a scalar load the loop then waited for: the step cost 4.8 pool steps, a sixth of the Sponza-class
                // frame's traversal cycles and more than a quarter of sample1's, tools/coop_stats.py.)
                const float4* npv = reinterpret_cast<const float4*>(A.tnodes + min(lane, A.topFlat - 1u));
                const float4 vb0 = npv[0], vb1 = npv[1];
                const uint4 vw = *reinterpret_cast<const uint4*>(npv + 2);
                uint32_t ibase = 0;                              // the 64 instances whose boxes the lanes hold
                float4 vmin = make_float4(0.f, 0.f, 0.f, -1.f), vmax = vmin;
                if (A.numInsts != 0u) {
                    vmin = *reinterpret_cast<const float4*>(A.insts[min(lane, A.numInsts - 1u)].worldMin);
                    vmax = *reinterpret_cast<const float4*>(A.insts[min(lane, A.numInsts - 1u)].worldMax);
                }
#define POOL_RLF(V, L) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(V), (int)(L)))
#define POOL_RLU(V, L) ((uint32_t)__builtin_amdgcn_readlane((int)(V), (int)(L)))
                const bool mine = isTop && tcur != COOP_NONE;
                RayInst W;
                W.o = o; W.d = d;
                W.rcp = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
                const float amin_ = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
                const float amax_ = fmaxf(fmaxf(fabsf(W.rcp.x), fabsf(W.rcp.y)), fabsf(W.rcp.z));
                W.exactOnly = !(amin_ > 1e-20f) || !(amax_ < 1e20f);
                const float oMax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
                const bool preOK = !W.exactOnly && (oMax < 1e20f);
                unsigned long long reach = mine ? 1ull : 0ull;
                for (uint32_t i = 0; i < A.topFlat; ++i) {
                    const uint32_t wx = POOL_RLU(vw.x, i), wy = POOL_RLU(vw.y, i), wz = POOL_RLU(vw.z, i);
                    const bool r = ((reach >> i) & 1ull) != 0ull;
                    if (__ballot(r) == 0ull) continue;          // no ray of the step gets here
                    if (!(wx & LEAF_BIT)) {
                        const f3 bmin = mk3(POOL_RLF(vb0.x, i), POOL_RLF(vb0.y, i), POOL_RLF(vb0.z, i));
                        const f3 bmax = mk3(POOL_RLF(vb1.x, i), POOL_RLF(vb1.y, i), POOL_RLF(vb1.z, i));
                        if (r && slab_fast(W, bmin, bmax)) reach |= (1ull << wx) | (1ull << wy);
                    } else if (wz == TYPE_INST) {
                        const uint32_t count = wx & 0x7fffffffu;
                        for (uint32_t b0 = 0; b0 < count; b0 += 16u) {
                            uint32_t m16 = 0;
                            for (uint32_t k = 0; k < min(16u, count - b0); ++k) {
                                const uint32_t ci = wy + b0 + k;              // the same instance for every lane here
                                if (ci - ibase >= 64u) {                      // (scenes of more than 64 instances: the next 64 boxes)
                                    ibase = ci & ~63u;
                                    const uint32_t li = min(ibase + lane, A.numInsts - 1u);
                                    vmin = *reinterpret_cast<const float4*>(A.insts[li].worldMin);
                                    vmax = *reinterpret_cast<const float4*>(A.insts[li].worldMax);
                                }
                                const uint32_t il = ci - ibase;
                                // conservative world-space pre-test (coop_inst_pretest), the instance's boxes broadcast from lane `il`
                                const float wc = POOL_RLF(vmin.w, il);
                                bool enter = true;
                                if (preOK && wc >= 0.0f) {
                                    const float m = wc * (oMax + POOL_RLF(vmax.w, il));
                                    const f3 tA = (mk3(POOL_RLF(vmin.x, il) - m, POOL_RLF(vmin.y, il) - m, POOL_RLF(vmin.z, il) - m) - o) * W.rcp;
                                    const f3 tB = (mk3(POOL_RLF(vmax.x, il) + m, POOL_RLF(vmax.y, il) + m, POOL_RLF(vmax.z, il) + m) - o) * W.rcp;
                                    const float tNear = fmaxf(fmaxf(fminf(tA.x, tB.x), fminf(tA.y, tB.y)), fminf(tA.z, tB.z));
                                    const float tFar = fminf(fminf(fmaxf(tA.x, tB.x), fmaxf(tA.y, tB.y)), fmaxf(tA.z, tB.z));
                                    const float n0 = fmaxf(tNear, 0.0f);
                                    const float band = 4.8e-7f * (fabsf(tFar) + n0) + 1e-30f;
                                    enter = !((tFar - n0) < -band);
                                }
                                if (enter) m16 |= 1u << k;
                            }
                            if (r && m16) POOL_FILE_INSTANCES(m16, wy + b0);
                        }
                    }
                }
                if (mine) POOL_TPOP();
#undef POOL_RLF
#undef POOL_RLU
                // the rays that found instances enter their first one right away, together with the lanes already waiting for an
                // instance step: one step that serves ~48 lanes instead of two that serve ~40 and ~23
                isInst = (tcur != COOP_NONE) && (tcur != POOL_INBLAS) && (tcur & TAG_MASK) == TAG_INST;
                nInst = __popcll(__ballot(isInst));
                chainInst = nInst > 0 && PCAP - poolTop >= 64u + RESERVE;
                if (!chainInst) continue;
                goto pool_instance_step;
            }
            if (A.topFlat != 0u) {       // (INL: scenes with single-leaf instances, handled on the spot; node and instance records as scalar loads)
                // Small top-level tree (<= 64 nodes): no walk.  Every node is looked at once, in index order (parents come
                // before their children, DFS pre-order), by all lanes of the step together -- the node is the same for all
                // of them, so its box comes through scalar loads and nothing is fetched along a dependent chain.  `reach`
                // says which nodes the reference's walk arrives at: the root, and the children of every reached inner node
                // whose box the ray hits (same decision rule); reached leaves file their instances (after the pre-test) as
                // mask entries.  Same visit set, 1 step instead of one per visited node.
                if (isTop && tcur != COOP_NONE) {
                    RayInst W;
                    W.o = o; W.d = d;
                    W.rcp = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
                    const float amin_ = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
                    const float amax_ = fmaxf(fmaxf(fabsf(W.rcp.x), fabsf(W.rcp.y)), fabsf(W.rcp.z));
                    W.exactOnly = !(amin_ > 1e-20f) || !(amax_ < 1e20f);
                    const float oMax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
                    const bool preOK = !W.exactOnly && (oMax < 1e20f);
                    unsigned long long reach = 1ull;
                    for (uint32_t i = 0; i < A.topFlat; ++i) {
                        const float4* np = reinterpret_cast<const float4*>(A.tnodes + i);
                        const uint4 w = *reinterpret_cast<const uint4*>(np + 2);
                        const bool r = ((reach >> i) & 1ull) != 0ull;
                        if (!(w.x & LEAF_BIT)) {
                            const float4 bmin = np[0], bmax = np[1];
                            if (r && slab_fast(W, mk3(bmin.x, bmin.y, bmin.z), mk3(bmax.x, bmax.y, bmax.z))) reach |= (1ull << w.x) | (1ull << w.y);
                        } else if (w.z == TYPE_INST && r) {
                            const uint32_t count = w.x & 0x7fffffffu;
                            if (INL && w.w != 0u) {
                                // this leaf holds instances whose BLAS is one small leaf (a quad: 2 triangles; flagged by
                                // derive_accel in the node's spare word).  They are dealt with on the spot: matrix and
                                // triangles are the same for all lanes of the step (scalar loads), each lane transforms its
                                // own ray with the reference's expressions and runs the branch-free Moeller-Trumbore -- no
                                // instance step, no parked ray, no queued tests for them.
                                for (uint32_t b0 = 0; b0 < count; b0 += 16u) {
                                    uint32_t m16 = 0;
                                    for (uint32_t k = 0; k < min(16u, count - b0); ++k) {
                                        const uint32_t ci = w.y + b0 + k;              // the same instance for every lane here
                                        const DInst& I = A.insts[ci];
                                        const uint32_t rcnt = wide_count(I.rootDesc1);
                                        if ((I.rootDesc1 & WIDE_LEAF) && rcnt <= 8u) {
                                            if (!((REC != 1) && anyHit && L.best[lane] != ~0ull)) {
                                                const f3 ro = mat4_mul3(I.inv, o.x, o.y, o.z, 1.0f);
                                                const f3 rd = mat4_mul3(I.inv, d.x, d.y, d.z, 0.0f);
                                                for (uint32_t j = 0; j < rcnt; ++j) {
                                                    float t, b1, b2;
                                                    if (coop_triangle(A, wide_slot(I.rootDesc0) + j, ro, rd, tmin, tmax, t, b1, b2)) {
                                                        const uint32_t low = (ci << COOP_INST_SHIFT) | (wide_slot(I.rootDesc0) + j - I._p0);
                                                        atomicMin(&L.best[lane], ((unsigned long long)__float_as_uint(t) << 32) | low);
                                                    }
                                                }
                                            }
                                        } else if (coop_inst_pretest(I, o, W.rcp, oMax, preOK)) m16 |= 1u << k;
                                    }
                                    if (m16) POOL_FILE_INSTANCES(m16, w.y + b0);
                                }
                            } else {
                                for (uint32_t b0 = 0; b0 < count; b0 += 16u) {
                                    uint32_t m16 = 0;
                                    for (uint32_t k = 0; k < min(16u, count - b0); ++k)
                                        if (coop_inst_pretest(A.insts[w.y + b0 + k], o, W.rcp, oMax, preOK)) m16 |= 1u << k;
                                    if (m16) POOL_FILE_INSTANCES(m16, w.y + b0);
                                }
                            }
                        }
                    }
                    POOL_TPOP();
                }
                isInst = (tcur != COOP_NONE) && (tcur != POOL_INBLAS) && (tcur & TAG_MASK) == TAG_INST;
                nInst = __popcll(__ballot(isInst));
                chainInst = nInst > 0 && PCAP - poolTop >= 64u + RESERVE;
                if (chainInst) goto pool_instance_step;
                continue;
            }
            if (isTop && tcur != COOP_NONE) {
                const float4* np = reinterpret_cast<const float4*>(A.ctnodes + (tcur & IDX_MASK));
                const float4 bmin = np[0], bmax = np[1];
                const uint4 w = *reinterpret_cast<const uint4*>(np + 2);
                RayInst W;
                W.o = o; W.d = d;
                W.rcp = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
                const float amin_ = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
                const float amax_ = fmaxf(fmaxf(fabsf(W.rcp.x), fabsf(W.rcp.y)), fabsf(W.rcp.z));
                W.exactOnly = !(amin_ > 1e-20f) || !(amax_ < 1e20f);
                if (!(w.x & LEAF_BIT)) {
                    if (slab_fast(W, mk3(bmin.x, bmin.y, bmin.z), mk3(bmax.x, bmax.y, bmax.z))) { tstack[tsp * 64u] = TAG_TLAS | w.y; ++tsp; tcur = TAG_TLAS | w.x; }
                    else POOL_TPOP();
                } else {
                    const uint32_t count = w.x & 0x7fffffffu;
                    if (w.z == TYPE_INST) {
                        const float oMax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
                        const bool preOK = !W.exactOnly && (oMax < 1e20f);
                        for (uint32_t b0 = 0; b0 < count; b0 += 16u) {
                            uint32_t m16 = 0;
                            for (uint32_t i = 0; i < min(16u, count - b0); ++i)
                                if (coop_inst_pretest(A.insts[w.y + b0 + i], o, W.rcp, oMax, preOK)) m16 |= 1u << i;
                            if (m16) POOL_FILE_INSTANCES(m16, w.y + b0);
                        }
                    }
                    POOL_TPOP();
                }
            }
            continue;
        }
        // ---- instance entry (radiance.cl:161-169): the BLAS root goes into the pool ----------------------------------
    pool_instance_step:
        if (chainInst || (nInst > 0 && nInst * POOL_W_INST >= nPool * 4 && PCAP - poolTop >= 64u + RESERVE)) {
            // a lane with instances of the shared-transform group pending enters one whatever it has in the pool or in the test queue
            // (same ray slot); any other instance waits for the queued tests of the instance left before (`markPrev`)
            const bool grpNext = grouping && POOL_GLEFT != 0u;
            const bool ready = isInst && (grpNext || (int32_t)(qHead - markPrev) >= 0);
            if (__ballot(ready) == 0ull) { POOL_TEST(); continue; }
            COOP_STAT(5, __popcll(__ballot(ready)));
            if (REC != 1) { if (anyHit && ready && L.best[lane] != ~0ull) POOL_DROP(); }
            uint32_t cntE = 0, stE = 0, rootNode = COOP_NONE;
            if (ready && tcur != COOP_NONE && tcur != POOL_INBLAS) {
                uint32_t ci = tcur & COOP_IFIRST_MASK;
                if (flatTop) {       // the lowest pending instance of the bitmap (of the group first)
                    ci = 0;
                    for (uint32_t w_ = 0; w_ < A.topNeed; ++w_) {
                        const uint32_t v_ = tstack[w_ * 64u];
                        const uint32_t c_ = grpNext ? (v_ & A.groupBits[w_]) : v_;
                        if (c_) { const uint32_t b_ = (uint32_t)__ffs((int)c_) - 1u; ci = w_ * 32u + b_; tstack[w_ * 64u] = v_ & ~(1u << b_); break; }
                    }
                    tsp -= grpNext ? 0x401u : 1u;
                } else {
                    const uint32_t m16 = (tcur >> COOP_IMASK_SHIFT) & 0xffffu, rest = m16 & (m16 - 1u);
                    ci += (uint32_t)__ffs((int)m16) - 1u;
                    if (rest) { tstack[tsp * 64u] = TAG_INST | (rest << COOP_IMASK_SHIFT) | (tcur & COOP_IFIRST_MASK); ++tsp; }
                }
                const float4* ip = reinterpret_cast<const float4*>(A.insts + ci);
                const uint4 rdsc = *reinterpret_cast<const uint4*>(ip + 9);    // rootDesc0, rootDesc1, triBase, -
                RayInst R;
                if (grpNext && (tsp & POOL_SLOTGROUP)) {
                    // the slot already holds the group's ray: no matrix, no transform, no slot write
                    const float4 ra = POOL_RA(rays, lane), rb = POOL_RB(rays, lane);
                    R.o = mk3(ra.x, ra.y, ra.z); R.d = mk3(rb.x, rb.y, rb.z);
                    R.rcp = mk3(__builtin_amdgcn_rcpf(R.d.x), __builtin_amdgcn_rcpf(R.d.y), __builtin_amdgcn_rcpf(R.d.z));
                    R.exactOnly = ((__float_as_uint(rb.w) >> 29) & 1u) != 0u;
                } else {
                    float m[16];
                    *reinterpret_cast<float4*>(m + 0) = ip[0];
                    *reinterpret_cast<float4*>(m + 4) = ip[1];
                    *reinterpret_cast<float4*>(m + 8) = ip[2];
                    *reinterpret_cast<float4*>(m + 12) = ip[3];
                    // the ray in the instance's space goes into this lane's slot: the queued tests of the instance left before are
                    // done (`ready`)
                    R.o = mat4_mul3(m, o.x, o.y, o.z, 1.0f);
                    R.d = mat4_mul3(m, d.x, d.y, d.z, 0.0f);
                    R.rcp = mk3(__builtin_amdgcn_rcpf(R.d.x), __builtin_amdgcn_rcpf(R.d.y), __builtin_amdgcn_rcpf(R.d.z));
                    const float amin = fminf(fminf(fabsf(R.d.x), fabsf(R.d.y)), fabsf(R.d.z));
                    const float amax = fmaxf(fmaxf(fabsf(R.rcp.x), fabsf(R.rcp.y)), fabsf(R.rcp.z));
                    R.exactOnly = !(amin > 1e-20f) || !(amax < 1e20f);
                    // slot words 3 / 7: instance slot (instances outside the group); flags << 29 (bit 0 exactOnly, bit 2 anyHit)
                    POOL_RA(rays, lane) = make_float4(R.o.x, R.o.y, R.o.z, __uint_as_float(ci));
                    POOL_RB(rays, lane) = make_float4(R.d.x, R.d.y, R.d.z, __uint_as_float(((R.exactOnly ? 1u : 0u) | (anyHit ? 4u : 0u)) << 29));
                    if (grpNext) tsp |= POOL_SLOTGROUP;
                }
                if (rdsc.y & WIDE_LEAF) {      // (never an instance of the group: derive_accel keeps leaf roots out of it)
                    cntE = wide_count(rdsc.y); stE = wide_slot(rdsc.x);
                    POOL_TPOP();
                } else {
                    const float4 rmin = ip[10], rmax = ip[11];
                    bool enter;
                    if (CULL) {
                        float tnRoot;
                        enter = slab_fast_t(R, mk3(rmin.x, rmin.y, rmin.z), mk3(rmax.x, rmax.y, rmax.z), tnRoot);
                        const uint32_t hb = reinterpret_cast<const uint32_t*>(L.best)[2u * lane + 1u];
                        // gate of the whole BLAS: the root descriptor's normal cone, every vertex is inside the root box
                        const CullGate G = cull_gate(R, cull_ray(R), rdsc.x, rdsc.y, false, mk3(rmin.x, rmin.y, rmin.z), mk3(rmax.x, rmax.y, rmax.z));
                        float tlim = tmax;
                        if (!anyHit && hb != 0xffffffffu) tlim = fminf(tlim, __uint_as_float(hb));
                        if (tnRoot > cull_limit(G, tlim)) enter = false;
                    } else enter = slab_fast(R, mk3(rmin.x, rmin.y, rmin.z), mk3(rmax.x, rmax.y, rmax.z));
                    if (enter) rootNode = rdsc.x;
                    if (!grpNext) {
                        if (enter) { pendN[lane] = 1u; tcur = POOL_INBLAS; }
                        else POOL_TPOP();
                    } else {
                        // the group's instances are all in the pool together: the count is the sum over them.  The lane asks for the
                        // next instance step while the group has more, then waits for the pool (top of the loop), then moves on
                        const uint32_t pn = pendN[lane] + (enter ? 1u : 0u);
                        if (enter) pendN[lane] = pn;
                        if (POOL_GLEFT == 0u) {
                            if (pn != 0u) tcur = POOL_INBLAS;
                            else { POOL_TPOP(); markPrev = qTail; }      // (the slot serves the group's queued tests until they are done)
                        }
                    }
                }
            }
            {
                const unsigned long long pm = __ballot(rootNode != COOP_NONE);
                if (rootNode != COOP_NONE) pool[poolTop + lanes_below(pm)] = (lane << POOL_LANE_SHIFT) | (rootNode & POOL_NODE_MASK);
                poolTop += (uint32_t)__popcll(pm);
            }
            const uint32_t tagBits = lane << COOP_LANE_SHIFT;
            const bool leafRoot = cntE != 0u;
            while (__any(cntE != 0u)) {      // (a leaf of more than POOL_PIECE triangles goes in pieces)
                const uint32_t c = min(cntE, POOL_PIECE);
                POOL_ENQ(tagBits, c, stE);
                stE += c; cntE -= c;
            }
            if (leafRoot) markPrev = qTail;      // the slot serves these tests until they are done
            continue;
        }
        // ---- pool step: up to 64 pending BLAS nodes, whichever rays they belong to -----------------------------------
        if (poolTop != 0u) {
          uint32_t rep = 0;
          if constexpr (QUAD && !CULL) {
            // ---- quad records (rdx_types.h DQuad): two levels of the tree per item; up to four children go back into the pool.
            // The two halves are fetched one after the other (same 128-byte line: the second is an L1 hit): a whole record in
            // registers costs 16 more than the kernel has.
            do {
                const uint32_t freeE = PCAP - poolTop;
                const uint32_t npop = min(min(64u, poolTop), freeE >= RESERVE + 3u ? (freeE - RESERVE) / 3u : 1u);
                COOP_STAT(6, npop);
                const bool valid = lane < npop;
                const uint32_t item = valid ? pool[poolTop - 1u - lane] : (lane << POOL_LANE_SHIFT);
                poolTop -= npop;
                const uint32_t wl = item >> POOL_LANE_SHIFT;
                const float4* qp = reinterpret_cast<const float4*>(A.quad + (item & POOL_NODE_MASK));
                const uint32_t slotBits = wl << POOL_LANE_SHIFT;
                uint32_t pushA = COOP_NONE, pushB = COOP_NONE, pushC = COOP_NONE, pushE = COOP_NONE, runA = 0, runB = 0, runC = 0, runE = 0;
                {
                    const float4 a0 = qp[0], a1 = qp[1], b0 = qp[2], b1 = qp[3];
                    const float4 ra = POOL_RA(rays, wl), rb = POOL_RB(rays, wl);
                    const uint32_t qf = __float_as_uint(rb.w) >> 29;
                    RayInst Q;
                    Q.o = mk3(ra.x, ra.y, ra.z);
                    Q.d = mk3(rb.x, rb.y, rb.z);
                    Q.rcp = mk3(__builtin_amdgcn_rcpf(Q.d.x), __builtin_amdgcn_rcpf(Q.d.y), __builtin_amdgcn_rcpf(Q.d.z));
                    Q.exactOnly = (qf & 1u) != 0u;
                    bool live = false;
                    if (valid) {
                        const uint32_t hb = reinterpret_cast<const uint32_t*>(L.best)[2u * wl + 1u];     // t bits of the owner's best candidate
                        live = !((REC != 1) && (qf & 4u) && hb != 0xffffffffu);                     // (a shadow ray already answered drops its items)
                    }
                    if (live) quad_half(Q, a0, a1, b0, b1, pushA, pushB, runA, runB);
#ifndef POOL_QUAD_ALL_AT_ONCE // the second half is fetched after the first is done (same 128-byte line): the record then costs 16
                              // registers, not 32, and the kernels fit 6 waves per SIMD -- all eight loads in flight at 5 waves
                              // measured slower (25.0 vs 23.7 ms, Sponza-class)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    const float4 c0 = qp[4], c1 = qp[5], e0 = qp[6], e1 = qp[7];
                    if (live) quad_half(Q, c0, c1, e0, e1, pushC, pushE, runC, runE);
                }
                {   // entry A ends on top: the host put the subtree with the smallest pool need there (derive_accel)
                    const unsigned long long mE = __ballot(pushE != COOP_NONE), mC = __ballot(pushC != COOP_NONE);
                    const unsigned long long mB = __ballot(pushB != COOP_NONE), mA = __ballot(pushA != COOP_NONE);
                    uint32_t at = poolTop;
                    if (pushE != COOP_NONE) pool[at + lanes_below(mE)] = slotBits | (pushE & POOL_NODE_MASK);
                    at += (uint32_t)__popcll(mE);
                    if (pushC != COOP_NONE) pool[at + lanes_below(mC)] = slotBits | (pushC & POOL_NODE_MASK);
                    at += (uint32_t)__popcll(mC);
                    if (pushB != COOP_NONE) pool[at + lanes_below(mB)] = slotBits | (pushB & POOL_NODE_MASK);
                    at += (uint32_t)__popcll(mB);
                    if (pushA != COOP_NONE) pool[at + lanes_below(mA)] = slotBits | (pushA & POOL_NODE_MASK);
                    poolTop = at + (uint32_t)__popcll(mA);
                    const int delta = (valid ? -1 : 0) + (pushA != COOP_NONE ? 1 : 0) + (pushB != COOP_NONE ? 1 : 0) + (pushC != COOP_NONE ? 1 : 0) + (pushE != COOP_NONE ? 1 : 0);
                    if (delta != 0) atomicAdd(&pendN[wl], (uint32_t)delta);
                }
                while (__any((runA | runB | runC | runE) >> WIDE_SLOT_BITS)) {
                    uint32_t ca = min(runA >> WIDE_SLOT_BITS, POOL_PIECE), cb = min(runB >> WIDE_SLOT_BITS, POOL_PIECE);
                    pool_enqueue2(A, L.queue, rays, L.best, lane, slotBits, ca, runA & WIDE_SLOT_MASK, cb, runB & WIDE_SLOT_MASK, qHead, qTail, tmin, tmax);
                    runA = runA - (ca << WIDE_SLOT_BITS) + ca; runB = runB - (cb << WIDE_SLOT_BITS) + cb;
                    ca = min(runC >> WIDE_SLOT_BITS, POOL_PIECE); cb = min(runE >> WIDE_SLOT_BITS, POOL_PIECE);
                    pool_enqueue2(A, L.queue, rays, L.best, lane, slotBits, ca, runC & WIDE_SLOT_MASK, cb, runE & WIDE_SLOT_MASK, qHead, qTail, tmin, tmax);
                    runC = runC - (ca << WIDE_SLOT_BITS) + ca; runE = runE - (cb << WIDE_SLOT_BITS) + cb;
                }
                if (qTail - qHead >= POOL_TEST_MIN) POOL_TEST();
            } while (++rep < POOL_REPEAT && poolTop >= 64u);
            continue;
          }
          do {
            const uint32_t freeE = PCAP - poolTop;
            const uint32_t npop = min(min(64u, poolTop), freeE > RESERVE ? freeE - RESERVE : 1u);
            COOP_STAT(6, npop);
            const bool valid = lane < npop;
            const uint32_t item = valid ? pool[poolTop - 1u - lane] : (lane << POOL_LANE_SHIFT);
            poolTop -= npop;
            const uint32_t wl = item >> POOL_LANE_SHIFT;
            // the wide node is requested FIRST (idle lanes read node 0), so that its memory latency runs beside the lane shuffles
            // below instead of behind them
#if POOL_QUAD_FETCH
            // quad-cooperative fetch (quad_gather64): a quarter of the cache-line / page look-ups of a per-lane gather
            float4 l0, l1, r0, r1;
            {
                uint4 nd[4];
                quad_gather64(reinterpret_cast<const uint4*>(A.wide), item & POOL_NODE_MASK, nd);
                l0 = make_float4(__uint_as_float(nd[0].x), __uint_as_float(nd[0].y), __uint_as_float(nd[0].z), __uint_as_float(nd[0].w));
                l1 = make_float4(__uint_as_float(nd[1].x), __uint_as_float(nd[1].y), __uint_as_float(nd[1].z), __uint_as_float(nd[1].w));
                r0 = make_float4(__uint_as_float(nd[2].x), __uint_as_float(nd[2].y), __uint_as_float(nd[2].z), __uint_as_float(nd[2].w));
                r1 = make_float4(__uint_as_float(nd[3].x), __uint_as_float(nd[3].y), __uint_as_float(nd[3].z), __uint_as_float(nd[3].w));
            }
#else
            const float4* wp = reinterpret_cast<const float4*>(A.wide + (item & POOL_NODE_MASK));
            const float4 l0 = wp[0], l1 = wp[1], r0 = wp[2], r1 = wp[3];
#endif
            // the object-space ray of the item, from its LDS slot
            const uint32_t slotBits = wl << POOL_LANE_SHIFT;
            const float4 ra = POOL_RA(rays, wl), rb = POOL_RB(rays, wl);
            const uint32_t qf = __float_as_uint(rb.w) >> 29;
            RayInst Q;
            Q.o = mk3(ra.x, ra.y, ra.z);
            Q.d = mk3(rb.x, rb.y, rb.z);
            Q.rcp = mk3(__builtin_amdgcn_rcpf(Q.d.x), __builtin_amdgcn_rcpf(Q.d.y), __builtin_amdgcn_rcpf(Q.d.z));
            Q.exactOnly = (qf & 1u) != 0u;
            uint32_t cntL = 0, stL = 0, cntR = 0, stR = 0, pushL = COOP_NONE, pushR = COOP_NONE;
            int delta = 0;
            if (valid) {
                delta = -1;
                const uint32_t hb = reinterpret_cast<const uint32_t*>(L.best)[2u * wl + 1u];     // t bits of the owner's best candidate
                const bool dropIt = (REC != 1) && (qf & 4u) && hb != 0xffffffffu;           // shadow ray already answered
                if (!dropIt) {
                    const uint32_t ld0 = __float_as_uint(l0.w), ld1 = __float_as_uint(l1.w);
                    const uint32_t rd0 = __float_as_uint(r0.w), rd1 = __float_as_uint(r1.w);
                    if (CULL) {
                        // each child's gate (kernels.hip "culled walk"): its normal cone against the ray, its box holds its vertices
                        const CullRay CR = cull_ray(Q);
                        const bool leafL = (ld1 & WIDE_LEAF) != 0u, leafR = (rd1 & WIDE_LEAF) != 0u;
                        const CullGate GL = cull_gate(Q, CR, ld0, ld1, leafL, mk3(l0.x, l0.y, l0.z), mk3(l1.x, l1.y, l1.z));
                        const CullGate GR = cull_gate(Q, CR, rd0, rd1, leafR, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z));
                        float tlim = tmax;
                        if (((REC == 1) || !(qf & 4u)) && hb != 0xffffffffu) tlim = fminf(tlim, __uint_as_float(hb));
                        float tnL = 0.f, tnR = 0.f;
                        if (cull_child(Q, mk3(l0.x, l0.y, l0.z), mk3(l1.x, l1.y, l1.z), leafL, GL, tlim, tnL)) {
                            if (leafL) { cntL = wide_count(ld1); stL = wide_slot(ld0); } else pushL = ld0;
                        }
                        if (cull_child(Q, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), leafR, GR, tlim, tnR)) {
                            if (leafR) { cntR = wide_count(rd1); stR = wide_slot(rd0); } else pushR = rd0;
                        }
                        // the nearer child goes on top of the LIFO (the "L" slot is written above the "R" slot below)
                        if (pushL != COOP_NONE && pushR != COOP_NONE && tnL > tnR) { const uint32_t t_ = pushL; pushL = pushR; pushR = t_; }
                    } else {
                        if (ld1 & WIDE_LEAF) { cntL = wide_count(ld1); stL = wide_slot(ld0); }
                        else if (slab_fast(Q, mk3(l0.x, l0.y, l0.z), mk3(l1.x, l1.y, l1.z))) pushL = ld0;
                        if (rd1 & WIDE_LEAF) { cntR = wide_count(rd1); stR = wide_slot(rd0); }
                        else if (slab_fast(Q, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z))) pushR = rd0;
                    }
#ifdef POOL_EXP_DOUBLE_SLAB       // experiment: what do two more slab tests cost? (results unchanged: ra.w never holds that pattern)
                    {
                        const bool x1 = slab_fast(Q, mk3(l0.y, l0.z, l0.x), mk3(l1.y, l1.z, l1.x));
                        const bool x2 = slab_fast(Q, mk3(r0.y, r0.z, r0.x), mk3(r1.y, r1.z, r1.x));
                        delta += (int)((uint32_t)x1 & (uint32_t)x2 & (uint32_t)(__float_as_uint(ra.w) == 0x12345678u));
                    }
#endif
                    delta += (pushL != COOP_NONE ? 1 : 0) + (pushR != COOP_NONE ? 1 : 0);
                }
            }
            {   // right children below left ones: the left half of a wide node holds the subtree with the smaller stack need
                // (CULL: the nearer child on top instead; the pool is then sized for any order)
                const unsigned long long mR = __ballot(pushR != COOP_NONE), mL = __ballot(pushL != COOP_NONE);
                const uint32_t nR = (uint32_t)__popcll(mR);
                if (pushR != COOP_NONE) pool[poolTop + lanes_below(mR)] = slotBits | (pushR & POOL_NODE_MASK);
                if (pushL != COOP_NONE) pool[poolTop + nR + lanes_below(mL)] = slotBits | (pushL & POOL_NODE_MASK);
                poolTop += nR + (uint32_t)__popcll(mL);
            }
            if (valid && delta != 0) atomicAdd(&pendN[wl], (uint32_t)delta);
            const uint32_t tagBits = slotBits;
            while (__any((cntL | cntR) != 0u)) {
                const uint32_t cl = min(cntL, POOL_PIECE), cr = min(cntR, POOL_PIECE);
                pool_enqueue2(A, L.queue, rays, L.best, lane, tagBits, cl, stL, cr, stR, qHead, qTail, tmin, tmax);
                stL += cl; cntL -= cl; stR += cr; cntR -= cr;
            }
            if (qTail - qHead >= POOL_TEST_MIN) POOL_TEST();
          } while (++rep < POOL_REPEAT && poolTop >= 64u);
            continue;
        }
        // (not reached: with an empty pool one of the two top-level branches above is always taken)
        if (qTail != qHead) POOL_TEST();
    }
    // the iteration bound was hit: the host reports an error.  (A plain system-scope store into the pinned host word: an
    // atomic would need PCIe atomics, and nothing else is ever written there.)
    if (!finished && lane == 0 && A.status) __hip_atomic_store(A.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef COOP_STATS
    if (lane < 8u) atomicAdd(&g_coop_state[lane], (unsigned long long)stState[lane]);
    if (lane < 8u) atomicAdd(&g_coop_cycles[lane], statC[lane]);
    if (lane < 7u) { atomicAdd(&g_coop_stats[lane], (unsigned long long)statN[lane]); atomicAdd(&g_coop_stats[8u + lane], (unsigned long long)statL[lane]); }
#endif
    pol.retire(st);
#undef POOL_TEST
#undef POOL_ENQ
#undef POOL_ILEFT
#undef POOL_GLEFT
#undef POOL_SLOTGROUP
#undef POOL_TPOP
#undef POOL_DROP
#undef POOL_FILE_INSTANCES
#undef POOL_INSTBITS
#undef POOL_START_RAY
}

} // namespace rdx

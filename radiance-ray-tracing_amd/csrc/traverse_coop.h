// traverse_coop.h -- wave-cooperative BVH traversal for wave64 (the production extend / shadow walk).
//
// Why this shape.  The reference's closest-hit walk (radiance/shader/radiance.cl:41-192) never culls
// by the best t found so far: which nodes and triangles a ray visits depends on the slab tests alone.
// Box traversal and triangle testing are therefore completely decoupled, and a triangle may be tested
// at any time, in any order, by anyone, as long as the winner is the candidate the reference would
// keep: the smallest t, ties going to the first one in the reference's DFS order = the lowest
// (instance slot, triangle slot) (bvh.cpp:487-497,551-563 number both in DFS-leaf order).
//
// A per-lane walk keeps only ~7-14 of 64 lanes busy on this workload (rocprofv3:
// SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU), because lanes drift apart between box tests,
// triangle runs of different lengths and the early-outs of Möller–Trumbore.  Here a wave instead
// alternates between two fully converged phases:
//
//   node step   every lane that holds a wide BLAS node does the same thing: one 64-byte fetch, two slab
//               tests.  Leaf children are not tested in place: their triangle slots are appended to a
//               per-wave queue in LDS (offsets from wave64 ballot prefix sums).
//   test step   when >= 64 triangle tests are queued (or nothing else is left), each lane takes ONE
//               queue entry -- possibly another lane's ray -- reads that ray's object-space origin /
//               direction from LDS, runs a branch-free Möller–Trumbore with the reference's arithmetic
//               and publishes an accepted candidate with a 64-bit LDS atomic-min on
//               (t bits << 32 | instance slot << 22 | local triangle slot).
//
// Top-level nodes and instance entries are rare and handled when no lane holds a BLAS node; the
// queue is drained before any lane switches instance, so a queued entry always refers to its owner's
// current object-space ray.  Any-hit (shadow) rays use the same machinery and drop their remaining
// work as soon as a candidate has been published.
//
// Limits (checked on the host, otherwise the per-lane kernels are used): <= 1024 instances,
// <= 4M triangles per BLAS, < 64M triangle slots in total.
#pragma once

namespace rdx {

#define COOP_QCAP 512u                 // queue ring capacity in entries (power of two)
#define COOP_LANE_SHIFT 26u            // queue entry = owner lane << 26 | absolute triangle slot
#define COOP_SLOT_MASK ((1u << 26) - 1u)
#define COOP_INST_SHIFT 22u            // key low word = instance slot << 22 | BLAS-local triangle slot
#define COOP_LOCAL_MASK ((1u << 22) - 1u)
#define COOP_NONE 0xffffffffu

__host__ __device__ inline uint32_t coop_words_per_wave(uint32_t need) { return need * 64u + COOP_QCAP + 6u * 64u + 64u + 64u + 128u; }

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

struct CoopLds {
    uint32_t* stack;                 // [need][64]
    uint32_t* queue;                 // [COOP_QCAP]
    float* ray;                      // [6][64] object-space origin.xyz, direction.xyz of each lane's current instance
    uint32_t* inst;                  // [64] current instance slot
    uint32_t* base;                  // [64] first triangle slot of the current BLAS
    unsigned long long* best;        // [64]
};

// branch-free Möller–Trumbore (radiance.cl:211-251) + the accept window of radiance.cl:90-91
__device__ __forceinline__ bool coop_triangle(const AccelView& A, uint32_t slot, f3 ro, f3 rd, float tmin, float tmax,
                                              float& tOut, float& b1Out, float& b2Out)
{
    const float4* tp = reinterpret_cast<const float4*>(A.tris + slot);
    const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
    const f3 e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q2.x, q2.y, q2.z);
    const f3 rce2 = cross3(rd, e2);
    const float det = dot3(e1, rce2);
    const float inv_det = 1.0f / det;
    const f3 s = ro - mk3(q0.x, q0.y, q0.z);
    const float b1 = inv_det * dot3(s, rce2);
    const f3 sce1 = cross3(s, e1);
    const float b2 = inv_det * dot3(rd, sce1);
    const float t = inv_det * dot3(e2, sce1);
    tOut = t; b1Out = b1; b2Out = b2;
    return (det != 0) & !(b1 < 0 || b1 > 1) & !(b2 < 0 || b1 + b2 > 1) & (t > 0) & (t > tmin) & (t < tmax);
}

// one test step: up to 64 queued (owner, triangle) pairs, one per lane
__device__ __forceinline__ void coop_test_step(const AccelView& A, const CoopLds& L, uint32_t lane, uint32_t& qHead,
                                               uint32_t qTail, float tmin, float tmax)
{
    const uint32_t n = min(64u, qTail - qHead);
    if (lane < n) {
        const uint32_t e = L.queue[(qHead + lane) & (COOP_QCAP - 1u)];
        const uint32_t owner = e >> COOP_LANE_SHIFT, slot = e & COOP_SLOT_MASK;
        const f3 ro = mk3(L.ray[0 * 64 + owner], L.ray[1 * 64 + owner], L.ray[2 * 64 + owner]);
        const f3 rd = mk3(L.ray[3 * 64 + owner], L.ray[4 * 64 + owner], L.ray[5 * 64 + owner]);
        float t, b1, b2;
        if (coop_triangle(A, slot, ro, rd, tmin, tmax, t, b1, b2)) {
            const uint32_t low = (L.inst[owner] << COOP_INST_SHIFT) | (slot - L.base[owner]);
            const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | low;
            atomicMin(&L.best[owner], key);
        }
    }
    qHead += n;
}

// append `cnt` (0..8) consecutive triangle slots starting at `start` for every lane; wave-uniform control
__device__ __forceinline__ void coop_enqueue(const AccelView& A, const CoopLds& L, uint32_t lane, uint32_t cnt, uint32_t start,
                                             uint32_t& qHead, uint32_t& qTail, float tmin, float tmax)
{
    uint32_t pre = 0, total = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const unsigned long long m = __ballot((cnt >> b) & 1u);
        pre += lanes_below(m) << b;
        total += (uint32_t)__popcll(m) << b;
    }
    if (total == 0) return;
    while (qTail - qHead + total > COOP_QCAP) coop_test_step(A, L, lane, qHead, qTail, tmin, tmax);
    const uint32_t at = qTail + pre;
    for (uint32_t k = 0; k < cnt; ++k) L.queue[(at + k) & (COOP_QCAP - 1u)] = (lane << COOP_LANE_SHIFT) | (start + k);
    qTail += total;
}

// All 64 lanes of the wave call this (inactive lanes pass active = false).
template <int REC>
__device__ __forceinline__ void traverse_coop(const AccelView& A, bool active, f3 o, f3 d, float tmin, float tmax,
                                              uint32_t* __restrict__ lds, uint32_t need, Best& B)
{
    const uint32_t lane = __lane_id();
    CoopLds L;
    L.stack = lds + lane;                                  // [level * 64]
    L.queue = lds + need * 64u;
    L.ray = reinterpret_cast<float*>(L.queue + COOP_QCAP);
    L.inst = reinterpret_cast<uint32_t*>(L.ray + 6 * 64);
    L.base = L.inst + 64;
    L.best = reinterpret_cast<unsigned long long*>(L.base + 64);
    L.best[lane] = ~0ull;
    L.inst[lane] = 0; L.base[lane] = 0;

    uint32_t qHead = 0, qTail = 0;                         // wave-uniform
    uint32_t sp = 0;
    uint32_t cur = active ? (TAG_TLAS | 0u) : COOP_NONE;
    RayInst R;
    R.o = o; R.d = d; R.rcp = mk3(0.f, 0.f, 0.f); R.exactOnly = true;

#define COOP_POP() do { if (sp == 0) cur = COOP_NONE; else { --sp; cur = L.stack[sp * 64u]; } } while (0)

    for (;;) {
        const bool isNode = (cur != COOP_NONE) && ((cur & TAG_MASK) == TAG_BLAS);
        if (__any(isNode)) {
            // ---- node step ------------------------------------------------------------------------
            uint32_t cntL = 0, stL = 0, cntR = 0, stR = 0, nextL = COOP_NONE, nextR = COOP_NONE;
            if (isNode) {
                const float4* wp = reinterpret_cast<const float4*>(A.wide + (cur & IDX_MASK));
                const float4 l0 = wp[0], l1 = wp[1], r0 = wp[2], r1 = wp[3];
                const uint32_t ld0 = __float_as_uint(l0.w), ld1 = __float_as_uint(l1.w);
                const uint32_t rd0 = __float_as_uint(r0.w), rd1 = __float_as_uint(r1.w);
                if (ld1 & WIDE_LEAF) {
                    cntL = ld1 & 0x7fffffffu; stL = ld0;
                    while (cntL > 8u) { L.stack[sp * 64u] = leaf_item(stL, 8u); ++sp; stL += 8u; cntL -= 8u; }   // oversized leaf: rare
                } else if (slab_fast(R, mk3(l0.x, l0.y, l0.z), mk3(l1.x, l1.y, l1.z))) {
                    nextL = ld0;
                }
                if (rd1 & WIDE_LEAF) {
                    cntR = rd1 & 0x7fffffffu; stR = rd0;
                    while (cntR > 8u) { L.stack[sp * 64u] = leaf_item(stR, 8u); ++sp; stR += 8u; cntR -= 8u; }
                } else if (slab_fast(R, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z))) {
                    nextR = rd0;
                }
                if (nextL != COOP_NONE) { if (nextR != COOP_NONE) { L.stack[sp * 64u] = nextR; ++sp; } cur = nextL; }
                else if (nextR != COOP_NONE) cur = nextR;
                else COOP_POP();
            }
            coop_enqueue(A, L, lane, cntL, stL, qHead, qTail, tmin, tmax);
            coop_enqueue(A, L, lane, cntR, stR, qHead, qTail, tmin, tmax);
            if (qTail - qHead >= 64u) {
                coop_test_step(A, L, lane, qHead, qTail, tmin, tmax);
                if (REC == 2) { if (cur != COOP_NONE && L.best[lane] != ~0ull) { cur = COOP_NONE; sp = 0; } }
            }
            continue;
        }
        const uint32_t tag = cur & TAG_MASK;
        const bool isLeaf = (cur != COOP_NONE) && tag == TAG_LEAF;
        if (__any(isLeaf)) {
            // ---- queued piece of an oversized leaf, or a leaf root -----------------------------------
            uint32_t cnt = 0, st = 0;
            if (isLeaf) { st = cur & LEAF_START_MASK; cnt = ((cur >> LEAF_START_BITS) & 7u) + 1u; COOP_POP(); }
            coop_enqueue(A, L, lane, cnt, st, qHead, qTail, tmin, tmax);
            continue;
        }
        const bool isTop = (cur != COOP_NONE) && tag == TAG_TLAS;
        if (__any(isTop)) {
            // ---- top-level node (radiance.cl:110-150) --------------------------------------------------
            if (isTop) {
                const float4* np = reinterpret_cast<const float4*>(A.tnodes + (cur & IDX_MASK));
                const float4 bmin = np[0], bmax = np[1];
                const uint4 w = *reinterpret_cast<const uint4*>(np + 2);
                if (!(w.x & LEAF_BIT)) {
                    if (slab_hit(o, d, bmin, bmax)) { L.stack[sp * 64u] = TAG_TLAS | w.y; ++sp; cur = TAG_TLAS | w.x; }
                    else COOP_POP();
                } else {
                    const uint32_t count = w.x & 0x7fffffffu;
                    if (w.z == TYPE_INST && count > 0) {
                        for (uint32_t i = count - 1; i >= 1; --i) { L.stack[sp * 64u] = TAG_INST | (w.y + i); ++sp; }
                        cur = TAG_INST | w.y;
                    } else COOP_POP();
                }
            }
            continue;
        }
        const bool isInst = (cur != COOP_NONE);                 // only TAG_INST items are left
        if (__any(isInst)) {
            // ---- instance entry: drain first, every queued entry refers to its owner's CURRENT instance ----
            while (qTail != qHead) coop_test_step(A, L, lane, qHead, qTail, tmin, tmax);
            if (REC == 2) { if (cur != COOP_NONE && L.best[lane] != ~0ull) { cur = COOP_NONE; sp = 0; } }
            if (cur != COOP_NONE) {
                const uint32_t ci = cur & IDX_MASK;
                const float4* ip = reinterpret_cast<const float4*>(A.insts + ci);
                float m[16];
                *reinterpret_cast<float4*>(m + 0) = ip[0];
                *reinterpret_cast<float4*>(m + 4) = ip[1];
                *reinterpret_cast<float4*>(m + 8) = ip[2];
                *reinterpret_cast<float4*>(m + 12) = ip[3];
                R.o = mat4_mul3(m, o.x, o.y, o.z, 1.0f);         // radiance.cl:161-169
                R.d = mat4_mul3(m, d.x, d.y, d.z, 0.0f);
                R.rcp = mk3(1.0f / R.d.x, 1.0f / R.d.y, 1.0f / R.d.z);
                const float amin = fminf(fminf(fabsf(R.d.x), fabsf(R.d.y)), fabsf(R.d.z));
                const float amax = fmaxf(fmaxf(fabsf(R.rcp.x), fabsf(R.rcp.y)), fabsf(R.rcp.z));
                R.exactOnly = !(amin > 1e-20f) || !(amax < 1e20f);
                L.ray[0 * 64 + lane] = R.o.x; L.ray[1 * 64 + lane] = R.o.y; L.ray[2 * 64 + lane] = R.o.z;
                L.ray[3 * 64 + lane] = R.d.x; L.ray[4 * 64 + lane] = R.d.y; L.ray[5 * 64 + lane] = R.d.z;
                const uint4 rdsc = *reinterpret_cast<const uint4*>(ip + 9);    // rootDesc0, rootDesc1, triBase, -
                L.inst[lane] = ci; L.base[lane] = rdsc.z;
                if (rdsc.y & WIDE_LEAF) {
                    uint32_t cnt = rdsc.y & 0x7fffffffu, st = rdsc.x;
                    while (cnt > 8u) { L.stack[sp * 64u] = leaf_item(st, 8u); ++sp; st += 8u; cnt -= 8u; }
                    if (cnt) cur = leaf_item(st, cnt); else COOP_POP();
                } else {
                    const float4 rmin = ip[10], rmax = ip[11];
                    if (slab_fast(R, mk3(rmin.x, rmin.y, rmin.z), mk3(rmax.x, rmax.y, rmax.z))) cur = rdsc.x;
                    else COOP_POP();
                }
            }
            continue;
        }
        // ---- nobody holds a work item: finish the queue and leave ----------------------------------------
        while (qTail != qHead) coop_test_step(A, L, lane, qHead, qTail, tmin, tmax);
        break;
    }
#undef COOP_POP

    // ---- decode the winner; b1 / b2 are recomputed with the arithmetic of the accepting test ------------
    B.t = FLT_MAX; B.b1 = 0.f; B.b2 = 0.f; B.slot = 0; B.inst = RDX_MISS; B.hit = false;
    const unsigned long long key = L.best[lane];
    if (active && key != ~0ull) {
        const uint32_t low = (uint32_t)key;
        const uint32_t inst = low >> COOP_INST_SHIFT;
        const DInst& I = A.insts[inst];
        const uint32_t slot = I._p0 + (low & COOP_LOCAL_MASK);
        B.hit = true; B.inst = inst; B.slot = slot;
        if (REC == 1) {
            const f3 ro = mat4_mul3(I.inv, o.x, o.y, o.z, 1.0f);
            const f3 rd = mat4_mul3(I.inv, d.x, d.y, d.z, 0.0f);
            float t, b1, b2;
            coop_triangle(A, slot, ro, rd, tmin, tmax, t, b1, b2);
            B.t = t; B.b1 = b1; B.b2 = b2;
        } else {
            B.t = __uint_as_float((uint32_t)(key >> 32));
        }
    }
}

} // namespace rdx

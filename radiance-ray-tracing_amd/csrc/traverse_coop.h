// traverse_coop.h -- persistent, wave-cooperative BVH traversal for wave64 (the production extend /
// shadow walk).
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
// cycles through a few fully converged steps:
//
//   node step   every lane that holds a wide BLAS node does the same thing: one 64-byte fetch, two slab
//               tests.  Leaf children are not tested in place: their triangle slots are appended to a
//               per-wave queue in LDS (offsets from wave64 ballot prefix sums).
//   test step   when >= 64 triangle tests are queued (or nothing else is left), each lane takes ONE
//               queue entry -- usually another lane's ray.  The object-space ray of the entry is fetched from
//               the walking lane's REGISTERS by lane shuffles (ds_bpermute); only if that lane has meanwhile
//               entered its next instance (the entry carries the parity of the instance it was queued under)
//               is it the ray parked in the lane's one LDS slot.  The lane runs a branch-free Möller–Trumbore
//               with the reference's arithmetic and publishes an accepted candidate with a 64-bit LDS
//               atomic-min on (t bits << 32 | instance slot << 22 | BLAS-local triangle slot).
//   top / instance steps   top-level nodes and instance entries.  The instances of a top-level leaf travel as one
//               16-bit mask entry; entering an instance parks the ray of the instance being left in LDS, so a
//               lane may run one instance ahead of its queued triangle tests (never two: markPrev).  A BLAS
//               that is a single leaf has its triangles queued right in the instance step.
//   Which step runs next is greedy on weighted lane counts: the kind the most lanes are waiting for, upstream
//   kinds (top, instance) counting 2x / 3x because they feed the node steps; waiting lanes batch up and every
//   step executes as converged as the moment allows (tools/coop_stats.py prints what the lanes do).
//   refill step the wave is persistent: a lane whose ray is finished (stack empty and all its queued
//               tests consumed) writes its result and takes the next ray index from the wave's reservation,
//               which is refilled from a global counter in chunks (guided self-scheduling), so lanes do not
//               idle while a long ray of the same wave is still walking.  The wave leaves when the counter is
//               exhausted, every lane is idle and the queue is empty -- an exit every wave reaches.
//
//   steal step  subtree sharing inside the wave.  Because nothing is culled by the best t, the subtrees on a
//               lane's stack are independent pieces of work whose only output is the atomic-min key of
//               the ray's owner -- so ANY lane may walk them.  Once the wave cannot get new rays (global
//               counter exhausted), free lanes take the bottom entry of a busy lane's stack (the largest
//               pending subtree) and walk it on behalf of that ray: the world ray comes over by lane shuffles,
//               a stolen BLAS-level entry re-enters the donor's instance through the ordinary instance step
//               (same matrix, same world ray => the same object-space ray), candidates go to best[owner], and
//               the owner hands its result over only when its helper count is back to 0.  Helpers donate in
//               turn, so one long ray spreads over the wave in a few steps and the tail of a launch is bounded
//               by the wave's remaining WORK / 64 instead of by its longest ray.
//
// LDS per wave = (stack need + 18) * 256 B: the stack [need][64], the 512-entry queue ring, the parked rays
// [7][64], best[64] (64-bit) and the helper counts [64].  Residency is LDS-bound and the kernel lives on residency
// (tools/occupancy_probe.sh), hence the small stack need (derive_accel: smaller subtree first) and the single slot.
//
// Any-hit (shadow) rays use the same machinery and drop their remaining work as soon as a candidate
// has been published.
//
// Limits (checked on the host, otherwise the per-lane kernels are used): <= 1024 instances,
// <= 4M triangles per BLAS, < 32M triangle slots in total.
#pragma once

namespace rdx {

#ifndef COOP_QCAP
#define COOP_QCAP 512u                 // queue ring capacity in entries: a power of two, and one enqueue (64 lanes x 8) must fit
#endif
static_assert(COOP_QCAP >= 512u && (COOP_QCAP & (COOP_QCAP - 1u)) == 0u, "queue ring too small for one enqueue");
#define COOP_LANE_SHIFT 26u            // queue entry = walking lane << 26 | instance parity << 25 | absolute triangle slot
#define COOP_PAR_SHIFT 25u
#define COOP_SLOT_MASK ((1u << 25) - 1u)
static_assert(COOP_SLOT_MASK + 1u == RDX_COOP_MAX_TRI_SLOTS, "the host's fallback rule (derive_accel) must match the queue entry layout");
#define COOP_INST_SHIFT 22u            // key low word = instance slot << 22 | BLAS-local triangle slot
#define COOP_LOCAL_MASK ((1u << 22) - 1u)
static_assert((1u << COOP_INST_SHIFT) == RDX_COOP_MAX_BLAS_TRIS && (RDX_COOP_MAX_INSTANCES << COOP_INST_SHIFT) == 0u, "key layout");
#define COOP_NONE 0xffffffffu
#define COOP_RAY_WORDS 7u              // per lane: o.xyz d.xyz (object space), instance slot | owner lane << 16 of the instance being left
#ifndef COOP_IDLE_BIAS
#define COOP_IDLE_BIAS 16              // a refill step needs this many more idle lanes than the busiest work kind has (tuned: profiles/)
#endif

#ifndef COOP_STEAL
#define COOP_STEAL 1                   // subtree sharing inside the wave (steal step)
#endif
#ifndef COOP_MIN_QUOTA
#define COOP_MIN_QUOTA 64u             // rays handed to a wave per refill, at least.  < 64 spreads small launches over more waves whose
                                       // other lanes help: faster below ~16 k rays, slower above (tools/gpu_variants.sh) -- off
#endif
#ifndef COOP_STEAL_MIN
#define COOP_STEAL_MIN 4               // a steal step needs this many (stealer, donor) pairs, or a quarter of the walking lanes
#endif
#ifndef COOP_TAIL_MIN
#define COOP_TAIL_MIN 4                // while stealing: finished lanes are handed over as soon as this many wait
#endif
#ifndef COOP_WATCHDOG
#define COOP_WATCHDOG 0               // debug builds: bound the iterations of a wave (costs ~3 % through code placement alone)
#endif
#ifndef COOP_MAX_ITER
#define COOP_MAX_ITER (1u << 24)
#endif
#ifndef COOP_W_TOP
#define COOP_W_TOP 8                   // weight (in quarters) of a lane waiting for a top-level step against one waiting for a node step
#endif
#ifndef COOP_W_INST
#define COOP_W_INST 12                 // the same for the instance step
#endif
#ifndef COOP_CHUNK_MAX
#define COOP_CHUNK_MAX 512u            // most ray indices a wave reserves with one atomic on the global counter (64 = one per refill)
#endif
#define COOP_OWNER_SHIFT 16u           // ray slot word 6 = instance slot | owner lane << 16
#define COOP_RESUME (1u << 29)         // TAG_INST work item of a helper: enter the instance, then continue with the stolen entry on the stack
// TAG_INST work item = up to 16 instances of one top-level leaf: first instance slot (13 bits) | 16-bit mask of the
// instances still to enter << 13.  One stack entry per 16 instances instead of one per instance.
#define COOP_IMASK_SHIFT 13u
#define COOP_IFIRST_MASK ((1u << COOP_IMASK_SHIFT) - 1u)

__host__ __device__ inline uint32_t coop_words_per_wave(uint32_t need) { return need * 64u + COOP_QCAP + COOP_RAY_WORDS * 64u + 128u + 64u; }

__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// COOP_STATS (experiment build only): per step kind, how often it ran and how many lanes it served, summed over the
// grid into g_coop_stats[kind] / [8 + kind]; kinds: 0 finish/refill, 1 shade, 2 steal, 3 leaf item, 4 top, 5 instance,
// 6 node, 7 test.  Read back with rdx_debug_coop_stats (tools/coop_stats.py).
#ifdef COOP_STATS
__device__ unsigned long long g_coop_stats[16];
__device__ unsigned long long g_coop_state[8];      // lane-iterations spent in: node, top, inst, leaf, finishing, done, free, iterations*64
__device__ unsigned long long g_coop_cycles[8];     // pool engine: wave cycles (s_memtime) per step kind; a test that follows a pool step counts as 6
#define COOP_STAT(kind, lanes) do { statN[kind] += 1u; statL[kind] += (uint32_t)(lanes); statKind = (kind); } while (0)
#else
#define COOP_STAT(kind, lanes) do {} while (0)
#endif

struct CoopLds {
    uint32_t* stack;                 // [need][64]   (already offset by the lane)
    uint32_t* queue;                 // [COOP_QCAP]
    float* ray;                      // [7 words][64 lanes]: the object-space ray of the instance a lane has just left
    unsigned long long* best;        // [64]
    uint32_t* pend;                  // [64] helpers still walking for the ray owned by this lane
};

// branch-free Möller–Trumbore (radiance.cl:211-251) + the accept window of radiance.cl:90-91, on a triangle record already in
// registers (q0 = v0 | primID, q1 = e1, q2 = e2)
__device__ __forceinline__ bool coop_triangle_regs(float4 q0, float4 q1, float4 q2, f3 ro, f3 rd, float tmin, float tmax,
                                                   float& tOut, float& b1Out, float& b2Out)
{
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
__device__ __forceinline__ bool coop_triangle(const AccelView& A, uint32_t slot, f3 ro, f3 rd, float tmin, float tmax,
                                              float& tOut, float& b1Out, float& b2Out)
{
    const float4* tp = reinterpret_cast<const float4*>(A.tris + slot);
    return coop_triangle_regs(tp[0], tp[1], tp[2], ro, rd, tmin, tmax, tOut, b1Out, b2Out);
}

// one test step: up to 64 queued (walking lane, triangle) pairs, one per lane.  The object-space ray of an entry is
// the walking lane's CURRENT one -- fetched from its registers by lane shuffles -- unless that lane has meanwhile
// entered its next instance (the entry's parity bit differs from the lane's): then it is the ray parked in the
// lane's LDS slot.  All 64 lanes must call this (the shuffles read the registers of active lanes).
__device__ __forceinline__ void coop_test_step(const AccelView& A, const CoopLds& L, uint32_t lane, uint32_t& qHead,
                                               uint32_t qTail, float tmin, float tmax, const RayInst& R, uint32_t par, uint32_t w6)
{
    const uint32_t n = min(64u, qTail - qHead);
#ifdef COOP_STATS
    if (lane == 0) { atomicAdd(&g_coop_stats[7], 1ull); atomicAdd(&g_coop_stats[15], (unsigned long long)n); }
#endif
#ifdef COOP_EXP_NOTEST            // timing experiment (results are wrong): consume the entries without testing them
    qHead += n; return;
#endif
    const uint32_t e = lane < n ? L.queue[(qHead + lane) & (COOP_QCAP - 1u)] : (lane << COOP_LANE_SHIFT);
    const uint32_t wl = e >> COOP_LANE_SHIFT;
    // the triangle record is requested FIRST (idle lanes read slot 0): its ~1 us of memory latency then runs beside the lane
    // shuffles below instead of behind them -- the LDS crossbar was on the critical path of every step
    const float4* tp = reinterpret_cast<const float4*>(A.tris + (e & COOP_SLOT_MASK));
    const float4 tq0 = tp[0], tq1 = tp[1], tq2 = tp[2];
    const uint32_t cpar = __shfl(par, wl);
    uint32_t w = __shfl(w6, wl);
    f3 ro = mk3(__shfl(R.o.x, wl), __shfl(R.o.y, wl), __shfl(R.o.z, wl));
    f3 rd = mk3(__shfl(R.d.x, wl), __shfl(R.d.y, wl), __shfl(R.d.z, wl));
    if (lane < n) {
        const uint32_t slot = e & COOP_SLOT_MASK;
        if (((e >> COOP_PAR_SHIFT) & 1u) != cpar) {
            const float* rs = L.ray + wl;
            ro = mk3(rs[0 * 64], rs[1 * 64], rs[2 * 64]);
            rd = mk3(rs[3 * 64], rs[4 * 64], rs[5 * 64]);
            w = __float_as_uint(rs[6 * 64]);
        }
        float t, b1, b2;
        if (coop_triangle_regs(tq0, tq1, tq2, ro, rd, tmin, tmax, t, b1, b2)) {
            const uint32_t inst = w & ((1u << COOP_OWNER_SHIFT) - 1u);
            const uint32_t low = (inst << COOP_INST_SHIFT) | (slot - A.insts[inst]._p0);      // BLAS-local triangle slot
            const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | low;
            atomicMin(&L.best[w >> COOP_OWNER_SHIFT], key);   // the candidate goes to the ray's owner
        }
    }
    qHead += n;
}

// append `cnt` (0..8) consecutive triangle slots starting at `start` for every lane; wave-uniform control
__device__ __forceinline__ void coop_enqueue(const AccelView& A, const CoopLds& L, uint32_t lane, uint32_t tagBits, uint32_t cnt,
                                             uint32_t start, uint32_t& qHead, uint32_t& qTail, float tmin, float tmax,
                                             const RayInst& R, uint32_t par, uint32_t w6)
{
#ifdef COOP_EXP_NOENQ             // timing experiment (results are wrong): box traversal only
    return;
#endif
    uint32_t pre = 0, total = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const unsigned long long m = __ballot((cnt >> b) & 1u);
        pre += lanes_below(m) << b;
        total += (uint32_t)__popcll(m) << b;
    }
    if (total == 0) return;
    while (qTail - qHead + total > COOP_QCAP) coop_test_step(A, L, lane, qHead, qTail, tmin, tmax, R, par, w6);
    const uint32_t at = qTail + pre;
#ifdef COOP_EXP_NOWRITE           // timing experiment (results are wrong): prefix sums only, nothing queued
    if (at == 0xffffffffu) qTail += total;
    return;
#endif
    for (uint32_t k = 0; k < cnt; ++k) L.queue[(at + k) & (COOP_QCAP - 1u)] = tagBits | (start + k);
    qTail += total;
}

// Conservative world-space rejection of an instance whose BLAS root is an inner node.  The reference
// would transform the ray and slab-test the root box in object space (radiance.cl:161-176, 61-63); if the
// world ray misses the world AABB of that box, grown by a margin that dominates every fp32 rounding the
// reference's own computation can incur, that test is known to fail and the instance entry
// can be skipped without changing any result.  Returns true when the instance must be entered.
__device__ __forceinline__ bool coop_inst_pretest(const DInst& I, f3 o, f3 rcpW, float oMax, bool preOK)
{
    const float4 wmin = *reinterpret_cast<const float4*>(I.worldMin), wmax = *reinterpret_cast<const float4*>(I.worldMax);
    if (!preOK || !(wmin.w >= 0.0f)) return true;
    const float m = wmin.w * (oMax + wmax.w);
    const f3 tA = (mk3(wmin.x - m, wmin.y - m, wmin.z - m) - o) * rcpW, tB = (mk3(wmax.x + m, wmax.y + m, wmax.z + m) - o) * rcpW;
    const float tNear = fmaxf(fmaxf(fminf(tA.x, tB.x), fminf(tA.y, tB.y)), fminf(tA.z, tB.z));
    const float tFar = fminf(fminf(fmaxf(tA.x, tB.x), fmaxf(tA.y, tB.y)), fmaxf(tA.z, tB.z));
    const float n0 = fmaxf(tNear, 0.0f);
    const float band = 4.8e-7f * (fabsf(tFar) + n0) + 1e-30f;
    return !((tFar - n0) < -band);          // NaN / inf fall through to "enter"
}

// Ray source / result sink of one use of the engine (extend, shadow, both in one launch, whole paths, test batch):
//   typedef ... State;                                  per-lane state the policy keeps while a work item lives
//   static constexpr bool kShades;                      does the policy use the shade step?
//   bool load(i, o, d, anyHit, State&)                  first ray of work item i; false: nothing to trace, finish()
//                                                       is called with a miss.  anyHit is honoured when REC == 3.
//   int  finish(i, B, o, d, anyHit, State&)             the lane's ray is complete.  COOP_RELEASE: item done;
//                                                       COOP_SHADE: wait for a shade step; COOP_NEWRAY: o, d, anyHit
//                                                       now hold the item's next ray
//   int  shade(i, o, d, anyHit, State&)                 converged shading step (COOP_RELEASE or COOP_NEWRAY)
//   void retire(State&)                                 called once by every lane when the wave leaves
// REC: 1 = every ray is a closest-hit ray, 2 = every ray is an any-hit (shadow) ray, 3 = per ray (policy).
//
// All 64 lanes of the wave call this; `counter` is a zero-initialised device word shared by the grid.
enum : int { COOP_RELEASE = 0, COOP_SHADE = 1, COOP_NEWRAY = 2 };
#ifndef COOP_SHADE_BIAS
#define COOP_SHADE_BIAS 8              // a shade step needs this many more waiting lanes than the busiest traversal step kind
#endif

template <int REC, class Policy>
__device__ __forceinline__ void traverse_coop(const AccelView& A, const Policy& pol, uint32_t n, uint32_t* __restrict__ counter,
                                              float tmin, float tmax, uint32_t* __restrict__ lds, uint32_t need)
{
    const uint32_t lane = __lane_id();
    CoopLds L;
    L.stack = lds + lane;                                  // [level * 64]
    L.queue = lds + need * 64u;
    L.ray = reinterpret_cast<float*>(L.queue + COOP_QCAP);
    L.best = reinterpret_cast<unsigned long long*>(L.ray + COOP_RAY_WORDS * 64u);
    L.pend = reinterpret_cast<uint32_t*>(L.best + 64);
    L.pend[lane] = 0u;
    // small launches are spread over the whole grid: a wave is handed at most `quota` rays per refill and its
    // other lanes help (steal step); launches with >= 64 rays per resident wave run as before until the tail
    const uint32_t nWavesGrid = gridDim.x * (blockDim.x >> 6);
    // (camera rays -- REC 1 launches -- are taken 64 at a time: measured faster than chunks, tools/gpu_variants.sh)
    const uint32_t chunkMax = (REC == 1) ? 64u : (uint32_t)COOP_CHUNK_MAX;
    uint32_t chunk = min(chunkMax, max(64u, (n / (4u * nWavesGrid)) & ~63u));
    const uint32_t quota = COOP_STEAL ? min(64u, max((uint32_t)COOP_MIN_QUOTA, (n + nWavesGrid - 1u) / nWavesGrid)) : 64u;

#ifdef COOP_STATS
    uint32_t statN[8] = {0, 0, 0, 0, 0, 0, 0, 0}, statL[8] = {0, 0, 0, 0, 0, 0, 0, 0}, statKind = 7u;      // wave-uniform
    uint32_t stState[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    uint32_t qHead = 0, qTail = 0;                         // wave-uniform, monotonically increasing
    bool exhausted = false;                                // wave-uniform: the global counter ran past n and the wave's own
                                                           // reservation is used up
    uint32_t resBase = 0, resEnd = 0;                      // wave-uniform: ray indices reserved by this wave, not yet handed out
    bool lastOfAll = false;                                // this wave's reservation reached the end of the launch
    // per-lane ray state
    uint32_t rayIdx = COOP_NONE;                           // ray being walked (COOP_NONE: lane is free)
    uint32_t cur = COOP_NONE, sp = 0;
    uint32_t par = 0;                                      // LDS ray slot of the current instance
    uint32_t w6 = lane << COOP_OWNER_SHIFT;                // instance slot being walked | << 16: the lane whose ray it is (this lane,
                                                           // or the lane it helps)
#define COOP_OWNER() (w6 >> COOP_OWNER_SHIFT)
    uint32_t spInst = 0;                                   // inside a BLAS: stack[0, spInst) are top-level entries
    uint32_t markPrev = 0;                                 // qTail when the previous instance was left
    uint32_t finMark = 0; bool finishing = false;
    bool anyHit = (REC == 2);                              // this lane's ray ends at its first accepted candidate
#if COOP_WATCHDOG
    uint32_t iter = 0;
#endif
    bool needShade = false;                                // the lane's item waits for a shade step
    typename Policy::State st{};
    f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
    RayInst R;
    R.o = o; R.d = d; R.rcp = mk3(0.f, 0.f, 0.f); R.exactOnly = true;

#define COOP_POP() do { if (sp == 0) cur = COOP_NONE; else { --sp; cur = L.stack[sp * 64u]; } } while (0)

    for (;;) {
#if COOP_WATCHDOG
        // watchdog (wave-uniform, scalar): three orders of magnitude above what any launch needs; a wave that gets here
        // leaves with work undone rather than spin (results are then wrong, which the parity tests would show)
        iter = __builtin_amdgcn_readfirstlane(iter + 1u);
        if (iter > COOP_MAX_ITER) break;
#endif
        // ---- lanes whose walk has ended wait for their queued tests, then hand the result over ----
        // (a helper's walk ends like a ray's; an owner additionally waits until its helpers are back)
        const bool stealPhase = COOP_STEAL && (exhausted || quota < 64u);
        if (cur == COOP_NONE && !finishing && !needShade &&
            (COOP_OWNER() != lane || (rayIdx != COOP_NONE && (!stealPhase || __hip_atomic_load(&L.pend[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == 0u)))) { finishing = true; finMark = qTail; }
        const bool done = finishing && (int32_t)(qHead - finMark) >= 0;
        const bool isFree = (rayIdx == COOP_NONE) && (COOP_OWNER() == lane);
        const unsigned long long doneMask = __ballot(done), freeMask = __ballot(isFree);
        const unsigned long long workMask = __ballot(cur != COOP_NONE);
        // free lanes count as idle while new rays can still be had -- in the steal phase only once the wave has
        // run out of work altogether (otherwise they help)
        const int nIdle = __popcll(doneMask) + ((!exhausted && (!stealPhase || workMask == 0ull)) ? __popcll(freeMask) : 0);
        const int nShade = Policy::kShades ? __popcll(__ballot(needShade)) : 0;
        // Step selection is greedy: of the step kinds that lanes are waiting for (refill, shade, top-level
        // node, instance entry, BLAS node) the wave takes the one with the most lanes, so every step runs
        // as converged as the moment allows and waiting lanes batch up instead of trickling through.
        const uint32_t tag = cur & TAG_MASK;
        const bool has = (cur != COOP_NONE);
        const bool isNode = has && tag == TAG_BLAS, isLeaf = has && tag == TAG_LEAF, isTop = has && tag == TAG_TLAS,
                   isInst = has && tag == TAG_INST;
        const unsigned long long nodeMask = __ballot(isNode), topMask = __ballot(isTop), instMask = __ballot(isInst);
        const int nNode = __popcll(nodeMask), nTop = __popcll(topMask), nInst = __popcll(instMask);
        const int nMaxWork = max(nNode, max(nTop, nInst));
#ifdef COOP_STATS
        stState[0] += nNode; stState[1] += nTop; stState[2] += nInst; stState[3] += __popcll(__ballot(isLeaf));
        stState[4] += __popcll(__ballot(finishing && !done)); stState[5] += __popcll(doneMask); stState[6] += __popcll(freeMask); stState[7] += 64;
#endif

        // (re)start the walk of the ray now in o, d
#define COOP_START_RAY(WALK) do {                                                                      \
            L.best[lane] = ~0ull;                                                                      \
            sp = 0; spInst = 0; par = 0; markPrev = qHead; finishing = false; needShade = false;       \
            cur = (WALK) ? (TAG_TLAS | 0u) : COOP_NONE;                                                \
        } while (0)

        if (nIdle > 0 && ((stealPhase ? nIdle >= min(nMaxWork, COOP_TAIL_MIN) : nIdle >= nMaxWork + COOP_IDLE_BIAS) || (workMask == 0ull && nShade == 0))) {
            COOP_STAT(0, __popcll(doneMask) + ((!exhausted) ? __popcll(freeMask) : 0));
            if (done && COOP_OWNER() != lane) {    // a helper is back: its subtree is walked and its queued tests are consumed
                atomicSub(&L.pend[COOP_OWNER()], 1u);
                w6 = lane << COOP_OWNER_SHIFT; finishing = false;
            } else if (done) {
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
                else if (act == COOP_SHADE) needShade = true;
                else { anyHit = (REC == 2) || (REC == 3 && ah); COOP_START_RAY(true); }
            }
            if (!exhausted) {
                const bool want = (rayIdx == COOP_NONE) && (COOP_OWNER() == lane);
                const unsigned long long wm = __ballot(want);
                uint32_t cnt = min((uint32_t)__popcll(wm), quota);
                if (cnt) {
                    // Ray indices are reserved from the global counter in chunks and handed out from the wave's own
                    // reservation: one atomic round trip per chunk instead of per refill.  Guided self-scheduling:
                    // the chunk is 1/4 of an even share of what was left at the previous reservation, between 64
                    // and COOP_CHUNK_MAX, so reservations shrink towards the end of the launch and no wave sits on
                    // a large private backlog while others have run dry.
                    if (resBase == resEnd) {
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
                        if (idx < n) {
                            rayIdx = idx;
                            bool ah = false;
                            const bool walk = pol.load(idx, o, d, ah, st);
                            anyHit = (REC == 2) || (REC == 3 && ah);
                            COOP_START_RAY(walk);
                        }
                    }
                }
            }
            continue;
        }
        // ---- shade step: every lane whose closest-hit ray found something runs the hit shader together ----
        if (Policy::kShades && nShade > 0 && (nShade >= nMaxWork + COOP_SHADE_BIAS || workMask == 0ull)) {
            COOP_STAT(1, nShade);
            if (needShade) {
                bool ah = false;
                const int act = pol.shade(rayIdx, o, d, ah, st);
                needShade = false;
                if (act == COOP_RELEASE) rayIdx = COOP_NONE;
                else { anyHit = (REC == 2) || (REC == 3 && ah); COOP_START_RAY(true); }
            }
            continue;
        }
        if (workMask == 0ull) {
            if (qTail != qHead) { coop_test_step(A, L, lane, qHead, qTail, tmin, tmax, R, par, w6); continue; }
            if (__ballot(rayIdx != COOP_NONE || COOP_OWNER() != lane) == 0ull) break;      // exhausted, every lane free, queue empty
            continue;                                              // lanes still finishing: next round hands them over
        }
        // ---- steal step: free lanes take the bottom stack entry (largest pending subtree) of busy lanes ------
        if (stealPhase) {
            const bool canSteal = isFree;
            const bool donor = has && sp > 0 && !(isInst && (cur & COOP_RESUME));   // (a resuming helper's stack entry has no instance yet)
            const unsigned long long sMask = __ballot(canSteal), dMask = __ballot(donor);
            const int pairs = min(__popcll(sMask), __popcll(dMask));
            if (pairs > 0 && pairs >= min(COOP_STEAL_MIN, max(1, __popcll(workMask) >> 2))) {
                // the donors' lane numbers go through the 64 ring slots behind the queue tail
                if (qTail - qHead + 64u > COOP_QCAP) { coop_test_step(A, L, lane, qHead, qTail, tmin, tmax, R, par, w6); continue; }
                COOP_STAT(2, pairs);
                const uint32_t dRank = lanes_below(dMask), sRank = lanes_below(sMask);
                uint32_t entry = COOP_NONE;
                if (donor && dRank < (uint32_t)pairs) {
                    L.queue[(qTail + dRank) & (COOP_QCAP - 1u)] = lane;
                    // stack = [top-level entries | entries of the current BLAS]; visiting order is free, so the hole
                    // is filled by moving the last entry of each segment down
                    const bool inBlas = (tag == TAG_BLAS || tag == TAG_LEAF);
                    const uint32_t split = inBlas ? spInst : sp;
                    entry = L.stack[0];
                    if (split > 0) {
                        L.stack[0] = L.stack[(split - 1u) * 64u];
                        if (sp > split) L.stack[(split - 1u) * 64u] = L.stack[(sp - 1u) * 64u];
                        if (inBlas) spInst = split - 1u;
                    } else {
                        L.stack[0] = L.stack[(sp - 1u) * 64u];
                    }
                    --sp;
                }
                // the k-th stealer is served by the k-th donor.  Every lane shuffles (the others from themselves), so the
                // donor's world ray lands directly in the stealer's registers
                const bool takes = canSteal && sRank < (uint32_t)pairs;
                const uint32_t src = takes ? (L.queue[(qTail + sRank) & (COOP_QCAP - 1u)] & 63u) : lane;
                o.x = __shfl(o.x, src); o.y = __shfl(o.y, src); o.z = __shfl(o.z, src);
                d.x = __shfl(d.x, src); d.y = __shfl(d.y, src); d.z = __shfl(d.z, src);
                const uint32_t sw6 = __shfl(w6, src);           // donor: owner of the ray | the instance it is in
                anyHit = __shfl((int)anyHit, src) != 0;
                const uint32_t e2 = __shfl(entry, src);
                if (takes) {
                    cur = e2; sp = 0; spInst = 0; finishing = false;
                    w6 = (sw6 & ~((1u << COOP_OWNER_SHIFT) - 1u)) | (w6 & ((1u << COOP_OWNER_SHIFT) - 1u));
                    // a stolen BLAS entry belongs to the donor's current instance: re-enter that instance through the
                    // ordinary instance step (same matrix, same world ray => the very same object-space ray), which
                    // then continues with the stolen entry instead of the BLAS root
                    const uint32_t t2 = e2 & TAG_MASK;
                    if (t2 == TAG_BLAS || t2 == TAG_LEAF) {
                        L.stack[0] = e2; sp = 1;
                        cur = TAG_INST | COOP_RESUME | (sw6 & COOP_IFIRST_MASK);
                    }
                    atomicAdd(&L.pend[COOP_OWNER()], 1u);
                }
                continue;
            }
        }

        // ---- queued piece of an oversized leaf, or a leaf root ----------------------------------------
        if (__any(isLeaf)) {
            COOP_STAT(3, __popcll(__ballot(isLeaf)));
            uint32_t cnt = 0, st = 0;
            if (isLeaf) { st = cur & LEAF_START_MASK; cnt = ((cur >> LEAF_START_BITS) & 7u) + 1u; COOP_POP(); }
            coop_enqueue(A, L, lane, (lane << COOP_LANE_SHIFT) | (par << COOP_PAR_SHIFT), cnt, st, qHead, qTail, tmin, tmax, R, par, w6);
            continue;
        }
        // ---- top-level node (radiance.cl:110-150) --------------------------------------------------------
        // The choice between top-level, instance and node step is greedy on WEIGHTED lane counts (quarters): upstream
        // kinds feed the downstream ones, so serving them a little early keeps more lanes walking than a plain
        // majority vote, which lets the minority kinds starve behind a stable majority of node lanes
        if (nTop > 0 && nTop * COOP_W_TOP >= nNode * 4 && nTop * COOP_W_TOP >= nInst * COOP_W_INST) {
            COOP_STAT(4, nTop);
            if (isTop) {
                const float4* np = reinterpret_cast<const float4*>(A.ctnodes + (cur & IDX_MASK));
                const float4 bmin = np[0], bmax = np[1];
                const uint4 w = *reinterpret_cast<const uint4*>(np + 2);
                // the world ray in the form the fast slab test takes (reciprocal + "exact form only" flag), rebuilt per
                // visit rather than kept in registers across the walk; the inner-node decision is the reference's
                // (division form inside the band), as for BLAS nodes
                RayInst W;
                W.o = o; W.d = d;
                W.rcp = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
                const float amin_ = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
                const float amax_ = fmaxf(fmaxf(fabsf(W.rcp.x), fabsf(W.rcp.y)), fabsf(W.rcp.z));
                W.exactOnly = !(amin_ > 1e-20f) || !(amax_ < 1e20f);
                if (!(w.x & LEAF_BIT)) {
                    if (slab_fast(W, mk3(bmin.x, bmin.y, bmin.z), mk3(bmax.x, bmax.y, bmax.z))) { L.stack[sp * 64u] = TAG_TLAS | w.y; ++sp; cur = TAG_TLAS | w.x; }
                    else COOP_POP();
                } else {
                    const uint32_t count = w.x & 0x7fffffffu;
                    if (w.z == TYPE_INST) {
                        // every instance of the leaf is entered by the reference (no per-instance box test);
                        // instances whose root box the ray provably misses are dropped here (coop_inst_pretest; the
                        // 1-ulp reciprocal is far inside its margin)
                        const f3 rcpW = W.rcp;
                        const float oMax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
                        const bool preOK = !W.exactOnly && (oMax < 1e20f);
                        for (uint32_t b0 = 0; b0 < count; b0 += 16u) {
                            uint32_t m16 = 0;
                            for (uint32_t i = 0; i < min(16u, count - b0); ++i)
                                if (coop_inst_pretest(A.insts[w.y + b0 + i], o, rcpW, oMax, preOK)) m16 |= 1u << i;
                            if (m16) { L.stack[sp * 64u] = TAG_INST | (m16 << COOP_IMASK_SHIFT) | (w.y + b0); ++sp; }
                        }
                    }
                    COOP_POP();
                }
            }
            continue;
        }
        // ---- instance entry (radiance.cl:161-169) ----------------------------------------------------------
        if (nInst > 0 && nInst * COOP_W_INST >= nNode * 4) {
            // the LDS ray slot about to be overwritten belongs to the instance before the previous one:
            // every queued test of it lies before markPrev
            const bool ready = isInst && (int32_t)(qHead - markPrev) >= 0;
            if (__ballot(ready) == 0ull) { coop_test_step(A, L, lane, qHead, qTail, tmin, tmax, R, par, w6); continue; }
            COOP_STAT(5, __popcll(__ballot(ready)));
            if (REC != 1) { if (anyHit && ready && L.best[COOP_OWNER()] != ~0ull) { cur = COOP_NONE; sp = 0; } }
            uint32_t cntE = 0, stE = 0;           // triangles of a leaf-root BLAS, queued at the end of the step
            if (ready && cur != COOP_NONE) {
                const bool resume = COOP_STEAL && (cur & COOP_RESUME);
                uint32_t ci = cur & COOP_IFIRST_MASK;
                if (!resume) {               // lowest instance of the mask; the rest of the entry goes back on the stack
                    const uint32_t m16 = (cur >> COOP_IMASK_SHIFT) & 0xffffu, rest = m16 & (m16 - 1u);
                    ci += (uint32_t)__ffs((int)m16) - 1u;
                    if (rest) { L.stack[sp * 64u] = TAG_INST | (rest << COOP_IMASK_SHIFT) | (cur & COOP_IFIRST_MASK); ++sp; }
                }
                const float4* ip = reinterpret_cast<const float4*>(A.insts + ci);
                float m[16];
                *reinterpret_cast<float4*>(m + 0) = ip[0];
                *reinterpret_cast<float4*>(m + 4) = ip[1];
                *reinterpret_cast<float4*>(m + 8) = ip[2];
                *reinterpret_cast<float4*>(m + 12) = ip[3];
                {   // park the ray of the instance being left: queued tests of it may still be pending
                    float* rs = L.ray + lane;
                    rs[0 * 64] = R.o.x; rs[1 * 64] = R.o.y; rs[2 * 64] = R.o.z;
                    rs[3 * 64] = R.d.x; rs[4 * 64] = R.d.y; rs[5 * 64] = R.d.z;
                    rs[6 * 64] = __uint_as_float(w6);
                }
                w6 = (w6 & ~((1u << COOP_OWNER_SHIFT) - 1u)) | ci;
                R.o = mat4_mul3(m, o.x, o.y, o.z, 1.0f);
                R.d = mat4_mul3(m, d.x, d.y, d.z, 0.0f);
                // hardware reciprocal (<= 1 ulp) instead of three IEEE divisions: R.rcp only feeds the fast slab test, whose
                // band covers it -- fl(s/d) and fl(s * rcp) then differ by at most 2^-23 + 2 * 2^-24 = 2^-22 relative, the
                // band is 4.8e-7 = 2 x 2^-22 of (|tFar| + max(tNear, 0)) -- and the decision inside the band is the division form's
                R.rcp = mk3(__builtin_amdgcn_rcpf(R.d.x), __builtin_amdgcn_rcpf(R.d.y), __builtin_amdgcn_rcpf(R.d.z));
                const float amin = fminf(fminf(fabsf(R.d.x), fabsf(R.d.y)), fabsf(R.d.z));
                const float amax = fmaxf(fmaxf(fabsf(R.rcp.x), fabsf(R.rcp.y)), fabsf(R.rcp.z));
                R.exactOnly = !(amin > 1e-20f) || !(amax < 1e20f);
                const uint4 rdsc = *reinterpret_cast<const uint4*>(ip + 9);    // rootDesc0, rootDesc1, triBase, -
                markPrev = qTail;            // everything queued so far belongs to instances being left
                par ^= 1u;
                spInst = sp;                 // everything on the stack now is a top-level entry
                if (resume) { spInst = sp - 1u; COOP_POP(); }
                else if (rdsc.y & WIDE_LEAF) {
                    // a BLAS that is a single leaf (a quad, a small box): its triangles are queued right here (below,
                    // converged) instead of costing a leaf-item step of their own -- 15 % of all steps on sample1
                    uint32_t cnt = wide_count(rdsc.y), st = wide_slot(rdsc.x);
                    while (cnt > 8u) { L.stack[sp * 64u] = leaf_item(st, 8u); ++sp; st += 8u; cnt -= 8u; }
                    cntE = cnt; stE = st;
                    COOP_POP();
                } else {
                    const float4 rmin = ip[10], rmax = ip[11];
                    if (slab_fast(R, mk3(rmin.x, rmin.y, rmin.z), mk3(rmax.x, rmax.y, rmax.z))) cur = rdsc.x;
                    else COOP_POP();
                }
            }
            coop_enqueue(A, L, lane, (lane << COOP_LANE_SHIFT) | (par << COOP_PAR_SHIFT), cntE, stE, qHead, qTail, tmin, tmax, R, par, w6);
            continue;
        }
        // ---- node step ----------------------------------------------------------------------------------------
        if (nodeMask != 0ull) {
            COOP_STAT(6, nNode);
            uint32_t cntL = 0, stL = 0, cntR = 0, stR = 0, nextL = COOP_NONE, nextR = COOP_NONE;
            if (isNode) {
                const float4* wp = reinterpret_cast<const float4*>(A.wide + (cur & IDX_MASK));
                const float4 l0 = wp[0], l1 = wp[1], r0 = wp[2], r1 = wp[3];
                const uint32_t ld0 = __float_as_uint(l0.w), ld1 = __float_as_uint(l1.w);
                const uint32_t rd0 = __float_as_uint(r0.w), rd1 = __float_as_uint(r1.w);
                if (ld1 & WIDE_LEAF) {
                    cntL = wide_count(ld1); stL = wide_slot(ld0);
                    while (cntL > 8u) { L.stack[sp * 64u] = leaf_item(stL, 8u); ++sp; stL += 8u; cntL -= 8u; }   // oversized leaf: rare
                } else if (slab_fast(R, mk3(l0.x, l0.y, l0.z), mk3(l1.x, l1.y, l1.z))) {
                    nextL = ld0;
                }
                if (rd1 & WIDE_LEAF) {
                    cntR = wide_count(rd1); stR = wide_slot(rd0);
                    while (cntR > 8u) { L.stack[sp * 64u] = leaf_item(stR, 8u); ++sp; stR += 8u; cntR -= 8u; }
                } else if (slab_fast(R, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z))) {
                    nextR = rd0;
                }
                if (nextL != COOP_NONE) { if (nextR != COOP_NONE) { L.stack[sp * 64u] = nextR; ++sp; } cur = nextL; }
                else if (nextR != COOP_NONE) cur = nextR;
                else COOP_POP();
            }
            const uint32_t tagBits = (lane << COOP_LANE_SHIFT) | (par << COOP_PAR_SHIFT);
            coop_enqueue(A, L, lane, tagBits, cntL, stL, qHead, qTail, tmin, tmax, R, par, w6);
            coop_enqueue(A, L, lane, tagBits, cntR, stR, qHead, qTail, tmin, tmax, R, par, w6);
            if (qTail - qHead >= 64u) {
                coop_test_step(A, L, lane, qHead, qTail, tmin, tmax, R, par, w6);
                if (REC != 1) { if (anyHit && cur != COOP_NONE && L.best[COOP_OWNER()] != ~0ull) { cur = COOP_NONE; sp = 0; } }
            }
            continue;
        }
        // lanes are waiting below their thresholds and nothing else can run: let the queue advance
        if (qTail != qHead) coop_test_step(A, L, lane, qHead, qTail, tmin, tmax, R, par, w6);
    }
#ifdef COOP_STATS
    if (lane < 8u) atomicAdd(&g_coop_state[lane], (unsigned long long)stState[lane]);
    if (lane < 7u) { atomicAdd(&g_coop_stats[lane], (unsigned long long)statN[lane]); atomicAdd(&g_coop_stats[8u + lane], (unsigned long long)statL[lane]); }
#endif
    pol.retire(st);                      // all 64 lanes, converged: per-lane tallies of the policy
#undef COOP_POP
#undef COOP_OWNER
#undef COOP_START_RAY
}

} // namespace rdx

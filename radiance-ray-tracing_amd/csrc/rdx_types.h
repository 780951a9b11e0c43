// rdx_types.h -- POD layouts shared by the host runtime, the BVH builder and the HIP kernels.
//
// "Blob" structs are the reference's on-device acceleration-structure format
// (radiance/shader/data.cl:4-83 == radiance/src/core.h:34-101); the product keeps producing and
// accepting that byte format at the API boundary (TLAS cache files, ReadBuffer of a TLAS).
// "D*" structs are the traversal-friendly layout derived from a blob at first use: same float
// values, laid out for 16-byte coalesced loads and without the dependent vertex fetch.
#pragma once
#include <stdint.h>

namespace rdx {

// ---- reference blob format -------------------------------------------------------------------
struct BlobTopHeader { uint32_t type, nodeByteOffset, instByteOffset, totalBufferSize; };   // core.h:34-40
struct BlobBotHeader { uint32_t type, nodeByteOffset, faceByteOffset, vertexOffset; };      // core.h:42-48
struct BlobNode {                                                                           // core.h:59-87 (48 B)
    float bottom[4];
    float top[4];
    uint32_t w0, w1, w2, w3;   // inner: {left, right, 0, 0}; leaf: {0x80000000|count, start, type, 0}
};
struct BlobTri  { uint32_t idx0, idx1, idx2, primID; };                                     // core.h:90-96
struct BlobInst { float m[16]; uint32_t SBTOffset, instanceID, customInstanceID, instanceOffset; }; // core.h:50-57 (80 B)

enum : uint32_t { TYPE_INST = 1, TYPE_TRIG = 2, TYPE_TOP_AS = 1, TYPE_BOT_AS = 2, LEAF_BIT = 0x80000000u };

// ---- host structs bound to the raygen stage (core.h:103-158) ----------------------------------
struct RayTraceProperties { uint32_t totalSamples, batchSize, depth, debug; };
struct Material { float albedo[4]; float metallic, roughness, transmission, ior;
                  int32_t albedoTexIdx, metallicTexIdx, roughnessTexIdx, normalTexIdx; };
struct MeshInfo { int32_t vertexOffset, indexOffset, uvOffset, normalOffset, materialIndex, _0, _1, _2; };
struct DirLight { float direction[4]; float color[4]; };
struct SceneProperties { uint32_t lightCount[4]; DirLight lights[5]; };
struct PhysicalCamera { float widthPixel, heightPixel, focalLength, sensorWidth, focalDistance, fStop,
                        x, y, z, wx, wy, wz; };

static_assert(sizeof(BlobNode) == 48 && sizeof(BlobTri) == 16 && sizeof(BlobInst) == 80, "blob layout");
static_assert(sizeof(Material) == 48 && sizeof(MeshInfo) == 32 && sizeof(SceneProperties) == 176 &&
              sizeof(PhysicalCamera) == 48 && sizeof(RayTraceProperties) == 16, "host struct layout");

// ---- derived device layout --------------------------------------------------------------------
// stack / node references carry a 2-bit tag in the top bits
enum : uint32_t { TAG_BLAS = 0u << 30, TAG_TLAS = 1u << 30, TAG_INST = 2u << 30, TAG_LEAF = 3u << 30, TAG_MASK = 3u << 30,
                  IDX_MASK = ~(3u << 30),
                  // TAG_LEAF work item: up to 8 consecutive triangle slots, (count-1) << 27 | first slot
                  LEAF_START_BITS = 27, LEAF_START_MASK = (1u << 27) - 1u };

struct alignas(16) DNode {        // 48 B; child / triangle indices are ABSOLUTE into the merged arrays
    float bmin[4];
    float bmax[4];
    uint32_t w0, w1, w2, w3;      // as BlobNode
};
struct alignas(16) DTri {         // 48 B: v0, e1 = v1 - v0, e2 = v2 - v0 (single IEEE subtractions)
    float v0[3]; uint32_t primID;
    float e1[3]; uint32_t _p0;
    float e2[3]; uint32_t _p1;
};
struct alignas(16) DInst {        // 224 B
    float inv[16];                // InverseMat4x4(object->world), math.cl:56-183 evaluated once on the host
    float fwd[16];                // object->world
    uint32_t SBTOffset, instanceID, customInstanceID, blasRoot;   // blasRoot: absolute DNode index (reference-order kernel)
    uint32_t rootDesc0, rootDesc1, _p0, _p1;                      // root of the wide layout, encoded like a DWide child (its cone = the whole BLAS's); _p0 = first
                                                                  // triangle slot of the BLAS
    float rootMin[4];             // root box (tested on entry iff the root is an inner node)
    float rootMax[4];
    // world-space AABB of the root box for the conservative instance pre-test (traverse_coop.h):
    // worldMin[3] = margin coefficient c (< 0: no pre-test for this instance), worldMax[3] = max |coordinate|
    float worldMin[4];
    float worldMax[4];
};

// "Wide" BLAS node: one record per INNER node of the reference tree, carrying the boxes of BOTH
// children, so a single 64-byte fetch decides both subtrees and leaf children need no node fetch at
// all.  Child descriptor (d0, d1): inner child -> d0 = DWide index, d1 = 0;
//                                  leaf child  -> d0 = first triangle slot, d1 = WIDE_LEAF | count.
// Child descriptor in full:
//   d1 = bit 31 leaf | bits 24-30 c7 | byte 2 cone y | byte 1 cone x | byte 0 cone z      c7 = triangle count (leaf) or cone T (inner)
//   d0 = leaf: cone T << 25 | first triangle slot (25 bits);   inner: DWide index
// The NORMAL CONE of a child (rdx_runtime.cpp derive_accel, kernels.hip "culled walk") bounds how close to parallel a ray can
// be to any triangle of the leaf / below the inner child: axis byte b = round(127.5 + 127 a), read with v_cvt_f32_ubyte0/1/2;
// threshold T (7 bits): the culled walk may skip the child only for rays with |d . (b - 127.5)| >= T |d|; T = 127: never.
enum : uint32_t { WIDE_LEAF = 0x80000000u, WIDE_MAX_LEAF_TRIS = 127u, WIDE_SLOT_BITS = 25u, WIDE_SLOT_MASK = (1u << 25) - 1u, WIDE_CONE_NEVER = 127u };
#if defined(__HIPCC__) || defined(__CUDACC__)
#define RDX_HD __host__ __device__
#else
#define RDX_HD
#endif
RDX_HD inline uint32_t wide_count(uint32_t d1) { return (d1 >> 24) & 0x7fu; }              // leaf children
RDX_HD inline uint32_t wide_slot(uint32_t d0) { return d0 & WIDE_SLOT_MASK; }              // leaf children
// cone = x | y << 8 | z << 16 | T << 24 (NormalCone::pack)
RDX_HD inline void wide_desc(bool leaf, uint32_t ref, uint32_t count, uint32_t cone, uint32_t& d0, uint32_t& d1)
{
    const uint32_t x = cone & 0xffu, y = (cone >> 8) & 0xffu, z = (cone >> 16) & 0xffu, T = (cone >> 24) & 0x7fu;
    d1 = (leaf ? WIDE_LEAF : 0u) | ((leaf ? count : T) << 24) | (y << 16) | (x << 8) | z;
    d0 = leaf ? ((T << WIDE_SLOT_BITS) | (ref & WIDE_SLOT_MASK)) : ref;
}
struct alignas(16) DWide {        // 64 B
    float lmin[3]; uint32_t ld0;
    float lmax[3]; uint32_t ld1;
    float rmin[3]; uint32_t rd0;
    float rmax[3]; uint32_t rd1;
};
// "Quad" record of the pool engine's exhaustive walk (rdx_runtime.cpp derive_accel, traverse_pool.h): record i belongs to the
// same inner node N as DWide record i and holds, for each of N's two children c, one HALF of two entries:
//   * c is an inner node whose box is exactly the union of its children's boxes (always, in a tree the reference's builder
//     made): the two children of c -- N's grandchildren -- each with its own box.  c itself is never fetched: a grandchild g's
//     box lies inside c's, and the reference's slab decision is monotone under inclusion for rays without zero direction
//     components, so "g's box is hit" implies "c's box is hit".  A LEAF grandchild is tested iff c's box is hit (the reference
//     never looks at a leaf's own box), so its entry carries c's box instead of its own.  Entries of such a half have QUAD_PAIR
//     set; rays with a (nearly) zero direction component test c's box -- a leaf entry's, or the union of the two -- as well.
//   * otherwise (c is a leaf, or its box is not that union): c itself in the first entry -- a leaf without QUAD_PAIR is
//     entered without a test, an inner node iff its box is hit -- and an empty second entry (leaf, count 0).
// One 128-byte fetch thus decides two levels of the reference's tree.  Entry = {min[3], d0, max[3], d1} like a DWide child;
// d1 = WIDE_LEAF | count << 24 | QUAD_PAIR, d0 = first triangle slot or the record index of the inner node (no cones).
enum : uint32_t { QUAD_PAIR = 1u };
struct alignas(16) DQuad { DWide half[2]; };
static_assert(sizeof(DWide) == 64 && sizeof(DQuad) == 128 && sizeof(DInst) == 224, "derived layout");

} // namespace rdx

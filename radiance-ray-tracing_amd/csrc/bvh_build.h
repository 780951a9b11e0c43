// bvh_build.h -- host BVH builder + blob packers (product code).
// Bit-exact with the reference's binned-SAH builder (radiance/src/bvh.cpp:46-597) and its
// BLAS/TLAS packers (radiance/src/radiance.cpp:318-425), but O(N log N) per node instead of
// O(N * bins): see bvh_build.cpp.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace rdx {

struct Blas {
    std::vector<uint8_t> data;   // [BlobBotHeader | BlobNode[] | BlobTri[] | float4 vertex[]]
    int maxDepth = 0;
};

struct InstanceDesc {
    float transform[16];
    uint32_t SBTOffset;
    uint32_t customInstanceID;
    const Blas* blas;
};

// GPU-assisted candidate evaluation (bvh_build.cpp): the binning pass of a node's candidate planes, done on the device for large
// nodes.  `upload`: the primitives of one BLAS, 9 floats each (box min, box max, centroid) -> handle (0 = refused).  `bin`: for
// the node's work list (indices into those primitives) and the candidate planes of up to three axes (an empty vector = axis
// skipped), `out` = per non-empty axis [K + 1] counts (as uint32 bits) followed by [K + 1] boxes {min.xyz, max.xyz}; bin b of a
// primitive = index of the first candidate greater than its centroid.  false = not done (the host bins).  Thread-safe.
struct GpuBinner {
    virtual ~GpuBinner() {}
    virtual uint64_t upload(const float* prims9, uint32_t n) = 0;
    virtual void release(uint64_t handle) = 0;
    virtual bool bin(uint64_t handle, const uint32_t* work, size_t n, const std::vector<float> cand[3], std::vector<float>& out) = 0;
};
void set_gpu_binner(GpuBinner* b, size_t minPrims);

// returns nullptr and sets err on failure (e.g. a split loop the reference would never leave)
Blas* build_blas(const float* verts_xyz, uint32_t nverts, const uint32_t* indices, uint32_t ntris,
                 std::string& err);
bool  build_tlas(const InstanceDesc* inst, uint32_t ninst, std::vector<uint8_t>& blob, int& maxDepth,
                 std::string& err);

} // namespace rdx

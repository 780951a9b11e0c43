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

// returns nullptr and sets err on failure (e.g. a split loop the reference would never leave)
Blas* build_blas(const float* verts_xyz, uint32_t nverts, const uint32_t* indices, uint32_t ntris,
                 std::string& err);
bool  build_tlas(const InstanceDesc* inst, uint32_t ninst, std::vector<uint8_t>& blob, int& maxDepth,
                 std::string& err);

} // namespace rdx

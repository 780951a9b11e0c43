// sceneBuilder.h -- RD::Scene::Load as the reference's tools/sceneBuilder.h:32-58 declares it (same struct, same
// members, same INCLUDE_SCENE_DESC / INCLUDE_SCENE_LAYOUT macros, same `<path>.cache` convention,
// tools/sceneBuilder.cpp:223-262), for Wavefront OBJ + MTL files and without assimp: the import is rdx_obj_load
// (include/rdx.h, csrc/scene_obj.cpp), which yields the very buffers the reference fills from its aiScene
// (sceneBuilder.cpp:69-219).  Header-only, like the rest of the facade.
//
// Differences a caller can see: only .obj is read (the reference reads whatever its assimp build supports; its own
// samples use .glb); textures are not loaded (textureData is an empty array and every *TexIdx is -1 -- the live
// shader stubs all texture fetches, samples/shader.cl:379-445); a failure prints the library's error text and
// exits like every other facade call, where the reference asserts.
#pragma once

#include <ctime>
#include <string>
#include <vector>

#include <radiance.h>

#define INCLUDE_SCENE_DESC(scene)   \
scene->meshInfoData,                \
scene->vertexData,                  \
scene->indexData,                   \
scene->uvData,                      \
scene->normalData,                  \
scene->materialData,                \
scene->textureData,                 \
scene->sampler,                     \
scene->topAccelStruct

#define INCLUDE_SCENE_LAYOUT        \
RD::BUFFER_TYPE,                    \
RD::BUFFER_TYPE,                    \
RD::BUFFER_TYPE,                    \
RD::BUFFER_TYPE,                    \
RD::BUFFER_TYPE,                    \
RD::BUFFER_TYPE,                    \
RD::TEX_ARRAY_TYPE,                 \
RD::IMAGE_SAMPLER_TYPE,             \
RD::ACCEL_STRUCT_TYPE

namespace RD
{

struct Scene
{
public:
    static Scene* Load(std::string path, RD::Platform* plt, bool loadFromCache = false);

    RD::Buffer meshInfoData;
    RD::Buffer vertexData;
    RD::Buffer indexData;
    RD::Buffer uvData;
    RD::Buffer normalData;

    RD::Buffer materialData;
    RD::ImageArray textureData;
    RD::Sampler sampler;

    RD::TopAccelStruct topAccelStruct;
};

inline Scene* Scene::Load(std::string path, RD::Platform* plt, bool loadFromCache)
{
    rdx_obj_scene o;
    if (rdx_obj_load(path.c_str(), &o)) detail::fatal("Scene::Load");

    auto upload = [&](const void* data, size_t bytes) {
        RD::Buffer b = RD::CreateBuffer(plt, (unsigned int)(bytes ? bytes : 16));
        if (bytes) RD::WriteBuffer(plt, b, bytes, const_cast<void*>(data));
        return b;
    };
    static_assert(sizeof(RD::MeshInfo) == sizeof(rdx_mesh_info) && sizeof(RD::Material) == sizeof(rdx_material), "layout");
    Scene* rdScene = new Scene();
    rdScene->textureData = RD::CreateImageArray(plt, 4096, 4096, 0);                  // sceneBuilder.cpp:40
    rdScene->sampler = RD::CreateSampler(plt, RD_ADDRESS_REPEAT, RD_FILTER_LINEAR);   // :41
    rdScene->meshInfoData = upload(o.meshInfo, (size_t)o.nmeshes * sizeof(RD::MeshInfo));
    rdScene->vertexData = upload(o.vertex, (size_t)o.nvertices * sizeof(RD::Vec3));
    rdScene->indexData = upload(o.index, (size_t)o.ntriangles * sizeof(RD::Triangle));
    rdScene->uvData = upload(o.uv, (size_t)o.nvertices * sizeof(RD::Vec3));
    rdScene->normalData = upload(o.normal, (size_t)o.nvertices * sizeof(RD::Vec3));
    rdScene->materialData = upload(o.materials, (size_t)o.nmaterials * sizeof(RD::Material));

    const std::string cachePath = path + ".cache";
    if (loadFromCache) {
        RD::FileToTopAccelStruct(plt, cachePath.c_str(), &rdScene->topAccelStruct);
    } else {
        time_t start_t, end_t;
        time(&start_t);
        std::vector<RD::BottomAccelStruct> rdBotASList;
        for (uint32_t i = 0; i < o.nmeshes; i++) {
            RD::Mesh rdMesh;
            const float* v = o.vertex + o.meshInfo[i].vertexOffset;
            const uint32_t* t = o.index + o.meshInfo[i].indexOffset;
            for (uint32_t k = 0; k < o.meshVertexCount[i]; k++) rdMesh.vertexData.push_back({v[3 * k], v[3 * k + 1], v[3 * k + 2]});
            for (uint32_t k = 0; k < o.meshTriangleCount[i]; k++) rdMesh.indexData.push_back({t[3 * k], t[3 * k + 1], t[3 * k + 2]});
            rdBotASList.push_back(RD::BuildAccelStruct(plt, rdMesh));
        }
        std::vector<RD::Instance> rdInstanceList;       // one per mesh, BuildInstance (sceneBuilder.cpp:287-315) on a flat file
        for (uint32_t i = 0; i < o.nmeshes; i++) {
            RD::Instance inst;
            inst.transform = RD::Mat4x4{};
            inst.SBTOffset = 0;
            inst.customInstanceID = (unsigned int)o.meshInfo[i].materialIndex;
            inst.bottomAccelStruct = rdBotASList[i];
            rdInstanceList.push_back(inst);
        }
        rdScene->topAccelStruct = RD::BuildAccelStruct(plt, rdInstanceList);
        RD::TopAccelStructToFile(plt, rdScene->topAccelStruct, cachePath.c_str());
        time(&end_t);
        printf("\nBVH build report:\n");
        printf("\tNumber of meshes: %u\n", o.nmeshes);
        printf("\tNumber of vertices: %u\n", o.nvertices);
        printf("\tNumber of triangles: %u\n", o.ntriangles);
        printf("\tBuild time cost: %f (sec)\n", difftime(end_t, start_t));
    }
    rdx_obj_free(&o);
    return rdScene;
}

} // namespace RD

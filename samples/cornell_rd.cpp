// cornell_rd.cpp -- the Radiance host API (namespace RD, include/radiance.h) driving the HIP core.
// Follows the flow of the reference's live sample (samples/sample1.cpp:363-498): create the buffers,
// build one BLAS per mesh and a TLAS over the instances, bind the 14 descriptors in kernel-parameter
// order, TraceRays, read the RGBA8 image back, bump totalSamples.  The scene is procedural (an open
// Cornell box with two boxes) because no asset / assimp exists offline; the output is a binary PPM.
//
// build: g++ -std=c++17 -Iinclude samples/cornell_rd.cpp -Lradiance-ray-tracing_amd -lrdx
//            -Wl,-rpath,$PWD/radiance-ray-tracing_amd -o cornell_rd
// run:   ./cornell_rd 320 180 8 out.ppm [frames per view = 2] [views = 1]
//
// With views > 1 it plays the host loop of the reference's interactive sample (samples/sample1.cpp:447-548) without a window:
// every view is an "edit" -- the camera is moved, the camera buffer rewritten and totalSamples reset to 0, which is what the
// reference's inspector does when a property changes -- followed by `frames` progressive frames (TraceRays; totalSamples +=
// batchSize); the last frame of each view is written to <out minus .ppm>_<view>.ppm and the frame time is printed.
#include <chrono>
#include <cstring>
#include <string>

#include "radiance.h"

struct HostScene {
    std::vector<RD::Mesh> meshes;
    std::vector<RD::Vec3> normals;      // concatenated, one per vertex
    std::vector<RD::Vec3> uvs;
    std::vector<RD::MeshInfo> infos;    // one per instance (the shader indexes it by instance id)
    std::vector<RD::Mat4x4> transforms;
    std::vector<int> meshOf, materialOf;
};

static void addQuad(RD::Mesh& m, std::vector<RD::Vec3>& nrm, RD::Vec3 a, RD::Vec3 b, RD::Vec3 c, RD::Vec3 d, RD::Vec3 n)
{
    const unsigned base = (unsigned)m.vertexData.size();
    for (RD::Vec3 v : {a, b, c, d}) { m.vertexData.push_back(v); nrm.push_back(n); }
    m.indexData.push_back({base, base + 1, base + 2});
    m.indexData.push_back({base, base + 2, base + 3});
}

static RD::Mesh makeBox(std::vector<RD::Vec3>& nrm, RD::Vec3 lo, RD::Vec3 hi)
{
    RD::Mesh m;
    const float x0 = lo.x, y0 = lo.y, z0 = lo.z, x1 = hi.x, y1 = hi.y, z1 = hi.z;
    addQuad(m, nrm, {x0, y0, z0}, {x0, y0, z1}, {x0, y1, z1}, {x0, y1, z0}, {-1, 0, 0});
    addQuad(m, nrm, {x1, y0, z0}, {x1, y1, z0}, {x1, y1, z1}, {x1, y0, z1}, {1, 0, 0});
    addQuad(m, nrm, {x0, y0, z0}, {x1, y0, z0}, {x1, y0, z1}, {x0, y0, z1}, {0, -1, 0});
    addQuad(m, nrm, {x0, y1, z0}, {x0, y1, z1}, {x1, y1, z1}, {x1, y1, z0}, {0, 1, 0});
    addQuad(m, nrm, {x0, y0, z0}, {x0, y1, z0}, {x1, y1, z0}, {x1, y0, z0}, {0, 0, -1});
    addQuad(m, nrm, {x0, y0, z1}, {x1, y0, z1}, {x1, y1, z1}, {x0, y1, z1}, {0, 0, 1});
    return m;
}

int main(int argc, char** argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 320, H = argc > 2 ? atoi(argv[2]) : 180;
    const unsigned spp = argc > 3 ? (unsigned)atoi(argv[3]) : 8;
    const std::string out = argc > 4 ? argv[4] : "cornell_rd.ppm";
    const int framesPerView = argc > 5 ? atoi(argv[5]) : 2, views = argc > 6 ? atoi(argv[6]) : 1;

    // ---- scene ------------------------------------------------------------------------------------
    HostScene S;
    std::vector<std::vector<RD::Vec3>> meshNormals;
    auto quadMesh = [&](RD::Vec3 a, RD::Vec3 b, RD::Vec3 c, RD::Vec3 d, RD::Vec3 n) {
        RD::Mesh m; std::vector<RD::Vec3> nn; addQuad(m, nn, a, b, c, d, n);
        S.meshes.push_back(m); meshNormals.push_back(nn);
    };
    const float X = 3, Y = 6, Z = 3;
    quadMesh({-X, 0, -Z}, {X, 0, -Z}, {X, 0, Z}, {-X, 0, Z}, {0, 1, 0});      // floor
    quadMesh({-X, Y, -Z}, {-X, Y, Z}, {X, Y, Z}, {X, Y, -Z}, {0, -1, 0});     // ceiling
    quadMesh({-X, 0, Z}, {X, 0, Z}, {X, Y, Z}, {-X, Y, Z}, {0, 0, -1});       // back
    quadMesh({-X, 0, -Z}, {-X, 0, Z}, {-X, Y, Z}, {-X, Y, -Z}, {1, 0, 0});    // left
    quadMesh({X, 0, -Z}, {X, Y, -Z}, {X, Y, Z}, {X, 0, Z}, {-1, 0, 0});       // right
    { std::vector<RD::Vec3> nn; S.meshes.push_back(makeBox(nn, {-0.8f, 0, -0.8f}, {0.8f, 3.2f, 0.8f})); meshNormals.push_back(nn); }
    { std::vector<RD::Vec3> nn; S.meshes.push_back(makeBox(nn, {-0.8f, 0, -0.8f}, {0.8f, 1.6f, 0.8f})); meshNormals.push_back(nn); }
    const int materialOf[7] = {0, 0, 0, 1, 2, 3, 0};
    RD::Mat4x4 tall(0.951f, 0, 0.309f, -1.2f, 0, 1, 0, 0, -0.309f, 0, 0.951f, 1.0f, 0, 0, 0, 1);
    RD::Mat4x4 shortT(0.956f, 0, -0.292f, 1.3f, 0, 1, 0, 0, 0.292f, 0, 0.956f, -0.6f, 0, 0, 0, 1);

    std::vector<RD::Vec3> vertexList, normalList, uvList;
    std::vector<RD::Triangle> indexList;
    std::vector<RD::MeshInfo> meshInfoList;
    for (size_t i = 0; i < S.meshes.size(); ++i) {          // one instance per mesh, as Scene::Load produces
        RD::MeshInfo mi{};
        mi.vertexOffset = (int)vertexList.size() * 3; mi.indexOffset = (int)indexList.size() * 3;
        mi.uvOffset = (int)uvList.size() * 3; mi.normalOffset = (int)normalList.size() * 3;
        mi.materialIndex = materialOf[i];
        for (auto& v : S.meshes[i].vertexData) { vertexList.push_back(v); uvList.push_back({0, 0, 0}); }
        for (auto& n : meshNormals[i]) normalList.push_back(n);
        for (auto& t : S.meshes[i].indexData) indexList.push_back(t);
        meshInfoList.push_back(mi);
    }
    RD::Material mats[4] = {};
    const float albedo[4][3] = {{0.73f, 0.73f, 0.73f}, {0.65f, 0.05f, 0.05f}, {0.12f, 0.45f, 0.15f}, {0.9f, 0.8f, 0.5f}};
    for (int i = 0; i < 4; ++i) {
        for (int c = 0; c < 3; ++c) mats[i].albedo[c] = albedo[i][c];
        mats[i].albedo[3] = 1; mats[i].metallic = i == 3 ? 0.9f : 0.0f; mats[i].roughness = i == 3 ? 0.2f : 0.9f;
        mats[i].transmission = 0; mats[i].ior = 1.45f;
        mats[i].albedoTexIdx = mats[i].metallicTexIdx = mats[i].roughnessTexIdx = mats[i].normalTexIdx = -1;
    }

    RD::PhysicalCamera camData{};
    camData.widthPixel = (float)W; camData.heightPixel = (float)H;
    camData.focalLength = 0.100f; camData.sensorWidth = 0.036f; camData.focalDistance = 14.0f; camData.fStop = 0.0f;
    camData.x = 0; camData.y = 6.5f; camData.z = -16.0f;
    camData.wx = 0.2617992f; camData.wy = -3.14159f; camData.wz = 0.0f;
    RD::SceneProperties sceneData{};
    sceneData.lightCount[0] = 1;
    const float ldir[4] = {0.0f, -0.7071063f, 0.7071073f, 0.0f};
    for (int c = 0; c < 4; ++c) { sceneData.lights[0].direction[c] = ldir[c]; sceneData.lights[0].color[c] = c < 3 ? 10.0f : 1.0f; }
    RD::RayTraceProperties RTProp = {0, spp, 8, 0};

    // ---- RD:: API, in the order of sample1.cpp:363-411 -------------------------------------------------
    RD::Platform* plt = RD::Platform::GetPlatform();
    const size_t imageSize = (size_t)W * H * RD_CHANNEL;
    std::vector<uint8_t> image(imageSize);

    RD::Buffer rdRTProp = RD::CreateBuffer(plt, sizeof(RD::RayTraceProperties));
    RD::WriteBuffer(plt, rdRTProp, sizeof(RD::RayTraceProperties), &RTProp);
    RD::Buffer rdImage = RD::CreateImage(plt, W, H);
    RD::Buffer rdImageScratch = RD::CreateBuffer(plt, (unsigned)(imageSize * sizeof(float)));
    RD::Buffer rdCamData = RD::CreateBuffer(plt, sizeof(camData));
    RD::WriteBuffer(plt, rdCamData, sizeof(camData), &camData);
    RD::Buffer rdSceneData = RD::CreateBuffer(plt, sizeof(RD::SceneProperties));
    RD::WriteBuffer(plt, rdSceneData, sizeof(sceneData), &sceneData);

    auto upload = [&](void* p, size_t bytes) { RD::Buffer b = RD::CreateBuffer(plt, (unsigned)bytes); RD::WriteBuffer(plt, b, bytes, p); return b; };
    RD::Buffer rdMeshInfo = upload(meshInfoList.data(), meshInfoList.size() * sizeof(RD::MeshInfo));
    RD::Buffer rdVertex = upload(vertexList.data(), vertexList.size() * sizeof(RD::Vec3));
    RD::Buffer rdIndex = upload(indexList.data(), indexList.size() * sizeof(RD::Triangle));
    RD::Buffer rdUV = upload(uvList.data(), uvList.size() * sizeof(RD::Vec3));
    RD::Buffer rdNormal = upload(normalList.data(), normalList.size() * sizeof(RD::Vec3));
    RD::Buffer rdMat = upload(mats, sizeof(mats));
    RD::ImageArray rdTextures = RD::CreateImageArray(plt, 4096, 4096, 0);
    RD::Sampler rdSampler = RD::CreateSampler(plt, RD_ADDRESS_REPEAT, RD_FILTER_LINEAR);

    std::vector<RD::BottomAccelStruct> botAS;
    for (auto& m : S.meshes) botAS.push_back(RD::BuildAccelStruct(plt, m));
    std::vector<RD::Instance> instances;
    for (size_t i = 0; i < S.meshes.size(); ++i) {
        RD::Instance inst = {i == 5 ? tall : (i == 6 ? shortT : RD::Mat4x4{}), 0, (unsigned)materialOf[i], botAS[i]};
        instances.push_back(inst);
    }
    RD::TopAccelStruct rdTopAS = RD::BuildAccelStruct(plt, instances);

    RD::DescriptorSet descSet = RD::CreateDescriptorSet({rdRTProp, rdImageScratch, rdImage, rdCamData, rdSceneData, rdMeshInfo,
                                                        rdVertex, rdIndex, rdUV, rdNormal, rdMat, rdTextures, rdSampler, rdTopAS});
    RD::PipelineLayout layout = RD::CreatePipelineLayout({RD::BUFFER_TYPE, RD::BUFFER_TYPE, RD::IMAGE_TYPE, RD::BUFFER_TYPE,
                                                          RD::BUFFER_TYPE, RD::BUFFER_TYPE, RD::BUFFER_TYPE, RD::BUFFER_TYPE,
                                                          RD::BUFFER_TYPE, RD::BUFFER_TYPE, RD::BUFFER_TYPE, RD::TEX_ARRAY_TYPE,
                                                          RD::IMAGE_SAMPLER_TYPE, RD::ACCEL_STRUCT_TYPE});
    char shaderCode[] = "__kernel void raygen(/* stock pipeline: samples/sbt.json */) {}";
    RD::ShaderModule shader = RD::CreateShaderModule(plt, shaderCode, (unsigned)strlen(shaderCode), "functName..");
    RD::Pipeline pipeline = RD::CreatePipeline({1, layout, {shader}, {}});
    RD::BindPipeline(plt, pipeline);
    RD::BindDescriptorSet(plt, descSet);

    // ---- the host loop (sample1.cpp:447-548): per view an edit + reset, then progressive frames -------------------------
    auto writePPM = [&](const std::string& path) {
        FILE* fp = fopen(path.c_str(), "wb");
        if (!fp) return false;
        fprintf(fp, "P6\n%d %d\n255\n", W, H);
        for (int i = 0; i < W * H; ++i) fwrite(&image[4 * i], 1, 3, fp);
        fclose(fp);
        printf("Writing image with extent: <%d, %d> to %s\n", W, H, path.c_str());
        return true;
    };
    for (int view = 0; view < views; ++view) {
        if (view > 0) {          // an edit: orbit the camera a little, rewrite its buffer, restart the accumulation
            camData.x = 2.5f * (float)view; camData.wy = -3.14159f + 0.15f * (float)view;
            RD::WriteBuffer(plt, rdCamData, sizeof(camData), &camData);
            RD::RayTraceProperties p;
            RD::ReadBuffer(plt, rdRTProp, sizeof p, &p);
            p.totalSamples = 0;
            RD::WriteBuffer(plt, rdRTProp, sizeof p, &p);
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (int frame = 0; frame < framesPerView; ++frame) {
            RD::TraceRays(plt, 0, 0, 0, W, H);
            RD::ReadBuffer(plt, rdImage, imageSize, image.data());
            RD::RayTraceProperties p;
            RD::ReadBuffer(plt, rdRTProp, sizeof p, &p);
            p.totalSamples += p.batchSize;
            RD::WriteBuffer(plt, rdRTProp, sizeof p, &p);
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (views > 1) printf("view %d: %d frames of %u spp, %.2f ms per frame incl. read-back\n", view, framesPerView, spp, ms / framesPerView);
        const std::string path = views > 1 ? out.substr(0, out.rfind(".ppm")) + "_" + std::to_string(view) + ".ppm" : out;
        if (!writePPM(path)) return 1;
    }
    return 0;
}

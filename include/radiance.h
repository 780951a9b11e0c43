// radiance.h -- the Radiance host API (`namespace RD`) as a thin inline C++ layer over the C ABI of
// librdx.so (include/rdx.h).  Same names, signatures and error behaviour as the reference's
// radiance/include/radiance.h (:86-174), so its callers (samples/sample1.cpp, tools/sceneBuilder.cpp)
// build against this header unchanged; underneath, TraceRays runs the hand-written HIP wavefront
// path tracer instead of an OpenCL megakernel.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "core.h"
#include "clcontext.h"

#ifndef SHADER_LIB_PATH
#define SHADER_LIB_PATH ""      // the reference passes -I<this> to its OpenCL JIT; so does the run-time compilation of user shader programs
#endif

namespace RD
{

typedef void* Handle;
typedef cl_mem TopAccelStruct;
typedef cl_mem Image;
typedef cl_mem ImageArray;
typedef cl_sampler Sampler;
typedef unsigned int Uniform;
typedef cl_mem Buffer;

enum DescriptorType { ACCEL_STRUCT_TYPE, IMAGE_TYPE, IMAGE_ARRAY_TYPE, IMAGE_SAMPLER_TYPE, BUFFER_TYPE, TEX_ARRAY_TYPE };

struct Mesh { std::vector<Vec3> vertexData; std::vector<Triangle> indexData; };

struct BVHNode;
struct _BottomAccelStruct { BVHNode* root; std::vector<char> data; rdx_blas handle; };
typedef _BottomAccelStruct* BottomAccelStruct;

struct Instance
{
    Mat4x4 transform;                 // row-major object->world
    unsigned int SBTOffset;
    unsigned int customInstanceID;
    BottomAccelStruct bottomAccelStruct;
};

typedef std::vector<Handle> DescriptorSet;
typedef std::vector<DescriptorType> PipelineLayout;
typedef cl_kernel ShaderModule;

#define SHADER_UNUSED (~0U)
struct ShaderGroup { ShaderModule generalShader, closestHitShader, anyHitShader; };
struct PipelineCreateInfo
{
    unsigned int maxRayRecursionDepth;
    PipelineLayout layout;
    std::vector<ShaderModule> modules;
    std::vector<ShaderGroup> groups;
};
typedef PipelineCreateInfo Pipeline;

#define CHANNEL 4
#define RD_CHANNEL CHANNEL

typedef uint32_t AddressingMode;
#define RD_ADDRESS_CLAMP_TO_EDGE CL_ADDRESS_CLAMP_TO_EDGE
#define RD_ADDRESS_CLAMP CL_ADDRESS_CLAMP
#define RD_ADDRESS_REPEAT CL_ADDRESS_REPEAT
#define RD_ADDRESS_MIRRORED_REPEAT CL_ADDRESS_MIRRORED_REPEAT
typedef uint32_t FilterMode;
#define RD_FILTER_NEAREST CL_FILTER_NEAREST
#define RD_FILTER_LINEAR CL_FILTER_LINEAR

struct Platform
{
    static Platform* GetPlatform()
    {
        static Platform ctx;
        if (!ctx.initialized) {
            ctx.clContext = CLContext::GetCLContext();
            ctx.initialized = true;
            printf("Platform initialized.\n");
        }
        return &ctx;
    }
    ~Platform() { printf("Platform destroyed.\n"); if (clContext) clContext->Cleanup(); }

    Pipeline activePipeline;
    CLContext* clContext = nullptr;
    bool initialized = false;

private:
    Platform() = default;
    void operator=(Platform&) = delete;
    Platform(Platform&) = delete;
};

namespace detail
{
[[noreturn]] inline void fatal(const char* what)
{
    printf("Radiance Error: %s: %s\n", what, rdx_last_error());
    rdx_shutdown();
    exit(-1);
}
template <class T> inline T* need(T* p, const char* what) { if (!p) fatal(what); return p; }
} // namespace detail

// ---- acceleration structures (blocking) -----------------------------------------------------------
inline BottomAccelStruct BuildAccelStruct(Platform*, Mesh& mesh)
{
    printf("\nStart building bottom level BVH\n\tVertex count:%ld\n\tTriangle count:%ld\n", (long)mesh.vertexData.size(),
           (long)mesh.indexData.size());
    static_assert(sizeof(Vec3) == 12 && sizeof(Triangle) == 12, "Mesh element layout");
    rdx_blas h = detail::need(rdx_blas_build(reinterpret_cast<const float*>(mesh.vertexData.data()), (uint32_t)mesh.vertexData.size(),
                                             reinterpret_cast<const uint32_t*>(mesh.indexData.data()), (uint32_t)mesh.indexData.size()),
                              "BuildAccelStruct(Mesh)");
    _BottomAccelStruct* as = new _BottomAccelStruct();
    as->root = nullptr;
    as->handle = h;
    uint32_t n = 0;
    const char* p = static_cast<const char*>(rdx_blas_data(h, &n));
    as->data.assign(p, p + n);
    printf("Max BVH depth is %d\n", rdx_blas_max_depth(h));
    return as;
}

inline TopAccelStruct BuildAccelStruct(Platform*, std::vector<Instance>& instances)
{
    printf("\nStart building top level BVH\n\tInstance count: %ld\n", (long)instances.size());
    std::vector<rdx_instance> in(instances.size());
    for (size_t i = 0; i < instances.size(); ++i) {
        const float* m = &instances[i].transform.a1;
        for (int k = 0; k < 16; ++k) in[i].transform[k] = m[k];
        in[i].SBTOffset = instances[i].SBTOffset;
        in[i].customInstanceID = instances[i].customInstanceID;
        in[i].bottomAccelStruct = instances[i].bottomAccelStruct ? instances[i].bottomAccelStruct->handle : nullptr;
    }
    return detail::need(rdx_tlas_build(in.data(), (uint32_t)in.size()), "BuildAccelStruct(instances)");
}

inline void TopAccelStructToFile(Platform*, TopAccelStruct accelStruct, const char* path) { if (rdx_tlas_to_file(accelStruct, path)) detail::fatal("TopAccelStructToFile"); }
inline void FileToTopAccelStruct(Platform*, const char* path, TopAccelStruct* accelStruct) { *accelStruct = detail::need(rdx_tlas_from_file(path), "FileToTopAccelStruct"); }

// ---- resources ----------------------------------------------------------------------------------------
inline Buffer CreateBuffer(Platform*, unsigned int size) { return detail::need(rdx_buffer_create(size), "CreateBuffer"); }
inline Image CreateImage(Platform*, unsigned int width, unsigned int height) { return detail::need(rdx_buffer_create((size_t)width * height * CHANNEL), "CreateImage"); }
// texture arrays / samplers (radiance.cpp:96-137, 202-224): RGBA8 2D image arrays; sampled by the stock shader when option
// "textures" is on (the live reference shader has its reads commented out, see rdx.h)
inline ImageArray CreateImageArray(Platform*, unsigned int width, unsigned int height, unsigned int arraySize) { return detail::need(rdx_image_array_create(width, height, arraySize), "CreateImageArray"); }
inline Sampler CreateSampler(Platform*, AddressingMode addressingMode, FilterMode filterMode) { return detail::need(rdx_sampler_create(addressingMode, filterMode), "CreateSampler"); }
inline void ReadImage(Platform*, ImageArray handle, unsigned int width, unsigned int height, size_t arrayIndex, void* data) { if (rdx_image_read(handle, width, height, arrayIndex, data)) detail::fatal("ReadImage"); }
inline void WriteImage(Platform*, ImageArray handle, unsigned int width, unsigned int height, size_t arrayIndex, void* data) { if (rdx_image_write(handle, width, height, arrayIndex, data)) detail::fatal("WriteImage"); }
inline void ReadBuffer(Platform*, Buffer handle, size_t size, void* data, size_t offset = 0) { if (rdx_buffer_read(handle, offset, size, data)) detail::fatal("ReadBuffer"); }
inline void WriteBuffer(Platform*, Buffer handle, size_t size, void* data, size_t offset = 0) { if (rdx_buffer_write(handle, offset, size, data)) detail::fatal("WriteBuffer"); }

// ---- pipeline -----------------------------------------------------------------------------------------
inline DescriptorSet CreateDescriptorSet(std::vector<Handle> handles) { return handles; }
inline PipelineLayout CreatePipelineLayout(std::vector<DescriptorType> descriptorTypes) { return descriptorTypes; }
inline ShaderModule CreateShaderModule(Platform*, char* code, unsigned int size, char* name)
{
    printf("build program and get raygen kernel\n");
    if (SHADER_LIB_PATH[0]) rdx_shader_include_path(SHADER_LIB_PATH);      // the reference's clBuildProgram("-g -I" SHADER_LIB_PATH)
    return detail::need(rdx_shader_module_create(code, size, name), "CreateShaderModule");
}
inline ShaderModule CreateShaderModule(Platform* p, char* code, unsigned int size, const char* name) { return CreateShaderModule(p, code, size, const_cast<char*>(name)); }
inline Pipeline CreatePipeline(PipelineCreateInfo pipelineCreateInfo) { return pipelineCreateInfo; }
inline void BindPipeline(Platform* platform, Pipeline pipeline)
{
    platform->activePipeline = pipeline;
    if (pipeline.modules.empty() || rdx_bind_pipeline(pipeline.modules[0])) detail::fatal("BindPipeline");
}
inline void BindDescriptorSet(Platform*, DescriptorSet descriptorSet)
{
    if (rdx_bind_descriptor_set(descriptorSet.data(), (uint32_t)descriptorSet.size())) detail::fatal("BindDescriptorSet");
}
inline void TraceRays(Platform*, unsigned int raygenGroupIndex, unsigned int missGroupIndex, unsigned int hitGroupIndex,
                      unsigned int width, unsigned int height)
{
    if (rdx_trace_rays(raygenGroupIndex, missGroupIndex, hitGroupIndex, width, height)) detail::fatal("TraceRays");
}

} // namespace RD

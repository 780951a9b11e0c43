// clcontext.h -- facade for the reference's radiance/src/clcontext.h.  There is no OpenCL underneath:
// the "context" is the HIP device/stream singleton that librdx.so owns (rdx_init).  Kept so that code
// written against the reference (`platform->clContext`, `RD::read_kernel_file_str`) still compiles.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <cstdlib>
#include <sys/stat.h>

#include "rdx.h"

// opaque handle spellings the reference inherits from <CL/cl.h>
#ifndef __OPENCL_CL_H
typedef struct rdx_buffer_s* cl_mem;
typedef struct rdx_shader_s* cl_kernel;
typedef struct rdx_sampler_s* cl_sampler;
#define CL_ADDRESS_CLAMP_TO_EDGE 0x1131
#define CL_ADDRESS_CLAMP 0x1132
#define CL_ADDRESS_REPEAT 0x1133
#define CL_ADDRESS_MIRRORED_REPEAT 0x1134
#define CL_FILTER_NEAREST 0x1140
#define CL_FILTER_LINEAR 0x1141
#endif

namespace RD
{

struct CLContext
{
    int device = -1;
    static CLContext* GetCLContext()
    {
        static CLContext ctx;
        if (rdx_init(-1) != 0) { printf("Radiance Error: %s\n", rdx_last_error()); exit(-1); }
        // RDX_DEVICES=<n>: an unchanged caller (samples/sample1.cpp) renders every TraceRays frame on n GPUs of the node
        // (rdx_init_devices: buffers replicated, frame sharded by image tiles, gathered to device 0 before TraceRays returns)
        if (const char* nd = getenv("RDX_DEVICES"))
            if (atoi(nd) > 1 && rdx_init_devices((uint32_t)atoi(nd), nullptr) != 0) { printf("Radiance Error: %s\n", rdx_last_error()); exit(-1); }
        return &ctx;
    }
    void Cleanup() { rdx_shutdown(); }
};

// error policy of the reference (clcontext.h:27-47): print, clean up, exit(-1)
#define RD_CHECK(_expr)                                                            \
    do {                                                                           \
        if ((_expr) == 0) break;                                                   \
        printf("Radiance Error: '%s' failed: %s\n", #_expr, rdx_last_error());     \
        rdx_shutdown();                                                            \
        exit(-1);                                                                  \
    } while (0)

// same contract as the reference's helper (clcontext.cpp:77-108): malloc'ed, not NUL-terminated
inline int read_kernel_file_str(const char* filename, char** data, size_t* size)
{
    if (!filename || !data || !size) return -1;
    printf("\nReading kernel file with name: %s\n", filename);
    FILE* fp = fopen(filename, "r");
    if (!fp) { fprintf(stderr, "Failed to load kernel."); return -1; }
    struct stat st;
    long fsize = (stat(filename, &st) == 0) ? (long)st.st_size : 0;
    *data = (char*)malloc(fsize > 0 ? fsize : 1);
    *size = fread(*data, 1, fsize, fp);
    printf("File size: %ld\n", (long)*size);
    fclose(fp);
    return 0;
}

} // namespace RD

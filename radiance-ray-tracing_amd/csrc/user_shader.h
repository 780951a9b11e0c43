// user_shader.h -- run-time compilation and launch of a user's OpenCL C raygen program (user_shader.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace rdx {

struct UserProgram {
    hipModule_t module = nullptr;
    hipFunction_t entry = nullptr;
    std::string log;                 // compiler output (warnings)
    bool stages = false;             // entry = the shade STAGE of the wavefront pipeline around the program's callHit / callMiss (not its raygen)
};

// compiles `text` (+ the runtime's forwarder) for `arch` (e.g. "gfx950"); nullptr and `err` (with the build log) on failure
// stages: compile the program's stage functions into the wavefront pipeline's shade stage (user_shader.cpp "stage mode") instead
// of its raygen megakernel
UserProgram* compile_user_shader(const std::string& text, const std::string& includePath, const std::string& arch, bool stages, std::string& err);
// one launch of the shade stage: scalars = pass, depth, maxDepth, nPixels, sampleBase, debug; ptrs: see kStageBody
int launch_user_stage(UserProgram* p, hipStream_t st, const uint32_t scalars[6], void* const ptrs[31], uint32_t nMax, std::string& err);
// ptrs: device addresses of descriptor slots 0..10 and 13; one work-item per pixel; blocks until the frame is done
int launch_user_shader(UserProgram* p, hipStream_t st, void* const ptrs[12], uint32_t npixels, uint32_t localSize, std::string& err);
void release_user_shader(UserProgram* p);

} // namespace rdx

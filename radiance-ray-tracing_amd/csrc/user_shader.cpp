// user_shader.cpp -- run-time compilation of a user's OpenCL C `raygen` program.
//
// The reference hands the text given to RD::CreateShaderModule to the OpenCL driver's JIT -- clCreateProgramWithSource +
// clBuildProgram("-g -I" SHADER_LIB_PATH) + clCreateKernel("raygen"), radiance/src/radiance.cpp:152-179 -- so ANY OpenCL C
// program with a `raygen` kernel of the 14-parameter binding contract (samples/shader.cl:175-190) runs.  The stock program
// (the stage functions of samples/sbt.json) is served by the hand-written wavefront pipeline of this library; every OTHER
// program is compiled here, for the GPU in use, by ROCm's own OpenCL C compiler (clang, -x cl, the ROCm OpenCL builtin
// library) and launched as what it is: a megakernel, one work-item per pixel.  It is a compatibility path -- a user who
// edits a shader gets the edited shader, at megakernel speed -- not the fast path.
//
// Mechanics.  The HIP runtime cannot launch kernels that take `image2d_array_t` / `sampler_t` parameters (its host side
// dereferences image objects that do not exist), and OpenCL C cannot spell a null image.  So the user's text gets a small
// function appended that forwards to `raygen` and keeps the two opaque parameters, a second translation unit declares that
// function with the two parameters as `__constant void*` -- on amdgcn both ARE pointers to constant-address-space descriptors --
// and holds the __kernel entry point that passes null for them; the two units are joined with llvm-link.  User programs
// therefore cannot sample textures yet (the live reference shader does not either, shader.cl:379-445).
// `#include "radiance.cl"` etc. resolve through the include path given to rdx_shader_include_path (the reference bakes
// SHADER_LIB_PATH into its binary, radiance.h:7) and then through the library's OWN device library,
// radiance-ray-tracing_amd/shader/{radiance,data,math,pbr}.cl next to librdx.so: own text with the reference library's
// interface, so a program that traces rays needs nothing from the reference's tree.
//
// Floating-point contract: user programs are compiled like the stock pipeline computes -- `-ffp-contract=off
// -cl-fp32-correctly-rounded-divide-sqrt` (contract p of DESIGN.md section 2), ONE contract behind the API.  RDX_JIT_FLAGS
// replaces those two flags (e.g. RDX_JIT_FLAGS="" for clang's OpenCL defaults, what clBuildProgram("-g -I...") would use).
//
// Compiled programs are cached: in the process by a hash of (text, flags, architecture, include path, the device library's
// files), and on disk (RDX_JIT_CACHE, default /tmp/rdx_jit_cache_<uid>) when no user include path is involved.
#include "user_shader.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <ftw.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <vector>

extern char** environ;

namespace rdx {

namespace {

const char* kForwarder =
    "\n\n/* ---- appended by the runtime: forwards to the program's raygen kernel (see user_shader.cpp) ---- */\n"
    "void rdx_user_raygen(__global void* a0, __global void* a1, __global void* a2, __global void* a3, __global void* a4,\n"
    "                     __global void* a5, __global void* a6, __global void* a7, __global void* a8, __global void* a9,\n"
    "                     __global void* a10, __global void* a13, uint npixels, image2d_array_t img, sampler_t smp)\n"
    "{\n"
    "    if (get_global_id(0) >= npixels) return;\n"
    "    raygen(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, img, smp, a13);\n"
    "}\n";

const char* kEntry =
    "typedef __global void* gp;\n"
    "typedef __constant void* op;\n"
    "void rdx_user_raygen(gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, uint, op, op);\n"
    "__kernel void rdx_user_entry(gp a0, gp a1, gp a2, gp a3, gp a4, gp a5, gp a6, gp a7, gp a8, gp a9, gp a10, gp a13, uint npixels)\n"
    "{\n"
    "    rdx_user_raygen(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a13, npixels, 0, 0);\n"
    "}\n";

// ---- stage mode: the shade stage of the wavefront pipeline around the program's own callHit / callMiss -------------------------
// (radiance-ray-tracing_amd/shader/radiance.cl "stage mode" has the record / replay traceRay this relies on.)  The function below
// restates, per path and per bounce, what the stock raygen does around its traceRay call (samples/shader.cl:207-260): the payload
// and the SceneData it hands to the shaders, `color += contribution * payload.color`, `contribution *= payload.nextFactor`, the
// primary-miss and later-miss rules, the depth / debug cut-off.  Streams are those of csrc/kernels.h PathStreams.
const char* kStageBody =
    "\n\n/* ---- appended by the runtime (user_shader.cpp, stage mode) ---- */\n"
    "#undef get_global_id\n"
    "void rdx_stage_shade(uint pass, uint depth, uint maxDepth, uint nPixels, uint sampleBase, uint debug,\n"
    "    __global const uint* nIn, __global uint* nOut, __global uint* status,\n"
    "    __global struct PhysicalCamera* camData, __global struct SceneProperties* scene, __global struct MeshInfo* meshInfoData,\n"
    "    __global float* vertexData, __global uint* indexData, __global float* uvData, __global float* normalData,\n"
    "    __global struct Material* materials, __global struct AccelStruct* topLevel, __global const float* insts,\n"
    "    __global const float4* rayO, __global const float4* rayD, __global const float4* thr, __global const float4* col,\n"
    "    __global const float4* hitA, __global const uint* hitInst, __global const float4* payC, __global const float4* payF,\n"
    "    __global float4* shO, __global float4* shD, __global const uint* shHit,\n"
    "    __global float4* nRayO, __global float4* nRayD, __global float4* nThr, __global float4* nCol, __global float4* nPayC,\n"
    "    __global float4* nPayF, __global float4* sampleColor, __local uint* wg, image2d_array_t img, sampler_t smp)\n"
    "{\n"
    "    const uint i = get_global_id(0);\n"
    "    const bool active = i < *nIn;\n"
    "    bool alive = false;\n"
    "    float3 color = (float3)(0.0f), contribution = (float3)(1.0f);\n"
    "    float4 ro = (float4)(0.0f), rd = (float4)(0.0f), th = (float4)(0.0f);\n"
    "    struct Payload payload;\n"
    "    if (active) {\n"
    "        ro = rayO[i]; rd = rayD[i]; th = thr[i];\n"
    "        const float4 cc = col[i];\n"
    "        color = cc.xyz; contribution = th.xyz;\n"
    "        const uint frameID = as_uint(rd.w), slot = as_uint(th.w);\n"
    "        struct { struct RdxStageCtx ctx; struct SceneData sd; } w;\n"
    "        w.ctx.mode = pass; w.ctx.calls = 0u; w.ctx.answer = 0u; w.ctx.error = 0u;\n"
    "        w.ctx.origin = (float4)(0.0f); w.ctx.direction = (float4)(0.0f); w.ctx.sbtRecordOffset = 0; w.ctx.missIndex = 0;\n"
    "        w.ctx.pixel = as_uint(ro.w); w.ctx.pad1 = 0u;\n"
    "        w.sd.camData = camData; w.sd.scene = scene; w.sd.meshInfoData = meshInfoData; w.sd.vertexData = vertexData;\n"
    "        w.sd.indexData = indexData; w.sd.uvData = uvData; w.sd.normalData = normalData; w.sd.materials = materials;\n"
    "        w.sd.topLevel = topLevel; w.sd.depth = (int)depth; w.sd.frameID = frameID; w.sd.debug = debug;\n"
    "        /* the payload as the raygen loop holds it before this bounce's traceRay */\n"
    "        payload.color = depth ? payC[i].xyz : (float3)(0.0f);\n"
    "        payload.nextFactor = depth ? payF[i].xyz : (float3)(1.0f);\n"
    "        payload.nextRayOrigin = ro.xyz; payload.nextRayDirection = rd.xyz;\n"
    "        payload.hit = false;\n"
    "        const uint inst = hitInst[i];\n"
    "        if (inst != 0xffffffffu) {\n"
    "            const float4 ha = hitA[i];\n"
    "            __global const float* I = insts + 56u * inst;\n"
    "            struct HitData hd;\n"
    "            const float lox = I[0] * ro.x + I[1] * ro.y + I[2] * ro.z + I[3] * 1.0f;\n"
    "            const float loy = I[4] * ro.x + I[5] * ro.y + I[6] * ro.z + I[7] * 1.0f;\n"
    "            const float loz = I[8] * ro.x + I[9] * ro.y + I[10] * ro.z + I[11] * 1.0f;\n"
    "            const float ldx = I[0] * rd.x + I[1] * rd.y + I[2] * rd.z + I[3] * 0.0f;\n"
    "            const float ldy = I[4] * rd.x + I[5] * rd.y + I[6] * rd.z + I[7] * 0.0f;\n"
    "            const float ldz = I[8] * rd.x + I[9] * rd.y + I[10] * rd.z + I[11] * 0.0f;\n"
    "            hd.hitPoint = (float3)(lox, loy, loz) + (float3)(ldx, ldy, ldz) * ha.x;\n"
    "            hd.distance = ha.x;\n"
    "            hd.primitiveIndex = as_uint(ha.w);\n"
    "            __global const uint* Iu = (__global const uint*)I;\n"
    "            hd.instanceSBTOffset = Iu[32]; hd.instanceIndex = Iu[33]; hd.instanceCustomIndex = Iu[34];\n"
    "            hd.barycentric = (float3)(1 - ha.y - ha.z, ha.y, ha.z);\n"
    "            hd.transform = vload16(1, I);\n"
    "            if (pass == 1u) w.ctx.answer = shHit[i];\n"
    "            callHit(1, &payload, &hd, &w.sd, img, smp);\n"
    "            if (w.ctx.error) atomic_or(status, 4u);\n"
    "            if (pass == 0u) {\n"
    "                const bool q = w.ctx.calls != 0u;\n"
    "                if (q && !(w.ctx.origin.w == 0.001f && w.ctx.direction.w == 1000.0f && w.ctx.sbtRecordOffset == 2)) atomic_or(status, 8u);\n"
    "                shO[i] = (float4)(w.ctx.origin.xyz, q ? 1.0f : 0.0f);\n"
    "                shD[i] = (float4)(w.ctx.direction.xyz, 0.0f);\n"
    "            }\n"
    "        } else if (pass == 1u) callMiss(3, &payload, &w.sd, img, smp);\n"
    "        else shO[i] = (float4)(0.0f);\n"
    "        if (pass == 1u) {\n"
    "            bool end = true;\n"
    "            if (payload.hit) {\n"
    "                color += contribution * payload.color;\n"
    "                contribution *= payload.nextFactor;\n"
    "                end = depth + 1u >= maxDepth;\n"
    "            } else if (depth == 0u) color = payload.color;\n"
    "            if (end) sampleColor[(size_t)(frameID - sampleBase) * nPixels + slot] = (float4)(color, 0.0f);\n"
    "            alive = !end;\n"
    "        }\n"
    "    }\n"
    "    if (pass == 0u) return;\n"
    "    /* compaction of the surviving paths: one cursor atomic per work-group */\n"
    "    const uint lid = get_local_id(0);\n"
    "    if (lid == 0u) wg[0] = 0u;\n"
    "    barrier(CLK_LOCAL_MEM_FENCE);\n"
    "    uint mine = 0u;\n"
    "    if (alive) mine = atomic_inc(wg);\n"
    "    barrier(CLK_LOCAL_MEM_FENCE);\n"
    "    if (lid == 0u) wg[1] = wg[0] ? atomic_add(nOut, wg[0]) : 0u;\n"
    "    barrier(CLK_LOCAL_MEM_FENCE);\n"
    "    if (!alive) return;\n"
    "    const uint j = wg[1] + mine;\n"
    "    nRayO[j] = (float4)(payload.nextRayOrigin, ro.w);\n"
    "    nRayD[j] = (float4)(payload.nextRayDirection, rd.w);\n"
    "    nThr[j] = (float4)(contribution, th.w);\n"
    "    nCol[j] = (float4)(color, 0.0f);\n"
    "    nPayC[j] = (float4)(payload.color, 0.0f);\n"
    "    nPayF[j] = (float4)(payload.nextFactor, 0.0f);\n"
    "}\n";

const char* kStageEntry =
    "typedef __global void* gp;\n"
    "typedef __constant void* op;\n"
    "void rdx_stage_shade(uint, uint, uint, uint, uint, uint, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, gp,\n"
    "                     gp, gp, gp, gp, gp, gp, gp, gp, gp, gp, __local uint*, op, op);\n"
    "__kernel void rdx_stage_entry(uint pass, uint depth, uint maxDepth, uint nPixels, uint sampleBase, uint debug,\n"
    "    gp a0, gp a1, gp a2, gp a3, gp a4, gp a5, gp a6, gp a7, gp a8, gp a9, gp a10, gp a11, gp a12, gp a13, gp a14, gp a15, gp a16, gp a17,\n"
    "    gp a18, gp a19, gp a20, gp a21, gp a22, gp a23, gp a24, gp a25, gp a26, gp a27, gp a28, gp a29, gp a30)\n"
    "{\n"
    "    __local uint wg[2];\n"
    "    rdx_stage_shade(pass, depth, maxDepth, nPixels, sampleBase, debug, a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15,\n"
    "                    a16, a17, a18, a19, a20, a21, a22, a23, a24, a25, a26, a27, a28, a29, a30, wg, 0, 0);\n"
    "}\n";

bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }

std::string slurp(const std::string& p)
{
    std::ifstream f(p);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// run argv[0] with stdout + stderr appended to `log`; 0 on success
int run(const std::vector<std::string>& argv, const std::string& log)
{
    std::vector<char*> av;
    for (auto& a : argv) av.push_back(const_cast<char*>(a.c_str()));
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) return -1;
    int status = 0;
    if (waitpid(pid, &status, 0) < 0) return -1;
    return (WIFEXITED(status) && WEXITSTATUS(status) == 0) ? 0 : -1;
}

int rm_entry(const char* path, const struct stat*, int, struct FTW*) { return remove(path); }
void remove_tree(const std::string& dir) { (void)nftw(dir.c_str(), rm_entry, 16, FTW_DEPTH | FTW_PHYS); }

uint64_t fnv1a(const std::string& t, uint64_t h = 1469598103934665603ull)
{
    for (unsigned char c : t) { h ^= c; h *= 1099511628211ull; }
    return h;
}

// directory of the library's own device library: <directory of librdx.so>/shader
std::string own_shader_dir()
{
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&own_shader_dir), &info) || !info.dli_fname) return std::string();
    std::string so = info.dli_fname;
    const size_t cut = so.find_last_of('/');
    return (cut == std::string::npos ? std::string(".") : so.substr(0, cut)) + "/shader";
}

bool copy_file(const std::string& from, const std::string& to)
{
    std::ifstream in(from, std::ios::binary);
    if (!in) return false;
    const std::string tmp = to + ".tmp" + std::to_string((long)getpid());
    { std::ofstream out(tmp, std::ios::binary); if (!out) return false; out << in.rdbuf(); if (!out) return false; }
    return rename(tmp.c_str(), to.c_str()) == 0;
}

std::mutex g_cacheLock;
std::map<uint64_t, UserProgram*> g_cache;          // programs compiled by this process, by key (never unloaded before shutdown)

UserProgram* load_code_object(const std::string& co, bool stages, std::string& err)
{
    auto* p = new UserProgram();
    p->stages = stages;
    hipError_t e = hipModuleLoad(&p->module, co.c_str());
    if (e == hipSuccess) e = hipModuleGetFunction(&p->entry, p->module, stages ? "rdx_stage_entry" : "rdx_user_entry");
    if (e != hipSuccess) {
        err = std::string("user shader: loading the compiled program failed: ") + hipGetErrorString(e);
        if (p->module) (void)hipModuleUnload(p->module);
        delete p;
        return nullptr;
    }
    return p;
}

} // namespace

UserProgram* compile_user_shader(const std::string& text, const std::string& includePath, const std::string& arch, bool stages, std::string& err)
{
    const char* envClang = std::getenv("RDX_CLANG");
    const std::string clang = envClang ? envClang : "/opt/rocm/lib/llvm/bin/clang";
    const std::string link = clang.substr(0, clang.find_last_of('/') + 1) + "llvm-link";
    const char* envRocm = std::getenv("ROCM_PATH");
    const std::string rocm = envRocm ? envRocm : "/opt/rocm";
    const char* envFlags = std::getenv("RDX_JIT_FLAGS");
    const std::string fpFlags = envFlags ? envFlags : "-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt";
    const std::string ownDir = own_shader_dir();

    // cache key: everything the code object depends on
    uint64_t key = fnv1a(text);
    key = fnv1a("|" + fpFlags + "|" + arch + "|" + includePath + "|" + clang + (stages ? "|stages" : "|megakernel"), key);
    for (const char* f : {"radiance.cl", "data.cl", "math.cl", "pbr.cl"}) key = fnv1a(slurp(ownDir + "/" + f), key);
    {
        std::lock_guard<std::mutex> lk(g_cacheLock);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) return it->second;
    }
    char keyHex[32];
    std::snprintf(keyHex, sizeof keyHex, "%016llx", (unsigned long long)key);
    const char* envCache = std::getenv("RDX_JIT_CACHE");
    const std::string cacheDir = envCache ? envCache : "/tmp/rdx_jit_cache_" + std::to_string((long)getuid());
    const bool diskCache = includePath.empty() && !(envCache && !*envCache);      // (a user include directory can change under us)
    const std::string cached = cacheDir + "/" + keyHex + ".co";
    if (diskCache && exists(cached) && !std::getenv("RDX_JIT_COMPILE_ONLY")) {
        std::string e2;
        if (UserProgram* p = load_code_object(cached, stages, e2)) {
            p->log = "(code object from the cache: " + cached + ")";
            std::lock_guard<std::mutex> lk(g_cacheLock);
            g_cache[key] = p;
            return p;
        }
    }

    if (!exists(clang) || !exists(link)) {
        err = "user shader: the OpenCL C compiler is not installed (" + clang + ", llvm-link; set RDX_CLANG)";
        return nullptr;
    }
    char tmpl[] = "/tmp/rdx_jit_XXXXXX";
    if (!mkdtemp(tmpl)) { err = "user shader: cannot create a temporary directory"; return nullptr; }
    const std::string dir = tmpl, log = dir + "/build.log";
    {
        std::ofstream f(dir + "/user.cl");
        // stage mode: work-items are compacted paths; the shader's get_global_id(0) has to stay the pixel (shader/radiance.cl)
        if (stages) f << "#define get_global_id(d) rdx_stage_gid((d), sceneData)\n#line 1\n";
        f << text << (stages ? kStageBody : kForwarder);
    }
    { std::ofstream f(dir + "/entry.cl"); f << (stages ? kStageEntry : kEntry); }
    std::vector<std::string> common = {clang, "-x", "cl", "-cl-std=CL1.2", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + arch, "-Xclang",
                                       "-finclude-default-header", "--rocm-path=" + rocm, "-O3"};
    {
        std::stringstream ss(fpFlags);
        std::string tok;
        while (ss >> tok) common.push_back(tok);
    }
    if (stages) common.push_back("-DRDX_WAVEFRONT_STAGES");
    auto cleanup = [&]() { if (!std::getenv("RDX_JIT_KEEP")) remove_tree(dir); };
    std::vector<std::string> a = common;
    // stage mode needs the product's own radiance.cl (its traceRay records / replays): own directory first
    if (stages && !ownDir.empty()) a.push_back("-I" + ownDir);
    if (!includePath.empty()) a.push_back("-I" + includePath);
    if (!stages && !ownDir.empty()) a.push_back("-I" + ownDir);
    // the printf lowering introduces a library call after the first link of the builtin bitcode
    for (const char* s : {"-Xclang", "-mlink-builtin-bitcode-postopt", "-emit-llvm", "-c"}) a.push_back(s);
    a.push_back(dir + "/user.cl"); a.push_back("-o"); a.push_back(dir + "/user.bc");
    std::vector<std::string> b = common;
    for (const char* s : {"-w", "-emit-llvm", "-c"}) b.push_back(s);
    b.push_back(dir + "/entry.cl"); b.push_back("-o"); b.push_back(dir + "/entry.bc");
    const std::vector<std::string> c = {link, dir + "/user.bc", dir + "/entry.bc", "-o", dir + "/all.bc"};
    const std::vector<std::string> d = {clang, "-target", "amdgcn-amd-amdhsa", "-mcpu=" + arch, "-O3", dir + "/all.bc", "-o", dir + "/user.co"};
    if (run(a, log) || run(b, log) || run(c, log) || run(d, log)) {
        std::string l = slurp(log);
        if (l.size() > 10000) l.resize(10000);                 // the reference prints at most 10 000 bytes of build log (radiance.cpp:170)
        err = "user shader: compilation failed\n" + l;
        cleanup();
        return nullptr;
    }
    if (std::getenv("RDX_JIT_COMPILE_ONLY")) { err = "compiled"; cleanup(); return nullptr; }      // rdx_debug_jit_compiles
    UserProgram* p = load_code_object(dir + "/user.co", stages, err);
    if (!p) { cleanup(); return nullptr; }
    p->log = slurp(log);
    if (diskCache) { (void)mkdir(cacheDir.c_str(), 0700); (void)copy_file(dir + "/user.co", cached); }
    cleanup();
    std::lock_guard<std::mutex> lk(g_cacheLock);
    g_cache[key] = p;
    return p;
}

int launch_user_shader(UserProgram* p, hipStream_t st, void* const ptrs[12], uint32_t npixels, uint32_t localSize, std::string& err)
{
    struct Args { void* p[12]; uint32_t n; uint32_t pad; } args;
    for (int i = 0; i < 12; ++i) args.p[i] = ptrs[i];
    args.n = npixels; args.pad = 0;
    size_t size = sizeof(void*) * 12 + sizeof(uint32_t);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    if (localSize == 0) localSize = 64;
    const uint32_t grid = (npixels + localSize - 1) / localSize;
    if (grid == 0) return 0;
    hipError_t e = hipModuleLaunchKernel(p->entry, grid, 1, 1, localSize, 1, 1, 0, st, nullptr, extra);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { err = std::string("user shader: launch failed: ") + hipGetErrorString(e); return -1; }
    return 0;
}

int launch_user_stage(UserProgram* p, hipStream_t st, const uint32_t scalars[6], void* const ptrs[31], uint32_t nMax, std::string& err)
{
    struct Args { uint32_t u[6]; void* p[31]; } args;      // (six dwords, then 8-byte aligned pointers: 24 bytes -> offset 24 is 8-aligned)
    for (int i = 0; i < 6; ++i) args.u[i] = scalars[i];
    for (int i = 0; i < 31; ++i) args.p[i] = ptrs[i];
    size_t size = sizeof(args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const uint32_t grid = (nMax + 255u) / 256u;
    if (grid == 0) return 0;
    const hipError_t e = hipModuleLaunchKernel(p->entry, grid, 1, 1, 256, 1, 1, 0, st, nullptr, extra);
    if (e != hipSuccess) { err = std::string("user stage: launch failed: ") + hipGetErrorString(e); return -1; }
    return 0;
}

void release_user_shader(UserProgram* p)
{
    if (!p) return;
    {   // programs are shared through the cache: drop the entry with the object
        std::lock_guard<std::mutex> lk(g_cacheLock);
        bool shared = false;
        for (auto it = g_cache.begin(); it != g_cache.end();) { if (it->second == p) { it = g_cache.erase(it); shared = true; } else ++it; }
        (void)shared;
    }
    if (p->module) (void)hipModuleUnload(p->module);
    delete p;
}

} // namespace rdx

/*
 * ref_harness.c -- ORACLE tooling (test infrastructure, NOT the product).
 *
 * Calls the *real* reference device code.  oracle/Makefile compiles the reference's
 * samples/shader.cl (which #includes radiance/shader/{radiance,data,math,pbr}.cl) where it lies
 * under /root/reference with the ROCm clang OpenCL-C front end for x86-64 and links it into
 * oracle/_ref/libref_shader.so.  The OpenCL builtins an OpenCL runtime would supply (dot, cross,
 * normalize, min, max, clamp, mix, pow, sin, cos, acos, sqrt, fabs, fmax, isinf, get_global_id)
 * stay UNRESOLVED: no stand-in is written for them, so only the functions that need none of them
 * are callable.  Those are exactly what this harness exposes, through lazy binding:
 *
 *     random_pcg3d       radiance/shader/math.cl:10-23
 *     MultiplyMat4Vec4   radiance/shader/math.cl:25-31
 *     MultiplyMat4Mat4   radiance/shader/math.cl:33-54
 *     InverseMat4x4      radiance/shader/math.cl:56-183
 *     D_GGX              radiance/shader/pbr.cl:6-13
 *
 * Must be compiled with the same clang (ext_vector_type ABI of float3/float4/float16).
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef unsigned int cl_uint3 __attribute__((ext_vector_type(3)));
typedef float cl_float3 __attribute__((ext_vector_type(3)));
typedef float cl_float4 __attribute__((ext_vector_type(4)));
typedef float cl_float16 __attribute__((ext_vector_type(16)));

static void* g_lib;
static cl_float3 (*p_pcg)(cl_uint3);
static void (*p_mv)(cl_float16*, cl_float4*, cl_float4*);
static void (*p_mm)(cl_float16*, cl_float16*, cl_float16*);
static _Bool (*p_inv)(cl_float16*, cl_float16*);
static float (*p_dggx)(float, float);

int ref_open(const char* path)
{
    if (g_lib) return 0;
    g_lib = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
    if (!g_lib) { fprintf(stderr, "ref_open: %s\n", dlerror()); return -1; }
    p_pcg = (cl_float3(*)(cl_uint3))dlsym(g_lib, "random_pcg3d");
    p_mv = (void (*)(cl_float16*, cl_float4*, cl_float4*))dlsym(g_lib, "MultiplyMat4Vec4");
    p_mm = (void (*)(cl_float16*, cl_float16*, cl_float16*))dlsym(g_lib, "MultiplyMat4Mat4");
    p_inv = (_Bool(*)(cl_float16*, cl_float16*))dlsym(g_lib, "InverseMat4x4");
    p_dggx = (float (*)(float, float))dlsym(g_lib, "D_GGX");
    return (p_pcg && p_mv && p_mm && p_inv && p_dggx) ? 0 : -2;
}

void ref_pcg3d(const uint32_t* in3, float* out3, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) {
        cl_uint3 v = {in3[3 * i], in3[3 * i + 1], in3[3 * i + 2]};
        cl_float3 r = p_pcg(v);
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}

int ref_inverse_mat4(const float* m16, float* out16)
{
    cl_float16 m, o = 0.0f;
    memcpy(&m, m16, 64);
    _Bool ok = p_inv(&m, &o);
    memcpy(out16, &o, 64);
    return ok ? 1 : 0;
}

void ref_mul_mat4_vec4(const float* m16, const float* v4, float* out4)
{
    cl_float16 m; cl_float4 v, o;
    memcpy(&m, m16, 64); memcpy(&v, v4, 16);
    p_mv(&m, &v, &o);
    memcpy(out4, &o, 16);
}

void ref_mul_mat4_mat4(const float* a16, const float* b16, float* out16)
{
    cl_float16 a, b, o;
    memcpy(&a, a16, 64); memcpy(&b, b16, 64);
    p_mm(&a, &b, &o);
    memcpy(out16, &o, 64);
}

float ref_d_ggx(float dotNH, float roughness) { return p_dggx(dotNH, roughness); }

/*
 * ref_gpu_kern.cl -- TEST INFRASTRUCTURE (oracle side), never part of the product path.
 *
 * Translation unit 2 of the "reference on the GPU" oracle: the __kernel entry points.  Each forwards to a
 * batch wrapper of ref_gpu_dev.cl (unit 1, which contains the real reference device code) and supplies
 * null descriptors for the reference's `image2d_array_t imageArray, sampler_t sampler` tail.  OpenCL C
 * cannot spell a null image, so this unit declares the wrappers with the two opaque parameters as
 * `__constant void*` -- on amdgcn both image and sampler types ARE pointers to constant-address-space
 * descriptors, so the two units agree at the LLVM-IR level, where they are joined with llvm-link
 * (oracle/Makefile).  The entry points therefore take global buffers and scalars only and can be launched
 * with hipModuleLaunchKernel from tests/refgpu_bind.py.
 */
typedef __global void* gp;
typedef __constant void* op;

void rdxref_trace(gp tlas, gp org, gp dir, uint n, float tmin, float tmax, int sbt, gp out, op img, op smp);
void rdxref_material(gp hits, gp raydir, gp frameIDs, gp depths, uint n, gp cam, gp scene, gp meshInfo, gp vertex,
                     gp index, gp uv, gp normal, gp materials, gp tlas, gp out, op img, op smp);
void rdxref_generate(gp cam, gp rnd3, uint n, gp org, gp dir);
void rdxref_aabb(gp in, uint n, gp out);
void rdxref_triangle(gp in, gp tris, gp verts, uint n, gp out);
void rdxref_brdf(gp in, uint n, gp out);
void rdxref_raygen(gp RTProp, gp imageScratch, gp image, gp camData, gp scene, gp meshInfoData, gp vertexData,
                   gp indexData, gp uvData, gp normalData, gp materials, gp topLevel, uint npixels, op img, op smp);

__kernel void k_ref_trace(gp tlas, gp org, gp dir, uint n, float tmin, float tmax, int sbt, gp out)
{
    rdxref_trace(tlas, org, dir, n, tmin, tmax, sbt, out, 0, 0);
}

__kernel void k_ref_material(gp hits, gp raydir, gp frameIDs, gp depths, uint n, gp cam, gp scene, gp meshInfo,
                             gp vertex, gp index, gp uv, gp normal, gp materials, gp tlas, gp out)
{
    rdxref_material(hits, raydir, frameIDs, depths, n, cam, scene, meshInfo, vertex, index, uv, normal, materials,
                    tlas, out, 0, 0);
}

__kernel void k_ref_generate(gp cam, gp rnd3, uint n, gp org, gp dir)
{
    rdxref_generate(cam, rnd3, n, org, dir);
}

__kernel void k_ref_aabb(gp in, uint n, gp out)
{
    rdxref_aabb(in, n, out);
}

__kernel void k_ref_triangle(gp in, gp tris, gp verts, uint n, gp out)
{
    rdxref_triangle(in, tris, verts, n, out);
}

__kernel void k_ref_brdf(gp in, uint n, gp out)
{
    rdxref_brdf(in, n, out);
}

__kernel void k_ref_raygen(gp RTProp, gp imageScratch, gp image, gp camData, gp scene, gp meshInfoData, gp vertexData,
                           gp indexData, gp uvData, gp normalData, gp materials, gp topLevel, uint npixels)
{
    rdxref_raygen(RTProp, imageScratch, image, camData, scene, meshInfoData, vertexData, indexData, uvData,
                  normalData, materials, topLevel, npixels, 0, 0);
}

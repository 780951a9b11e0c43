#!/usr/bin/env python3
"""make_golden_gpu.py -- regenerates tests/golden/refgpu_*.npz by running the REAL reference device code on an MI355X.

The code object oracle/_ref/ref_shader_gfx950_p.co is the reference's own samples/shader.cl (+ radiance/shader/*.cl)
compiled where it lies by oracle/Makefile (ROCm clang, OpenCL C, gfx950, `-ffp-contract=off
-cl-fp32-correctly-rounded-divide-sqrt`, ROCm's own OpenCL builtin library) with the batch wrappers of
oracle/ref_gpu_dev.cl / ref_gpu_kern.cl.  It is built in the build container (where /root/reference exists) and
travels to the GPU box; run there:

    gpurun -- python tests/golden/make_golden_gpu.py        # writes gpurun_out/golden/refgpu_*.npz
    cp gpurun_out/golden/refgpu_*.npz tests/golden/

Every array in the fixtures is either an input (seeded numpy data, scene buffers from scenes.py) or an output of the
reference code object -- nothing here comes from the product or from the CPU oracle, except the TLAS blob, which the
product's host builder makes (the reference's builder needs assimp and cannot be built; its SHA-256 is stored so the
tests notice if the blob ever changes).

refgpu_kat.npz        intersectAABB, intersectTriangle, microfacetBRDF + sampleMicrofacetBRDF_transm on unit inputs
refgpu_<scene>.npz    per small scene (tests/golden_cases.py): primary rays of generateRay, HitData of a 4096-ray batch
                      (closest hit: sbt 1; any hit: sbt 2), `material` payloads on captured hits, imageScratch + RGBA8
                      after TraceRays call 1 and 2 (progressive mean)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import golden_cases as gc          # noqa: E402
import refgpu_bind as rg           # noqa: E402
import rrt_amd                     # noqa: E402,F401
from radiance_ray_tracing_amd import rd, scenes    # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = rg.RefGpu("p")
    k = gc.kat_inputs()
    out = dict(k)
    out["aabb_hit"] = ref.aabb(k["aabb_o"], k["aabb_d"], k["aabb_lo"], k["aabb_hi"]).astype(np.uint8)
    hit, t, pt, bary = ref.triangle(k["tri_o"], k["tri_d"], k["tri_v0"], k["tri_v1"], k["tri_v2"])
    out.update(tri_hit=hit.astype(np.uint8), tri_t=t, tri_pt=pt, tri_bary=bary)
    out["brdf_out"] = ref.brdf(k["brdf_in"])
    np.savez_compressed(os.path.join(OUT, "refgpu_kat.npz"), **out)
    print("refgpu_kat.npz: aabb hits %d / %d, triangle hits %d / %d" % (out["aabb_hit"].sum(), out["aabb_hit"].size,
                                                                       out["tri_hit"].sum(), out["tri_hit"].size), flush=True)
    for name in gc.SCENES:
        s = gc.small_scene(scenes, name)
        blob = gc.scene_blob(rd, s)
        rs = rg.RefScene(ref, s, blob)
        out = {"blob_sha256": gc.sha(blob)}
        # primary rays: generateRay for every pixel
        npix = s.width * s.height
        rnd = gc.generate_inputs(npix, 3)
        go, gd = rs.generate(rnd)
        out.update(gen_rnd=rnd, gen_o=go, gen_d=gd)
        if name == "c1":                 # thin lens variant
            s2 = gc.small_scene(scenes, name, fstop=2.8)
            rs2 = rg.RefScene(ref, s2, blob)
            lo, ld = rs2.generate(rnd)
            out.update(lens_o=lo, lens_d=ld)
        # traversal batch
        sel = gc.spread(npix, gc.N_PRIMARY)
        po, pd = go[sel], gd[sel]
        ph = rs.trace(po, pd)
        o, d = gc.derived_rays(5, po, pd, ph["hit"], ph["distance"])
        h1 = rs.trace(o, d, 0.001, 1000.0, 1)
        h2 = rs.trace(o, d, 0.001, 1000.0, 2)
        out.update(ray_o=o, ray_d=d, hits=h1.view(np.uint8).reshape(h1.shape[0], -1), shadow_hit=h2["hit"].astype(np.uint8))
        # material on the closest hits of primary rays (item i is shaded as pixel i, the reference's RNG input)
        # (the hit of item i need not be pixel i's own: rays spread over the image)
        sel = gc.spread(npix, gc.N_MATERIAL)
        n = sel.shape[0]
        md = np.ascontiguousarray(gd[sel])
        mh = rs.trace(go[sel], md)
        frames, depths = gc.material_inputs(n)
        pay = rs.material_batch(mh, md, frames, depths)
        out.update(mat_hits=mh.view(np.uint8).reshape(n, -1), mat_dir=md, mat_payload=pay.view(np.uint8).reshape(n, -1))
        # two progressive frames
        for f in range(2):
            rs.frame()
            out["scratch%d" % f] = rs.read_scratch()
            out["image%d" % f] = rs.read_image()
        np.savez_compressed(os.path.join(OUT, "refgpu_%s.npz" % name), **out)
        print("refgpu_%s.npz: %d rays, %d closest hits, %d shadow hits, %d material hits" %
              (name, o.shape[0], int(h1["hit"].sum()), int(h2["hit"].sum()), int(mh["hit"].sum())), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""make_golden.py -- regenerates the fixtures under tests/golden/.  Run in the BUILD container
(needs /root/reference and oracle/_ref); the fixtures themselves are plain data and travel.

ref_kat.npz        inputs -> outputs of the REAL reference device functions that need no OpenCL
                   builtin (random_pcg3d, InverseMat4x4, MultiplyMat4Vec4, MultiplyMat4Mat4, D_GGX),
                   executed from the reference's samples/shader.cl compiled for x86
                   (oracle/Makefile -> oracle/_ref, oracle/ref_harness.c).
ref_gensbt.txt     stdout of the reference's own tools/genSBT.py run on its samples/sbt.json (only
                   its two hard-coded author paths are redirected; its ./tmp.cl lands in a temp dir).
oracle_kat.npz     outputs of the CPU oracle (oracle/rt_oracle.c) on small seeded cases: blob hashes,
                   ray batches -> HitData, shading I/O, small frames.  These pin the ORACLE against
                   regressions; they are not reference outputs ("parity unpinned").
"""
import contextlib
import hashlib
import io
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def make_ref_kat():
    import oracle_bind as ob
    R = ob.ref()
    assert R is not None, "oracle/_ref is not built (run `make -C oracle`)"
    rng = np.random.default_rng(20261004)
    pcg_in = rng.integers(0, 2**32, size=(1024, 3), dtype=np.uint64).astype(np.uint32)
    pcg_in[:8] = [[0, 0, 0], [1, 2, 3], [0xffffffff] * 3, [0, 0, 1], [5, 0, 2073599], [7, 2073599, 3], [1, 0, 0], [0, 1, 0]]
    pcg_out = np.zeros((1024, 3), np.float32)
    R.ref_pcg3d(pcg_in.ctypes.data, pcg_out.ctypes.data, 1024)

    mats = rng.normal(size=(64, 16)).astype(np.float32)
    # realistic instance transforms: rotation * scale + translation, last row 0 0 0 1
    for k in range(32):
        a, b = rng.uniform(0, 6.28, 2)
        ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
        m = np.eye(4)
        m[:3, :3] = (ry @ rx) * rng.uniform(0.2, 3.0)
        m[:3, 3] = rng.uniform(-20, 20, 3)
        mats[k] = m.astype(np.float32).reshape(16)
    mats[32] = np.eye(4, dtype=np.float32).reshape(16)
    mats[33] = 0.0                                           # singular
    inv_out = np.zeros((64, 16), np.float32)
    inv_ok = np.zeros(64, np.int32)
    mats = np.ascontiguousarray(mats)
    for k in range(64):
        inv_ok[k] = R.ref_inverse_mat4(mats[k].ctypes.data, inv_out[k].ctypes.data)
    vecs = rng.normal(size=(64, 4)).astype(np.float32)
    mv_out = np.zeros((64, 4), np.float32)
    mm_out = np.zeros((64, 16), np.float32)
    for k in range(64):
        R.ref_mul_mat4_vec4(mats[k].ctypes.data, vecs[k].ctypes.data, mv_out[k].ctypes.data)
        R.ref_mul_mat4_mat4(mats[k].ctypes.data, mats[(k + 1) % 64].ctypes.data, mm_out[k].ctypes.data)
    gg_in = rng.uniform(0, 1, size=(256, 2)).astype(np.float32)
    gg_out = np.array([R.ref_d_ggx(float(a), float(b)) for a, b in gg_in], np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_kat.npz"), pcg_in=pcg_in, pcg_out=pcg_out, mats=mats, inv_out=inv_out,
                        inv_ok=inv_ok, vecs=vecs, mv_out=mv_out, mm_out=mm_out, gg_in=gg_in, gg_out=gg_out)
    print("ref_kat.npz written")


def make_ref_gensbt():
    src = open(os.path.join(REF, "tools", "genSBT.py")).read()
    src = src.replace("/home/zekailin00/Desktop/ray-tracing/framework/samples/sbt.json", os.path.join(REF, "samples", "sbt.json"))
    src = src.replace("/home/zekailin00/Desktop/ray-tracing/framework/samples/shader.cl", os.path.join(REF, "samples", "shader.cl"))
    out = io.StringIO()
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            with contextlib.redirect_stdout(out):
                exec(compile(src, "genSBT.py", "exec"), {"__name__": "__main__"})
        finally:
            os.chdir(cwd)
    text = out.getvalue()
    lines = [l for l in text.splitlines() if l.strip().startswith("case ") or l.startswith("1. raygen")]
    with open(os.path.join(HERE, "ref_gensbt.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("ref_gensbt.txt written:", len(lines), "lines")


def make_oracle_kat():
    import oracle_bind as ob
    import rrt_amd  # noqa: F401
    from radiance_ray_tracing_amd import scenes
    out = {}
    rng = np.random.default_rng(7)
    for name, kw in (("c0", dict(width=32, height=32, spp=2, depth=3)),
                     ("c1", dict(width=32, height=18, spp=2, depth=4, sphere_subdiv=2))):
        s = (scenes.c0_two_boxes if name == "c0" else scenes.c1_cornell)(**kw)
        blob, depth, _ = ob.scene_tlas(s)
        out[name + "_blob_sha256"] = np.frombuffer(hashlib.sha256(blob).digest(), np.uint8)
        out[name + "_blob"] = np.frombuffer(blob, np.uint8)
        osc = ob.OracleScene(s, blob)
        n = 256
        px = rng.integers(0, s.width * s.height, n).astype(np.uint32)
        rin = np.stack([np.zeros(n, np.uint32), np.zeros(n, np.uint32), px], 1)
        o, d = osc.generate_rays(px, rin)
        hits = ob.trace_batch(blob, o, d, 0.001, 1000.0, 1)
        sh = ob.trace_batch(blob, o, d, 0.001, 1000.0, 2)
        out[name + "_ray_o"], out[name + "_ray_d"] = o, d
        out[name + "_hits"] = hits.view(np.uint8).reshape(n, -1)
        out[name + "_shadow_hit"] = sh["hit"].astype(np.uint8)
        osc.frame()
        osc.frame()
        out[name + "_scratch"] = osc.scratch.copy()
        out[name + "_image"] = osc.image.copy()
    np.savez_compressed(os.path.join(HERE, "oracle_kat.npz"), **out)
    print("oracle_kat.npz written")


if __name__ == "__main__":
    make_ref_kat()
    make_ref_gensbt()
    make_oracle_kat()

"""ref_gpu_probe.py -- first contact with the reference-on-GPU oracle (oracle/_ref/ref_shader_gfx950_*.co):
runs the REAL reference device code on the MI355X next to the product and the CPU oracle and prints where
they differ.  Diagnostic tool, not a test:  python tools/ref_gpu_probe.py [--time]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_bind as ob          # noqa: E402
import refgpu_bind as rg          # noqa: E402
import rrt_amd                    # noqa: E402,F401
from radiance_ray_tracing_amd import rd, scenes   # noqa: E402


def bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def ray_batch(osc, n, seed):
    rng = np.random.default_rng(seed)
    s = osc.scene
    px = rng.integers(0, s.width * s.height, n).astype(np.uint32)
    rin = np.stack([rng.integers(0, 8, n).astype(np.uint32), np.zeros(n, np.uint32), px], 1)
    o, d = osc.generate_rays(px, rin)
    hits = ob.trace_batch(osc.tlas.tobytes(), o, d)
    hp = o + d * hits["distance"][:, None]
    ok = hits["hit"] == 1
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = np.where(ok[:, None], hp, o).astype(np.float32)
    d3 = np.zeros((n, 3), np.float32)
    d3[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    o3 = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    d4 = d2.copy(); d4[:, 1] *= 1e-4
    return (np.concatenate([o, o2, o3, o2]).astype(np.float32), np.concatenate([d, d2, d3, d4]).astype(np.float32))


def cmp_hits(tag, a, b):
    fields = ("distance", "primitiveIndex", "instanceIndex", "instanceCustomIndex", "instanceSBTOffset", "barycentric",
              "hitPoint", "transform")
    hf = int((a["hit"] != b["hit"]).sum())
    h = (a["hit"] == 1) & (b["hit"] == 1)
    out = ["%s: n=%d hits=%d hitflag-diff=%d" % (tag, a.shape[0], int(h.sum()), hf)]
    for f in fields:
        x, y = a[f][h], b[f][h]
        ne = (bits(x).reshape(x.shape[0], -1) != bits(y).reshape(y.shape[0], -1)).any(1)
        out.append("%s=%d" % (f[:5], int(ne.sum())))
    print(" ".join(out), flush=True)
    return h


def main():
    plt = rd.Platform.GetPlatform()
    print("device:", rd.Platform.device_name(), flush=True)
    refs = {b: rg.RefGpu(b) for b in ("p", "d") if rg.available(b)}
    print("reference builds loaded:", list(refs), flush=True)
    small = {
        "c0": scenes.c0_two_boxes(64, 64, spp=2, depth=3),
        "c1": scenes.c1_cornell(96, 54, spp=2, depth=4, sphere_subdiv=3),
        "c2": scenes.c2_atrium(96, 54, spp=2, depth=4, detail=0.2),
    }
    # ---- unit KATs: slab test and triangle test against the CPU oracle's restatement
    rng = np.random.default_rng(11)
    n = 200000
    o = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 8, 0] = 0; d[n // 8: n // 4, 1] = 0
    lo = rng.uniform(-4, 0, (n, 3)).astype(np.float32)
    hi = lo + rng.uniform(0, 4, (n, 3)).astype(np.float32)
    o[: n // 16] = lo[: n // 16]          # origins on box planes with a zero direction component -> NaN lanes
    L = ob.lib()
    for b, r in refs.items():
        got = r.aabb(o, d, lo, hi)
        exp = np.array([L.orc_intersect_aabb(o[i].ctypes.data, d[i].ctypes.data, lo[i].ctypes.data, hi[i].ctypes.data)
                        for i in range(20000)], np.uint32)
        print("aabb[%s] vs oracle: %d / 20000 differ (ref hits %d)" % (b, int((got[:20000] != exp).sum()), int(got.sum())), flush=True)
    v0 = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    v1 = v0 + rng.normal(size=(n, 3)).astype(np.float32)
    v2 = v0 + rng.normal(size=(n, 3)).astype(np.float32)
    tgt = v0 * 0.3 + v1 * 0.3 + v2 * 0.4
    dd = (tgt - o).astype(np.float32) + (rng.normal(size=(n, 3)) * 0.2).astype(np.float32)
    for b, r in refs.items():
        hit, t, pt, bary = r.triangle(o, dd, v0, v1, v2)
        m = 20000
        eh = np.zeros(m, np.uint32); et = np.zeros(m, np.float32); ep = np.zeros((m, 3), np.float32); eb = np.zeros((m, 3), np.float32)
        for i in range(m):
            eh[i] = L.orc_intersect_triangle(o[i].ctypes.data, dd[i].ctypes.data, v0[i].ctypes.data, v1[i].ctypes.data,
                                             v2[i].ctypes.data, et[i:i + 1].ctypes.data, ep[i].ctypes.data, eb[i].ctypes.data)
        k = (hit[:m] == 1) & (eh == 1)
        print("triangle[%s] vs oracle: hitflag diff %d, both-hit %d, t-bits diff %d, bary diff %d, point diff %d"
              % (b, int((hit[:m] != eh).sum()), int(k.sum()), int((t[:m][k].view(np.uint32) != et[k].view(np.uint32)).sum()),
                 int((bits(bary[:m][k]).reshape(-1, 12) != bits(eb[k]).reshape(-1, 12)).any(1).sum()),
                 int((bits(pt[:m][k]).reshape(-1, 12) != bits(ep[k]).reshape(-1, 12)).any(1).sum())), flush=True)
    # ---- traversal on the small scenes
    for name, s in small.items():
        dev = scenes.DeviceScene(s)
        blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
        osc = ob.OracleScene(s, blob)
        o, d = ray_batch(osc, 4096, 5)
        for rec in (1, 2):
            orc = ob.trace_batch(blob, o, d, 0.001, 1000.0, rec)
            prod = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
            for b, r in refs.items():
                rs = rg.RefScene(r, s, blob)
                got = rs.trace(o, d, 0.001, 1000.0, rec)
                cmp_hits("%s rec%d ref[%s] vs oracle " % (name, rec, b), got, orc)
                cmp_hits("%s rec%d ref[%s] vs product" % (name, rec, b), got, prod)
        # frames
        osc2 = ob.OracleScene(s, blob)
        for b, r in refs.items():
            rs = rg.RefScene(r, s, blob)
            dev.set_rtprop(totalSamples=0); dev.clear_scratch()
            osc2.set_rtprop(totalSamples=0); osc2.scratch[:] = 0
            for frame in range(2):
                ms = rs.frame()
                img = dev.render()
                osc2.frame()
                a = rs.read_scratch().astype(np.float64)
                p = dev.read_scratch().reshape(-1).astype(np.float64)
                q = osc2.scratch.astype(np.float64)
                print("%s frame%d ref[%s] %.2f ms: RMSE ref-product %.3g  ref-oracle %.3g  product-oracle %.3g  max|ref-product| %.3g  identical px %.4f"
                      % (name, frame, b, ms, np.sqrt(np.mean((a - p) ** 2)), np.sqrt(np.mean((a - q) ** 2)),
                         np.sqrt(np.mean((p - q) ** 2)), np.abs(a - p).max(),
                         float((a.reshape(-1, 4)[:, :3] == p.reshape(-1, 4)[:, :3]).all(1).mean())), flush=True)
                i8 = rs.read_image().astype(int)
                print("      RGBA8 |ref-product| <= 1 on %.5f, == on %.5f" % ((np.abs(i8 - img.reshape(-1).astype(int)) <= 1).mean(),
                                                                              (i8 == img.reshape(-1).astype(int)).mean()), flush=True)
        # material on captured hits (item i is shaded as pixel i)
        m = 4000
        px = np.arange(m, dtype=np.uint32)
        rin = np.stack([np.zeros(m, np.uint32), np.zeros(m, np.uint32), px], 1)
        po, pd = osc.generate_rays(px % (s.width * s.height), rin)
        hits = ob.trace_batch(blob, po, pd)
        hits["instanceSBTOffset"] = 0
        frames = (np.arange(m) % 7).astype(np.uint32)
        depths = (np.arange(m) % 5).astype(np.int32)
        prod = rd.MaterialBatch(hits, pd, px, frames, depths)
        orc = osc.material_batch(hits, pd, px, frames, depths)
        k = hits["hit"] == 1
        for b, r in refs.items():
            rs = rg.RefScene(r, s, blob)
            got = rs.material_batch(hits, pd, frames, depths)
            for f in ("nextFactor", "nextRayOrigin", "nextRayDirection", "color"):
                x, y, z = got[f][k], prod[f][k], orc[f][k]
                print("%s material ref[%s] %-16s max|ref-product| %.3g (bit-equal rows %.4f)   max|ref-oracle| %.3g"
                      % (name, b, f, np.abs(x - y).max(), float((bits(x).reshape(-1, 12) == bits(y).reshape(-1, 12)).all(1).mean()),
                         np.abs(x - z).max()), flush=True)
    if "--time" in sys.argv:
        s = scenes.c1_cornell(480, 270, spp=4, depth=8)
        dev = scenes.DeviceScene(s)
        blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
        for b, r in refs.items():
            for local in (64, 1):
                rs = rg.RefScene(r, s, blob)
                t0 = time.time()
                ms = rs.frame(local)
                print("timing c1 480x270 4spp depth8 ref[%s] local=%d: %.1f ms (wall %.2f s)" % (b, local, ms, time.time() - t0), flush=True)
        dev.render(); dev.set_rtprop(totalSamples=0); dev.render()
        print("product same frame: %.2f ms" % rd.GetTraceStats().ms_total, flush=True)


if __name__ == "__main__":
    main()

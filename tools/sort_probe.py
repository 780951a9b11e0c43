#!/usr/bin/env python3
"""sort_probe.py -- does per-bounce ray sorting pay?  Times the cooperative extend kernel on the same
secondary rays in (a) the order the shade stage emits them (pixel-coherent origins, random directions),
(b) random order, (c) sorted by a Morton code of the origin plus the direction octant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from radiance_ray_tracing_amd import rd, scenes

def morton3(q):
    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0x3ff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x30000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x300f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x9249249)
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2_atrium"
w, h = 1920, 1080
s = scenes.CONFIGS[cfg](w, h, 1, 8)
dev = scenes.DeviceScene(s)
px = np.arange(w * h, dtype=np.uint32)
o, d = rd.GenerateBatch(px, np.stack([np.zeros_like(px), np.zeros_like(px), px], 1))
hits = rd.TraceBatch(dev.topAccelStruct, o, d)
rng = np.random.default_rng(1)
ok = hits["hit"] == 1
hp = (o + d * hits["distance"][:, None])[ok]
d2 = rng.normal(size=hp.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
o2 = (hp + 1e-3 * d2).astype(np.float32)
lo, hi = o2.min(0), o2.max(0)
q = np.clip(((o2 - lo) / (hi - lo + 1e-9) * 1023).astype(np.int64), 0, 1023)
octant = ((d2[:, 0] > 0).astype(np.uint64) | ((d2[:, 1] > 0).astype(np.uint64) << np.uint64(1)) | ((d2[:, 2] > 0).astype(np.uint64) << np.uint64(2)))
orders = {"emitted": np.arange(o2.shape[0]), "random": rng.permutation(o2.shape[0]),
          "morton(origin)": np.argsort(morton3(q), kind="stable"),
          "octant|morton": np.argsort((octant << np.uint64(30)) | morton3(q), kind="stable")}
for name, idx in orders.items():
    oo, dd = np.ascontiguousarray(o2[idx]), np.ascontiguousarray(d2[idx])
    ts = []
    for _ in range(4):
        rd.TraceBatch(dev.topAccelStruct, oo, dd, 0.001, 1000.0, 1)
        ts.append(rd.GetTraceStats().ms_extend)
    print("%s %-16s n=%d  %.3f ms" % (cfg, name, oo.shape[0], min(ts)), flush=True)

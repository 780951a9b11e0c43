#!/usr/bin/env python3
"""trav_scale.py -- kernel time of one traversal launch as a function of the ray count (GPU box):
exposes the fixed cost (ramp + tail) of the persistent cooperative kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from radiance_ray_tracing_amd import rd, scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else "c1_cornell"
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
perm = rng.permutation(o2.shape[0])
o2, d2 = o2[perm], d2[perm]
kernels = [int(k) for k in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3]
for n in (1, 64, 1024, 16384, 65536, 131072, 262144, 524288, 1048576, o2.shape[0]):
    for kernel in kernels:
        rd.SetOption("kernel", kernel)
        ts = []
        for _ in range(5):
            rd.TraceBatch(dev.topAccelStruct, o2[:n], d2[:n], 0.001, 1000.0, 1)
            ts.append(rd.GetTraceStats().ms_extend)
        print("%s n=%8d kernel=%d  %.4f ms  (%.3f ns/ray)" % (cfg, n, kernel, min(ts), 1e6 * min(ts) / n), flush=True)
rd.SetOption("kernel", 3)

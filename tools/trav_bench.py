#!/usr/bin/env python3
"""trav_bench.py -- A/B timing of the traversal kernels on captured ray batches (GPU box).
Rays: primary rays of the config + one generation of scattered secondaries from their hit points.
Prints kernel ms (HIP events) for the production (wide) and the reference-order kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from radiance_ray_tracing_amd import rd, scenes

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2_atrium"
    w, h = 960, 540
    s = scenes.CONFIGS[cfg](w, h, 1, 8)
    dev = scenes.DeviceScene(s)
    px = np.arange(w * h, dtype=np.uint32)
    rin = np.stack([np.zeros_like(px), np.zeros_like(px), px], 1)
    o, d = rd.GenerateBatch(px, rin)
    hits = rd.TraceBatch(dev.topAccelStruct, o, d, reference_order=True)
    rng = np.random.default_rng(1)
    ok = hits["hit"] == 1
    hp = (o + d * hits["distance"][:, None])[ok]
    d2 = rng.normal(size=hp.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = (hp + 1e-3 * d2).astype(np.float32)
    for name, (oo, dd) in (("primary", (o, d)), ("scattered", (o2, d2))):
        for rec in (1, 2):
            res = {}
            for mode, key in ((2, "coop"), (1, "wide"), (0, "reforder")):
                ts = []
                for _ in range(5):
                    rd.SetOption("kernel", mode); out = rd.TraceBatch(dev.topAccelStruct, oo, dd, 0.001, 1000.0, rec)
                    ts.append(rd.GetTraceStats().ms_extend)
                res[key] = (min(ts), int(out["hit"].sum()))
            print("%s %-9s rec=%d n=%d  coop %.3f  wide %.3f  reforder %.3f ms  (hits %d/%d/%d)" % (cfg, name, rec, oo.shape[0], res["coop"][0], res["wide"][0], res["reforder"][0], res["coop"][1], res["wide"][1], res["reforder"][1]), flush=True)

if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""coop_stats.py -- step statistics of the cooperative traversal engine (experiment build -DCOOP_STATS, selected with
RDX_LIB=.../librdx_stats.so): how often each step kind ran and how many lanes it served, over one frame."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from radiance_ray_tracing_amd import rd, scenes, _lib

KINDS = ["finish/refill", "shade", "steal", "leaf item", "top", "instance", "node / pool", "test"]
L = _lib.lib()
fn = L.rdx_debug_coop_stats
fn.restype = ctypes.c_int
if os.environ.get("RDX_CULL"):
    rd.SetOption("cull", int(os.environ["RDX_CULL"]))
CFGS = [("c1_cornell", 1920, 1080), ("c2_atrium", 1920, 1080), ("c1_cornell", 680, 381)]
if os.environ.get("RDX_STATS_CFG"):          # e.g. RDX_STATS_CFG=c2_atrium,c4_atrium_10m
    CFGS = [(c, 1920, 1080) for c in os.environ["RDX_STATS_CFG"].split(",")]
for cfg, w, h in CFGS:
    s = scenes.CONFIGS[cfg](w, h, 4, 8)
    dev = scenes.DeviceScene(s)
    dev.render()
    out = (ctypes.c_ulonglong * 32)()
    fn(out)                                   # clear what the warm-up frame counted
    dev.set_rtprop(totalSamples=0); dev.render()
    st = rd.GetTraceStats()
    fn(out)
    n = np.array(out[:8], np.float64); l = np.array(out[8:], np.float64)
    rays = st.rays_primary + st.rays_bounce + st.rays_shadow
    print("%s %dx%d: %d rays, %.0f wave-steps (%.1f per 64 rays)" % (cfg, w, h, rays, n.sum(), n.sum() / (rays / 64)))
    stt = np.array(out[16:24], np.float64)
    names = ["node (pool engine: instance in the pool)", "top", "instance", "leaf", "finishing (tests pending)", "done (waits for hand-over)", "free"]
    print("  lane states per iteration: " + ", ".join("%s %.1f" % (names[k], 64 * stt[k] / stt[7]) for k in range(7))
          + ", other %.1f" % (64 - 64 * stt[:7].sum() / stt[7]))
    cyc = np.array(out[24:32], np.float64)
    if cyc.sum():
        print("  wave cycles by step kind (a test that follows a pool step counts with it; 'test' = the rest): "
              + ", ".join("%s %.1f %% (%.0f cycles / step)" % (KINDS[k], 100 * cyc[k] / cyc.sum(), cyc[k] / max(n[k], 1)) for k in range(8) if cyc[k]))
    for k in range(8):
        if n[k]:
            print("  %-14s %5.1f %% of steps, %5.1f lanes / step, %6.2f lane-steps per ray" % (KINDS[k], 100 * n[k] / n.sum(), l[k] / n[k], l[k] / rays))

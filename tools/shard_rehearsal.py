#!/usr/bin/env python3
"""shard_rehearsal.py -- what one rank of an N-GPU run has to do, measured on one GPU: rd.SetShard(rank, N) on the 1080p x 4 spp x
depth 8 frame (the interleaved 64x64 tiles the multi-GPU paths use), wall time per TraceRays without per-stage events.
Prints the full-frame time, the 1/N times and the projected strong-scaling factor (before the gather, ~1 MB per rank)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import rrt_amd  # noqa: F401
from radiance_ray_tracing_amd import rd, scenes
import ctypes as C
from radiance_ray_tracing_amd import _lib

plt = rd.Platform.GetPlatform(0)
L = _lib.lib()
L.rdx_buffer_read.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
for opt in sys.argv[1:]:
    k, v = opt.split("=")
    rd.SetOption(k, int(v))


def timed(dev, n=12):
    for _ in range(3):
        dev.set_rtprop(totalSamples=0); rd.TraceRays(plt, 0, 0, 0, 1920, 1080)
    t0 = time.perf_counter()
    for _ in range(n):
        dev.set_rtprop(totalSamples=0); rd.TraceRays(plt, 0, 0, 0, 1920, 1080)
    return (time.perf_counter() - t0) / n * 1e3


for name in ("c1_cornell", "c2_atrium"):
    dev = scenes.DeviceScene(scenes.CONFIGS[name](1920, 1080, 4, 8), plt)
    rd.SetShard(0, 1, 64, 64)
    full = timed(dev)
    out = ["%s full %.2f ms" % (name, full)]
    for world in (2, 4, 8):
        ts = []
        for rank in sorted(set((0, world - 1, world // 2))):
            rd.SetShard(rank, world, 64, 64)
            ts.append(timed(dev))
        out.append("1/%d: %.2f ms (x%.2f)" % (world, max(ts), full / max(ts)))
    rd.SetShard(0, 1, 64, 64)
    print("; ".join(out), flush=True)

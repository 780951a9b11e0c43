#!/usr/bin/env python3
"""shard_frame.py <workload cfg> <rank> <world> [opt=value ...] -- a few frames of one shard, for rocprofv3 --kernel-trace (tools/gpu_shard_timeline.sh)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import rrt_amd  # noqa: F401
from radiance_ray_tracing_amd import rd, scenes
plt = rd.Platform.GetPlatform(0)
for opt in sys.argv[4:]:
    k, v = opt.split("="); rd.SetOption(k, int(v))
dev = scenes.DeviceScene(scenes.CONFIGS[sys.argv[1]](1920, 1080, 4, 8), plt)
rd.SetShard(int(sys.argv[2]), int(sys.argv[3]), 64, 64)
for _ in range(5):
    dev.set_rtprop(totalSamples=0); rd.TraceRays(plt, 0, 0, 0, 1920, 1080)

#!/usr/bin/env python3
"""multi_device_check.py <n> <out.npz> -- renders two progressive frames of a small scene on n logical devices from ONE process
(rdx_init_devices; with RDX_ALLOW_VIRTUAL_DEVICES=1 they may share a GPU) and saves imageScratch, the RGBA8 image and the ray
counts.  tests/test_gpu_parity.py compares the file with a one-device render: bit-identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import rrt_amd  # noqa: E402,F401
from radiance_ray_tracing_amd import rd, scenes  # noqa: E402


def render(n):
    plt = rd.Platform.InitDevices(n) if n > 1 else rd.Platform.GetPlatform()
    s = scenes.c1_cornell(200, 120, spp=2, depth=4, sphere_subdiv=3)
    dev = scenes.DeviceScene(s, plt)
    out = {}
    for f in range(2):
        img = dev.render()
        st = rd.GetTraceStats()
        out["scratch%d" % f] = dev.read_scratch().copy()
        out["image%d" % f] = img.copy()
        out["rays%d" % f] = np.array([st.rays_primary, st.rays_bounce, st.rays_shadow, st.closest_hits, st.pixels], np.int64)
    return out


if __name__ == "__main__":
    n = int(sys.argv[1])
    np.savez(sys.argv[2], **render(n))
    print("rendered on %d logical device(s)" % n)

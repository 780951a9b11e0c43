#!/usr/bin/env python3
"""xcc_probe.py -- HW_REG_XCC_ID of the first 64 workgroups of a launch (GPU box)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import rrt_amd  # noqa: F401
from radiance_ray_tracing_amd import _lib, rd
rd.Platform.GetPlatform(0)
out = (ctypes.c_uint32 * 64)()
print("rc", _lib.lib().rdx_debug_xcc_probe(out, 64))
print("raw", [hex(v) for v in out[:16]])
print("xcc id (low 4 bits) of blocks 0..63:", [v & 15 for v in out])

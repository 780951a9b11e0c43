#!/usr/bin/env python3
"""gather_probe.py -- time of 2^26 random 64-byte record reads per lane (the traversal kernels' node fetch) as a per-lane gather
(4 x 16-byte loads per lane) and as the quad-cooperative gather (kernels.hip quad_gather64), for tables resident in L2, in the
Infinity Cache and in HBM.  GPU only."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import rrt_amd  # noqa: F401
from radiance_ray_tracing_amd import _lib, rd
rd.Platform.GetPlatform(0)
L = _lib.lib()
L.rdx_debug_gather_probe.restype = ctypes.c_float
L.rdx_debug_gather_probe.argtypes = [ctypes.c_uint32, ctypes.c_ulonglong, ctypes.c_uint32, ctypes.c_uint32]
n = 1 << 26
for name, table in (("2 MiB (L2)", 2 << 20), ("16 MiB", 16 << 20), ("64 MiB (Infinity Cache)", 64 << 20), ("8 GiB (HBM)", 8 << 30)):
    a = L.rdx_debug_gather_probe(64, table, n, 4)
    b = L.rdx_debug_gather_probe(65, table, n, 4)
    print("%-24s per-lane gather %.3f ms (%.1f G records/s)   quad-cooperative %.3f ms (%.1f G records/s)%s" %
          (name, a, n / a / 1e6, b, n / b / 1e6, "  WRONG DATA" if b == -2.0 else ""), flush=True)

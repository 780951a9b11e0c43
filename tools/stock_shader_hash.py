#!/usr/bin/env python3
"""Prints the constants the runtime recognises the reference's stock shader by.

    python tools/stock_shader_hash.py [/root/reference/samples/shader.cl]

* the whole-text hash (comments and white space removed) selects the compiled-in HIP stages;
* the *reduced* hash -- the same text with the bodies of `material`, `shadow`, `environment`, `callHit`, `callMiss`,
  `callAnyHit` blanked -- says "this program's raygen IS the stock raygen; only its stage functions differ", which is what
  lets a user's stage functions run on the wavefront pipeline (csrc/user_shader.cpp, stage mode).
Reads the reference's file as data; nothing of it is stored here but two 64-bit numbers.
"""
import ctypes, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(root, "radiance-ray-tracing_amd", os.environ.get("RDX_LIB_NAME", "librdx.so")))
lib.rdx_debug_stage_reduced_hash.restype = ctypes.c_ulonglong
lib.rdx_debug_stage_reduced_hash.argtypes = [ctypes.c_char_p, ctypes.c_uint32]
path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/samples/shader.cl"
text = open(path, "rb").read()
print("#define RDX_STOCK_REDUCED_HASH 0x%016xull" % lib.rdx_debug_stage_reduced_hash(text, len(text)))

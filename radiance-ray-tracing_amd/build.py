"""build.py -- compiles the HIP ray-tracing core into radiance-ray-tracing_amd/librdx.so (in-tree).

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is part of the numerical
contract: traversal / intersection results must be the IEEE values of the
expressions as written.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librdx.so")
# experiment builds: RDX_DEFINES="-DFOO -DBAR" RDX_LIB_NAME=librdx_foo.so python build.py --force
EXTRA = os.environ.get("RDX_DEFINES", "").split()
if os.environ.get("RDX_LIB_NAME"):
    LIB = os.path.join(HERE, os.environ["RDX_LIB_NAME"])
SOURCES = ["kernels.hip", "rdx_runtime.cpp", "bvh_build.cpp", "scene_obj.cpp", "user_shader.cpp"]
HEADERS = ["kernels.h", "stages.h", "device_math.h", "rdx_types.h", "bvh_build.h", "sbt_generated.h", "traverse_coop.h", "traverse_pool.h", "user_shader.h",
           os.path.join("..", "..", "include", "rdx.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-x", "hip"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    # regenerate the SBT tables from samples/sbt.json -- or, for a library with another shader binding table
    # (RDX_SBT_JSON=<file> RDX_LIB_NAME=<name>.so), into a header next to that library, selected with -DRDX_SBT_HEADER
    gen = os.path.join(HERE, "..", "tools", "genSBT.py")
    sbt = []
    if os.environ.get("RDX_SBT_JSON"):
        hdr = LIB[:-3] + "_sbt.h"
        subprocess.check_call([sys.executable, gen, os.environ["RDX_SBT_JSON"], hdr], stdout=subprocess.DEVNULL)
        sbt = ['-DRDX_SBT_HEADER="%s"' % hdr]
    else:
        subprocess.check_call([sys.executable, gen], stdout=subprocess.DEVNULL)
    cmd = [HIPCC] + FLAGS + EXTRA + sbt + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)

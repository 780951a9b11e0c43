"""radiance-ray-tracing_amd -- MI355X-native ray-tracing core behind the Radiance host API.

The directory name carries a hyphen (it is the name the project brief asks for), so it is loaded
through the tiny `rrt_amd` shim at the repository root:

    import rrt_amd                                   # registers `radiance_ray_tracing_amd`
    from radiance_ray_tracing_amd import rd, scenes

Contents
    csrc/        HIP kernels (wavefront stages), C-ABI runtime, bit-exact BVH builder
    librdx.so    built in-tree by build.py (hipcc --offload-arch=gfx950)
    _lib.py      ctypes view of include/rdx.h
    rd.py        Python mirror of the reference's `namespace RD` host API (radiance/include/radiance.h)
    scenes.py    deterministic procedural scenes standing in for the reference's .glb assets
    dist.py      image-tile sharding + RCCL framebuffer gather (torch.distributed)
"""
from . import _lib, rd, scenes  # noqa: F401

__all__ = ["_lib", "rd", "scenes"]

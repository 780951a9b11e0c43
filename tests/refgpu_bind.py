"""ctypes launcher for the REFERENCE-ON-GPU oracle (oracle/_ref/ref_shader_gfx950*.co) -- test infrastructure only.

The code object is the reference's own OpenCL C device code (samples/shader.cl + radiance/shader/*.cl), compiled
where it lies by oracle/Makefile with ROCm clang for gfx950 and linked against ROCm's OpenCL builtin library, plus
thin batch wrappers (oracle/ref_gpu_dev.cl, ref_gpu_kern.cl).  This module loads it with the HIP module API
(hipModuleLoad / hipModuleLaunchKernel through ctypes on libamdhip64) and exposes numpy-in / numpy-out calls.
Nothing under radiance-ray-tracing_amd/ imports it; tests/, tests/golden/make_golden_gpu.py and bench.py's
informational `reference_kernel` leg use it as the checker / the thing timed beside the product.

Two builds of the same source:
  "p"  pinned:  -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt   (the parity contract)
  "d"  default: clang's OpenCL defaults (contraction on, 2.5-ulp divide)     (what clBuildProgram("-g -I..") would give)
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
CO = {"p": os.path.join(REF_DIR, "ref_shader_gfx950_p.co"), "d": os.path.join(REF_DIR, "ref_shader_gfx950_d.co"),
      "um": os.path.join(REF_DIR, "ref_shader_gfx950_um.co")}      # um: the reference program with tests/golden/user_material.inc as `material`

HIT_WORDS, PAYLOAD_WORDS = 28, 13
_hip = None


def available(build="p"):
    return os.path.exists(CO[build])


def hip():
    global _hip
    if _hip is None:
        last = None
        for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError as e:      # pragma: no cover
                last = e
        if _hip is None:
            raise last
        h = _hip
        h.hipGetErrorString.restype = C.c_char_p
        h.hipGetErrorString.argtypes = [C.c_int]
        h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        h.hipFree.argtypes = [C.c_void_p]
        h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        h.hipModuleLoad.argtypes = [C.POINTER(C.c_void_p), C.c_char_p]
        h.hipModuleGetFunction.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_char_p]
        h.hipModuleLaunchKernel.argtypes = [C.c_void_p] + [C.c_uint] * 6 + [C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        h.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        h.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        h.hipEventSynchronize.argtypes = [C.c_void_p]
        h.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        h.hipEventDestroy.argtypes = [C.c_void_p]
    return _hip


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, hip().hipGetErrorString(rc).decode()))


class DevBuf:
    def __init__(self, nbytes):
        self.nbytes = max(int(nbytes), 16)
        self.ptr = C.c_void_p()
        _ck(hip().hipMalloc(C.byref(self.ptr), self.nbytes), "hipMalloc")

    @classmethod
    def of(cls, arr):
        a = np.ascontiguousarray(arr)
        b = cls(a.nbytes)
        if a.nbytes:
            _ck(hip().hipMemcpy(b.ptr, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        return b

    def zero(self):
        _ck(hip().hipMemset(self.ptr, 0, self.nbytes), "hipMemset")
        return self

    def read(self, dtype, count):
        out = np.empty(count, dtype)
        if out.nbytes:
            _ck(hip().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def __del__(self):
        try:
            if self.ptr:
                hip().hipFree(self.ptr)
        except Exception:
            pass


class RefGpu:
    """One loaded code object; `launch` packs the explicit kernel arguments the way the OpenCL / HSA ABI lays them
    out (natural alignment, in order) and lets the HIP runtime fill the hidden ones from the kernel's metadata."""

    def __init__(self, build="p"):
        if not available(build):
            raise FileNotFoundError(CO[build] + " (build it with `make -C oracle` where /root/reference exists)")
        self.build = build
        self.module = C.c_void_p()
        _ck(hip().hipModuleLoad(C.byref(self.module), CO[build].encode()), "hipModuleLoad")
        self._fn = {}
        self.last_ms = 0.0

    def fn(self, name):
        if name not in self._fn:
            f = C.c_void_p()
            _ck(hip().hipModuleGetFunction(C.byref(f), self.module, name.encode()), "hipModuleGetFunction " + name)
            self._fn[name] = f
        return self._fn[name]

    def launch(self, name, args, n, local=64):
        """args: DevBuf | None (null pointer) | np.uint32 / np.int32 / np.float32 scalars"""
        blob = bytearray()
        for a in args:
            if isinstance(a, DevBuf) or a is None:
                while len(blob) % 8:
                    blob += b"\0"
                blob += np.uint64(a.ptr.value if a is not None else 0).tobytes()
            else:
                assert isinstance(a, (np.uint32, np.int32, np.float32)), type(a)
                blob += a.tobytes()
        while len(blob) % 8:
            blob += b"\0"
        buf = (C.c_uint8 * len(blob)).from_buffer(blob)
        size = C.c_size_t(len(blob))
        extra = (C.c_void_p * 5)(1, C.addressof(buf), 2, C.addressof(size), 3)
        grid = (int(n) + local - 1) // local
        if grid == 0:
            return 0.0
        e0, e1 = C.c_void_p(), C.c_void_p()
        hip().hipEventCreate(C.byref(e0)); hip().hipEventCreate(C.byref(e1))
        hip().hipEventRecord(e0, None)
        _ck(hip().hipModuleLaunchKernel(self.fn(name), grid, 1, 1, local, 1, 1, 0, None, None, extra), "launch " + name)
        hip().hipEventRecord(e1, None)
        _ck(hip().hipEventSynchronize(e1), "sync after " + name)
        ms = C.c_float(0)
        hip().hipEventElapsedTime(C.byref(ms), e0, e1)
        hip().hipEventDestroy(e0); hip().hipEventDestroy(e1)
        self.last_ms = float(ms.value)
        return self.last_ms

    # ---- batch calls ------------------------------------------------------------------------------------
    def trace(self, tlas, origins, dirs, tmin=0.001, tmax=1000.0, sbtRecordOffset=1):
        """tlas: DevBuf with the TLAS blob.  Returns HIT_DTYPE records (see tests/oracle_bind.py)."""
        from oracle_bind import HIT_DTYPE
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        bo, bd, out = DevBuf.of(o), DevBuf.of(d), DevBuf(n * HIT_WORDS * 4).zero()
        self.launch("k_ref_trace", [tlas, bo, bd, np.uint32(n), np.float32(tmin), np.float32(tmax),
                                    np.int32(sbtRecordOffset), out], n)
        return out.read(np.uint32, n * HIT_WORDS).view(HIT_DTYPE)

    def aabb(self, o, d, lo, hi):
        a = np.concatenate([np.asarray(x, np.float32).reshape(-1, 3) for x in (o, d, lo, hi)], 1)
        n = a.shape[0]
        out = DevBuf(n * 4).zero()
        self.launch("k_ref_aabb", [DevBuf.of(a), np.uint32(n), out], n)
        return out.read(np.uint32, n)

    def triangle(self, o, d, v0, v1, v2):
        a = np.concatenate([np.asarray(x, np.float32).reshape(-1, 3) for x in (o, d)], 1)
        n = a.shape[0]
        verts = np.zeros((n, 3, 4), np.float32)
        for k, v in enumerate((v0, v1, v2)):
            verts[:, k, :3] = np.asarray(v, np.float32).reshape(-1, 3)
        tris = np.zeros((n, 4), np.uint32)
        tris[:, 1], tris[:, 2], tris[:, 3] = 1, 2, np.arange(n)
        out = DevBuf(n * 32).zero()
        self.launch("k_ref_triangle", [DevBuf.of(a), DevBuf.of(tris), DevBuf.of(verts), np.uint32(n), out], n)
        r = out.read(np.uint32, n * 8).reshape(n, 8)
        return r[:, 0].copy(), r[:, 1].copy().view(np.float32), r[:, 2:5].copy().view(np.float32), r[:, 5:8].copy().view(np.float32)

    def brdf(self, packed19):
        a = np.ascontiguousarray(packed19, np.float32).reshape(-1, 19)
        n = a.shape[0]
        out = DevBuf(n * 36).zero()
        self.launch("k_ref_brdf", [DevBuf.of(a), np.uint32(n), out], n)
        return out.read(np.float32, n * 9).reshape(n, 9)


class RefScene:
    """The buffers of a scenes.Scene uploaded for the reference kernels (same arrays the product uploads; the TLAS
    blob is passed in -- the reference's own builder needs assimp and cannot be built here)."""

    def __init__(self, ref, scene, tlas_blob):
        self.ref, self.scene = ref, scene
        b = scene.buffers()
        w, h = scene.width, scene.height
        self.width, self.height = w, h
        self.rtprop_host = np.array(scene.rtprop).reshape(1).copy()
        self.rtprop = DevBuf.of(self.rtprop_host.view(np.uint8))
        self.cam = DevBuf.of(np.array(scene.camera).reshape(1).view(np.uint8))
        self.props = DevBuf.of(np.array(scene.sceneProps).reshape(1).view(np.uint8))
        self.meshInfo = DevBuf.of(b["meshInfo"].view(np.uint8)); self.vertex = DevBuf.of(b["vertex"])
        self.index = DevBuf.of(b["index"]); self.uv = DevBuf.of(b["uv"]); self.normal = DevBuf.of(b["normal"])
        self.material = DevBuf.of(b["material"].view(np.uint8))
        self.tlas = DevBuf.of(np.frombuffer(tlas_blob, np.uint8))
        self.scratch = DevBuf(w * h * 16).zero()
        self.image = DevBuf(w * h * 4).zero()

    def set_rtprop(self, **kw):
        for k, v in kw.items():
            self.rtprop_host[0][k] = v
        _ck(hip().hipMemcpy(self.rtprop.ptr, self.rtprop_host.ctypes.data, 16, 1), "hipMemcpy H2D")

    def frame(self, local=64):
        """one host frame like sample1.cpp:447-498: raygen over all pixels, then totalSamples += batchSize.
        The reference launches with local_work_size 1 (radiance.cpp:250-259); results do not depend on it."""
        n = self.width * self.height
        ms = self.ref.launch("k_ref_raygen", [self.rtprop, self.scratch, self.image, self.cam, self.props, self.meshInfo,
                                              self.vertex, self.index, self.uv, self.normal, self.material, self.tlas,
                                              np.uint32(n)], n, local)
        self.set_rtprop(totalSamples=int(self.rtprop_host[0]["totalSamples"]) + int(self.rtprop_host[0]["batchSize"]))
        return ms

    def read_scratch(self):
        return self.scratch.read(np.float32, self.width * self.height * 4)

    def read_image(self):
        return self.image.read(np.uint8, self.width * self.height * 4)

    def generate(self, rand3):
        """primary rays for pixels 0..n-1 (generateRay takes the pixel from get_global_id)"""
        r = np.ascontiguousarray(rand3, np.uint32).reshape(-1, 3)
        n = r.shape[0]
        o, d = DevBuf(n * 12), DevBuf(n * 12)
        self.ref.launch("k_ref_generate", [self.cam, DevBuf.of(r), np.uint32(n), o, d], n)
        return o.read(np.float32, n * 3).reshape(n, 3), d.read(np.float32, n * 3).reshape(n, 3)

    def material_batch(self, hits, ray_dirs, frame_ids, depths):
        """`material` on captured hits; item i is shaded as pixel i"""
        from oracle_bind import HIT_DTYPE, PAYLOAD_DTYPE
        h = np.ascontiguousarray(hits, HIT_DTYPE)
        n = h.shape[0]
        out = DevBuf(n * PAYLOAD_WORDS * 4).zero()
        self.ref.launch("k_ref_material", [DevBuf.of(h.view(np.uint8)), DevBuf.of(np.ascontiguousarray(ray_dirs, np.float32)),
                                           DevBuf.of(np.ascontiguousarray(frame_ids, np.uint32)),
                                           DevBuf.of(np.ascontiguousarray(depths, np.int32)), np.uint32(n),
                                           self.cam, self.props, self.meshInfo, self.vertex, self.index, self.uv, self.normal,
                                           self.material, self.tlas, out], n)
        return out.read(np.uint32, n * PAYLOAD_WORDS).view(PAYLOAD_DTYPE)

    def trace(self, origins, dirs, tmin=0.001, tmax=1000.0, sbtRecordOffset=1):
        return self.ref.trace(self.tlas, origins, dirs, tmin, tmax, sbtRecordOffset)

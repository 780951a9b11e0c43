"""rd.py -- Python mirror of the reference's host API `namespace RD`
(radiance/include/radiance.h:86-174), over the C ABI of include/rdx.h.

Same function names, argument order and meaning as the reference, so the tests read like the
reference's own sample (samples/sample1.cpp:363-498).  Differences, all deliberate:

  * errors raise `RadianceError` (the reference prints and calls exit(-1), clcontext.h:27-47;
    the C++ facade in include/radiance.h keeps that behaviour);
  * `Mesh.vertexData` / `indexData` are numpy arrays ((N,3) float32 / (M,3) uint32) instead of
    std::vector<Vec3> / std::vector<Triangle>;
  * buffers accept numpy arrays or bytes for `data`.
"""
import ctypes as C

import numpy as np

from . import _lib

CHANNEL = 4
RD_CHANNEL = CHANNEL

# enum DescriptorType (radiance.h:21-29)
ACCEL_STRUCT_TYPE, IMAGE_TYPE, IMAGE_ARRAY_TYPE, IMAGE_SAMPLER_TYPE, BUFFER_TYPE, TEX_ARRAY_TYPE = range(6)

# addressing / filter modes keep the OpenCL values the reference forwards (radiance.h:94-112)
RD_ADDRESS_CLAMP_TO_EDGE, RD_ADDRESS_CLAMP, RD_ADDRESS_REPEAT, RD_ADDRESS_MIRRORED_REPEAT = 0x1131, 0x1132, 0x1133, 0x1134
RD_FILTER_NEAREST, RD_FILTER_LINEAR = 0x1140, 0x1141


class RadianceError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise RadianceError(_lib.last_error())


def _handle(h, what):
    if not h:
        raise RadianceError("%s: %s" % (what, _lib.last_error()))
    return h


# ---- POD structs (radiance/src/core.h:103-158) as numpy dtypes ----------------------------------
RayTraceProperties = np.dtype([("totalSamples", "<u4"), ("batchSize", "<u4"), ("depth", "<u4"), ("debug", "<u4")])
Material = np.dtype([("albedo", "<f4", 4), ("metallic", "<f4"), ("roughness", "<f4"), ("transmission", "<f4"),
                     ("ior", "<f4"), ("albedoTexIdx", "<i4"), ("metallicTexIdx", "<i4"),
                     ("roughnessTexIdx", "<i4"), ("normalTexIdx", "<i4")])
MeshInfo = np.dtype([("vertexOffset", "<i4"), ("indexOffset", "<i4"), ("uvOffset", "<i4"), ("normalOffset", "<i4"),
                     ("materialIndex", "<i4"), ("_0", "<i4"), ("_1", "<i4"), ("_2", "<i4")])
DirLight = np.dtype([("direction", "<f4", 4), ("color", "<f4", 4)])
SceneProperties = np.dtype([("lightCount", "<u4", 4), ("lights", DirLight, 5)])
PhysicalCamera = np.dtype([("widthPixel", "<f4"), ("heightPixel", "<f4"), ("focalLength", "<f4"),
                           ("sensorWidth", "<f4"), ("focalDistance", "<f4"), ("fStop", "<f4"),
                           ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("wx", "<f4"), ("wy", "<f4"), ("wz", "<f4")])
assert (RayTraceProperties.itemsize, Material.itemsize, MeshInfo.itemsize, DirLight.itemsize,
        SceneProperties.itemsize, PhysicalCamera.itemsize) == (16, 48, 32, 32, 176, 48)


class Mesh:
    """struct Mesh (radiance.h:44-48)"""

    def __init__(self, vertexData=None, indexData=None):
        self.vertexData = np.zeros((0, 3), np.float32) if vertexData is None else vertexData
        self.indexData = np.zeros((0, 3), np.uint32) if indexData is None else indexData


class BottomAccelStruct:
    def __init__(self, handle):
        self.handle = handle

    @property
    def data(self):
        """the host blob (struct _BottomAccelStruct::data, radiance.h:52-58)"""
        n = C.c_uint32(0)
        p = _lib.lib().rdx_blas_data(self.handle, C.byref(n))
        return bytes((C.c_uint8 * n.value).from_address(p))

    @property
    def max_depth(self):
        return _lib.lib().rdx_blas_max_depth(self.handle)


class Instance:
    """struct Instance (radiance.h:64-74); transform is a row-major 4x4"""

    def __init__(self, transform=None, SBTOffset=0, customInstanceID=0, bottomAccelStruct=None):
        self.transform = np.eye(4, dtype=np.float32) if transform is None else np.asarray(transform, np.float32).reshape(4, 4)
        self.SBTOffset = SBTOffset
        self.customInstanceID = customInstanceID
        self.bottomAccelStruct = bottomAccelStruct


class Buffer:
    """RD::Buffer / Image / TopAccelStruct (all `cl_mem` in the reference, radiance.h:12-19)"""

    def __init__(self, handle, size, keepalive=None):
        self.handle = handle
        self.size = size
        self._keepalive = keepalive     # e.g. the torch tensor whose memory is wrapped

    @property
    def device_ptr(self):
        return _lib.lib().rdx_buffer_device_ptr(self.handle)


class PipelineCreateInfo:
    """struct PipelineCreateInfo (radiance.h:81-88)"""

    def __init__(self, maxRayRecursionDepth=1, layout=(), modules=(), groups=()):
        self.maxRayRecursionDepth = maxRayRecursionDepth
        self.layout = list(layout)
        self.modules = list(modules)
        self.groups = list(groups)


class Platform:
    """struct Platform (radiance.h:146-174): process-wide singleton owning the device and stream."""
    _instance = None

    def __init__(self):
        self.activePipeline = None
        self.initialized = False

    @staticmethod
    def GetPlatform(device=-1):
        if Platform._instance is None:
            Platform._instance = Platform()
        p = Platform._instance
        if not p.initialized:
            _check(_lib.lib().rdx_init(device))
            p.initialized = True
        return p

    @staticmethod
    def InitDevices(n, ordinals=None):
        """Extension: render every TraceRays frame on `n` devices from this one process (rdx_init_devices); call before any
        buffer is created.  Returns the platform."""
        p = Platform.GetPlatform(-1 if ordinals is None else ordinals[0])
        arr = (C.c_int * n)(*ordinals) if ordinals is not None else None
        _check(_lib.lib().rdx_init_devices(int(n), arr))
        return p

    @staticmethod
    def device_name():
        buf = C.create_string_buffer(256)
        _check(_lib.lib().rdx_device_name(buf, 256))
        return buf.value.decode()


def _as_bytes_ptr(data, size):
    if isinstance(data, np.ndarray):
        arr = np.ascontiguousarray(data)
        if arr.nbytes < size:
            raise RadianceError("host array (%d bytes) smaller than the requested transfer (%d)" % (arr.nbytes, size))
        return arr, arr.ctypes.data
    if isinstance(data, (bytes, bytearray)):
        buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
        if len(data) < size:
            raise RadianceError("host bytes smaller than the requested transfer")
        return buf, C.addressof(buf)
    raise TypeError("data must be a numpy array or bytes")


# ---- acceleration structures (radiance.h:88-92) ---------------------------------------------------
def BuildAccelStruct(platform, what):
    """Both overloads of RD::BuildAccelStruct: Mesh -> BottomAccelStruct, [Instance] -> TopAccelStruct."""
    L = _lib.lib()
    if isinstance(what, Mesh):
        v = np.ascontiguousarray(what.vertexData, np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(what.indexData, np.uint32).reshape(-1, 3)
        h = L.rdx_blas_build(v.ctypes.data, v.shape[0], i.ctypes.data, i.shape[0])
        return BottomAccelStruct(_handle(h, "BuildAccelStruct(Mesh)"))
    insts = list(what)
    arr = _instance_array(insts)
    h = L.rdx_tlas_build(arr, len(insts))
    _handle(h, "BuildAccelStruct(instances)")
    return Buffer(h, L.rdx_buffer_size(h))


def BuildAccelStructs(platform, meshes):
    """[Mesh] -> [BottomAccelStruct], built concurrently inside the library (rdx_blas_build_many); same results as
    BuildAccelStruct(platform, mesh) for each mesh in turn."""
    L = _lib.lib()
    meshes = list(meshes)
    n = len(meshes)
    if n == 0:
        return []
    vs = [np.ascontiguousarray(m.vertexData, np.float32).reshape(-1, 3) for m in meshes]
    ts = [np.ascontiguousarray(m.indexData, np.uint32).reshape(-1, 3) for m in meshes]
    vp = (C.c_void_p * n)(*[v.ctypes.data for v in vs])
    tp = (C.c_void_p * n)(*[t.ctypes.data for t in ts])
    nv = (C.c_uint32 * n)(*[v.shape[0] for v in vs])
    nt = (C.c_uint32 * n)(*[t.shape[0] for t in ts])
    out = (C.c_void_p * n)()
    _check(L.rdx_blas_build_many(n, vp, nv, tp, nt, out))
    return [BottomAccelStruct(_handle(out[i], "BuildAccelStructs")) for i in range(n)]


def _instance_array(instances):
    arr = (_lib.rdx_instance * max(len(instances), 1))()
    for k, inst in enumerate(instances):
        m = np.asarray(inst.transform, np.float32).reshape(16)
        for j in range(16):
            arr[k].transform[j] = float(m[j])
        arr[k].SBTOffset = inst.SBTOffset
        arr[k].customInstanceID = inst.customInstanceID
        arr[k].bottomAccelStruct = inst.bottomAccelStruct.handle if inst.bottomAccelStruct else None
    return arr


def BuildTopAccelStructBlob(instances):
    """Host-only TLAS build (no GPU): returns (blob bytes, max depth).  Extension used by the CPU tests."""
    L = _lib.lib()
    instances = list(instances)
    arr = _instance_array(instances)
    n, d = C.c_uint32(0), C.c_int(0)
    p = L.rdx_tlas_build_blob(arr, len(instances), C.byref(n), C.byref(d))
    _handle(p, "BuildAccelStruct(instances)")
    blob = bytes((C.c_uint8 * n.value).from_address(p))
    L.rdx_free(p)
    return blob, d.value


def TopAccelStructToFile(platform, accelStruct, path):
    _check(_lib.lib().rdx_tlas_to_file(accelStruct.handle, str(path).encode()))


def FileToTopAccelStruct(platform, path):
    """The reference returns through an out-parameter (radiance.h:92); here the handle is returned."""
    L = _lib.lib()
    h = _handle(L.rdx_tlas_from_file(str(path).encode()), "FileToTopAccelStruct")
    return Buffer(h, L.rdx_buffer_size(h))


# ---- resources (radiance.h:115-128) ----------------------------------------------------------------
def CreateBuffer(platform, size):
    h = _handle(_lib.lib().rdx_buffer_create(int(size)), "CreateBuffer")
    return Buffer(h, int(size))


def CreateImage(platform, width, height):
    """radiance.cpp:86-93: an RGBA8 image is a width*height*4-byte buffer"""
    return CreateBuffer(platform, int(width) * int(height) * CHANNEL)


class Sampler:
    def __init__(self, handle, addressingMode, filterMode):
        self.handle, self.addressingMode, self.filterMode = handle, addressingMode, filterMode


def CreateImageArray(platform, width, height, arraySize):
    """radiance.cpp:96-120: `arraySize` RGBA8 images of width x height (CL_RGBA / CL_UNSIGNED_INT8)"""
    h = _handle(_lib.lib().rdx_image_array_create(int(width), int(height), int(arraySize)), "CreateImageArray")
    b = Buffer(h, int(width) * int(height) * int(arraySize) * 4)
    b.width, b.height, b.arraySize = int(width), int(height), int(arraySize)
    return b


def CreateSampler(platform, addressingMode, filterMode):
    """radiance.cpp:122-130 (normalized coordinates)"""
    return Sampler(_handle(_lib.lib().rdx_sampler_create(int(addressingMode), int(filterMode)), "CreateSampler"), addressingMode, filterMode)


def WriteImage(platform, handle, width, height, arrayIndex, data):
    """radiance.cpp:214-224: the (width, height) region at the origin of layer `arrayIndex`, RGBA8, rows tightly packed"""
    keep, ptr = _as_bytes_ptr(data, int(width) * int(height) * 4)
    _check(_lib.lib().rdx_image_write(handle.handle, int(width), int(height), int(arrayIndex), ptr))


def ReadImage(platform, handle, width, height, arrayIndex, data=None):
    """radiance.cpp:202-212"""
    if data is None:
        data = np.empty((int(height), int(width), 4), np.uint8)
    _check(_lib.lib().rdx_image_read(handle.handle, int(width), int(height), int(arrayIndex), data.ctypes.data))
    return data


def WrapDeviceMemory(platform, device_ptr, size, keepalive=None):
    """Extension: adopt device memory owned by the caller (e.g. a torch tensor) as an RD::Buffer."""
    h = _handle(_lib.lib().rdx_buffer_wrap(C.c_void_p(int(device_ptr)), int(size)), "WrapDeviceMemory")
    return Buffer(h, int(size), keepalive)


def WriteBuffer(platform, handle, size, data, offset=0):
    keep, ptr = _as_bytes_ptr(data, size)
    _check(_lib.lib().rdx_buffer_write(handle.handle, int(offset), int(size), ptr))


def ReadBuffer(platform, handle, size, data=None, offset=0):
    """Reads `size` bytes; fills `data` (numpy array) if given and returns it, else returns bytes-like uint8 array."""
    if data is None:
        data = np.empty(int(size), np.uint8)
    if not data.flags["C_CONTIGUOUS"] or data.nbytes < size:
        raise RadianceError("ReadBuffer: destination must be C-contiguous and large enough")
    _check(_lib.lib().rdx_buffer_read(handle.handle, int(offset), int(size), data.ctypes.data))
    return data


# ---- pipeline (radiance.h:130-144) -------------------------------------------------------------------
def CreateDescriptorSet(handles):
    return list(handles)


def CreatePipelineLayout(descriptorTypes):
    return list(descriptorTypes)


def CreateShaderModule(platform, code, size, name):
    if isinstance(code, str):
        code = code.encode()
    h = _handle(_lib.lib().rdx_shader_module_create(code, int(size), name.encode() if isinstance(name, str) else name),
                "CreateShaderModule")
    return h


def SetShaderIncludePath(path):
    """-I directory for user shader programs (the reference's SHADER_LIB_PATH, radiance.h:7)"""
    _check(_lib.lib().rdx_shader_include_path(str(path).encode() if path else None))


def CreatePipeline(pipelineCreateInfo):
    return pipelineCreateInfo


def BindPipeline(platform, pipeline):
    platform.activePipeline = pipeline
    if not pipeline.modules:
        raise RadianceError("BindPipeline: pipeline has no shader module")
    _check(_lib.lib().rdx_bind_pipeline(pipeline.modules[0]))


def BindDescriptorSet(platform, descriptorSet):
    n = len(descriptorSet)
    arr = (C.c_void_p * max(n, 1))()
    for i, h in enumerate(descriptorSet):
        arr[i] = h.handle if isinstance(h, (Buffer, Sampler)) else None
    _check(_lib.lib().rdx_bind_descriptor_set(arr, n))


def TraceRays(platform, raygenGroupIndex, missGroupIndex, hitGroupIndex, width, height):
    _check(_lib.lib().rdx_trace_rays(raygenGroupIndex, missGroupIndex, hitGroupIndex, int(width), int(height)))


# ---- extensions ----------------------------------------------------------------------------------------
def GetTraceStats():
    st = _lib.rdx_trace_stats()
    _check(_lib.lib().rdx_get_trace_stats(C.byref(st)))
    return st


def GetVisitProfile(max_bounces=64):
    """(bounces, 2, 4) uint64: per bounce, per ray class (radiance, shadow): top nodes, instance visits,
    bottom nodes, triangle tests of the reference algorithm (after a frame traced with count_visits)"""
    out = np.zeros(8 * max_bounces, np.uint64)
    n = _lib.lib().rdx_get_visit_profile(out.ctypes.data, max_bounces)
    if n < 0:
        raise RadianceError(_lib.last_error())
    return out[:8 * n].reshape(n, 2, 4)


def GetBounceCounts(n=16):
    out = np.zeros(n, np.uint64)
    _check(_lib.lib().rdx_get_bounce_counts(out.ctypes.data, n))
    return out


def SetOption(name, value):
    _check(_lib.lib().rdx_set_option(name.encode(), int(value)))


def SetProfiling(on):
    _check(_lib.lib().rdx_set_profiling(1 if on else 0))


def SetShard(rank, world, tile_w=64, tile_h=64):
    _check(_lib.lib().rdx_set_shard(rank, world, tile_w, tile_h))


HIT_DTYPE = np.dtype([("hitPoint", "<f4", 3), ("distance", "<f4"), ("primitiveIndex", "<u4"), ("instanceIndex", "<u4"),
                      ("instanceCustomIndex", "<u4"), ("instanceSBTOffset", "<u4"), ("barycentric", "<f4", 3),
                      ("hit", "<u4"), ("transform", "<f4", 16)])
PAYLOAD_DTYPE = np.dtype([("color", "<f4", 3), ("hit", "<u4"), ("nextFactor", "<f4", 3),
                          ("nextRayOrigin", "<f4", 3), ("nextRayDirection", "<f4", 3)])
assert HIT_DTYPE.itemsize == C.sizeof(_lib.rdx_hit) and PAYLOAD_DTYPE.itemsize == C.sizeof(_lib.rdx_payload)


def TraceBatch(tlas, origins, dirs, tmin=0.001, tmax=1000.0, sbtRecordOffset=1, count_visits=False, reference_order=False):
    """Test seam: closest-hit (1) / any-hit (2) traversal of explicit rays -> structured array of HitData.
    Default = the production kernel; reference_order / count_visits use the reference-order kernel."""
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    out = np.zeros(o.shape[0], HIT_DTYPE)
    visit = np.zeros(4, np.uint64)
    mode = 1 if (reference_order or count_visits) else 0
    _check(_lib.lib().rdx_trace_batch(tlas.handle, o.ctypes.data, d.ctypes.data, o.shape[0], tmin, tmax,
                                      sbtRecordOffset, mode, out.ctypes.data, visit.ctypes.data if count_visits else None))
    return (out, visit) if count_visits else out


def MaterialBatch(hits, ray_dirs, pixels, frame_ids, depths):
    h = np.ascontiguousarray(hits, HIT_DTYPE)
    d = np.ascontiguousarray(ray_dirs, np.float32).reshape(-1, 3)
    p = np.ascontiguousarray(pixels, np.uint32)
    f = np.ascontiguousarray(frame_ids, np.uint32)
    dp = np.ascontiguousarray(depths, np.int32)
    out = np.zeros(h.shape[0], PAYLOAD_DTYPE)
    _check(_lib.lib().rdx_material_batch(h.ctypes.data, d.ctypes.data, p.ctypes.data, f.ctypes.data, dp.ctypes.data,
                                         h.shape[0], out.ctypes.data))
    return out


def GenerateBatch(pixels, rand_inputs):
    p = np.ascontiguousarray(pixels, np.uint32)
    r = np.ascontiguousarray(rand_inputs, np.uint32).reshape(-1, 3)
    o = np.zeros((p.shape[0], 3), np.float32)
    d = np.zeros((p.shape[0], 3), np.float32)
    _check(_lib.lib().rdx_generate_batch(p.ctypes.data, r.ctypes.data, p.shape[0], o.ctypes.data, d.ctypes.data))
    return o, d


def Pcg3dBatch(inputs):
    r = np.ascontiguousarray(inputs, np.uint32).reshape(-1, 3)
    out = np.zeros((r.shape[0], 3), np.float32)
    _check(_lib.lib().rdx_pcg3d_batch(r.ctypes.data, out.ctypes.data, r.shape[0]))
    return out

"""scenes.py -- deterministic procedural scenes.

The reference loads .glb assets through assimp (tools/sceneBuilder.cpp:27-285); neither assimp nor
any asset exists offline, so measurement and tests use generated scenes whose buffers have exactly
the layout `RD::Scene::Load` produces:

    vertexData / normalData / uvData : float32, 3 floats per vertex, meshes concatenated
                                       (sceneBuilder.cpp:42-47,83-88: uv is stored as a Vec3 too)
    indexData                        : uint32, 3 per triangle, mesh-local vertex indices
    meshInfoData                     : one MeshInfo per INSTANCE -- the live shader indexes it with
                                       hitData->instanceIndex (samples/shader.cl:310), which equals the
                                       mesh index only when instances and meshes are one-to-one
    materialData                     : Material[], factor-only (all *TexIdx = -1: texture fetches are
                                       stubbed to 0 in the live shader, shader.cl:379-445)

Everything is derived from integer seeds; the oracle and the GPU path always consume the same arrays
generated in the same process.

Configs (BASELINE.json / SURVEY.md 8d):
    c0_two_boxes      two 12-triangle boxes, 256x256, 1 spp, depth 1        (plumbing)
    c1_cornell        Cornell-like box + 2 boxes + icosphere, 8 instances    (sample1 stand-in)
    c2_atrium         Sponza-class atrium, ~262k unique triangles, 25 instances
    c4_atrium_10m     San-Miguel-scale: the same generator at 10.4 M unique triangles
"""
import math
import os

import numpy as np

from . import rd

F = np.float32


# ---------------------------------------------------------------------------------------------------
# mesh primitives: each returns (vertices (N,3) f32, triangles (M,3) u32, normals (N,3) f32, uvs (N,3) f32)
# ---------------------------------------------------------------------------------------------------
def _finish(v, t, n, uv=None):
    v = np.ascontiguousarray(v, F) + F(0.0)          # +0.0 turns -0.0 into +0.0
    n = np.ascontiguousarray(n, F) + F(0.0)
    if uv is None:
        uv = np.zeros_like(v)
    return v, np.ascontiguousarray(t, np.uint32), n, np.ascontiguousarray(uv, F)


def quad(p0, p1, p2, p3, normal):
    """two triangles p0-p1-p2, p0-p2-p3"""
    v = np.array([p0, p1, p2, p3], F)
    t = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    n = np.tile(np.array(normal, F), (4, 1))
    uv = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], F)
    return _finish(v, t, n, uv)


def box(lo, hi):
    """axis-aligned box, 24 vertices (flat normals), 12 triangles"""
    lo = np.array(lo, F)
    hi = np.array(hi, F)
    faces = []
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces.append(([x0, y0, z0], [x0, y0, z1], [x0, y1, z1], [x0, y1, z0], [-1, 0, 0]))
    faces.append(([x1, y0, z0], [x1, y1, z0], [x1, y1, z1], [x1, y0, z1], [1, 0, 0]))
    faces.append(([x0, y0, z0], [x1, y0, z0], [x1, y0, z1], [x0, y0, z1], [0, -1, 0]))
    faces.append(([x0, y1, z0], [x0, y1, z1], [x1, y1, z1], [x1, y1, z0], [0, 1, 0]))
    faces.append(([x0, y0, z0], [x0, y1, z0], [x1, y1, z0], [x1, y0, z0], [0, 0, -1]))
    faces.append(([x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1], [0, 0, 1]))
    vs, ts, ns, us = [], [], [], []
    for k, (a, b, c, d, nn) in enumerate(faces):
        v, t, n, uv = quad(a, b, c, d, nn)
        vs.append(v); ts.append(t + 4 * k); ns.append(n); us.append(uv)
    return _finish(np.concatenate(vs), np.concatenate(ts), np.concatenate(ns), np.concatenate(us))


def icosphere(subdiv, radius=1.0):
    """unit icosahedron subdivided `subdiv` times: 20*4^subdiv triangles, smooth normals"""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
             [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]]
    faces = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
             [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11],
             [6, 2, 10], [8, 6, 7], [9, 8, 1]]
    v = np.array(verts, np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(faces, np.int64)
    for _ in range(subdiv):
        edges = {}
        vl = list(map(tuple, v))
        nf = []

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in edges:
                m = (np.array(vl[a]) + np.array(vl[b])) * 0.5
                m /= np.linalg.norm(m)
                vl.append(tuple(m))
                edges[key] = len(vl) - 1
            return edges[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v = np.array(vl, np.float64)
        f = np.array(nf, np.int64)
    n = v.astype(F)
    return _finish((v * radius).astype(F), f.astype(np.uint32), n)


def _hash01(ix, iy, seed):
    """deterministic per-vertex noise in [0,1): PCG-style integer hash (same constants as math.cl:12-15)"""
    x = (ix.astype(np.uint64) * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xffffffff)
    y = (iy.astype(np.uint64) * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xffffffff)
    z = (np.uint64(seed) * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xffffffff)
    x = (x + y * z) & np.uint64(0xffffffff)
    y = (y + z * x) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(16)
    x = (x + y * np.uint64(747796405)) & np.uint64(0xffffffff)
    x ^= x >> np.uint64(15)
    return (x.astype(np.float64) / 4294967296.0)


def heightfield(origin, du, dv, normal, nu, nv, amplitude, seed, wave=0.0):
    """nu x nv cells spanning origin + a*du + b*dv (a,b in [0,1]), displaced along `normal` by noise
    (+ optional sinusoidal `wave`).  2*nu*nv triangles, smooth normals."""
    origin = np.array(origin, np.float64); du = np.array(du, np.float64); dv = np.array(dv, np.float64)
    normal = np.array(normal, np.float64)
    a = np.arange(nu + 1, dtype=np.float64) / nu
    b = np.arange(nv + 1, dtype=np.float64) / nv
    A, B = np.meshgrid(a, b, indexing="ij")
    ia, ib = np.meshgrid(np.arange(nu + 1), np.arange(nv + 1), indexing="ij")
    h = amplitude * (_hash01(ia, ib, seed) - 0.5)
    if wave:
        h = h + wave * np.sin(A * 9.0 + seed) * np.cos(B * 7.0 - seed)
    P = origin[None, None, :] + A[..., None] * du + B[..., None] * dv + h[..., None] * normal
    # smooth normals from central differences
    dPa = np.gradient(P, axis=0)
    dPb = np.gradient(P, axis=1)
    N = np.cross(dPa, dPb)
    N /= np.maximum(np.linalg.norm(N, axis=2, keepdims=True), 1e-20)
    if np.dot(N[0, 0], normal) < 0:
        N = -N
    idx = (ia * (nv + 1) + ib)
    q00 = idx[:-1, :-1].ravel(); q10 = idx[1:, :-1].ravel(); q11 = idx[1:, 1:].ravel(); q01 = idx[:-1, 1:].ravel()
    t = np.concatenate([np.stack([q00, q10, q11], 1), np.stack([q00, q11, q01], 1)])
    uv = np.stack([A.ravel(), B.ravel(), np.zeros(A.size)], 1)
    return _finish(P.reshape(-1, 3).astype(F), t.astype(np.uint32), N.reshape(-1, 3).astype(F), uv.astype(F))


def cylinder(base, radius, height, segs, rings, bump, seed):
    """vertical tessellated column with radial noise: 2*segs*rings triangles"""
    base = np.array(base, np.float64)
    th = np.arange(segs + 1, dtype=np.float64) / segs * 2.0 * math.pi
    hh = np.arange(rings + 1, dtype=np.float64) / rings
    T, H = np.meshgrid(th, hh, indexing="ij")
    it, ih = np.meshgrid(np.arange(segs + 1) % segs, np.arange(rings + 1), indexing="ij")
    r = radius * (1.0 + bump * (_hash01(it, ih, seed) - 0.5) + 0.08 * np.sin(H * 40.0))
    P = np.stack([base[0] + r * np.cos(T), base[1] + H * height, base[2] + r * np.sin(T)], 2)
    N = np.stack([np.cos(T), np.zeros_like(T), np.sin(T)], 2)
    idx = np.arange((segs + 1) * (rings + 1)).reshape(segs + 1, rings + 1)
    q00 = idx[:-1, :-1].ravel(); q10 = idx[1:, :-1].ravel(); q11 = idx[1:, 1:].ravel(); q01 = idx[:-1, 1:].ravel()
    t = np.concatenate([np.stack([q00, q11, q10], 1), np.stack([q00, q01, q11], 1)])
    uv = np.stack([T.ravel() / (2 * math.pi), H.ravel(), np.zeros(T.size)], 1)
    return _finish(P.reshape(-1, 3).astype(F), t.astype(np.uint32), N.reshape(-1, 3).astype(F), uv.astype(F))


# ---------------------------------------------------------------------------------------------------
# scene container
# ---------------------------------------------------------------------------------------------------
def material(albedo, metallic=0.0, roughness=0.5, transmission=0.0, ior=1.45):
    m = np.zeros((), rd.Material)
    m["albedo"] = (albedo[0], albedo[1], albedo[2], 1.0)
    m["metallic"], m["roughness"], m["transmission"], m["ior"] = metallic, roughness, transmission, ior
    m["albedoTexIdx"] = m["metallicTexIdx"] = m["roughnessTexIdx"] = m["normalTexIdx"] = -1
    return m


def translate(x, y, z):
    m = np.eye(4, dtype=F)
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    return m


def rotate_y(deg):
    c, s = math.cos(math.radians(deg)), math.sin(math.radians(deg))
    m = np.eye(4, dtype=F)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def scale(sx, sy, sz):
    m = np.eye(4, dtype=F)
    m[0, 0], m[1, 1], m[2, 2] = sx, sy, sz
    return m


# samples/sample1.cpp:38-88 : Blender -> renderer conventions of the reference's sample
def blender_camera(width, height, focalLength, sensorWidth, focalDistance, fStop, loc, rot_deg):
    def rad(deg):
        return F(F(deg) / F(180.0) * F(3.14159))
    cam = np.zeros((), rd.PhysicalCamera)
    cam["widthPixel"], cam["heightPixel"] = width, height
    cam["focalLength"], cam["sensorWidth"], cam["focalDistance"], cam["fStop"] = focalLength, sensorWidth, focalDistance, fStop
    cam["x"], cam["y"], cam["z"] = loc[0], loc[2], -loc[1]
    cam["wx"] = rad(-90.0 - rot_deg[0])
    cam["wy"] = rad(rot_deg[2] - 180.0)
    cam["wz"] = rad(rot_deg[1] - 180.0)
    return cam


def blender_dir_light(xDeg, zDeg, intensity):
    xr = (-90.0 - (-xDeg)) / 180.0 * 3.14159
    zr = zDeg / 180.0 * 3.14159
    d = np.array([0.0, 0.0, -1.0])
    rx = np.array([[1, 0, 0], [0, math.cos(xr), -math.sin(xr)], [0, math.sin(xr), math.cos(xr)]])
    ry = np.array([[math.cos(zr), 0, math.sin(zr)], [0, 1, 0], [-math.sin(zr), 0, math.cos(zr)]])
    d = ry @ (rx @ d)
    sp = np.zeros((), rd.SceneProperties)
    sp["lightCount"][0] = 1
    sp["lights"][0]["direction"] = (d[0], d[1], d[2], 0.0)
    sp["lights"][0]["color"] = (intensity, intensity, intensity, 1.0)
    return sp


class Scene:
    """Host-side description; `.buffers()` yields the exact arrays Scene::Load would upload."""

    def __init__(self, name):
        self.name = name
        self.meshes = []        # (v, t, n, uv)
        self.instances = []     # (mesh index, 4x4 transform, material index)
        self.materials = []
        self.camera = None
        self.sceneProps = None
        self.rtprop = np.zeros((), rd.RayTraceProperties)
        self.sbt_offsets = {}   # instance number -> SBTOffset (default 0, as the reference's loader writes: sceneBuilder.cpp:302)

    def add_mesh(self, mesh):
        self.meshes.append(mesh)
        return len(self.meshes) - 1

    def add_instance(self, mesh_index, transform=None, material_index=0, sbt_offset=0):
        self.instances.append((mesh_index, np.eye(4, dtype=F) if transform is None else np.asarray(transform, F), material_index))
        if sbt_offset:
            self.sbt_offsets[len(self.instances) - 1] = int(sbt_offset)     # instanceShaderBindingTableRecordOffset

    @property
    def width(self):
        return int(self.camera["widthPixel"])

    @property
    def height(self):
        return int(self.camera["heightPixel"])

    def triangle_count(self, unique=True):
        if unique:
            return int(sum(m[1].shape[0] for m in self.meshes))
        return int(sum(self.meshes[i][1].shape[0] for i, _, _ in self.instances))

    def buffers(self):
        vo, io = [0], [0]
        for v, t, n, uv in self.meshes:
            vo.append(vo[-1] + v.shape[0] * 3)
            io.append(io[-1] + t.shape[0] * 3)
        vertex = np.concatenate([m[0].reshape(-1) for m in self.meshes]).astype(F)
        index = np.concatenate([m[1].reshape(-1) for m in self.meshes]).astype(np.uint32)
        normal = np.concatenate([m[2].reshape(-1) for m in self.meshes]).astype(F)
        uv = np.concatenate([m[3].reshape(-1) for m in self.meshes]).astype(F)
        info = np.zeros(len(self.instances), rd.MeshInfo)
        for k, (mi, _, mat) in enumerate(self.instances):
            info[k]["vertexOffset"] = vo[mi]
            info[k]["indexOffset"] = io[mi]
            info[k]["uvOffset"] = vo[mi]
            info[k]["normalOffset"] = vo[mi]
            info[k]["materialIndex"] = mat
        mats = np.array(self.materials, rd.Material)
        return dict(meshInfo=info, vertex=vertex, index=index, uv=uv, normal=normal, material=mats)


class DeviceScene:
    """A Scene uploaded through the RD:: API, wired exactly like samples/sample1.cpp:363-411."""

    def __init__(self, scene, plt=None, shader_text="__kernel void raygen() {}"):
        self.scene = scene
        self.rtprop = np.array(scene.rtprop).copy()     # device-side RayTraceProperties as last written
        self.plt = plt or rd.Platform.GetPlatform()
        plt = self.plt
        w, h = scene.width, scene.height
        self.width, self.height = w, h
        b = scene.buffers()
        self.rdRTProp = rd.CreateBuffer(plt, rd.RayTraceProperties.itemsize)
        rd.WriteBuffer(plt, self.rdRTProp, 16, self.rtprop)
        self.rdImage = rd.CreateImage(plt, w, h)
        self.rdImageScratch = rd.CreateBuffer(plt, w * h * rd.CHANNEL * 4)
        self.rdCamData = rd.CreateBuffer(plt, rd.PhysicalCamera.itemsize)
        rd.WriteBuffer(plt, self.rdCamData, 48, np.array(scene.camera))
        self.rdSceneData = rd.CreateBuffer(plt, rd.SceneProperties.itemsize)
        rd.WriteBuffer(plt, self.rdSceneData, 176, np.array(scene.sceneProps))

        def up(arr):
            buf = rd.CreateBuffer(plt, max(arr.nbytes, 16))
            rd.WriteBuffer(plt, buf, arr.nbytes, arr)
            return buf
        self.meshInfoData = up(b["meshInfo"]); self.vertexData = up(b["vertex"]); self.indexData = up(b["index"])
        self.uvData = up(b["uv"]); self.normalData = up(b["normal"]); self.materialData = up(b["material"])
        self.blas = rd.BuildAccelStructs(plt, [rd.Mesh(m[0], m[1]) for m in scene.meshes])
        insts = [rd.Instance(tf, scene.sbt_offsets.get(k, 0), mat, self.blas[mi]) for k, (mi, tf, mat) in enumerate(scene.instances)]
        self.topAccelStruct = rd.BuildAccelStruct(plt, insts)
        self.descSet = rd.CreateDescriptorSet([
            self.rdRTProp, self.rdImageScratch, self.rdImage, self.rdCamData, self.rdSceneData,
            self.meshInfoData, self.vertexData, self.indexData, self.uvData, self.normalData, self.materialData,
            None, None, self.topAccelStruct])
        layout = rd.CreatePipelineLayout([rd.BUFFER_TYPE, rd.BUFFER_TYPE, rd.IMAGE_TYPE, rd.BUFFER_TYPE, rd.BUFFER_TYPE] +
                                         [rd.BUFFER_TYPE] * 6 + [rd.TEX_ARRAY_TYPE, rd.IMAGE_SAMPLER_TYPE, rd.ACCEL_STRUCT_TYPE])
        shader = rd.CreateShaderModule(plt, shader_text, len(shader_text), "functName..")
        self.pipeline = rd.CreatePipeline(rd.PipelineCreateInfo(1, layout, [shader], []))
        self.bind()

    def bind(self):
        rd.BindPipeline(self.plt, self.pipeline)
        rd.BindDescriptorSet(self.plt, self.descSet)

    def set_rtprop(self, totalSamples=None, batchSize=None, depth=None, debug=None):
        p = self.rtprop.copy()
        for k, v in (("totalSamples", totalSamples), ("batchSize", batchSize), ("depth", depth), ("debug", debug)):
            if v is not None:
                p[k] = v
        self.rtprop = p
        rd.WriteBuffer(self.plt, self.rdRTProp, 16, p)

    def render(self):
        """one frame as samples/sample1.cpp:447-498 does it: TraceRays, read the image, bump totalSamples"""
        rd.TraceRays(self.plt, 0, 0, 0, self.width, self.height)
        img = rd.ReadBuffer(self.plt, self.rdImage, self.width * self.height * 4).reshape(self.height, self.width, 4)
        p = np.zeros((), rd.RayTraceProperties)
        rd.ReadBuffer(self.plt, self.rdRTProp, 16, p.reshape(1).view(np.uint8))
        p = p.copy()
        p["totalSamples"] += p["batchSize"]
        rd.WriteBuffer(self.plt, self.rdRTProp, 16, np.array(p))
        self.rtprop = p
        return img

    def read_scratch(self):
        a = np.empty(self.width * self.height * 4, F)
        rd.ReadBuffer(self.plt, self.rdImageScratch, a.nbytes, a.view(np.uint8))
        return a.reshape(self.height, self.width, 4)

    def clear_scratch(self):
        z = np.zeros(self.width * self.height * 4, F)
        rd.WriteBuffer(self.plt, self.rdImageScratch, z.nbytes, z)


# ---------------------------------------------------------------------------------------------------
# the configs
# ---------------------------------------------------------------------------------------------------
def _rtprop(total, batch, depth, debug=0):
    p = np.zeros((), rd.RayTraceProperties)
    p["totalSamples"], p["batchSize"], p["depth"], p["debug"] = total, batch, depth, debug
    return p


def c0_two_boxes(width=256, height=256, spp=1, depth=1):
    """BASELINE config 0: the two-mesh / two-instance scene sketched in samples/sample0.cpp:25-46."""
    s = Scene("c0_two_boxes")
    b0 = s.add_mesh(box([-1, -1, -1], [1, 1, 1]))
    b1 = s.add_mesh(box([-0.5, -0.5, -0.5], [0.5, 0.5, 0.5]))
    s.materials = [material((0.8, 0.3, 0.3), 0.0, 0.6), material((0.3, 0.3, 0.8), 0.0, 0.4)]
    s.add_instance(b0, None, 0)
    s.add_instance(b1, translate(2.0, 0.25, -0.5), 1)
    s.camera = blender_camera(width, height, 0.05, 0.036, 6.0, 0.0, (0.5, 7.0, 1.5), (-100.0, 180.0, 0.0))
    s.sceneProps = blender_dir_light(-45.0, 20.0, 5.0)
    s.rtprop = _rtprop(0, spp, depth)
    return s


def c1_cornell(width=1920, height=1080, spp=4, depth=8, sphere_subdiv=5, fstop=0.0):
    """BASELINE config 1 (sample1.cpp stand-in): open-front Cornell box, 2 boxes, an icosphere
    (20*4^subdiv triangles; 5 -> 20480), 8 meshes / 8 instances, camera = sample1's buddha preset."""
    s = Scene("c1_cornell")
    X, Y, Z = 3.0, 6.0, 3.0
    floor = s.add_mesh(quad([-X, 0, -Z], [X, 0, -Z], [X, 0, Z], [-X, 0, Z], [0, 1, 0]))
    ceil_ = s.add_mesh(quad([-X, Y, -Z], [-X, Y, Z], [X, Y, Z], [X, Y, -Z], [0, -1, 0]))
    back = s.add_mesh(quad([-X, 0, Z], [X, 0, Z], [X, Y, Z], [-X, Y, Z], [0, 0, -1]))
    left = s.add_mesh(quad([-X, 0, -Z], [-X, 0, Z], [-X, Y, Z], [-X, Y, -Z], [1, 0, 0]))
    right = s.add_mesh(quad([X, 0, -Z], [X, Y, -Z], [X, Y, Z], [X, 0, Z], [-1, 0, 0]))
    tall = s.add_mesh(box([-0.8, 0.0, -0.8], [0.8, 3.2, 0.8]))
    short = s.add_mesh(box([-0.8, 0.0, -0.8], [0.8, 1.6, 0.8]))
    ball = s.add_mesh(icosphere(sphere_subdiv, 1.0))
    s.materials = [material((0.73, 0.73, 0.73), 0.0, 0.9),      # white
                   material((0.65, 0.05, 0.05), 0.0, 0.9),      # red
                   material((0.12, 0.45, 0.15), 0.0, 0.9),      # green
                   material((0.9, 0.8, 0.5), 0.9, 0.2),         # metal
                   material((0.95, 0.95, 0.95), 0.0, 0.08, 1.0, 1.45)]   # glass
    s.add_instance(floor, None, 0)
    s.add_instance(ceil_, None, 0)
    s.add_instance(back, None, 0)
    s.add_instance(left, None, 1)
    s.add_instance(right, None, 2)
    s.add_instance(tall, translate(-1.2, 0.0, 1.0) @ rotate_y(18.0), 3)
    s.add_instance(short, translate(1.3, 0.0, -0.6) @ rotate_y(-17.0), 0)
    s.add_instance(ball, translate(0.9, 2.45, -0.6) @ scale(0.85, 0.85, 0.85), 4)
    s.camera = blender_camera(width, height, 0.100, 0.036, 14.0, fstop, (0.0, 16.0, 6.5), (-105.0, 180.0, 0.0))
    s.sceneProps = blender_dir_light(-45.0, 0.0, 10.0)
    s.rtprop = _rtprop(0, spp, depth)
    return s


def c2_atrium(width=1920, height=1080, spp=4, depth=8, detail=1.0):
    """BASELINE config 2 ("Sponza-class"): colonnaded atrium, ~262k unique triangles at detail=1,
    25 meshes / 25 instances (TLAS depth stays well under the reference's 8-entry stack)."""
    s = Scene("c2_atrium")
    g = lambda n: max(2, int(round(n * detail)))
    LX, LY, LZ = 12.0, 9.0, 20.0
    m = []
    m.append((heightfield([-LX, 0, -LZ], [2 * LX, 0, 0], [0, 0, 2 * LZ], [0, 1, 0], g(128), g(128), 0.05, 11), 0))   # floor
    # roof over the two side aisles only: the nave is open to the sky (and to the directional light)
    m.append((heightfield([-LX, LY, -LZ], [0, 0, 2 * LZ], [LX - 4.5, 0, 0], [0, -1, 0], g(128), g(64), 0.30, 12, 0.4), 1))
    m.append((heightfield([4.5, LY, -LZ], [0, 0, 2 * LZ], [LX - 4.5, 0, 0], [0, -1, 0], g(128), g(64), 0.30, 17, 0.4), 1))
    m.append((heightfield([-LX, 0, -LZ], [0, 0, 2 * LZ], [0, LY, 0], [1, 0, 0], g(128), g(64), 0.08, 13), 2))        # left wall
    m.append((heightfield([LX, 0, -LZ], [0, LY, 0], [0, 0, 2 * LZ], [-1, 0, 0], g(64), g(128), 0.08, 14), 2))        # right wall
    m.append((heightfield([-LX, 0, LZ], [2 * LX, 0, 0], [0, LY, 0], [0, 0, -1], g(64), g(64), 0.08, 15), 3))         # back wall
    m.append((heightfield([-LX, 0, -LZ], [0, LY, 0], [2 * LX, 0, 0], [0, 0, 1], g(64), g(64), 0.08, 16), 3))         # front wall
    for k in range(16):
        side = -1.0 if k < 8 else 1.0
        z = -LZ + 2.5 + (k % 8) * (2 * LZ - 5.0) / 7.0
        m.append((cylinder([side * 7.0, 0.0, z], 0.55, LY - 0.4, g(64), g(64), 0.06, 100 + k), 4 + (k % 3)))
    m.append((heightfield([-5.0, 6.5, -6.0], [4.0, -1.5, 0], [0, 0, 12.0], [0, 1, 0], g(64), g(64), 0.02, 31, 0.35), 7))  # cloth
    m.append((heightfield([1.0, 5.0, -6.0], [4.0, 1.5, 0], [0, 0, 12.0], [0, 1, 0], g(64), g(64), 0.02, 32, 0.35), 8))    # cloth
    s.materials = [material((0.55, 0.5, 0.45), 0.0, 0.7), material((0.7, 0.68, 0.62), 0.0, 0.85),
                   material((0.62, 0.55, 0.48), 0.0, 0.8), material((0.5, 0.45, 0.4), 0.0, 0.8),
                   material((0.75, 0.72, 0.68), 0.0, 0.55), material((0.6, 0.58, 0.55), 0.0, 0.35),
                   material((0.8, 0.75, 0.6), 1.0, 0.25), material((0.7, 0.15, 0.12), 0.0, 0.9),
                   material((0.15, 0.25, 0.65), 0.0, 0.9)]
    for mesh, mat in m:
        s.add_instance(s.add_mesh(mesh), None, mat)
    # camera inside, near the front wall, looking down the nave (+z)
    s.camera = blender_camera(width, height, 0.028, 0.036, 20.0, 0.0, (0.6, 17.0, 3.2), (-92.0, 180.0, 2.0))
    s.sceneProps = blender_dir_light(-60.0, 25.0, 8.0)
    s.rtprop = _rtprop(0, spp, depth)
    return s


def c4_atrium_10m(width=1920, height=1080, spp=4, depth=8, detail=6.3, foliage=64):
    """BASELINE config 4 ("San-Miguel-scale"): the atrium generator at 6.3x tessellation = 10.4 M unique
    triangles in 25 instances, plus `foliage` instances of ONE shared 5120-triangle BLAS (rotated / non-uniformly scaled
    bushes along the nave and the aisles: the BLAS-deduplication path of the TLAS packer, radiance/src/bvh.cpp:575-588);
    the acceleration blob (446 MB) exceeds L2 and the 256 MB Infinity Cache."""
    s = c2_atrium(width, height, spp, depth, detail=detail)
    s.name = "c4_atrium_10m"
    if foliage:
        bush = s.add_mesh(icosphere(4, 1.0))
        s.materials = list(s.materials) + [material((0.18, 0.42, 0.12), 0.0, 0.75)]
        mat = len(s.materials) - 1
        k = np.arange(foliage)
        h = _hash01(k, k * 7 + 3, 977)
        h2 = _hash01(k * 5 + 1, k, 978)
        for i in range(foliage):
            lane = (-9.5, -3.0, 3.0, 9.5)[i % 4]
            z = -18.0 + 36.0 * (i // 4) / max(1, (foliage - 1) // 4)
            sx, sy, sz = 0.45 + 0.5 * h[i], 0.5 + 0.9 * h2[i], 0.45 + 0.5 * h[(i * 3) % foliage]
            tf = translate(lane + 0.8 * (h2[i] - 0.5), 0.55 * sy, z + 0.6 * (h[i] - 0.5)) @ rotate_y(360.0 * h[i]) @ scale(sx, sy, sz)
            s.add_instance(bush, tf, mat)
    return s


def c2_atrium_400(width=1920, height=1080, spp=4, depth=8, detail=1.0, pieces=16):
    """The Sponza-class atrium as a loader that makes one instance per mesh would deliver it (the reference's does:
    tools/sceneBuilder.cpp:287-315; SURVEY a5: "Sponza-class: 1 ... ~400 instances"): the same 262 k triangles, every mesh of
    c2_atrium cut into `pieces` runs of consecutive triangles -- 25 x 16 = 400 meshes / 400 instances, identity transforms."""
    base = c2_atrium(width, height, spp, depth, detail)
    s = Scene("c2_atrium_400")
    s.materials = base.materials
    for (mi, _, mat) in base.instances:
        v, t, n, uv = base.meshes[mi]
        nt = t.shape[0]
        for k in range(pieces):
            a, b = nt * k // pieces, nt * (k + 1) // pieces
            if b <= a:
                continue
            tt = t[a:b]
            used, inv = np.unique(tt.reshape(-1), return_inverse=True)
            s.add_instance(s.add_mesh(_finish(v[used], inv.reshape(-1, 3).astype(np.uint32), n[used], uv[used])), None, mat)
    s.camera, s.sceneProps, s.rtprop = base.camera, base.sceneProps, base.rtprop
    return s


CONFIGS = {"c0_two_boxes": c0_two_boxes, "c1_cornell": c1_cornell, "c2_atrium": c2_atrium, "c4_atrium_10m": c4_atrium_10m,
           "c2_atrium_400": c2_atrium_400}


# ---------------------------------------------------------------------------------------------------
# Wavefront OBJ + MTL (scene ingestion without assimp; parser = rdx_obj_load in librdx.so)
# ---------------------------------------------------------------------------------------------------
def load_obj(path, width=1920, height=1080, spp=4, depth=8, camera=None, light=None):
    """Scene from an OBJ (+ MTL) file: the buffers RD::Scene::Load (tools/sceneBuilder.cpp:27-258) would upload, one
    instance per mesh with the identity transform.  `camera` / `light` default to a view of the scene's bounding box
    from the front-top and the sample1 light (an OBJ file carries neither)."""
    import ctypes as C
    from . import _lib
    L = _lib.lib()
    o = _lib.rdx_obj_scene()
    if L.rdx_obj_load(os.fsencode(path), C.byref(o)):
        raise rd.RadianceError(_lib.last_error())
    try:
        def arr(ptr, n, dt):
            return np.ctypeslib.as_array(ptr, shape=(n,)).view(dt).copy() if n else np.zeros(0, dt)
        info = arr(C.cast(o.meshInfo, C.POINTER(C.c_int32)), o.nmeshes * 8, np.int32).view(rd.MeshInfo)
        vertex = arr(o.vertex, o.nvertices * 3, F).reshape(-1, 3)
        index = arr(o.index, o.ntriangles * 3, np.uint32).reshape(-1, 3)
        uv = arr(o.uv, o.nvertices * 3, F).reshape(-1, 3)
        normal = arr(o.normal, o.nvertices * 3, F).reshape(-1, 3)
        mats = arr(C.cast(o.materials, C.POINTER(C.c_int32)), o.nmaterials * 12, np.int32).view(rd.Material)
        vcount = arr(o.meshVertexCount, o.nmeshes, np.uint32)
        tcount = arr(o.meshTriangleCount, o.nmeshes, np.uint32)
    finally:
        L.rdx_obj_free(C.byref(o))
    s = Scene(os.path.basename(path))
    s.materials = [mats[i] for i in range(mats.shape[0])]
    for k in range(info.shape[0]):
        v0, t0 = int(info[k]["vertexOffset"]) // 3, int(info[k]["indexOffset"]) // 3
        nv, nt = int(vcount[k]), int(tcount[k])
        m = s.add_mesh((vertex[v0:v0 + nv].copy(), index[t0:t0 + nt].copy(), normal[v0:v0 + nv].copy(), uv[v0:v0 + nv].copy()))
        s.add_instance(m, None, int(info[k]["materialIndex"]))
    lo, hi = vertex.min(0), vertex.max(0)
    c, ext = (lo + hi) / 2, float(np.max(hi - lo))
    s.camera = camera if camera is not None else blender_camera(
        width, height, 0.050, 0.036, 2.0 * ext, 0.0, (float(c[0]), float(c[2]) + 2.0 * ext, float(c[1]) + 0.35 * ext), (-100.0, 180.0, 0.0))
    s.sceneProps = light if light is not None else blender_dir_light(-45.0, 0.0, 10.0)
    s.rtprop = _rtprop(0, spp, depth)
    return s


def save_obj(scene, path):
    """Writes a Scene whose instances are all untransformed as OBJ + MTL with full-precision floats (%.9g round-trips fp32), one `o` + `usemtl` per mesh, faces as v/vt/vn
    with equal indices -- the file `load_obj` turns back into the same buffers.  Test and export helper."""
    mtl = os.path.splitext(path)[0] + ".mtl"
    with open(mtl, "w") as f:
        for i, m in enumerate(scene.materials):
            f.write("newmtl m%d\nKd %.9g %.9g %.9g\nd %.9g\nPm %.9g\nPr %.9g\nTf %.9g %.9g %.9g\nNi %.9g\n\n" % (
                i, m["albedo"][0], m["albedo"][1], m["albedo"][2], m["albedo"][3], m["metallic"], m["roughness"],
                m["transmission"], m["transmission"], m["transmission"], m["ior"]))
    with open(path, "w") as f:
        f.write("mtllib %s\n" % os.path.basename(mtl))
        base = 0
        for k, (mi, tf, mat) in enumerate(scene.instances):
            if not np.array_equal(np.asarray(tf, F), np.eye(4, dtype=F)):
                raise ValueError("save_obj: instance %d is transformed; OBJ has no instancing" % k)
            v, t, n, uv = scene.meshes[mi]
            f.write("o mesh%d\nusemtl m%d\n" % (k, mat))
            for a in v:
                f.write("v %.9g %.9g %.9g\n" % tuple(a))
            for a in uv:
                f.write("vt %.9g %.9g\n" % (a[0], a[1]))
            for a in n:
                f.write("vn %.9g %.9g %.9g\n" % tuple(a))
            for a in t:
                i0, i1, i2 = (int(x) + base + 1 for x in a)
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (i0, i0, i0, i1, i1, i1, i2, i2, i2))
            base += v.shape[0]

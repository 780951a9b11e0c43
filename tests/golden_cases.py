"""Shared definitions of the golden cases: the generator (tests/golden/make_golden_gpu.py, runs the REAL reference
code object on an MI355X) and the tests (CPU oracle vs fixtures, HIP product vs fixtures) must build byte-identical
scenes and inputs, so both take them from here."""
import hashlib

import numpy as np

SCENES = ("c0", "c1", "c2")
N_PRIMARY = 1024          # primary rays per scene in the traversal batch (x4 with the derived rays)
N_MATERIAL = 2048


def spread(npix, n):
    """n pixel indices spread evenly over the image"""
    return np.unique(np.linspace(0, npix - 1, n).astype(np.int64))


def small_scene(scenes, name, fstop=0.0):
    if name == "c0":
        return scenes.c0_two_boxes(64, 64, spp=2, depth=3)
    if name == "c1":
        return scenes.c1_cornell(96, 54, spp=2, depth=4, sphere_subdiv=3, fstop=fstop)
    if name == "c2":
        return scenes.c2_atrium(96, 54, spp=2, depth=4, detail=0.2)
    raise KeyError(name)


def scene_blob(rd, scene):
    """TLAS blob of a scenes.Scene through the product's host-only builder (needs no GPU)."""
    blas = rd.BuildAccelStructs(None, [rd.Mesh(m[0], m[1]) for m in scene.meshes])
    insts = [rd.Instance(tf, scene.sbt_offsets.get(k, 0), mat, blas[mi]) for k, (mi, tf, mat) in enumerate(scene.instances)]
    return rd.BuildTopAccelStructBlob(insts)[0]


def sha(blob):
    return np.frombuffer(hashlib.sha256(bytes(blob)).digest(), np.uint8)


def derived_rays(seed, o, d, hit, t):
    """primary rays + scattered secondaries from their hit points + axis-aligned rays (zeros in the direction exercise
    the inf / NaN slab paths) + grazing rays; everything float32, deterministic from `seed`"""
    rng = np.random.default_rng(seed)
    n = o.shape[0]
    hp = (o + d * t[:, None]).astype(np.float32)
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True).astype(np.float32)
    o2 = np.where((hit == 1)[:, None], hp, o).astype(np.float32)
    d3 = np.zeros((n, 3), np.float32)
    d3[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    o3 = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    o3[: n // 8] = np.round(o3[: n // 8])          # origins on integer planes: exact hits of box planes
    d4 = d2.copy()
    d4[:, 1] *= np.float32(1e-4)
    return (np.ascontiguousarray(np.concatenate([o, o2, o3, o2]), np.float32),
            np.ascontiguousarray(np.concatenate([d, d2, d3, d4]), np.float32))


def kat_inputs():
    """unit inputs for intersectAABB / intersectTriangle / the BRDF pair"""
    rng = np.random.default_rng(20261004)
    n = 4096
    o = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    lo = rng.uniform(-4, 0, (n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0, 4, (n, 3))).astype(np.float32)
    aim = (lo + (hi - lo) * rng.uniform(-0.3, 1.3, (n, 3))).astype(np.float32)       # half the rays are aimed at the box
    d[n // 2:] = (aim - o)[n // 2:]
    d[: n // 8, 0] = 0; d[n // 8: n // 4, 1] = 0; d[n // 4: n // 4 + 64] = 0
    o[: n // 16] = lo[: n // 16]                    # origin on a box plane AND a zero direction component: 0/0
    o[n // 16: n // 8, 0] = hi[n // 16: n // 8, 0]
    hi[n // 2: n // 2 + 64, 2] = lo[n // 2: n // 2 + 64, 2]    # flat boxes
    v0 = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    v1 = (v0 + rng.normal(size=(n, 3))).astype(np.float32)
    v2 = (v0 + rng.normal(size=(n, 3))).astype(np.float32)
    w = rng.dirichlet([1, 1, 1], n).astype(np.float32)
    tgt = v0 * w[:, :1] + v1 * w[:, 1:2] + v2 * w[:, 2:]
    to = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    td = ((tgt - to) + rng.normal(size=(n, 3)) * 0.2).astype(np.float32)
    td[: 64] = (v1 - v0)[: 64]                      # rays parallel to the triangle plane: det == 0 or garbage
    e = slice(64, 128)                              # rays aimed at the edge v0-v1
    td[e] = ((v0[e] * w[e, :1] + v1[e] * (1 - w[e, :1])) - to[e]).astype(np.float32)
    # BRDF: unit-ish vectors around a normal, all material classes
    m = 2048
    N = rng.normal(size=(m, 3)); N /= np.linalg.norm(N, axis=1, keepdims=True)
    def hemi():
        v = rng.normal(size=(m, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        s = np.sign((v * N).sum(1, keepdims=True)); s[s == 0] = 1
        return v * s
    L, V = hemi(), hemi()
    V[: m // 8] *= -1                                # back-facing views (glass from inside)
    N[-32:] = [1.0, 0.0, 0.0]                        # the GetNormalSpace special case
    albedo = rng.uniform(0.2, 0.9, (m, 3))
    metallic = (rng.uniform(size=m) < 0.3).astype(np.float64) * rng.uniform(0.5, 1, m)
    rough = rng.uniform(0.05, 1, m)
    transm = (rng.uniform(size=m) < 0.3).astype(np.float64)
    ior = rng.uniform(1.1, 1.8, m)
    rnd = rng.uniform(0, 1, (m, 3))
    brdf = np.concatenate([L, V, N, albedo, metallic[:, None], rough[:, None], transm[:, None], ior[:, None], rnd], 1).astype(np.float32)
    return dict(aabb_o=o, aabb_d=d, aabb_lo=lo, aabb_hi=hi, tri_o=to, tri_d=td, tri_v0=v0, tri_v1=v1, tri_v2=v2,
                brdf_in=np.ascontiguousarray(brdf))


def material_inputs(n):
    frames = (np.arange(n) % 7).astype(np.uint32)
    depths = (np.arange(n) % 5).astype(np.int32)
    return frames, depths


def generate_inputs(n, seed):
    rng = np.random.default_rng(seed)
    r = rng.integers(0, 2**32, size=(n, 3), dtype=np.uint64).astype(np.uint32)
    r[:4] = 0
    return r

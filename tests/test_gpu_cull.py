"""GPU suite, part 3 (-m gpu): the CULLED walk (option "cull", automatic for scenes of >= 16 k inner nodes, so the default on
the headline scene) held to the reference's exhaustive walk where its margins are thinnest.

The culled walk skips (a) subtrees a closest-hit ray enters beyond its best t and (b) leaves whose box the ray misses, each
with a 2^-8 relative margin (csrc/kernels.hip "culled walk"; the bound is in DESIGN.md 4.1c: exact for every ray / triangle
pair whose Moeller-Trumbore determinant is not dominated by rounding, i.e. |det| >= 2^-12 |d||e1||e2|).  The cases below aim
at what is left: badly conditioned pairs.  Comparand: the reference's own `intersectTop` (radiance/shader/radiance.cl:110-192)
compiled for gfx950 (oracle/_ref, build p) when the code object is present, and in any case the product's reference-order
kernel (rdx_trace_batch mode 1), which the suite holds bit-identical to that code (tests/test_gpu_reference.py).

  * sliver triangles (aspect ratio 1e6 .. 1e7) in random orientations, hit along and across the long edge
  * stacks of coplanar tessellated planes, TILTED (no exact zeros in normals or edges), 1e-5 .. 1e-3 apart
  * rays INSIDE those planes: direction = difference of two points of the plane, exactly as computed in fp32, and perturbed by
    1e-7 .. 1e-3; rays along triangle edges; rays that start on vertices
  * far-away origins (|o| ~ 1e4) aimed at all of the above
  * the families of tools/cull_stress.py on the Sponza-class scene (grazing, surface to surface, axis-parallel)
  * 500 seeds of tools/fuzz_parity.py's random scenes (culled pool kernel against the reference-order kernel)

Every batch is traced as closest-hit and as any-hit rays with cull = 1 and cull = 0; HitData must match bit for bit."""
import os
import sys

import numpy as np
import pytest

import oracle_bind as ob  # noqa: F401  (HIT_DTYPE lives there)
import refgpu_bind as rg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32
FIELDS = ("distance", "primitiveIndex", "instanceIndex", "instanceCustomIndex", "barycentric", "hitPoint", "transform")


@pytest.fixture(scope="module")
def mods(gpu):
    import rrt_amd  # noqa: F401
    from radiance_ray_tracing_amd import rd, scenes
    return rd, scenes


@pytest.fixture(scope="module")
def ref(gpu):
    return rg.RefGpu("p") if rg.available("p") else None


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(a.shape[0], -1) if a.shape[0] else np.zeros((0, 1), np.uint8)


def _mismatches(want, got, closest):
    bad = want["hit"] != got["hit"]
    if closest:
        h = (want["hit"] == 1) & ~bad
        for f in FIELDS:
            d = np.zeros(want.shape[0], bool)
            d[h] = (_bits(want[f][h]) != _bits(got[f][h])).any(1)
            bad |= d
    return bad


def check_batches(rd, dev, ref, batches, tag):
    """every batch x {closest, any hit} x {cull 1, cull 0} against the live reference (if built) and the reference-order kernel"""
    tl = None
    if ref is not None:
        blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
        tl = rg.DevBuf.of(np.frombuffer(blob, np.uint8))
    total = 0
    for name, (o, d) in batches.items():
        o = np.ascontiguousarray(o, F); d = np.ascontiguousarray(d, F)
        ok = np.isfinite(o).all(1) & np.isfinite(d).all(1) & (np.abs(d).sum(1) > 0)
        o, d = o[ok], d[ok]
        for rec in (1, 2):
            want = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec, reference_order=True)
            if tl is not None:
                live = ref.trace(tl, o, d, 0.001, 1000.0, rec)
                bad = _mismatches(live, want, rec == 1)
                assert not bad.any(), "%s / %s rec %d: reference-order kernel differs from the live reference on %d rays" % (tag, name, rec, int(bad.sum()))
            for cull in (1, 0):
                rd.SetOption("kernel", 3); rd.SetOption("cull", cull)
                try:
                    got = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
                finally:
                    rd.SetOption("cull", -1)
                bad = _mismatches(want, got, rec == 1)
                if bad.any():
                    i = int(np.flatnonzero(bad)[0])
                    raise AssertionError("%s / %s rec %d cull %d: %d of %d rays differ; first: o=%r d=%r want t=%r prim=%d inst=%d hit=%d, got t=%r prim=%d inst=%d hit=%d"
                                         % (tag, name, rec, cull, int(bad.sum()), o.shape[0], o[i].tolist(), d[i].tolist(),
                                            float(want["distance"][i]), int(want["primitiveIndex"][i]), int(want["instanceIndex"][i]), int(want["hit"][i]),
                                            float(got["distance"][i]), int(got["primitiveIndex"][i]), int(got["instanceIndex"][i]), int(got["hit"][i])))
            total += o.shape[0]
    return total


# ---------------------------------------------------------------------------------------------------------------------
# adversarial geometry
# ---------------------------------------------------------------------------------------------------------------------
def _rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a - b * b - c * c + d * d]])


def sliver_mesh(rng, n, aspect):
    """n needle triangles: two vertices `length` apart, the third off their midpoint by length / aspect"""
    v = np.zeros((n, 3, 3)); t = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    for k in range(n):
        R = _rot(rng)
        c = rng.uniform(-4, 4, 3)
        L = rng.uniform(1.0, 6.0)
        w = L / aspect * rng.uniform(1.0, 10.0)
        p = np.array([[-L / 2, 0, 0], [L / 2, 0, 0], [rng.uniform(-0.4, 0.4) * L, w, 0]])
        v[k] = p @ R.T + c
    v = v.reshape(-1, 3).astype(F)
    return v, t, np.tile(np.array([[0, 0, 1]], F), (3 * n, 1)), np.zeros_like(v)


def tilted_plane_stack(rng, res, layers, gap):
    """`layers` copies of a res x res grid of quads in ONE tilted plane orientation, `gap` apart along the normal; vertices are
    computed in float64 and rounded, so the triangles are coplanar only up to rounding -- and nothing is axis-aligned"""
    R = _rot(rng)
    u, w, nrm = R[:, 0], R[:, 1], R[:, 2]
    c = rng.uniform(-1, 1, 3)
    vs, ts = [], []
    for l in range(layers):
        g = np.linspace(-3, 3, res + 1)
        P = c[None, None, :] + g[:, None, None] * u[None, None, :] + g[None, :, None] * w[None, None, :] + (l * gap) * nrm[None, None, :]
        base = len(vs) * (res + 1) ** 2
        vs.append(P.reshape(-1, 3))
        idx = lambda i, j: base + i * (res + 1) + j
        for i in range(res):
            for j in range(res):
                ts.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
                ts.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    v = np.concatenate(vs).astype(F)
    return (v, np.array(ts, np.uint32), np.tile(nrm.astype(F), (v.shape[0], 1)), np.zeros_like(v)), (u, w, nrm, c)


def _unit(x):
    return (x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)).astype(F)


def _scene(scenes, name, meshes_and_tfs):
    s = scenes.Scene(name)
    s.materials = [scenes.material((0.7, 0.7, 0.7))]
    for mesh, tfs in meshes_and_tfs:
        mi = s.add_mesh(mesh)
        for tf in tfs:
            s.add_instance(mi, tf, 0)
    s.camera = scenes.blender_camera(64, 48, 0.05, 0.036, 9.0, 0.0, (0.5, 14.0, 1.0), (-96.0, 180.0, 0.0))
    s.sceneProps = scenes.blender_dir_light(-45.0, 20.0, 5.0)
    s.rtprop = scenes._rtprop(0, 1, 2)
    return s


def _tf(rng, identity=False):
    M = np.eye(4)
    if not identity:
        M[:3, :3] = _rot(rng) @ np.diag(rng.uniform(0.5, 2.0, 3))
        M[:3, 3] = rng.uniform(-3, 3, 3)
    return M.astype(F)


def _world_points(mesh, tf, rng, n):
    """n points on the triangles of `mesh` under `tf` (fp32 arithmetic like a shader's hit position: v0 + b1 e1 + b2 e2)"""
    v, t = mesh[0], mesh[1]
    k = rng.integers(0, t.shape[0], n)
    b = rng.uniform(0, 1, (n, 2)); fl = b.sum(1) > 1; b[fl] = 1 - b[fl]
    p = v[t[k, 0]] + (v[t[k, 1]] - v[t[k, 0]]) * b[:, :1].astype(F) + (v[t[k, 2]] - v[t[k, 0]]) * b[:, 1:].astype(F)
    return (p @ tf[:3, :3].T + tf[:3, 3]).astype(F), k


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_slivers_identical_to_reference(mods, ref, seed):
    rd, scenes = mods
    rng = np.random.default_rng(9000 + seed)
    mesh = sliver_mesh(rng, 3000, 10.0 ** rng.uniform(6.0, 7.0))
    tfs = [_tf(rng, identity=True)] + [_tf(rng) for _ in range(3)]
    # a few well-shaped boxes around, so that closest-hit rays have a best t to cull with
    s = _scene(scenes, "slivers%d" % seed, [(mesh, tfs), (scenes.box([-5, -5, -5], [-4.5, 5, 5]), [_tf(rng, identity=True)]),
                                           (scenes.icosphere(3, 1.5), [_tf(rng) for _ in range(2)])])
    dev = scenes.DeviceScene(s)
    n = 60000
    tf = tfs[int(rng.integers(0, len(tfs)))]
    tgt, k = _world_points(mesh, tf, rng, n)
    o = rng.uniform(-9, 9, (n, 3)).astype(F)
    aimed = _unit(tgt - o)
    # along the needle: start on one sliver, head for another point of the SAME sliver (in its plane), exact and perturbed
    a, _ = _world_points(mesh, tf, rng, n)
    v, t = mesh[0], mesh[1]
    e = ((v[t[k, 1]] - v[t[k, 0]]) @ tf[:3, :3].T).astype(F)
    along = _unit(e + (rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-7, -3, (n, 1))).astype(F))
    far = (tgt + _unit(rng.normal(size=(n, 3)).astype(F)) * F(1.0e4)).astype(F)
    batches = {"aimed at slivers": (o, aimed), "along the long edge": (tgt, along), "far origin": (far, _unit(tgt - far)),
               "sliver to sliver": (a, _unit(tgt - a))}
    assert check_batches(rd, dev, ref, batches, "slivers seed %d" % seed) > 4 * n


@pytest.mark.parametrize("seed,gap", [(1, 1e-5), (2, 1e-4), (3, 1e-3), (4, 0.0)])
def test_tilted_coplanar_stacks_identical_to_reference(mods, ref, seed, gap):
    rd, scenes = mods
    rng = np.random.default_rng(9100 + seed)
    mesh, (u, w, nrm, c) = tilted_plane_stack(rng, 24, 4, gap)
    tfs = [_tf(rng, identity=True), _tf(rng)]
    s = _scene(scenes, "stack%d" % seed, [(mesh, tfs), (scenes.icosphere(2, 0.7), [_tf(rng) for _ in range(3)])])
    dev = scenes.DeviceScene(s)
    n = 60000
    tf = tfs[0]
    a, _ = _world_points(mesh, tf, rng, n)
    b, _ = _world_points(mesh, tf, rng, n)
    inplane = _unit(b - a)                                             # difference of two points of the plane, as fp32 computes it
    eps = (rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-7, -3, (n, 1))).astype(F)
    nearplane = _unit(inplane + eps)
    v, t = mesh[0], mesh[1]
    k = rng.integers(0, t.shape[0], n)
    edge = _unit(v[t[k, 1]] - v[t[k, 0]])                              # along a triangle edge, from its first vertex
    vert = v[t[k, 0]]
    off = (a + (nrm * 1e-5).astype(F)).astype(F)                       # the closest-hit shader's offset origin (shader.cl:465)
    o = rng.uniform(-9, 9, (n, 3)).astype(F)
    far = (b + _unit(rng.normal(size=(n, 3)).astype(F)) * F(1.0e4)).astype(F)
    grazing_far = (a - inplane * F(1.0e4)).astype(F)
    batches = {"in plane": (a, inplane), "near plane": (a, nearplane), "offset origin, in plane": (off, inplane),
               "along edges from vertices": (vert, edge), "aimed": (o, _unit(b - o)), "far origin": (far, _unit(b - far)),
               "far origin, in plane": (grazing_far, inplane)}
    assert check_batches(rd, dev, ref, batches, "stack seed %d gap %g" % (seed, gap)) > 7 * n


def test_cull_stress_families_on_the_sponza_class_scene(mods, ref):
    """tools/cull_stress.py's ray families, 2^17 rays each, on BASELINE config 2's scene"""
    rd, scenes = mods
    s = scenes.CONFIGS["c2_atrium"](1920, 1080, 4, 8)
    dev = scenes.DeviceScene(s)
    rng = np.random.default_rng(77)
    n = 1 << 17
    px = rng.choice(1920 * 1080, n, replace=False).astype(np.uint32)
    po, pd = rd.GenerateBatch(px, rng.integers(0, 2 ** 32, size=(n, 3), dtype=np.uint64).astype(np.uint32))
    ph = rd.TraceBatch(dev.topAccelStruct, po, pd, reference_order=True)
    hit = ph["hit"] == 1
    hp = (po + pd * ph["distance"][:, None]).astype(F)
    d2 = _unit(rng.normal(size=(n, 3)).astype(F))
    o2 = np.where(hit[:, None], hp, po).astype(F)
    h2 = rd.TraceBatch(dev.topAccelStruct, o2, d2, reference_order=True)
    hp2 = (o2 + d2 * h2["distance"][:, None]).astype(F)
    tng = _unit(np.cross(pd, d2).astype(F))
    graze = (tng + (rng.normal(size=(n, 3)) * 1e-3).astype(F)).astype(F)
    ax = np.zeros((n, 3), F); ax[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    batches = {"primary": (po, pd), "scattered": (o2, d2), "grazing": (o2, graze), "surface to surface": (hp, (hp2 - hp).astype(F)), "axis": (o2, ax)}
    assert check_batches(rd, dev, ref, batches, "c2_atrium") > 4 * n


def test_fuzz_500_seeds_culled_pool_kernel(mods):
    """500 random scenes of tools/fuzz_parity.py: the culled pool kernel against the reference-order kernel, closest and any hit"""
    rd, scenes = mods
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as fz
    bad = 0
    for seed in range(70000, 70500):
        s, o, d, rng = fz.random_case(seed)
        dev = scenes.DeviceScene(s)
        first = rd.TraceBatch(dev.topAccelStruct, o, d, reference_order=True)
        o, d = fz.with_surface_rays(rng, o, d, first)
        for rec in (1, 2):
            want = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec, reference_order=True)
            rd.SetOption("kernel", 3); rd.SetOption("cull", 1)
            try:
                got = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
            finally:
                rd.SetOption("cull", -1)
            bad += int(_mismatches(want, got, rec == 1).sum())
    assert bad == 0

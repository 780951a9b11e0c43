#!/usr/bin/env python3
"""fuzz_parity.py [seeds] -- random scenes against the reference-order kernel, bit for bit: random meshes (quads, boxes,
icospheres, height fields, stacks of coincident triangles that the SAH builder cannot split -> leaves of more than 8
triangles), 1-70 instances with random affine transforms (rotation, non-uniform scale, shear, shared BLASes -> top-level
leaves, instance masks), rays from everywhere including axis-aligned ones and rays starting on surfaces.  Every
production kernel (3 pool -- culled walk, quad records, 64-byte records --, 2 per-lane stacks, 1 per-lane wide) must return the reference-order kernel's HitData exactly
(closest hit) and its hit flag (any hit)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rrt_amd
from radiance_ray_tracing_amd import rd, scenes

F = np.float32

def rand_tf(rng):
    a, b, c = rng.uniform(0, 2 * np.pi, 3)
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(c), -np.sin(c)], [0, np.sin(c), np.cos(c)]])
    S = np.diag(rng.uniform(0.3, 2.5, 3))
    if rng.random() < 0.3:
        S[0, 1] = rng.uniform(-0.5, 0.5)                  # shear
    M = np.eye(4)
    M[:3, :3] = Rz @ Ry @ Rx @ S
    M[:3, 3] = rng.uniform(-6, 6, 3)
    if rng.random() < 0.15:
        M = np.eye(4); M[:3, 3] = rng.uniform(-6, 6, 3)   # pure translation
    return M.astype(F)

def soup(rng, k):
    """k coincident / nearly coincident triangles: centroids equal -> the builder ends in one big leaf"""
    p = rng.uniform(-1, 1, (3, 3)).astype(F)
    v = np.tile(p, (k, 1)).astype(F)
    if rng.random() < 0.5:
        v += rng.uniform(-1e-6, 1e-6, v.shape).astype(F)
    t = np.arange(3 * k, dtype=np.uint32).reshape(k, 3)
    n = np.tile(np.array([[0, 0, 1]], F), (3 * k, 1))
    return v, t, n, np.zeros_like(v)

def rand_mesh(rng):
    kind = rng.integers(0, 6)
    if kind == 0:
        return scenes.quad([-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1], [0, 1, 0])
    if kind == 1:
        lo = rng.uniform(-1.5, -0.2, 3); hi = rng.uniform(0.2, 1.5, 3)
        return scenes.box(lo.tolist(), hi.tolist())
    if kind == 2:
        return scenes.icosphere(int(rng.integers(0, 4)), float(rng.uniform(0.3, 1.5)))
    if kind == 3:
        return scenes.heightfield([-1.5, 0, -1.5], [3.0 / 12, 0, 0], [0, 0, 3.0 / 12], [0, 1, 0], 12, 12, float(rng.uniform(0.05, 0.6)), int(rng.integers(0, 1000)))
    if kind == 4:
        return soup(rng, int(rng.integers(2, 40)))
    return scenes.cylinder([0, -1, 0], 0.5, 2.0, 12, 6, 0.05, int(rng.integers(0, 1000)))

def random_case(seed):
    """scene + first-stage rays of one seed (no GPU needed)"""
    rng = np.random.default_rng(1000 + seed)
    s = scenes.Scene("fuzz%d" % seed)
    meshes = [s.add_mesh(rand_mesh(rng)) for _ in range(int(rng.integers(1, 7)))]
    s.materials = [scenes.material((0.7, 0.7, 0.7))]
    for _ in range(int(rng.integers(1, 71))):
        s.add_instance(meshes[int(rng.integers(0, len(meshes)))], rand_tf(rng), 0)
    s.camera = scenes.blender_camera(64, 48, 0.05, 0.036, 9.0, 0.0, (0.5, 14.0, 1.0), (-96.0, 180.0, 0.0))
    s.sceneProps = scenes.blender_dir_light(-45.0, 20.0, 5.0)
    s.rtprop = scenes._rtprop(0, 1, 2)
    n = 6000
    o = rng.uniform(-9, 9, (n, 3)).astype(F)
    d = rng.normal(size=(n, 3)).astype(F); d /= np.linalg.norm(d, axis=1, keepdims=True)
    tgt = rng.uniform(-6, 6, (n, 3)).astype(F)
    d[: n // 2] = (tgt[: n // 2] - o[: n // 2]); d[: n // 2] /= np.linalg.norm(d[: n // 2], axis=1, keepdims=True)
    ax = np.zeros((n // 6, 3), F); ax[np.arange(n // 6), rng.integers(0, 3, n // 6)] = rng.choice([-1.0, 1.0], n // 6)
    d[-(n // 6):] = ax
    return s, o, d, rng

def with_surface_rays(rng, o, d, hits):
    """append rays that start on the surfaces the first-stage rays hit"""
    hp = (o + d * hits["distance"][:, None])[hits["hit"] == 1]
    if hp.shape[0]:
        d2 = rng.normal(size=hp.shape).astype(F); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
        o = np.concatenate([o, hp.astype(F)]); d = np.concatenate([d, d2])
    return np.ascontiguousarray(o, F), np.ascontiguousarray(d, F)

def run(nseeds, first_seed=0, verbose=True):
    bad = 0
    fields = ("distance", "primitiveIndex", "instanceIndex", "instanceCustomIndex", "barycentric", "hitPoint", "transform")
    for seed in range(first_seed, first_seed + nseeds):
        s, o, d, rng = random_case(seed)
        dev = scenes.DeviceScene(s)
        ref = rd.TraceBatch(dev.topAccelStruct, o, d, reference_order=True)
        o, d = with_surface_rays(rng, o, d, ref)
        for rec in (1, 2):
            ref = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec, reference_order=True)
            h = ref["hit"] == 1
            for kernel, cull, quad in ((3, 1, 1), (3, 0, 1), (3, 0, 0), (2, 0, 1), (1, 0, 1)):      # (3, 0, 1): the quad-record walk
                rd.SetOption("kernel", kernel); rd.SetOption("cull", cull); rd.SetOption("quad", quad)
                got = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
                rd.SetOption("kernel", 3); rd.SetOption("cull", -1); rd.SetOption("quad", 1)
                ok = np.array_equal(ref["hit"], got["hit"])
                if ok and rec == 1:
                    ok = all(np.array_equal(ref[f][h].view(np.uint8), got[f][h].view(np.uint8)) for f in fields)
                if not ok:
                    bad += 1
                    print("MISMATCH seed %d rec %d kernel %d cull %d quad %d (%d instances, %d rays, %d hits)" % (seed, rec, kernel, cull, quad, len(s.instances), o.shape[0], int(h.sum())), flush=True)
        if verbose and seed % 10 == 9:
            print("seed %d done, %d instances, %d rays, %d hits, mismatches so far %d" % (seed, len(s.instances), o.shape[0], int(h.sum()), bad), flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    b = run(n, first)
    print("fuzz: %d seeds from %d, %d mismatches" % (n, first, b))
    sys.exit(1 if b else 0)

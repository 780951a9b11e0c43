"""GPU parity suite, part 1 (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Integer / index work (RNG, hit primitive and instance ids, visit counters) and the traversal of GIVEN rays (t,
barycentrics, hit points: + - * / and fma only) must be BIT-EXACT against the oracle.  Anything behind a normalize
(v_rsq_f32 on the GPU) or sin / cos / acos / pow (OCML vs glibc) -- primary rays, shading, frames -- carries a tolerance
HERE, because the CPU cannot reproduce those bit for bit; the bit-exact comparison of those against the reference's own
device code running on the GPU is part 2, tests/test_gpu_reference.py."""
import os

import numpy as np
import pytest

import oracle_bind as ob

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def mods(gpu):
    import rrt_amd
    from radiance_ray_tracing_amd import rd, scenes
    return rd, scenes


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint8)


def _near_oracle(got, ref, frac=0.99, rmse=2e-3):
    """frames vs the CPU oracle: last-bit differences of normalize / transcendentals flip a handful of glass / mirror
    paths at silhouettes (the oracle itself is that far from the reference build, tests/test_cpu_oracle.py), so the bulk
    of the pixels must agree to 1e-5 and the rest may not drag the RMSE beyond 2e-3; the 1e-4 RMSE bar of the north star
    is held against the reference itself in tests/test_gpu_reference.py (where the difference is exactly 0)"""
    got = np.asarray(got, np.float64).reshape(-1, 4)
    ref = np.asarray(ref, np.float64).reshape(-1, 4)
    d = np.abs(got - ref).max(1)
    ok = (d < 1e-5).mean() >= frac and np.sqrt(np.mean((got - ref) ** 2)) < rmse
    if not ok:
        print("near_oracle: %.4f of the pixels within 1e-5 (need %.2f), RMSE %.3g" % ((d < 1e-5).mean(), frac, np.sqrt(np.mean((got - ref) ** 2))))
    return ok


def _small_scenes(scenes):
    return {
        "c0": scenes.c0_two_boxes(64, 64, spp=2, depth=3),
        "c1": scenes.c1_cornell(96, 54, spp=2, depth=4, sphere_subdiv=3),
        "c2": scenes.c2_atrium(96, 54, spp=2, depth=4, detail=0.2),
    }


def _ray_batch(osc, n, seed):
    """primary rays + scattered secondaries + axis-aligned / grazing / on-surface rays"""
    rng = np.random.default_rng(seed)
    s = osc.scene
    px = rng.integers(0, s.width * s.height, n).astype(np.uint32)
    rin = np.stack([rng.integers(0, 8, n).astype(np.uint32), np.zeros(n, np.uint32), px], 1)
    o, d = osc.generate_rays(px, rin)
    hits = ob.trace_batch(osc.tlas.tobytes(), o, d)
    # secondaries: start at the world-space hit point of the primaries, random directions
    hp = o + d * hits["distance"][:, None]
    ok = hits["hit"] == 1
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = np.where(ok[:, None], hp, o).astype(np.float32)
    # axis-aligned and grazing directions (zeros in the direction exercise the inf / NaN slab paths)
    d3 = np.zeros((n, 3), np.float32)
    d3[np.arange(n), rng.integers(0, 3, n)] = rng.choice([-1.0, 1.0], n)
    o3 = rng.uniform(-4, 4, size=(n, 3)).astype(np.float32)
    d4 = d2.copy(); d4[:, 1] *= 1e-4
    return (np.concatenate([o, o2, o3, o2]).astype(np.float32), np.concatenate([d, d2, d3, d4]).astype(np.float32))


def test_pcg3d_bit_exact(mods):
    rd, _ = mods
    g = np.load(os.path.join(GOLD, "ref_kat.npz"))
    out = rd.Pcg3dBatch(g["pcg_in"])
    assert np.array_equal(_bits(out), _bits(g["pcg_out"]))          # vs the real reference function
    x = np.random.default_rng(1).integers(0, 2**32, size=(100000, 3), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(_bits(rd.Pcg3dBatch(x)), _bits(ob.pcg3d(x)))


@pytest.mark.parametrize("name", ["c0", "c1", "c2"])
def test_traversal_bit_exact(mods, name):
    """closest-hit and any-hit HitData (t, bary, hit point, primitive / instance ids, transform) are
    bit-identical to the oracle, and so are the visit counters of the exhaustive walk"""
    rd, scenes = mods
    s = _small_scenes(scenes)[name]
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    blob_o, _, _ = ob.scene_tlas(s)
    assert blob == blob_o                                            # device blob == oracle blob
    osc = ob.OracleScene(s, blob_o)
    o, d = _ray_batch(osc, 4096, 5)
    fields = ("distance", "primitiveIndex", "instanceIndex", "instanceCustomIndex", "instanceSBTOffset",
              "barycentric", "hitPoint", "transform")
    for rec in (1, 2):
        ref, ctr = ob.trace_batch(blob_o, o, d, 0.001, 1000.0, rec, counters=True)
        # reference-order kernel: everything bit-exact, including the visit counters
        got, visit = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec, count_visits=True)
        assert np.array_equal(ref["hit"], got["hit"])
        h = ref["hit"] == 1
        for f in fields:
            assert np.array_equal(_bits(ref[f][h]), _bits(got[f][h])), (rec, f)
        c = ctr.as_dict()
        assert list(visit) == [c["top_nodes"][rec - 1], c["inst_visits"][rec - 1], c["bot_nodes"][rec - 1], c["tri_tests"][rec - 1]]
        # production kernels: 2 = wave-cooperative (default), 1 = per-lane wide nodes; both use the fast
        # slab test and a free visiting order, and must still land on the reference's exact HitData
        for kernel in (3, 2, 1):
            rd.SetOption("kernel", kernel)
            try:
                fast = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
            finally:
                rd.SetOption("kernel", 3)
            assert np.array_equal(ref["hit"], fast["hit"]), kernel
            if rec == 1:        # a shadow ray only reports whether a candidate exists
                for f in fields:
                    assert np.array_equal(_bits(ref[f][h]), _bits(fast[f][h])), (kernel, f)
    assert h.sum() > 500      # the batch really exercises hits


def test_primary_rays(mods):
    """against the CPU oracle primary rays agree to 1e-6 (normalize = v * v_rsq_f32 on the GPU, sin / cos of the camera
    angles and of the disk sample from OCML); bit-exactness against the reference build is tested in part 2"""
    rd, scenes = mods
    for fstop, exact in ((0.0, False), (2.8, False)):
        s = scenes.c1_cornell(96, 54, sphere_subdiv=1, fstop=fstop)
        dev = scenes.DeviceScene(s)
        osc = ob.OracleScene(s)
        rng = np.random.default_rng(2)
        px = rng.integers(0, 96 * 54, 5000).astype(np.uint32)
        rin = rng.integers(0, 2**32, size=(5000, 3), dtype=np.uint64).astype(np.uint32)
        o, d = osc.generate_rays(px, rin)
        go, gd = rd.GenerateBatch(px, rin)
        if exact:
            assert np.array_equal(_bits(o), _bits(go)) and np.array_equal(_bits(d), _bits(gd))
        else:
            assert np.abs(o - go).max() < 1e-6 and np.abs(d - gd).max() < 1e-6


def test_material_shading(mods):
    """closest-hit `material` on captured hits (diffuse, metal, glass): float tolerance 2e-5 relative
    to the value scale (transcendentals differ between glibc and OCML by ulps)"""
    rd, scenes = mods
    s = scenes.c1_cornell(96, 54, sphere_subdiv=3)
    dev = scenes.DeviceScene(s)
    osc = ob.OracleScene(s)
    rng = np.random.default_rng(9)
    n = 6000
    px = rng.integers(0, 96 * 54, n).astype(np.uint32)
    rin = np.stack([np.zeros(n, np.uint32), np.zeros(n, np.uint32), px], 1)
    o, d = osc.generate_rays(px, rin)
    hits = ob.trace_batch(osc.tlas.tobytes(), o, d)
    k = hits["hit"] == 1
    hits, d, px = hits[k], d[k], px[k]
    frames = rng.integers(0, 64, hits.shape[0]).astype(np.uint32)
    depths = rng.integers(0, 8, hits.shape[0]).astype(np.int32)
    ref = osc.material_batch(hits, d, px, frames, depths)
    got = rd.MaterialBatch(hits, d, px, frames, depths)
    assert np.array_equal(ref["hit"], got["hit"])
    for f in ("nextFactor", "nextRayOrigin", "nextRayDirection"):
        scale = np.maximum(1.0, np.abs(ref[f]))
        assert (np.abs(ref[f] - got[f]) / scale).max() < 2e-5, f
    # the oracle's colour includes its shadow test; the GPU seam reports the light-visible outcome.
    # So the oracle colour is either that (tolerance) or exactly the ambient term albedo * 0.1.
    albedo = np.array(s.materials)["albedo"][hits["instanceCustomIndex"], :3]
    ambient = (albedo * np.float32(0.1)).astype(np.float32)
    rel = np.abs(ref["color"] - got["color"]) / np.maximum(1.0, np.abs(ref["color"]))
    lit = np.all(rel < 2e-5, axis=1)
    occluded = np.all(ref["color"] == ambient, axis=1)
    assert np.all(lit | occluded)
    assert lit.sum() > 100 and occluded.sum() > 100
    assert set(np.unique(hits["instanceCustomIndex"])) >= {0, 3, 4}      # diffuse, metal and glass were hit


@pytest.mark.parametrize("name", ["c0", "c1", "c2"])
def test_frame_radiance_rmse(mods, name):
    """whole frames against the CPU oracle: >= 99 % of the pixels within 1e-5 (see _near_oracle), RGBA8 within 1 LSB on
    >= 99.9 % of the bytes; second TraceRays call exercises the progressive running mean (totalSamples > 0)"""
    rd, scenes = mods
    s = _small_scenes(scenes)[name]
    dev = scenes.DeviceScene(s)
    osc = ob.OracleScene(s)
    for frame in range(2):
        img = dev.render()
        osc.frame()
        assert _near_oracle(dev.read_scratch(), osc.scratch), (name, frame)
        diff = np.abs(img.reshape(-1).astype(int) - osc.image.astype(int))
        assert (diff <= 1).mean() >= 0.999
        st = rd.GetTraceStats()
        assert st.rays_primary == s.width * s.height * int(s.rtprop["batchSize"])


def test_ray_counts_match_oracle(mods):
    """rays actually traced == the reference algorithm's traceRay count minus the one duplicate
    re-trace the reference issues after each primary miss (shader.cl:243-252 does not break at
    depth 0, the re-trace misses again and ends the path).  Exact at depth 1 (no transcendental is
    involved in a primary ray); at depth 3 a handful of bounce rays flip between hit and miss
    because sin/cos/acos differ by ulps between glibc and OCML -> 1 % tolerance."""
    rd, scenes = mods
    for depth, tol in ((1, 0.0), (3, 0.01)):
        s = scenes.c0_two_boxes(64, 64, spp=1, depth=depth)
        osc = ob.OracleScene(s)
        dev = scenes.DeviceScene(s)
        dev.render()
        st = rd.GetTraceStats()
        c = osc.render(counters=True).as_dict()
        px = np.arange(64 * 64, dtype=np.uint32)
        o, d = osc.generate_rays(px, np.stack([np.zeros_like(px), np.zeros_like(px), px], 1))
        primary_misses = int((ob.trace_batch(osc.tlas.tobytes(), o, d)["hit"] == 0).sum()) if depth > 1 else 0
        assert st.rays_primary == c["primary"] == 64 * 64
        assert c["shadow"] == c["rays"][1] == c["hits"] and st.rays_shadow == st.closest_hits
        assert c["rays"][0] == c["primary"] + c["bounce"]
        assert abs(int(st.rays_shadow) - c["shadow"]) <= tol * c["shadow"]
        assert abs(int(st.rays_bounce) + primary_misses - c["bounce"]) <= tol * max(1, c["bounce"])


def test_debug_and_degenerate_rtprops(mods):
    rd, scenes = mods
    s = scenes.c0_two_boxes(48, 48, spp=2, depth=4)
    dev = scenes.DeviceScene(s)
    osc = ob.OracleScene(s)
    for kw in (dict(debug=1), dict(debug=0, depth=0), dict(depth=1, batchSize=1), dict(batchSize=0, depth=3)):
        dev.set_rtprop(totalSamples=0, **kw); osc.set_rtprop(totalSamples=0, **kw)
        dev.clear_scratch(); osc.scratch[:] = 0
        img = dev.render(); osc.frame()
        assert np.sqrt(np.mean((dev.read_scratch().reshape(-1) - osc.scratch) ** 2)) < 1e-5, kw
        assert (np.abs(img.reshape(-1).astype(int) - osc.image.astype(int)) <= 1).mean() >= 0.999, kw


def test_chunked_samples_identical(mods):
    """splitting a batch into sample chunks (bounded paths in flight) does not change a single bit"""
    rd, scenes = mods
    s = scenes.c1_cornell(64, 36, spp=4, depth=4, sphere_subdiv=2)
    dev = scenes.DeviceScene(s)
    dev.render()
    a = dev.read_scratch().copy()
    rd.SetOption("chunk_paths", 64 * 36)        # one sample per chunk
    try:
        dev.set_rtprop(totalSamples=0)
        dev.clear_scratch()
        dev.render()
        b = dev.read_scratch().copy()
    finally:
        rd.SetOption("chunk_paths", 16 << 20)
    assert np.array_equal(_bits(a), _bits(b))


def test_tile_sharding_bit_identical(mods):
    """rendering rank r of w ranks in turn and merging == the unsharded frame, bit for bit
    (pixel / RNG indices stay global); pack -> unpack of the owned tiles round-trips"""
    rd, scenes = mods
    s = scenes.c1_cornell(100, 60, spp=2, depth=3, sphere_subdiv=2)      # not a multiple of the tile size
    dev = scenes.DeviceScene(s)
    dev.render()
    full = dev.read_scratch().copy()
    full_img = rd.ReadBuffer(dev.plt, dev.rdImage, 100 * 60 * 4).copy()
    W, H, world = 100, 60, 3
    merged = rd.CreateBuffer(dev.plt, W * H * 16)
    merged_img = rd.CreateBuffer(dev.plt, W * H * 4)
    merged_multi = rd.CreateBuffer(dev.plt, W * H * 4)
    packed_imgs = []
    try:
        for r in range(world):
            rd.SetShard(r, world, 16, 16)
            dev.set_rtprop(totalSamples=0)
            dev.clear_scratch()
            dev.render()
            from radiance_ray_tracing_amd import _lib
            L = _lib.lib()
            ntiles = ((W + 15) // 16) * ((H + 15) // 16)
            owned = (ntiles - r + world - 1) // world
            packed = rd.CreateBuffer(dev.plt, owned * 256 * 16)
            packed8 = rd.CreateBuffer(dev.plt, owned * 256 * 4)
            assert L.rdx_pack_tiles(dev.rdImageScratch.handle, packed.handle, W, H, 16, r, world) == 0
            assert L.rdx_unpack_tiles(packed.handle, merged.handle, W, H, 16, r, world) == 0
            assert L.rdx_pack_tiles(dev.rdImage.handle, packed8.handle, W, H, 4, r, world) == 0
            assert L.rdx_unpack_tiles(packed8.handle, merged_img.handle, W, H, 4, r, world) == 0
            packed_imgs.append(packed8)
        # the gathering rank's one-call variant: every rank's packed buffer in one go
        import ctypes as C
        arr = (C.c_void_p * world)(*[p.handle for p in packed_imgs])
        assert L.rdx_unpack_tiles_multi(arr, 0, world, merged_multi.handle, W, H, 4, world) == 0
        assert L.rdx_unpack_tiles_multi(arr, 1, world, merged_multi.handle, W, H, 4, world) != 0      # rank out of range
    finally:
        rd.SetShard(0, 1, 64, 64)
    got = np.empty(W * H * 4, np.float32)
    rd.ReadBuffer(dev.plt, merged, got.nbytes, got.view(np.uint8))
    assert np.array_equal(_bits(got[np.arange(W * H * 4) % 4 != 3]), _bits(full.reshape(-1)[np.arange(W * H * 4) % 4 != 3]))
    assert np.array_equal(rd.ReadBuffer(dev.plt, merged_img, W * H * 4), full_img)
    assert np.array_equal(rd.ReadBuffer(dev.plt, merged_multi, W * H * 4), full_img)


def test_tlas_cache_file_roundtrip(mods, tmp_path):
    """TopAccelStructToFile / FileToTopAccelStruct (radiance.cpp:428-479): raw blob, byte-identical,
    and the reloaded structure traces identically"""
    rd, scenes = mods
    s = scenes.c1_cornell(32, 18, sphere_subdiv=2)
    dev = scenes.DeviceScene(s)
    path = str(tmp_path / "scene.cache")
    rd.TopAccelStructToFile(dev.plt, dev.topAccelStruct, path)
    blob_o, _, _ = ob.scene_tlas(s)
    assert open(path, "rb").read() == blob_o
    tl = rd.FileToTopAccelStruct(dev.plt, path)
    osc = ob.OracleScene(s, blob_o)
    o, d = _ray_batch(osc, 512, 3)
    assert np.array_equal(_bits(rd.TraceBatch(tl, o, d)), _bits(rd.TraceBatch(dev.topAccelStruct, o, d)))
    with pytest.raises(rd.RadianceError):
        rd.FileToTopAccelStruct(dev.plt, str(tmp_path / "missing.cache"))
    open(str(tmp_path / "short.cache"), "wb").write(blob_o[:100])
    with pytest.raises(rd.RadianceError):
        rd.FileToTopAccelStruct(dev.plt, str(tmp_path / "short.cache"))
    # side-car (<path>.meta: magic, version, byte count, FNV-1a hash): written next to the raw blob, verified on load;
    # a blob that contradicts it is refused, a cache without side-car (as the reference writes them) still loads
    meta = open(path + ".meta").read().split()
    assert meta[:2] == ["RDXCACHE", "1"] and int(meta[3]) == len(blob_o)
    bad = bytearray(blob_o); bad[len(bad) // 2] ^= 0x40
    open(path, "wb").write(bytes(bad))
    with pytest.raises(rd.RadianceError, match="side-car"):
        rd.FileToTopAccelStruct(dev.plt, path)
    os.remove(path + ".meta")
    open(path, "wb").write(blob_o)
    rd.FileToTopAccelStruct(dev.plt, path)


def test_error_paths(mods):
    rd, scenes = mods
    plt = rd.Platform.GetPlatform()
    b = rd.CreateBuffer(plt, 64)
    with pytest.raises(rd.RadianceError):
        rd.WriteBuffer(plt, b, 128, np.zeros(128, np.uint8))
    with pytest.raises(rd.RadianceError):
        rd.ReadBuffer(plt, b, 32, np.zeros(32, np.uint8), offset=48)
    with pytest.raises(rd.RadianceError):
        rd.CreateShaderModule(plt, "void main() {}", 14, "x")         # no raygen kernel
    s = scenes.c0_two_boxes(16, 16)
    dev = scenes.DeviceScene(s)
    with pytest.raises(rd.RadianceError):
        rd.TraceRays(plt, 0, 0, 0, 64, 64)                            # buffers sized for 16x16
    # a non-TLAS buffer in slot 13 is rejected
    ds = list(dev.descSet); ds[13] = dev.rdCamData
    rd.BindDescriptorSet(plt, ds)
    with pytest.raises(rd.RadianceError):
        rd.TraceRays(plt, 0, 0, 0, 16, 16)
    dev.bind()
    rd.TraceRays(plt, 0, 0, 0, 16, 16)


def test_full_size_properties(mods):
    """BASELINE config 1 at full size (1920x1080, 4 spp, depth 8): size-independent properties --
    determinism (two runs bit-identical), progressive mean (frame 2 = exact running mean of two
    independent batches), shadow rays == closest hits, finite radiance, and a 4096-pixel random
    subset re-rendered by the CPU oracle agrees (>= 98 % of the pixels within 1e-5 after 8 bounces)"""
    rd, scenes = mods
    s = scenes.c1_cornell()
    dev = scenes.DeviceScene(s)
    dev.render()
    a = dev.read_scratch().copy()
    st = rd.GetTraceStats()
    assert st.rays_primary == 1920 * 1080 * 4 and st.rays_shadow == st.closest_hits
    assert np.isfinite(a).all()
    dev.set_rtprop(totalSamples=0); dev.clear_scratch()
    dev.render()
    assert np.array_equal(_bits(a), _bits(dev.read_scratch()))
    # oracle on a pixel subset of the same frame
    osc = ob.OracleScene(s)
    px = np.random.default_rng(4).choice(1920 * 1080, 4096, replace=False).astype(np.uint32)
    osc.render(pixels=px)
    assert _near_oracle(a.reshape(-1, 4)[px], osc.scratch.reshape(-1, 4)[px], 0.98)


def test_cpp_facade_sample_runs(mods, tmp_path):
    """samples/cornell_rd.cpp drives the core through include/radiance.h (namespace RD) exactly like
    the reference's sample1.cpp: build, two progressive frames, read-back; deterministic output"""
    import subprocess
    from test_cpu_oracle import _build_cpp_sample
    exe = _build_cpp_sample(tmp_path)
    outs = []
    for k in range(2):
        p = str(tmp_path / ("o%d.ppm" % k))
        r = subprocess.run([exe, "96", "54", "2", p], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        data = open(p, "rb").read()
        assert data.startswith(b"P6\n96 54\n255\n") and len(data) == len(b"P6\n96 54\n255\n") + 96 * 54 * 3
        outs.append(data)
    assert outs[0] == outs[1]
    px = np.frombuffer(outs[0][len(b"P6\n96 54\n255\n"):], np.uint8)
    assert px.std() > 10            # an actual picture, not a constant
    # the windowless version of the reference's interactive host loop (sample1.cpp:447-548): 3 views = 2 camera edits, each
    # resetting totalSamples, 2 progressive frames per view.  View 0 is the run above; the edited views differ from it; and the
    # accumulation really restarts: view 2 rendered after the edits equals view 2 rendered alone is not expressible with this
    # CLI, so the check is that the whole sequence is deterministic.
    seqs = []
    for k in range(2):
        p = str(tmp_path / ("v%d.ppm" % k))
        r = subprocess.run([exe, "96", "54", "2", p, "2", "3"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.count("ms per frame") == 3
        seqs.append([open(str(tmp_path / ("v%d_%d.ppm" % (k, v))), "rb").read() for v in range(3)])
    assert seqs[0] == seqs[1]
    assert seqs[0][0] == outs[0] and seqs[0][1] != seqs[0][0] and seqs[0][2] != seqs[0][1]


@pytest.mark.parametrize("cfg", ["c1_cornell", "c2_atrium"])
def test_full_size_kernels_agree_bitwise(mods, cfg):
    """BASELINE configs 1 and 2 at full size (1920x1080, 4 spp, depth 8; ~66 M rays): the production
    wave-cooperative kernel, the per-lane wide-node kernel and the reference-order kernel (whose HitData
    and visit counters are pinned to the oracle bit for bit at small sizes) produce bit-identical
    imageScratch and RGBA8 -- every one of the ~37 M closest hits and ~29 M shadow queries agrees."""
    rd, scenes = mods
    s = scenes.CONFIGS[cfg]()
    dev = scenes.DeviceScene(s)
    ref = None
    try:
        for kernel in (0, 3, 2, 1):
            rd.SetOption("kernel", kernel)
            dev.set_rtprop(totalSamples=0); dev.clear_scratch()
            img = dev.render().copy()
            scr = dev.read_scratch().copy()
            st = rd.GetTraceStats()
            cur = (scr, img, st.rays_bounce, st.rays_shadow)
            if ref is None:
                ref = cur
            else:
                assert np.array_equal(_bits(ref[0]), _bits(cur[0])), kernel
                assert np.array_equal(ref[1], cur[1]) and ref[2:] == cur[2:], kernel
    finally:
        rd.SetOption("kernel", 3)


def test_fused_and_split_schedules_identical(mods):
    """shadow(d) + extend(d+1) traced by one fused launch or by two launches give bit-identical frames; so
    do 2 concurrent sample groups, the two-stream overlap, and the whole-path pipeline (one launch per chunk)"""
    rd, scenes = mods
    s = scenes.c2_atrium(160, 90, spp=4, depth=6, detail=0.2)
    dev = scenes.DeviceScene(s)
    outs = []
    try:
        for opts in ({"fuse": 1}, {"fuse": 0}, {"fuse": 0, "overlap": 1}, {"fuse": 0, "groups": 2}, {"groups": 2}, {"groups": 0},
                     {"groups": 4}, {"pipeline": 1}):
            for k, v in {"fuse": -1, "overlap": 0, "groups": 1, "pipeline": 0, **opts}.items():
                rd.SetOption(k, v)
            dev.set_rtprop(totalSamples=0); dev.clear_scratch()
            dev.render()
            st = rd.GetTraceStats()
            # 0 = automatic: two groups for a chunk this small; the whole-path pipeline has no groups
            assert st.groups == (opts.get("groups", 1) or 2) or "pipeline" in opts
            outs.append((dev.read_scratch().copy(), st.rays_bounce, st.rays_shadow))
    finally:
        rd.SetOption("fuse", -1); rd.SetOption("overlap", 0); rd.SetOption("groups", 0); rd.SetOption("pipeline", 0)
    for o in outs[1:]:
        assert np.array_equal(_bits(outs[0][0]), _bits(o[0])) and outs[0][1:] == o[1:]


def test_4k_chunked_frame(mods):
    """BASELINE config 3 geometry at 3840x2160 with 6 spp on one GPU: 50 M paths run as sample chunks of
    <= 16 M paths in flight; chunking must not change a bit (compared with 8 M-path chunks) and the
    statistics must add up"""
    rd, scenes = mods
    s = scenes.c2_atrium(3840, 2160, spp=6, depth=3, detail=0.3)
    dev = scenes.DeviceScene(s)
    dev.render()
    a = dev.read_scratch().copy()
    st = rd.GetTraceStats()
    assert st.rays_primary == 3840 * 2160 * 6 and st.rays_shadow == st.closest_hits and np.isfinite(a).all()
    try:
        rd.SetOption("chunk_paths", 8 << 20)
        dev.set_rtprop(totalSamples=0); dev.clear_scratch()
        dev.render()
        assert np.array_equal(_bits(a), _bits(dev.read_scratch()))
    finally:
        rd.SetOption("chunk_paths", 16 << 20)


def test_instanced_grid_and_ragged_batches(mods):
    """45 instances of two shared BLASes (the 9-instance grid of samples/sample2.cpp:404-505, enlarged so that
    top-level leaves and the cooperative kernel's instance-mask stack entries, subtree stealing and ray-index
    reservation all see many instances): the wave-cooperative kernel, the per-lane wide kernel and the
    reference-order kernel agree with the oracle bit for bit on ragged batch sizes -- 0, 1, 63, 64, 65, 1000 and
    20 000 rays (less than one wave, exactly one, one and a bit, less than one ray per resident wave, many)"""
    rd, scenes = mods
    s = scenes.Scene("grid")
    ball = s.add_mesh(scenes.icosphere(2, 0.45))
    cube = s.add_mesh(scenes.box([-0.35, -0.35, -0.35], [0.35, 0.35, 0.35]))
    s.materials = [scenes.material((0.7, 0.7, 0.7), 0.0, 0.5), scenes.material((0.9, 0.8, 0.5), 0.9, 0.2)]
    k = 0
    for ix in range(5):
        for iy in range(3):
            for iz in range(3):
                tf = scenes.translate(1.3 * (ix - 2), 1.3 * (iy - 1), 1.3 * (iz - 1)) @ scenes.rotate_y(11.0 * k) @ \
                    scenes.scale(1.0 + 0.05 * (k % 3), 1.0, 1.0 - 0.04 * (k % 2))
                s.add_instance(ball if k % 2 == 0 else cube, tf, k % 2)
                k += 1
    s.camera = scenes.blender_camera(64, 48, 0.05, 0.036, 9.0, 0.0, (0.5, 9.0, 1.0), (-96.0, 180.0, 0.0))
    s.sceneProps = scenes.blender_dir_light(-45.0, 20.0, 5.0)
    s.rtprop = scenes._rtprop(0, 2, 4)
    dev = scenes.DeviceScene(s)
    blob_o, _, _ = ob.scene_tlas(s)
    assert rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes() == blob_o
    osc = ob.OracleScene(s, blob_o)
    o, d = _ray_batch(osc, 5000, 11)
    fields = ("distance", "primitiveIndex", "instanceIndex", "instanceCustomIndex", "barycentric", "hitPoint", "transform")
    hits_total = 0
    for n in (0, 1, 63, 64, 65, 1000, 20000):
        for rec in (1, 2):
            ref = ob.trace_batch(blob_o, o[:n], d[:n], 0.001, 1000.0, rec)
            for kernel in (3, 2, 1, 0):
                rd.SetOption("kernel", kernel)
                try:
                    got = rd.TraceBatch(dev.topAccelStruct, o[:n], d[:n], 0.001, 1000.0, rec)
                finally:
                    rd.SetOption("kernel", 3)
                assert got.shape[0] == n and np.array_equal(ref["hit"], got["hit"]), (n, rec, kernel)
                h = ref["hit"] == 1
                if rec == 1:
                    for f in fields:
                        assert np.array_equal(_bits(ref[f][h]), _bits(got[f][h])), (n, kernel, f)
            hits_total += int(h.sum())
    assert hits_total > 5000
    # and a frame of it against the oracle (progressive: two TraceRays calls)
    osc.frame(); osc.frame()
    dev.render(); dev.set_rtprop(totalSamples=2); dev.render()
    assert _near_oracle(dev.read_scratch(), osc.scratch)


def _obj_test_scene(scenes, w, h):
    s = scenes.Scene("objtest")
    ball = s.add_mesh(scenes.icosphere(3, 1.0))
    slab = s.add_mesh(scenes.box([-3.0, -1.5, -3.0], [3.0, -1.0, 3.0]))
    wall = s.add_mesh(scenes.quad([-3, -1, 3], [-3, 4, 3], [3, 4, 3], [3, -1, 3], [0, 0, -1]))     # behind the ball, seen from -z
    s.materials = [scenes.material((0.8, 0.3, 0.3), 0.0, 0.6), scenes.material((0.9, 0.8, 0.5), 0.9, 0.2),
                   scenes.material((0.95, 0.95, 0.95), 0.0, 0.08, 1.0, 1.45)]
    s.add_instance(ball, None, 2); s.add_instance(slab, None, 0); s.add_instance(wall, None, 1)
    s.camera = scenes.blender_camera(w, h, 0.05, 0.036, 8.0, 0.0, (0.5, 9.0, 1.0), (-96.0, 180.0, 0.0))
    s.sceneProps = scenes.blender_dir_light(-45.0, 20.0, 5.0)
    s.rtprop = scenes._rtprop(0, 2, 4)
    return s


def test_obj_scene_renders_like_the_oracle(mods, tmp_path):
    """a scene written as OBJ + MTL and read back by rdx_obj_load (scenes.load_obj) renders within
    the radiance tolerance of the oracle fed with the same loaded buffers, and traces bit-identically to it"""
    rd, scenes = mods
    src = _obj_test_scene(scenes, 96, 54)
    path = str(tmp_path / "scene.obj")
    scenes.save_obj(src, path)
    s = scenes.load_obj(path, 96, 54, 2, 4, camera=src.camera, light=src.sceneProps)
    assert s.triangle_count() == src.triangle_count() and len(s.materials) == 3
    dev = scenes.DeviceScene(s)
    blob_o, _, _ = ob.scene_tlas(s)
    assert rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes() == blob_o
    osc = ob.OracleScene(s, blob_o)
    o, d = _ray_batch(osc, 2048, 9)
    ref = ob.trace_batch(blob_o, o, d)
    got = rd.TraceBatch(dev.topAccelStruct, o, d)
    assert np.array_equal(_bits(ref), _bits(got))
    osc.frame()
    dev.render()
    assert _near_oracle(dev.read_scratch(), osc.scratch, 0.99, 1e-2)      # a glass ball fills a third of this small picture


def test_cpp_scene_loader_sample(mods, tmp_path):
    """samples/obj_rd.cpp: RD::Scene::Load (include/sceneBuilder.h) + INCLUDE_SCENE_DESC as in the reference's sample1;
    the second run takes the TLAS from the .cache file the first one wrote and must give the same picture"""
    import subprocess
    rd, scenes = mods
    from test_cpu_oracle import _build_cpp_sample
    exe = _build_cpp_sample(tmp_path, "obj_rd")
    path = str(tmp_path / "scene.obj")
    scenes.save_obj(_obj_test_scene(scenes, 96, 54), path)
    outs = []
    for k, extra in enumerate(([], ["cache"])):
        p = str(tmp_path / ("o%d.ppm" % k))
        r = subprocess.run([exe, path, "96", "54", p] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(open(p, "rb").read())
        assert ("BVH build report" in r.stdout) == (k == 0)
    assert os.path.exists(path + ".cache") and os.path.exists(path + ".cache.meta")
    assert outs[0] == outs[1]
    assert np.frombuffer(outs[0][len(b"P6\n96 54\n255\n"):], np.uint8).std() > 10


def test_fuzz_random_scenes(mods):
    """tools/fuzz_parity.py, 12 seeds: random meshes (incl. stacks of coincident triangles -> leaves of more than 8
    triangles), up to 70 instances with random affine transforms and shared BLASes, rays from everywhere: every
    production kernel returns the reference-order kernel's HitData bit for bit (that kernel is pinned to the oracle by
    the tests above)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run(12, first_seed=500, verbose=False) == 0


def test_large_and_walked_top_level_trees(mods):
    """the pool kernel evaluates top-level trees of <= 64 nodes all at once and walks larger ones: a 343-instance grid
    (more than 64 top-level nodes -> walked) and the same kernel with `top_flat` off on a 27-instance grid (walked by
    option) both agree with the reference-order kernel bit for bit, closest hit and any hit"""
    rd, scenes = mods
    for side, flat in ((7, 1), (3, 0), (3, 1)):
        s = scenes.Scene("grid%d" % side)
        ball = s.add_mesh(scenes.icosphere(1, 0.3))
        cube = s.add_mesh(scenes.box([-0.25, -0.25, -0.25], [0.25, 0.25, 0.25]))
        s.materials = [scenes.material((0.7, 0.7, 0.7), 0.0, 0.5)]
        k = 0
        for ix in range(side):
            for iy in range(side):
                for iz in range(side):
                    tf = scenes.translate(0.9 * (ix - side // 2), 0.9 * (iy - side // 2), 0.9 * (iz - side // 2)) @ scenes.rotate_y(7.0 * k)
                    s.add_instance(ball if k % 3 else cube, tf, 0)
                    k += 1
        s.camera = scenes.blender_camera(64, 48, 0.05, 0.036, 9.0, 0.0, (0.5, 9.0, 1.0), (-96.0, 180.0, 0.0))
        s.sceneProps = scenes.blender_dir_light(-45.0, 20.0, 5.0)
        s.rtprop = scenes._rtprop(0, 1, 2)
        dev = scenes.DeviceScene(s)
        rng = np.random.default_rng(side)
        n = 8000
        o = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
        d = (rng.uniform(-2, 2, (n, 3)).astype(np.float32) - o); d /= np.linalg.norm(d, axis=1, keepdims=True)
        try:
            rd.SetOption("top_flat", flat)
            for rec in (1, 2):
                ref = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec, reference_order=True)
                got = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
                assert np.array_equal(ref["hit"], got["hit"]), (side, flat, rec)
                if rec == 1:
                    assert np.array_equal(_bits(ref), _bits(got)), (side, flat)
                assert ref["hit"].sum() > 1000
        finally:
            rd.SetOption("top_flat", 1)


def test_single_leaf_instances_inline_or_not(mods):
    """instances whose BLAS is one small leaf (the quads of sample1) are tested inside the top-level step by default (`INL`
    kernels) and go through the instance step and the test queue with `inline_leaf_roots` 0 (the other instantiation of the
    pool kernels, whose top-level step reads its boxes by lane broadcast): same HitData as the reference-order kernel and
    the same frame, bit for bit, either way"""
    rd, scenes = mods
    s = scenes.c1_cornell(160, 90, spp=2, depth=4, sphere_subdiv=3)
    dev = scenes.DeviceScene(s)
    rng = np.random.default_rng(11)
    n = 20000
    o = rng.uniform(-1.2, 1.2, (n, 3)).astype(np.float32) + np.array([0, 1, 0], np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    frames = []
    try:
        for inl in (1, 0):
            rd.SetOption("inline_leaf_roots", inl)
            for cull in (0, 1):
                rd.SetOption("cull", cull)
                for rec in (1, 2):
                    ref = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec, reference_order=True)
                    got = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
                    assert np.array_equal(ref["hit"], got["hit"]), (inl, cull, rec)
                    if rec == 1:
                        assert np.array_equal(_bits(ref), _bits(got)), (inl, cull)
                    assert ref["hit"].sum() > 1000
                dev.set_rtprop(totalSamples=0)
                dev.render()
                frames.append(dev.read_scratch().copy())
    finally:
        rd.SetOption("inline_leaf_roots", 1); rd.SetOption("cull", -1)
    for f in frames[1:]:
        assert np.array_equal(frames[0].view(np.uint32), f.view(np.uint32))


def test_foreign_blob_is_refused(mods):
    """a TLAS blob whose nodes are not in the reference's DFS pre-order (a foreign or corrupted cache file) is refused with a
    clear error at first use instead of being mis-sized on the GPU"""
    rd, scenes = mods
    s = scenes.c1_cornell(32, 18, sphere_subdiv=2)
    dev = scenes.DeviceScene(s)
    blob = bytearray(rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes())
    words = np.frombuffer(blob, np.uint32).copy()
    # top-level root = node 0 at byte 16: words 4..15; children in words 12, 13 -> swap them
    assert not (words[12] & 0x80000000)
    words[12], words[13] = words[13], words[12]
    rd.WriteBuffer(dev.plt, dev.topAccelStruct, words.nbytes, words.view(np.uint8))
    with pytest.raises(rd.RadianceError, match="DFS pre-order"):
        rd.TraceRays(dev.plt, 0, 0, 0, 32, 18)
    words[12], words[13] = words[13], words[12]
    rd.WriteBuffer(dev.plt, dev.topAccelStruct, words.nbytes, words.view(np.uint8))
    rd.TraceRays(dev.plt, 0, 0, 0, 32, 18)


def _shard_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank); os.environ["WORLD_SIZE"] = str(world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import torch
    import torch.distributed as tdist
    import rrt_amd  # noqa: F401
    from radiance_ray_tracing_amd import dist as rdist, rd, scenes
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    plt = rd.Platform.GetPlatform(0)
    W, H = 200, 120
    s = scenes.c1_cornell(W, H, spp=1, depth=3, sphere_subdiv=2)
    dev = scenes.DeviceScene(s, plt)
    sh = rdist.FrameSharder(rd, plt, W, H, rank, world, 32, 32, torch.device("cuda", 0))
    frames = []
    for f in range(4):                      # progressive: every frame is a different picture
        rd.TraceRays(plt, 0, 0, 0, W, H)
        sh.gather_image(dev.rdImage)
        dev.set_rtprop(totalSamples=f + 1)
        if rank == 0:
            frames.append(rd.ReadBuffer(plt, dev.rdImage, W * H * 4).copy())
    if rank == 0:
        np.save(os.path.join(out_dir, "sharded.npy"), np.stack(frames))
    tdist.barrier()
    tdist.destroy_process_group()


def test_two_ranks_gather_every_frame(mods, tmp_path):
    """two ranks (sharing this box's one GPU, gloo transport: same code path as RCCL up to the transfer) render four
    DIFFERENT progressive frames, each gathered to rank 0 through FrameSharder's alternating staging buffers: every frame
    rank 0 sees must equal the unsharded frame of the same sample count"""
    import torch.multiprocessing as mp
    rd, scenes = mods
    port = 29600 + (os.getpid() % 1000)
    mp.get_context("spawn")
    mp.spawn(_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(str(tmp_path / "sharded.npy"))
    W, H = 200, 120
    dev = scenes.DeviceScene(scenes.c1_cornell(W, H, spp=1, depth=3, sphere_subdiv=2))
    rd.SetShard(0, 1, 64, 64)
    for f in range(4):
        rd.TraceRays(dev.plt, 0, 0, 0, W, H)
        dev.set_rtprop(totalSamples=f + 1)
        assert np.array_equal(rd.ReadBuffer(dev.plt, dev.rdImage, W * H * 4), got[f]), f
    assert not np.array_equal(got[0], got[3])


def _textured_scene(scenes, w, h):
    """two textured quads + a textured box under the sample1 light: albedo / roughness / metallic / normal texture indices"""
    s = scenes.Scene("textured")
    floor = s.add_mesh(scenes.quad([-3, 0, -3], [3, 0, -3], [3, 0, 3], [-3, 0, 3], [0, 1, 0]))
    wall = s.add_mesh(scenes.quad([-3, 0, 3], [3, 0, 3], [3, 4, 3], [-3, 4, 3], [0, 0, -1]))
    cube = s.add_mesh(scenes.box([-0.8, 0.0, -0.8], [0.8, 1.6, 0.8]))
    m0 = scenes.material((0.7, 0.7, 0.7), 0.0, 0.6); m0["albedoTexIdx"] = 0
    m1 = scenes.material((0.7, 0.7, 0.7), 0.0, 0.6); m1["albedoTexIdx"] = 1; m1["roughnessTexIdx"] = 2; m1["metallicTexIdx"] = 2
    m2 = scenes.material((0.9, 0.8, 0.5), 0.2, 0.4); m2["albedoTexIdx"] = 0; m2["normalTexIdx"] = 1
    s.materials = [m0, m1, m2]
    s.add_instance(floor, None, 0); s.add_instance(wall, None, 1); s.add_instance(cube, scenes.translate(0.3, 0.0, 0.2) @ scenes.rotate_y(25.0), 2)
    s.camera = scenes.blender_camera(w, h, 0.05, 0.036, 8.0, 0.0, (0.5, 9.0, 2.5), (-100.0, 180.0, 0.0))
    s.sceneProps = scenes.blender_dir_light(-45.0, 20.0, 6.0)
    s.rtprop = scenes._rtprop(0, 2, 3)
    return s


def _test_textures(size=64):
    yy, xx = np.mgrid[0:size, 0:size]
    t = np.zeros((3, size, size, 4), np.uint8)
    t[0, ..., 0] = np.where(((xx // 8) + (yy // 8)) % 2, 230, 40); t[0, ..., 1] = 120; t[0, ..., 2] = (xx * 4) % 256; t[0, ..., 3] = 255
    t[1, ..., 0] = (xx * 3 + yy) % 256; t[1, ..., 1] = (yy * 5) % 256; t[1, ..., 2] = 200; t[1, ..., 3] = 255
    t[2, ..., 0] = 17; t[2, ..., 1] = 60 + (xx % 16) * 8; t[2, ..., 2] = np.where(yy % 32 < 16, 0, 255); t[2, ..., 3] = 255
    return t


def test_image_array_and_texture_path(mods):
    """SURVEY 8(f) rank 3.  (1) CreateImageArray / WriteImage / ReadImage (radiance.cpp:96-137,202-224): region writes at the
    origin of a layer, round trip, errors.  (2) option "textures" 0 (default) reproduces the LIVE reference shader, whose
    texture reads are commented out -> texel 0: frames of a scene with textured materials equal the oracle's stubbed path.
    (3) "textures" 1 performs the read (coord (u, 1-v, texIdx), shader2.cl:255-265) for every addressing / filter mode:
    `material` payloads and frames agree with the oracle's OpenCL-1.2-spec sampler"""
    rd, scenes = mods
    plt = rd.Platform.GetPlatform()
    ia = rd.CreateImageArray(plt, 16, 8, 3)
    a = np.arange(16 * 8 * 4, dtype=np.uint8).reshape(8, 16, 4)
    rd.WriteImage(plt, ia, 16, 8, 1, a)
    assert np.array_equal(rd.ReadImage(plt, ia, 16, 8, 1), a)
    assert not rd.ReadImage(plt, ia, 16, 8, 0).any() and not rd.ReadImage(plt, ia, 16, 8, 2).any()
    sub = (np.arange(5 * 3 * 4, dtype=np.uint8) + 100).reshape(3, 5, 4)
    rd.WriteImage(plt, ia, 5, 3, 2, sub)                                 # a region smaller than the image, at its origin
    full = rd.ReadImage(plt, ia, 16, 8, 2)
    assert np.array_equal(full[:3, :5], sub) and not full[3:].any() and not full[:, 5:].any()
    raw = rd.ReadBuffer(plt, ia, 16 * 8 * 4 * 3).reshape(3, 8, 16, 4)    # layer-major, rows tightly packed
    assert np.array_equal(raw[1], a)
    with pytest.raises(rd.RadianceError):
        rd.WriteImage(plt, ia, 16, 8, 3, a)                              # layer out of range
    with pytest.raises(rd.RadianceError):
        rd.ReadImage(plt, ia, 17, 8, 0)
    with pytest.raises(rd.RadianceError):
        rd.CreateSampler(plt, 7, rd.RD_FILTER_LINEAR)

    W, H = 96, 54
    s = _textured_scene(scenes, W, H)
    tex = _test_textures()
    dev = scenes.DeviceScene(s)
    img = rd.CreateImageArray(plt, 64, 64, 3)
    for l in range(3):
        rd.WriteImage(plt, img, 64, 64, l, tex[l])
    osc = ob.OracleScene(s)
    px = np.arange(W * H, dtype=np.uint32)
    o, d = rd.GenerateBatch(px, np.stack([np.zeros_like(px), np.zeros_like(px), px], 1))
    hits = rd.TraceBatch(dev.topAccelStruct, o, d)
    k = hits["hit"] == 1
    assert set(np.unique(hits["instanceCustomIndex"][k])) == {0, 1, 2}
    frames = (px % 5).astype(np.uint32); depths = (px % 3).astype(np.int32)
    modes = [(rd.RD_ADDRESS_REPEAT, rd.RD_FILTER_LINEAR, 0, True), (rd.RD_ADDRESS_REPEAT, rd.RD_FILTER_NEAREST, 0, False),
             (rd.RD_ADDRESS_CLAMP_TO_EDGE, rd.RD_FILTER_LINEAR, 1, True), (rd.RD_ADDRESS_CLAMP, rd.RD_FILTER_NEAREST, 2, False),
             (rd.RD_ADDRESS_MIRRORED_REPEAT, rd.RD_FILTER_LINEAR, 3, True)]
    try:
        # (2) default: texel 0, whatever is bound
        ds = list(dev.descSet); ds[11] = img; ds[12] = rd.CreateSampler(plt, rd.RD_ADDRESS_REPEAT, rd.RD_FILTER_LINEAR)
        rd.BindDescriptorSet(plt, ds)
        dev.render(); osc.frame()
        stub = dev.read_scratch().copy()
        assert _near_oracle(stub, osc.scratch)
        # (3) sampled
        rd.SetOption("textures", 1)
        outs = []
        for addr, filt, omode, olin in modes:
            ds[12] = rd.CreateSampler(plt, addr, filt)
            rd.BindDescriptorSet(plt, ds)
            osc.bind_textures(tex, omode, olin)
            got = rd.MaterialBatch(hits, d, px, frames, depths)
            ref = osc.material_batch(hits, d, px, frames, depths)
            for f in ("nextFactor", "nextRayOrigin", "nextRayDirection"):
                scale = np.maximum(1.0, np.abs(ref[f][k]))
                assert (np.abs(ref[f][k] - got[f][k]) / scale).max() < 2e-5, (addr, filt, f)
            dev.set_rtprop(totalSamples=0); dev.clear_scratch(); osc.set_rtprop(totalSamples=0); osc.scratch[:] = 0
            dev.render(); osc.frame()
            outs.append(dev.read_scratch().copy())
            # (a uv within an ulp of a texel boundary picks the neighbouring texel on one side: a few pixels differ visibly)
            assert _near_oracle(outs[-1], osc.scratch, 0.985, 1e-2), (addr, filt)
        assert not np.array_equal(outs[0], stub) and not np.array_equal(outs[0], outs[1])      # the textures are really read
    finally:
        rd.SetOption("textures", 0)


def test_instance_sbt_offsets(mods, tmp_path):
    """dispatch index = instanceSBTOffset + sbtRecordOffset (radiance.cl:281, shader.cl:574-605).
    (a) stock table, one instance with SBTOffset 1: its radiance-ray hits dispatch row 2 -- closest-hit `shadow` (payload.hit,
        colour 0, next ray untouched) and any-hit `anyShadow`, which ENDS the walk at the first accepted candidate in the
        reference's DFS order; shadow rays onto it dispatch row 3, which has no hit shaders.  Such scenes are traced by the
        reference-order kernel: HitData (incl. the first-candidate rule) and frames equal the CPU oracle's, which keeps the
        reference's megakernel structure; test_gpu_reference.py repeats it against the reference's own device code.
    (b) a library built for a two-table sbt.json (tests/golden/sbt_two_tables.json): instances with SBTOffset 4 reach the same
        shaders through rows 5 / 6 -> frames bit-identical to offset 0."""
    import subprocess
    import sys
    rd, scenes = mods
    s = scenes.c1_cornell(96, 54, spp=2, depth=4, sphere_subdiv=3)
    s.sbt_offsets = {5: 1, 7: 1}                       # the tall box and the sphere
    dev = scenes.DeviceScene(s)
    blob_o, _, _ = ob.scene_tlas(s)
    assert rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes() == blob_o
    osc = ob.OracleScene(s, blob_o)
    o, d = _ray_batch(osc, 4096, 21)
    for rec in (1, 2):
        ref = ob.trace_batch(blob_o, o, d, 0.001, 1000.0, rec)
        got = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
        assert np.array_equal(_bits(ref), _bits(got)), rec
    assert (ref["instanceSBTOffset"][ref["hit"] == 1] == 1).sum() > 200
    for f in range(2):
        dev.render(); osc.frame()
        assert _near_oracle(dev.read_scratch(), osc.scratch), f
    # (b)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "radiance-ray-tracing_amd", "librdx_sbt2.so")
    if not os.path.exists(lib):
        env = dict(os.environ, RDX_SBT_JSON=os.path.join(root, "tests", "golden", "sbt_two_tables.json"), RDX_LIB_NAME="librdx_sbt2.so")
        subprocess.check_call([sys.executable, os.path.join(root, "radiance-ray-tracing_amd", "build.py"), "--force"], env=env, stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sbt_two_tables_check.py")], env=dict(os.environ, RDX_LIB=lib),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


def test_single_process_multi_device(mods, tmp_path):
    """rdx_init_devices (SURVEY 8b "Threading", 8e): one process, N logical devices, one blocking TraceRays call that returns with
    the gathered frame on device 0.  Only one GPU is present here, so three logical devices share it (RDX_ALLOW_VIRTUAL_DEVICES:
    own contexts, streams, path buffers and buffer replicas each; the peer copies become same-device copies) -- the scheduler,
    the replication of writes, the tile gather of image + accumulator and the statistics are exercised, not the xGMI transfer.
    Two progressive frames must be bit-identical to the one-device render"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "md.npz")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "multi_device_check.py"), "3", out],
                       env=dict(os.environ, RDX_ALLOW_VIRTUAL_DEVICES="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.load(out)
    sys.path.insert(0, os.path.join(root, "tools"))
    import multi_device_check as mdc
    ref = mdc.render(1)
    for k in ref:
        assert np.array_equal(_bits(np.asarray(ref[k])), _bits(np.asarray(got[k]))), k
    # without the override a machine with fewer GPUs refuses instead of silently sharing
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "multi_device_check.py"), "3", out], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "RDX_ALLOW_VIRTUAL_DEVICES" in (r.stdout + r.stderr)


def test_user_shader_program_is_compiled_and_run(mods):
    """a shader text that is neither the reference's stock program nor a parameterless placeholder is
    compiled at run time (ROCm clang, OpenCL C, the GPU in use) and launched as a megakernel with the 14 descriptors bound by
    position -- tests/golden/user_shader.cl writes values derived from every binding.  A text that does not compile fails with
    the build log; the stock placeholder still selects the wavefront pipeline"""
    rd, scenes = mods
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "tests", "golden", "user_shader.cl")).read()
    s = scenes.c1_cornell(80, 48, spp=3, depth=5, sphere_subdiv=1)
    dev = scenes.DeviceScene(s, shader_text=text)
    for local in (64, 1):                      # the reference launches with local_work_size 1 (radiance.cpp:250-259)
        rd.SetOption("user_shader_local_size", local)
        dev.clear_scratch()
        rd.TraceRays(dev.plt, 0, 0, 0, 80, 48)
        scr = dev.read_scratch().reshape(-1, 4)
        img = rd.ReadBuffer(dev.plt, dev.rdImage, 80 * 48 * 4).reshape(-1, 4)
        b = s.buffers()
        i = np.arange(80 * 48); x = i % 80; y = i // 80
        blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, 16).view(np.uint32)
        assert np.abs(scr[:, 0] - np.sin(x * np.float32(0.1)) * np.cos(y * np.float32(0.07))).max() < 1e-6
        e1 = np.float32(3) + b["vertex"][0] + b["normal"][1] + b["uv"][2]
        e2 = np.float32(blob[0]) + np.float32(b["index"][1]) + np.array(s.materials)["albedo"][0][0] + \
            np.array(s.sceneProps).reshape(1).view(np.float32)[4] + np.float32(b["meshInfo"].view(np.int32)[4])
        assert np.allclose(scr[:, 1], e1, atol=1e-6) and np.allclose(scr[:, 2], e2, atol=1e-5)
        assert np.array_equal(img[:, 0], (x & 255).astype(np.uint8)) and np.array_equal(img[:, 1], (y & 255).astype(np.uint8))
        assert (img[:, 2] == ((5 * 16 + blob[1]) & 255)).all() and (img[:, 3] == 255).all()
    rd.SetOption("user_shader_local_size", 64)
    with pytest.raises(rd.RadianceError, match="compilation failed"):
        rd.CreateShaderModule(dev.plt, "__kernel void raygen(__global float* a) { this is not OpenCL C; }", 70, "x")
    # the placeholder (no parameters) and the stock pipeline are unaffected
    dev2 = scenes.DeviceScene(s)
    dev2.render()
    assert rd.GetTraceStats().rays_primary == 80 * 48 * 3


def test_user_program_traces_rays_with_the_product_library(mods):
    """A user raygen that #includes "radiance.cl" and calls traceRay() -- compiled at run time against the library's OWN device
    library (radiance-ray-tracing_amd/shader/, no reference file on the include path) under the pinned floating-point contract
    -- reproduces the reference's HitData bit for bit: the committed outputs of the reference's intersectTop on the 4096 rays
    of tests/golden/refgpu_c1.npz, closest hit (sbtRecordOffset 1) and any hit (2: the program's any-hit callback ends the walk
    at the first accepted candidate, which only decides the hit flag)."""
    rd, scenes = mods
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import golden_cases as gc
    import oracle_bind as ob
    G = np.load(os.path.join(root, "tests", "golden", "refgpu_c1.npz"))
    s = gc.small_scene(scenes, "c1")
    text = open(os.path.join(root, "tests", "golden", "user_trace.cl")).read()
    rd.SetShaderIncludePath("")                       # nothing but the library's own directory
    dev = scenes.DeviceScene(s, shader_text=text)
    o, d = np.ascontiguousarray(G["ray_o"], np.float32), np.ascontiguousarray(G["ray_d"], np.float32)
    n = o.shape[0]
    rays = np.concatenate([o, d], 1).reshape(-1).astype(np.float32)
    plt = dev.plt
    bRays = rd.CreateBuffer(plt, rays.nbytes); rd.WriteBuffer(plt, bRays, rays.nbytes, rays)
    bOut = rd.CreateBuffer(plt, n * 28 * 4)
    ref_h = np.ascontiguousarray(G["hits"]).view(ob.HIT_DTYPE).reshape(-1)
    for rec in (1, 2):
        prop = np.zeros((), rd.RayTraceProperties)
        prop["batchSize"], prop["depth"] = n, rec
        rd.WriteBuffer(plt, dev.rdRTProp, 16, np.array(prop))
        rd.WriteBuffer(plt, bOut, n * 28 * 4, np.zeros(n * 28, np.uint32))
        rd.BindDescriptorSet(plt, rd.CreateDescriptorSet([dev.rdRTProp, bOut, dev.rdImage, dev.rdCamData, dev.rdSceneData, dev.meshInfoData, bRays,
                                                           dev.indexData, dev.uvData, dev.normalData, dev.materialData, None, None, dev.topAccelStruct]))
        rd.TraceRays(plt, 0, 0, 0, n, 1)
        got = rd.ReadBuffer(plt, bOut, n * 28 * 4).view(ob.HIT_DTYPE).reshape(-1)
        if rec == 1:
            assert np.array_equal(ref_h["hit"], got["hit"])
            h = ref_h["hit"] == 1
            assert h.sum() > 500
            for f in ("distance", "primitiveIndex", "instanceIndex", "instanceCustomIndex", "instanceSBTOffset", "barycentric", "hitPoint", "transform"):
                assert np.array_equal(_bits(ref_h[f][h]), _bits(got[f][h])), f
        else:
            assert np.array_equal(got["hit"].astype(np.uint8), G["shadow_hit"])


def test_gpu_assisted_builder_blobs_identical(mods):
    """SURVEY 8(f) rank 1, second half: candidate evaluation of the BVH builder on the GPU (the binning pass of large nodes,
    csrc/kernels.hip k_bvh_bin behind bvh_build.h GpuBinner; reference: the candidate loop of radiance/src/bvh.cpp:90-205).
    With the threshold lowered to 64 primitives almost every inner node's candidates are evaluated on the device; the blob must
    equal, byte for byte, the one the host path builds (which test_cpu_oracle.py holds to the reference algorithm's literal
    restatement) -- meshes of three kinds, an instanced top level on top."""
    rd, scenes = mods
    plt = rd.Platform.GetPlatform()
    s = scenes.c2_atrium(64, 36, 1, 1, detail=0.6)
    meshes = sorted(s.meshes, key=lambda m: -len(m[1]))[:4] + [scenes.c1_cornell(64, 36, 1, 1, sphere_subdiv=4).meshes[-1]]
    try:
        rd.SetOption("gpu_build", 0)
        host = [rd.BuildAccelStruct(plt, rd.Mesh(m[0], m[1])).data for m in meshes]
        rd.SetOption("gpu_build", 1); rd.SetOption("gpu_build_min", 64)
        import ctypes
        from radiance_ray_tracing_amd import _lib
        calls = _lib.lib().rdx_debug_gpu_bin_calls
        calls.restype = ctypes.c_ulonglong
        before = calls()
        dev = [rd.BuildAccelStruct(plt, rd.Mesh(m[0], m[1])).data for m in meshes]
        assert calls() - before > 300, "the device did not bin the nodes (%d calls)" % (calls() - before)
        many = [b.data for b in rd.BuildAccelStructs(plt, [rd.Mesh(m[0], m[1]) for m in meshes])]      # builder threads share the device
    finally:
        rd.SetOption("gpu_build", 1); rd.SetOption("gpu_build_min", 32768)
    for k, (a, b, c) in enumerate(zip(host, dev, many)):
        assert len(a) > 1000 and bytes(a) == bytes(b) == bytes(c), "mesh %d (%d triangles): GPU-assisted build differs" % (k, len(meshes[k][1]))


def test_axis_parallel_rays_in_box_planes_match_the_reference_order_walk(mods):
    """The quad-record walk skips a level of the reference's tree on the strength of "a grandchild's box lies inside the child's and
    the slab decision is monotone under inclusion" -- which NaNs strain: a ray with a zero direction component whose origin lies
    exactly in a face plane of a box (0 / 0 in the reference's division form).  Such rays take the walk's exact path (the skipped
    node's box is tested too; traverse_pool.h quad_half explains why the results would agree even without it).  Scene: identity instances of axis-aligned boxes and grids, so that BVH boxes share planes with
    vertices; rays: origins with one, two or three coordinates equal to vertex coordinates, directions along the axes and in the
    coordinate planes.  Every engine must return the reference-order walk's HitData, closest hit and any hit."""
    rd, scenes = mods
    rng = np.random.default_rng(77)
    s = scenes.Scene("planes")
    s.materials = [scenes.material((0.7, 0.7, 0.7))]
    I = np.eye(4, dtype=np.float32)
    verts = []
    for k in range(6):
        lo = np.round(rng.uniform(-3, 0, 3) * 4) / 4; hi = lo + np.round(rng.uniform(0.5, 3, 3) * 4) / 4
        m = scenes.box(lo.tolist(), hi.tolist()); s.add_instance(s.add_mesh(m), I, 0); verts.append(np.asarray(m[0], np.float32).reshape(-1, 3))
    m = scenes.heightfield([-3, -1, -3], [0.25, 0, 0], [0, 0, 0.25], [0, 1, 0], 24, 24, 0.0, 1)       # a flat 24 x 24 grid: 1152 triangles in one plane
    s.add_instance(s.add_mesh(m), I, 0); verts.append(np.asarray(m[0], np.float32).reshape(-1, 3))
    m = scenes.icosphere(3, 1.0); s.add_instance(s.add_mesh(m), I, 0); verts.append(np.asarray(m[0], np.float32).reshape(-1, 3))
    s.camera = scenes.blender_camera(64, 48, 0.05, 0.036, 9.0, 0.0, (0.5, 14.0, 1.0), (-96.0, 180.0, 0.0))
    s.sceneProps = scenes.blender_dir_light(-45.0, 20.0, 5.0)
    s.rtprop = scenes._rtprop(0, 1, 2)
    V = np.concatenate(verts)
    n = 20000
    o = V[rng.integers(0, len(V), n)].copy()
    keep = rng.integers(1, 8, n)                     # which coordinates stay exactly on a vertex coordinate (bit mask, >= 1)
    for a in range(3):
        move = (keep >> a) & 1 == 0
        o[move, a] += rng.uniform(-4, 4, int(move.sum())).astype(np.float32)
    d = np.zeros((n, 3), np.float32)
    kind = rng.integers(0, 3, n)
    ax = rng.integers(0, 3, n); sg = rng.choice([-1.0, 1.0], n).astype(np.float32)
    d[np.arange(n), ax] = sg                          # along an axis
    two = kind == 1                                   # in a coordinate plane: one component exactly zero
    d[two] = rng.normal(size=(int(two.sum()), 3)).astype(np.float32); d[two, ax[two]] = 0.0
    neg0 = kind == 2                                  # along an axis with the zeros signed
    d[neg0] = d[neg0] + np.float32(-0.0)
    d[neg0 & (rng.random(n) < 0.5)] *= np.float32(1.0)
    nz = np.linalg.norm(d, axis=1) > 0
    o, d = np.ascontiguousarray(o[nz]), np.ascontiguousarray(d[nz] / np.linalg.norm(d[nz], axis=1, keepdims=True))
    dev = scenes.DeviceScene(s)
    fields = ("distance", "primitiveIndex", "instanceIndex", "barycentric", "hitPoint")
    try:
        for rec in (1, 2):
            ref = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec, reference_order=True)
            h = ref["hit"] == 1
            assert 0.05 < h.mean() < 0.95
            for kernel, quad in ((3, 1), (3, 0), (2, 1), (1, 1)):
                rd.SetOption("kernel", kernel); rd.SetOption("quad", quad)
                got = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
                assert np.array_equal(ref["hit"], got["hit"]), (rec, kernel, quad, int((ref["hit"] != got["hit"]).sum()))
                if rec == 1:
                    for f in fields:
                        assert np.array_equal(ref[f][h].view(np.uint8), got[f][h].view(np.uint8)), (kernel, quad, f)
    finally:
        rd.SetOption("kernel", 3); rd.SetOption("quad", 1)

"""GPU suite, part 2 (-m gpu): the HIP product against the REAL reference device code.

Two comparands, both the reference's own samples/shader.cl + radiance/shader/*.cl compiled for gfx950 by ROCm clang
against ROCm's OpenCL builtin library (oracle/Makefile -> oracle/_ref/ref_shader_gfx950_p.co, flags `-ffp-contract=off
-cl-fp32-correctly-rounded-divide-sqrt` = the floating-point contract of DESIGN.md section 2):

  * committed outputs of that code object (tests/golden/refgpu_*.npz, made by tests/golden/make_golden_gpu.py) --
    always available;
  * the code object itself, launched here next to the product (tests/refgpu_bind.py) -- it is built in the build
    container, where /root/reference exists, and travels to the GPU box like the product's own librdx.so; tests that need
    it skip when it is absent.

Bar: BIT-EXACT everywhere -- HitData, primary rays, `material` payloads, whole imageScratch and RGBA8 frames, at the small
golden sizes and at BASELINE's full sizes (configs 1-4)."""
import os

import numpy as np
import pytest

import golden_cases as gc
import oracle_bind as ob
import refgpu_bind as rg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ("distance", "primitiveIndex", "instanceIndex", "instanceCustomIndex", "instanceSBTOffset", "barycentric",
          "hitPoint", "transform")


@pytest.fixture(scope="module")
def mods(gpu):
    import rrt_amd  # noqa: F401
    from radiance_ray_tracing_amd import rd, scenes
    return rd, scenes


@pytest.fixture(scope="module")
def ref(gpu):
    if not rg.available("p"):
        pytest.skip("oracle/_ref/ref_shader_gfx950_p.co is not built (needs /root/reference: `make -C oracle`)")
    return rg.RefGpu("p")


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint8).reshape(a.shape[0], -1) if a.ndim else a.view(np.uint8)


def _same_hits(ref_h, got, closest=True, tag=""):
    assert np.array_equal(ref_h["hit"], got["hit"]), tag
    if closest:
        h = ref_h["hit"] == 1
        for f in FIELDS:
            assert np.array_equal(_bits(ref_h[f][h]), _bits(got[f][h])), (tag, f)


def _ambient(scene, hits):
    albedo = np.array(scene.materials)["albedo"][hits["instanceCustomIndex"], :3]
    return (albedo * np.float32(0.1)).astype(np.float32)


def _check_material(scene, hits, ref_pay, got):
    """the product's seam reports the light-visible colour (its shadow query is a separate stage); the reference's colour
    includes the shadow test -> it is bit-equal to the product's, or exactly the ambient term when the light is occluded"""
    k = hits["hit"] == 1
    assert np.array_equal(ref_pay["hit"][k], got["hit"][k])
    for f in ("nextFactor", "nextRayOrigin", "nextRayDirection"):
        assert np.array_equal(_bits(ref_pay[f][k]), _bits(got[f][k])), f
    lit = (_bits(ref_pay["color"][k]) == _bits(got["color"][k])).all(1)
    occluded = (_bits(ref_pay["color"][k]) == _bits(_ambient(scene, hits[k]))).all(1)
    assert np.all(lit | occluded)
    return int(lit.sum()), int(occluded.sum())


# ---------------------------------------------------------------------------------------------------------------------
# committed reference outputs
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["c0", "c1", "c2"])
def test_product_matches_reference_goldens(mods, name):
    rd, scenes = mods
    G = np.load(os.path.join(GOLD, "refgpu_%s.npz" % name))
    s = gc.small_scene(scenes, name)
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    assert np.array_equal(gc.sha(blob), G["blob_sha256"])
    # traversal: all four kernels, closest hit and any hit
    ref_h = np.ascontiguousarray(G["hits"]).view(ob.HIT_DTYPE).reshape(-1)
    o, d = G["ray_o"], G["ray_d"]
    _same_hits(ref_h, rd.TraceBatch(dev.topAccelStruct, o, d, reference_order=True), tag="reference-order kernel")
    for kernel, cull in ((3, 1), (3, 0), (2, 0), (1, 0)):      # pool engine with the culled and the exhaustive walk, ...
        rd.SetOption("kernel", kernel); rd.SetOption("cull", cull)
        try:
            _same_hits(ref_h, rd.TraceBatch(dev.topAccelStruct, o, d), tag="kernel %d cull %d" % (kernel, cull))
            sh = rd.TraceBatch(dev.topAccelStruct, o, d, sbtRecordOffset=2)
        finally:
            rd.SetOption("kernel", 3); rd.SetOption("cull", -1)
        assert np.array_equal(sh["hit"].astype(np.uint8), G["shadow_hit"]), (kernel, cull)
    assert (ref_h["hit"] == 1).sum() > 500
    # primary rays: generateRay of every pixel (pinhole; thin lens for c1)
    npix = s.width * s.height
    px = np.arange(npix, dtype=np.uint32)
    go, gd = rd.GenerateBatch(px, G["gen_rnd"])
    assert np.array_equal(_bits(go), _bits(G["gen_o"])) and np.array_equal(_bits(gd), _bits(G["gen_d"]))
    # `material` payloads on captured hits
    hits = np.ascontiguousarray(G["mat_hits"]).view(ob.HIT_DTYPE).reshape(-1)
    ref_pay = np.ascontiguousarray(G["mat_payload"]).view(ob.PAYLOAD_DTYPE).reshape(-1)
    n = hits.shape[0]
    frames, depths = gc.material_inputs(n)
    got = rd.MaterialBatch(hits, G["mat_dir"], np.arange(n, dtype=np.uint32), frames, depths)
    lit, occ = _check_material(s, hits, ref_pay, got)
    assert lit > 100
    # two progressive frames: imageScratch and RGBA8 bit for bit, with the culled and with the exhaustive walk (the latter
    # also with the per-bounce ray sort, which only changes the order rays are handed to the traversal launches)
    for cull in (1, 0):
        rd.SetOption("cull", cull); rd.SetOption("sort", 1 - cull)
        try:
            dev.set_rtprop(totalSamples=0); dev.clear_scratch()
            for f in range(2):
                img = dev.render()
                assert np.array_equal(_bits(dev.read_scratch().reshape(-1)), _bits(G["scratch%d" % f])), (cull, f)
                assert np.array_equal(img.reshape(-1), G["image%d" % f]), (cull, f)
        finally:
            rd.SetOption("cull", -1); rd.SetOption("sort", -1)
    if name == "c1":
        dev2 = scenes.DeviceScene(gc.small_scene(scenes, name, fstop=2.8))
        lo, ld = rd.GenerateBatch(px, G["gen_rnd"])
        assert np.array_equal(_bits(lo), _bits(G["lens_o"])) and np.array_equal(_bits(ld), _bits(G["lens_d"]))
        del dev2


# ---------------------------------------------------------------------------------------------------------------------
# the live reference code object
# ---------------------------------------------------------------------------------------------------------------------
def _frames_identical(rd, dev, rs, frames=1, rows=None):
    """render `frames` TraceRays calls on both sides; compare imageScratch + RGBA8 of the first `rows` image rows (all if
    None: the reference kernel is launched for exactly those pixels)"""
    w, h = dev.width, dev.height
    n = w * (rows or h)
    out = []
    for f in range(frames):
        ms = rs.ref.launch("k_ref_raygen", [rs.rtprop, rs.scratch, rs.image, rs.cam, rs.props, rs.meshInfo, rs.vertex, rs.index,
                                            rs.uv, rs.normal, rs.material, rs.tlas, np.uint32(n)], n)
        rs.set_rtprop(totalSamples=int(rs.rtprop_host[0]["totalSamples"]) + int(rs.rtprop_host[0]["batchSize"]))
        img = dev.render()
        st = rd.GetTraceStats()
        a = rs.read_scratch()[: n * 4]
        b = dev.read_scratch().reshape(-1)[: n * 4]
        same = (a.view(np.uint32).reshape(-1, 4) == b.view(np.uint32).reshape(-1, 4)).all(1)
        assert same.all(), "frame %d: %d of %d pixels differ (max |diff| %g)" % (f, int((~same).sum()), n, float(np.abs(a - b).max()))
        assert np.array_equal(rs.read_image()[: n * 4], img.reshape(-1)[: n * 4])
        out.append((ms, st.ms_total))
    return out


@pytest.mark.parametrize("cfg,cull", [("c1_cornell", 1), ("c1_cornell", 0), ("c2_atrium", 1), ("c2_atrium", 0)])
def test_full_size_frames_identical_to_reference(mods, ref, cfg, cull):
    """BASELINE configs 1 and 2 at full size (1920x1080, 4 spp, depth 8): every pixel of imageScratch and of the RGBA8 image
    equals the reference megakernel's, two progressive TraceRays calls (8.3 M paths, ~66 M rays each); culled and
    exhaustive walk"""
    rd, scenes = mods
    s = scenes.CONFIGS[cfg]()
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    rs = rg.RefScene(ref, s, blob)
    rd.SetOption("cull", cull); rd.SetOption("sort", cull)       # (sorted hand-out together with the culled walk)
    try:
        t = _frames_identical(rd, dev, rs, frames=2)
    finally:
        rd.SetOption("cull", -1); rd.SetOption("sort", -1)
    print("%s: reference kernel %.0f ms, product %.1f ms per frame" % (cfg, t[0][0], t[0][1]))


def user_stage_program():
    """tests/golden/user_stages.cl with its closest-hit body spliced in (what the C preprocessor would do with the #include)"""
    text = open(os.path.join(GOLD, "user_stages.cl")).read()
    return text.replace('#include "user_material.inc"', open(os.path.join(GOLD, "user_material.inc")).read()).replace('#include "user_environment.inc"', open(os.path.join(GOLD, "user_environment.inc")).read())


@pytest.mark.parametrize("cfg,kw", [("c1_cornell", dict(width=480, height=270, spp=3, depth=6, sphere_subdiv=3)),
                                    ("c2_atrium", dict(width=480, height=270, spp=2, depth=8, detail=0.5))])
def test_user_closest_hit_shader_on_the_wavefront_pipeline(mods, cfg, kw):
    """A user's own closest-hit shader (tests/golden/user_material.inc: a shadow query in the middle of the function, the
    per-pixel RNG, next ray, throughput) and miss shader (user_environment.inc: a sky that depends on the ray) run on the
    wavefront pipeline -- compiled at run time, recorded / replayed around the
    pipeline's shadow-walk stage (csrc/user_shader.cpp "stage mode") -- and the frame equals, bit for bit, the REFERENCE
    program's megakernel with that same function body in the place of its `material` (oracle/patch_material.py ->
    oracle/_ref/ref_shader_gfx950_um.co): imageScratch and RGBA8, three progressive TraceRays calls."""
    rd, scenes = mods
    if not rg.available("um"):
        pytest.skip("oracle/_ref/ref_shader_gfx950_um.co is not built (needs /root/reference: `make -C oracle`)")
    refum = rg.RefGpu("um")
    s = getattr(scenes, cfg)(**kw)
    rd.SetShaderIncludePath("")
    rd.SetOption("user_stages", 2)
    try:
        dev = scenes.DeviceScene(s, shader_text=user_stage_program())
    finally:
        rd.SetOption("user_stages", 1)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    rs = rg.RefScene(refum, s, blob)
    t = _frames_identical(rd, dev, rs, frames=3)
    st = rd.GetTraceStats()
    assert st.launches_extend >= kw["depth"] and st.launches_shadow == st.launches_extend, "the program did not run on the wavefront pipeline"
    print("%s with a user closest-hit shader: reference megakernel %.0f ms, product stage mode %.1f ms per frame" % (cfg, t[0][0], t[0][1]))


def test_user_stage_functions_the_pipeline_cannot_serve_are_refused(mods):
    """Stage mode serves ONE nested traceRay per closest-hit shader, and only the stock shadow query (row 2, miss 4, 0.001 / 1000):
    a second query, or another kind of query, raises a status bit in the stage kernel and TraceRays fails naming the option that
    runs the program as a megakernel -- no wrong frame is ever returned.  A program whose stage function uses get_global_id()
    outside the scope of `sceneData` does not compile as a stage kernel; asserted eligibility then fails at module creation."""
    rd, scenes = mods
    s = scenes.c1_cornell(96, 54, spp=1, depth=3, sphere_subdiv=2)
    base = user_stage_program()
    twice = base.replace("const float3 tint =", "traceRay(sceneData->topLevel, 2, 4, O, L, 0.001f, 1000, &query, sceneData, imageArray, sampler);\n    const float3 tint =")
    other = base.replace("traceRay(sceneData->topLevel, 2, 4, O, L, 0.001f, 1000, &query", "traceRay(sceneData->topLevel, 2, 4, O, L, 0.01f, 1000, &query")
    assert twice != base and other != base
    rd.SetShaderIncludePath("")
    rd.SetOption("user_stages", 2)
    try:
        for text, what in ((twice, "more than once"), (other, "not the stock shadow query")):
            dev = scenes.DeviceScene(s, shader_text=text)
            with pytest.raises(rd.RadianceError, match=what):
                rd.TraceRays(dev.plt, 0, 0, 0, 96, 54)
        helper = base.replace("void material(", "uint my_pixel(void) { return (uint)get_global_id(0); }\nvoid material(")
        with pytest.raises(rd.RadianceError, match="compilation failed"):
            scenes.DeviceScene(s, shader_text=helper)
    finally:
        rd.SetOption("user_stages", 1)
    # the stock pipeline is unaffected by the failed frames
    dev2 = scenes.DeviceScene(s)
    dev2.render()
    assert rd.GetTraceStats().rays_primary == 96 * 54


def test_many_instance_scene_identical_to_reference(mods, ref):
    """The Sponza-class scene as a one-instance-per-mesh loader delivers it (tools/sceneBuilder.cpp:287-315): 400 instances,
    a top level of ~200 nodes -- too large for the flat top-level step, so the pool engine walks top level, instances and BLASes
    as ONE tree (rdx_runtime.cpp "unified tree").  HitData of 64 k primary + scattered rays (closest and any hit) and a 640x360
    x 2 spp x depth 8 frame against the reference's own kernels, with the unified tree and with the walked top level."""
    rd, scenes = mods
    s = scenes.CONFIGS["c2_atrium_400"](640, 360, 2, 8)
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    rs = rg.RefScene(ref, s, blob)
    rng = np.random.default_rng(5)
    n = 1 << 15
    px = rng.choice(640 * 360, n, replace=False).astype(np.uint32)
    po, pd = rd.GenerateBatch(px, rng.integers(0, 2 ** 32, size=(n, 3), dtype=np.uint64).astype(np.uint32))
    first = rs.trace(po, pd)
    hp = (po + pd * first["distance"][:, None]).astype(np.float32)[first["hit"] == 1]
    d2 = rng.normal(size=hp.shape).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o = np.ascontiguousarray(np.concatenate([po, hp]), np.float32); d = np.ascontiguousarray(np.concatenate([pd, d2]), np.float32)
    want1, want2 = rs.trace(o, d), rs.trace(o, d, sbtRecordOffset=2)
    assert (want1["hit"] == 1).sum() > 20000
    try:
        for unified in (1, 0):
            rd.SetOption("unified_tree", unified)
            for cull in (0, 1):
                rd.SetOption("cull", cull)
                _same_hits(want1, rd.TraceBatch(dev.topAccelStruct, o, d), tag="unified %d cull %d" % (unified, cull))
                _same_hits(want2, rd.TraceBatch(dev.topAccelStruct, o, d, sbtRecordOffset=2), closest=False, tag="any hit, unified %d cull %d" % (unified, cull))
            rd.SetOption("cull", -1)
            if unified:
                _frames_identical(rd, dev, rs, frames=1)
    finally:
        rd.SetOption("unified_tree", 1); rd.SetOption("cull", -1)


def rmse_vs_default_build(rd, scenes, cfg, w=160, h=90, spp=4, depth=8):
    """(RMSE of the product's imageScratch against build d, RMSE of build p against build d, product == build p) on one frame"""
    s = scenes.CONFIGS[cfg](w, h, spp, depth)
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    dev.set_rtprop(totalSamples=0); dev.clear_scratch(); dev.render()
    got = dev.read_scratch().reshape(-1).astype(np.float64)
    out = {}
    for build in ("p", "d"):
        rs = rg.RefScene(rg.RefGpu(build), s, blob)
        rs.frame()
        out[build] = rs.read_scratch().astype(np.float64)
    rmse = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
    return rmse(got, out["d"]), rmse(out["p"], out["d"]), bool(np.array_equal(got, out["p"]))


@pytest.mark.parametrize("cfg", ["c1_cornell", "c2_atrium"])
def test_distance_to_the_default_opencl_build(mods, ref, cfg):
    """The parity contract is build p of the reference's shader (-ffp-contract=off, correctly rounded divide / sqrt): the product
    is bit-identical to it.  What the reference's own clBuildProgram("-g -I...") (radiance.cpp:165-167) would run is build d --
    clang's OpenCL defaults: fused multiply-adds, 2.5-ulp divide.  This test states the distance: the product differs from build
    d by EXACTLY what build p differs from build d (glass / mirror paths flip under last-bit changes of a direction, so two builds
    of the same source are apart by more than the north star's 1e-4 on some frames), i.e. "radiance RMSE < 1e-4 against the
    reference" is met -- with RMSE 0 -- under the pinned contract, and is not a meaningful bar between two contracts."""
    if not rg.available("d"):
        pytest.skip("oracle/_ref/ref_shader_gfx950_d.co is not built")
    rd, scenes = mods
    r_prod, r_pd, same = rmse_vs_default_build(rd, scenes, cfg)
    print("%s 160x90 x 4 spp x depth 8: RMSE product vs build d %.3g, build p vs build d %.3g, product == build p: %s" % (cfg, r_prod, r_pd, same))
    assert same and r_prod == r_pd
    assert r_prod < 5e-2          # sanity: the two contracts render the same picture up to a few flipped paths


def test_c3_4k_64spp_identical_to_reference(mods, ref):
    """BASELINE config 3 on one GPU: the Sponza-class scene at 3840x2160 with 64 spp (531 M paths in 34 sample chunks).  The
    product renders the whole frame; the reference megakernel renders the first 36 image rows (138 k pixels x 64 spp =
    8.8 M paths, what a whole 1080p x 4 spp frame costs it) and those pixels must be bit-identical -- the sample-chunk
    boundaries and the running mean across 64 samples included"""
    rd, scenes = mods
    s = scenes.c2_atrium(3840, 2160, spp=64, depth=8)
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    rs = rg.RefScene(ref, s, blob)
    _frames_identical(rd, dev, rs, frames=1, rows=36)
    st = rd.GetTraceStats()
    assert st.rays_primary == 3840 * 2160 * 64 and st.rays_shadow == st.closest_hits


@pytest.fixture(scope="module")
def c4(mods):
    rd, scenes = mods
    s = scenes.c4_atrium_10m(3840, 2160, spp=4, depth=8)
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    return s, dev, blob


def test_c4_traversal_identical_to_reference_and_oracle(mods, ref, c4):
    """BASELINE config 4 (10.4 M triangles + 64 instances of a shared foliage BLAS, 446 MB blob): 24 k primary and
    scattered rays -- closest hit and any hit -- through the product's pool kernel, the reference's intersectTop on the
    GPU and the CPU oracle walking the same product-built blob: HitData bit-identical three ways"""
    rd, scenes = mods
    s, dev, blob = c4
    assert s.triangle_count() > 10_000_000 and len(s.instances) == 89
    rs = rg.RefScene(ref, s, blob)
    npix = s.width * s.height
    sel = gc.spread(npix, 6000)
    # primary rays of the selected pixels from the product's own generator (bit-identical to the reference's, see above)
    po, pd = rd.GenerateBatch(sel.astype(np.uint32), gc.generate_inputs(sel.shape[0], 9))
    ph = rd.TraceBatch(dev.topAccelStruct, po, pd)
    o, d = gc.derived_rays(17, po, pd, ph["hit"], ph["distance"])
    for rec in (1, 2):
        r = rs.trace(o, d, 0.001, 1000.0, rec)
        c = ob.trace_batch(blob, o, d, 0.001, 1000.0, rec)
        _same_hits(r, c, closest=True, tag="oracle rec %d" % rec)          # the oracle keeps the reference's DFS order
        for kernel in (3, 0):
            rd.SetOption("kernel", kernel)
            try:
                g = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
            finally:
                rd.SetOption("kernel", 3)
            _same_hits(r, g, closest=(rec == 1), tag="kernel %d rec %d" % (kernel, rec))
    assert (r["hit"] == 1).sum() > 8000


def test_c4_progressive_frames_identical_to_reference(mods, ref, c4):
    """config 4 at 3840x2160, progressive: three TraceRays calls of 4 spp with the host-side totalSamples += 4 of
    samples/sample1.cpp:480-490 in between; the first 24 rows (92 k pixels) against the reference megakernel, bit for bit"""
    rd, scenes = mods
    s, dev, blob = c4
    rs = rg.RefScene(ref, s, blob)
    dev.set_rtprop(totalSamples=0); dev.clear_scratch()
    _frames_identical(rd, dev, rs, frames=3, rows=24)
    assert int(dev.rtprop["totalSamples"]) == 12


def test_random_scenes_identical_to_reference(mods, ref):
    """random meshes (incl. stacks of coincident triangles), up to 70 instances with random affine transforms, shear and
    shared BLASes, rays from everywhere incl. axis-aligned ones: the product's kernels == the reference's intersectTop"""
    import importlib.util
    rd, scenes = mods
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    total = 0
    for seed in range(700, 724):
        s, o, d, rng = fz.random_case(seed)
        dev = scenes.DeviceScene(s)
        blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
        tl = rg.DevBuf.of(np.frombuffer(blob, np.uint8))
        o, d = fz.with_surface_rays(rng, o, d, ref.trace(tl, o, d))
        for rec in (1, 2):
            r = ref.trace(tl, o, d, 0.001, 1000.0, rec)
            for kernel, cull in ((3, 1), (3, 0), (2, 0), (1, 0), (0, 0)):
                rd.SetOption("kernel", kernel); rd.SetOption("cull", cull)
                try:
                    g = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
                finally:
                    rd.SetOption("kernel", 3); rd.SetOption("cull", -1)
                _same_hits(r, g, closest=(rec == 1), tag="seed %d kernel %d cull %d rec %d" % (seed, kernel, cull, rec))
        total += int((r["hit"] == 1).sum())
    assert total > 2000


def test_textured_materials_render_like_the_live_reference(mods, ref):
    """materials WITH texture indices, default options: the live reference shader reads texel 0 for them (its read_imageui
    calls are commented out, shader.cl:379-445) and so does the product -- frames bit-identical to the reference kernel"""
    from test_gpu_parity import _textured_scene
    rd, scenes = mods
    s = _textured_scene(scenes, 160, 90)
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    rs = rg.RefScene(ref, s, blob)
    _frames_identical(rd, dev, rs, frames=2)


def test_instance_sbt_offset_like_the_reference(mods, ref):
    """stock table, instances with SBTOffset 1 (see test_gpu_parity.test_instance_sbt_offsets): HitData of closest-hit and
    any-hit batches -- the first-accepted-candidate rule of a terminating any-hit shader included -- bit-identical to the
    reference's own intersectTop.  (Frames are not compared: a shadow ray that hits such an instance dispatches row 3, which
    has no hit shader, and the reference then reads `shadowPayload.hit` uninitialised -- shader.cl:497-503 -- so its picture
    is undefined there; the product and the oracle define it as "not occluded".)"""
    rd, scenes = mods
    s = scenes.c1_cornell(160, 90, spp=2, depth=4, sphere_subdiv=3)
    s.sbt_offsets = {5: 1, 7: 1}
    dev = scenes.DeviceScene(s)
    blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    rs = rg.RefScene(ref, s, blob)
    npix = s.width * s.height
    sel = gc.spread(npix, 3000)
    po, pd = rd.GenerateBatch(sel.astype(np.uint32), gc.generate_inputs(sel.shape[0], 5))
    ph = rd.TraceBatch(dev.topAccelStruct, po, pd)
    o, d = gc.derived_rays(23, po, pd, ph["hit"], ph["distance"])
    for rec in (1, 2):
        r = rs.trace(o, d, 0.001, 1000.0, rec)
        g = rd.TraceBatch(dev.topAccelStruct, o, d, 0.001, 1000.0, rec)
        _same_hits(r, g, closest=True, tag="rec %d" % rec)
    assert (r["instanceSBTOffset"][r["hit"] == 1] == 1).sum() > 100

#!/usr/bin/env python3
"""user_stage_bench.py -- what a user's own closest-hit shader costs: the fixture program tests/golden/user_stages.cl at the
metric's frame (1920x1080, 4 spp, depth 8) on the product's wavefront pipeline (stage mode), next to the
reference program's megakernel with the same function body (oracle/_ref/ref_shader_gfx950_um.co, when built) and to the stock
pipeline with the compiled-in HIP stages.  GPU only.   python tools/user_stage_bench.py [c1_cornell|c2_atrium]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import rrt_amd  # noqa: F401
from radiance_ray_tracing_amd import rd, scenes
import refgpu_bind as rg

GOLD = os.path.join(ROOT, "tests", "golden")
text = open(os.path.join(GOLD, "user_stages.cl")).read().replace('#include "user_material.inc"', open(os.path.join(GOLD, "user_material.inc")).read()).replace('#include "user_environment.inc"', open(os.path.join(GOLD, "user_environment.inc")).read())
out = {}
for cfg in (sys.argv[1:] or ["c1_cornell", "c2_atrium"]):
    s = scenes.CONFIGS[cfg]()
    rd.SetShaderIncludePath("")
    rd.SetOption("user_stages", 2)
    dev = scenes.DeviceScene(s, shader_text=text)
    rd.SetOption("user_stages", 1)
    ms = []
    for f in range(6):
        dev.render()
        st = rd.GetTraceStats()
        ms.append(st.ms_total)
    rays = st.rays_primary + st.rays_bounce + st.rays_shadow
    r = {"stage_mode_ms": round(float(np.median(ms[1:])), 2), "paths": int(st.rays_primary)}
    if rg.available("um"):
        blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
        rs = rg.RefScene(rg.RefGpu("um"), s, blob)
        n = dev.width * dev.height
        t = [rs.ref.launch("k_ref_raygen", [rs.rtprop, rs.scratch, rs.image, rs.cam, rs.props, rs.meshInfo, rs.vertex, rs.index,
                                            rs.uv, rs.normal, rs.material, rs.tlas, np.uint32(n)], n) for _ in range(2)]
        r["reference_megakernel_ms"] = round(float(min(t)), 1)
    stock = scenes.DeviceScene(s)
    ms = []
    for f in range(6):
        stock.render(); ms.append(rd.GetTraceStats().ms_total)
    r["stock_pipeline_ms"] = round(float(np.median(ms[1:])), 2)
    out[cfg] = r
    print(cfg, json.dumps(r), flush=True)
print(json.dumps(out))

#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (primary + bounce + shadow) and ms/frame at 1920x1080,
4 spp, depth 8 (BASELINE.json).

A step = one RD::TraceRays frame (RayTraceProperties{totalSamples 0, batchSize 4, depth 8}) of the
workload, scene and accumulators already resident in HBM.  Default workload = the Sponza-class scene (BASELINE.json
configs[2], the configuration the north star's 1-GPU target is stated on); the sample1 scene (configs[1]) and the
10.4 M-triangle scene (configs[4] geometry) are always timed as well and reported under `also` (N = 1).
N > 1: one process per GPU (torch.distributed / RCCL), the frame is sharded by interleaved 64x64 image tiles, no
collective while rendering, one RGBA8 gather to rank 0 at frame end -- inside the timed region.  The headline at every N is
the metric's own frame (1920x1080, 4 spp, depth 8) split N ways: STRONG scaling, what the north star's ">= 6x at 8 GPUs" is
about.  The weak-scaling figure (N x 4 samples per pixel, every GPU keeps the rays of the N = 1 frame: BASELINE configs 3 / 4
are such frames) is timed as well and reported under `also.weak_scaling_n_x_spp`; --scaling weak makes it the headline.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline      dominant kernel (k_fused_pool = shadow rays of bounce d + closest-hit rays of bounce d+1 per launch).
                `achieved` = fabric bytes per launch (rocprofv3 PMC passes FETCH_SIZE / WRITE_SIZE of this same workload,
                collected by this run in child processes after the timed region; units as
                /opt/skills/guides/MI355X_MICROARCH.md prescribes, FETCH_SIZE at face value as calibrated for this kernel's
                record gathers by tools/fetch_calibrate.py) / the kernel's average launch duration (HIP events on
                the library's own stream inside the timed region), against the 8 TB/s HBM peak -- a fraction that cannot
                exceed 1.  The SURVEY 8(d) byte model (every visit of the REFERENCE's exhaustive walk charged as an
                uncached read) is kept beside it as `reference_walk_equiv_GBps`: a throughput normalisation without a
                peak, since the product neither makes those visits nor misses the caches on the ones it makes.
                `issue` = the instruction-issue side of the same launches (SQ counters): the kernel is issue-bound.
  cpu_baseline  the CPU oracle (a port of the reference algorithm) on a bounded pixel sample of the same workload, on
                this box's host cores
  reference_on_gpu   informational: the reference's OWN OpenCL megakernel (oracle/_ref, built from the reference's
                sources by ROCm clang) rendering the same frame on this same GPU, launched as the reference launches it
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
N_SIMD = 256 * 4             # 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles (same guide)

WORKLOADS = {
    # BASELINE.json configs[1]: "sample1.cpp scene, 1920x1080, 4 spp, depth 8, 1xMI355X"
    "sample1": ("c1_cornell", "sample1 scene (procedural Cornell stand-in, 20.5k tris, 8 instances)"),
    # BASELINE.json configs[2]: "Sponza via assimp (~260k tris) ..." -- no asset offline, procedural atrium
    "sponza": ("c2_atrium", "Sponza-class procedural atrium (262k tris, 25 instances)"),
    # BASELINE.json configs[4] geometry ("San-Miguel-scale ~10M tris"): BVH (446 MB) beyond L2 + Infinity Cache
    "sanmiguel": ("c4_atrium_10m", "San-Miguel-scale procedural atrium (10.4M tris, 89 instances incl. 64 of one shared foliage BLAS)"),
    # the Sponza-class scene as a one-instance-per-mesh loader delivers it (tools/sceneBuilder.cpp:287-315): 400 instances
    "sponza400": ("c2_atrium_400", "Sponza-class procedural atrium, one instance per mesh (262k tris, 400 instances)"),
}


def workload_label(key, a, spp=None):
    return "%s, %dx%d, %d spp, depth %d" % (WORKLOADS[key][1], a.width, a.height, a.spp if spp is None else spp, a.depth)


def algorithmic_bytes(top, inst, bot, tri, rays):
    """B = 16/ray + 48/top node + (80+16)/instance visit + 48/bottom node + 64/triangle,
    every visit of the REFERENCE algorithm's exhaustive walk charged as an uncached read."""
    return 16 * rays + 48 * top + 96 * inst + 48 * bot + 64 * tri


ENGINE = "pool"          # suffix of the traversal kernels' names: k_fused_pool (library default, kernel option 3) / k_fused_coop (2)


def traversal_roofline(acc, visits, steps, depth):
    """(kernel name, reference-walk bytes of its launches per frame, seconds of those launches over the timed region,
    launches) of the dominant traversal kernel"""
    if acc.get("ms_path", 0.0) > 0.0:
        b = sum(algorithmic_bytes(visits["visit_top_nodes"][c], visits["visit_instances"][c], visits["visit_bot_nodes"][c],
                                  visits["visit_triangles"][c], r)
                for c, r in ((0, (acc["primary"] + acc["bounce"]) / steps), (1, acc["shadow"] / steps)))
        return ("k_path_coop", b + 184 * acc["hits"] / steps, acc["ms_path"] * 1e-3, max(1, acc["launches_extend"]))
    if acc["ms_fused"] > 0.0:
        # shadow(d) and extend(d+1) share one launch (D-1 launches per frame): its reference-walk bytes are those of the
        # shadow rays of bounces 0..D-2 plus the closest-hit rays of bounces 1..D-1
        prof, cnt = visits["profile"], visits["counts"]
        D = prof.shape[0]

        def bytes_of(d, cls):
            v = prof[d, cls]
            return algorithmic_bytes(v[0], v[1], v[2], v[3], cnt[d] if cls == 0 else cnt[d + 1])
        return ("k_fused_" + ENGINE, sum(bytes_of(d, 1) + bytes_of(d + 1, 0) for d in range(D - 1)), acc["ms_fused"] * 1e-3,
                max(1, steps * (D - 1)))
    rays = (acc["primary"] + acc["bounce"]) / steps
    b = algorithmic_bytes(visits["visit_top_nodes"][0], visits["visit_instances"][0], visits["visit_bot_nodes"][0],
                          visits["visit_triangles"][0], rays)
    return ("k_extend_" + ENGINE, b, acc["ms_extend"] * 1e-3, max(1, acc["launches_extend"]))


# ---- rocprofv3 PMC passes of this same workload, run as child processes after the timed region ---------------------------
PMC_PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "issue": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES",
              "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE"],
}


def _rocprof():
    for c in (shutil.which("rocprofv3"), "/opt/rocm/bin/rocprofv3"):
        if c and os.path.exists(c):
            return c
    return None


PROFILER_ENV_MARKS = ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_", "ROCPROF_", "ROCP_", "HSA_TOOLS_LIB")


def under_profiler():
    """True when this process already runs under rocprofv3 (its tool library is preloaded and has initialised the GPU): a nested
    profiler pass -- a `#!/usr/bin/env python3` launcher that would exec its target from a GPU-initialised process -- must not
    be started from here (tools/gpu_profile.sh and friends profile `bench.py --no-pmc --no-reference` themselves)."""
    if any(k.startswith(PROFILER_ENV_MARKS) for k in os.environ):
        return True
    return "rocprof" in os.environ.get("LD_PRELOAD", "")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if not k.startswith(PROFILER_ENV_MARKS)}
    if "rocprof" in env.get("LD_PRELOAD", ""):
        env.pop("LD_PRELOAD")
    env["TMPDIR"] = "/tmp"
    return env


def pmc_pass(counters, child_args, timeout_s=300):
    """-> {kernel name: {"launches": n, counter: average per launch}} or None"""
    prof = _rocprof()
    if prof is None or under_profiler():
        return None
    tmp = tempfile.mkdtemp(prefix="rdx_pmc_", dir="/tmp")
    env = _clean_env()
    cmd = [prof, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "p", "--",
                                        "python3", os.path.join(ROOT, "bench.py"), "--pmc-child"] + child_args
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(files[0])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
        return {k: dict({c: v / max(1, len(seen[k])) for c, v in d.items()}, launches=len(seen[k])) for k, d in agg.items()}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_child(args):
    """the workload of a PMC pass: one warm-up and two frames, nothing printed"""
    import rrt_amd  # noqa: F401
    from radiance_ray_tracing_amd import rd, scenes
    plt = rd.Platform.GetPlatform(0)
    apply_options(rd, args)
    scene = scenes.CONFIGS[WORKLOADS[args.workload][0]](args.width, args.height, args.spp, args.depth)
    dev = scenes.DeviceScene(scene, plt)
    for _ in range(3):
        dev.set_rtprop(totalSamples=0)
        rd.TraceRays(plt, 0, 0, 0, args.width, args.height)


def apply_options(rd, args):
    global ENGINE
    rd.SetOption("fuse", args.fuse)
    if args.pipeline >= 0:
        rd.SetOption("pipeline", args.pipeline)
    if args.top_flat >= 0:
        rd.SetOption("top_flat", args.top_flat)
    if args.groups >= 1:
        rd.SetOption("groups", args.groups)
    if args.cull >= 0:
        rd.SetOption("cull", args.cull)
    if args.sort >= 0:
        rd.SetOption("sort", args.sort)
    for kv in (args.opt or "").split(","):          # experiments: any library option by name
        if kv:
            rd.SetOption(kv.split("=")[0], int(kv.split("=")[1]))
    if args.kernel >= 0:
        rd.SetOption("kernel", args.kernel)
        ENGINE = "pool" if args.kernel == 3 else "coop"


def option_args(args):
    out = ["--workload", args.workload, "--width", str(args.width), "--height", str(args.height), "--spp", str(args.spp),
           "--depth", str(args.depth), "--fuse", str(args.fuse), "--pipeline", str(args.pipeline), "--kernel", str(args.kernel),
           "--top-flat", str(args.top_flat), "--groups", str(args.groups), "--cull", str(args.cull), "--sort", str(args.sort)] + (["--opt", args.opt] if args.opt else [])
    return out


def measured_roofline(kernel_name, launches_s, launches, child_args):
    """fabric traffic and instruction issue of `kernel_name` from PMC passes of this workload; durations from the timed region"""
    full = "rdx::" + kernel_name
    res = {}
    found = None
    for name, ctrs in PMC_PASSES.items():
        r = pmc_pass(ctrs, child_args) or {}
        # the walk over quad records (library default where the culled walk is off) runs kernels named ..._q
        cands = [k for k in (full + "_q", full) if k in r]
        key = max(cands, key=lambda k: r[k].get("launches", 0)) if cands else None
        res[name] = r.get(key) if key else None
        found = found or key
    avg_s = launches_s / launches if launches else 0.0
    out = {"traffic": None, "achieved": None, "issue": None, "kernel_profiled": found}
    f, w = res.get("fetch"), res.get("write")
    if f and w and avg_s > 0:
        # guide: both counters are in KiB; on gfx950 FETCH_SIZE tallies wide coalesced reads (128-B requests) at 64 B, hence its
        # "x 2".  That correction is calibrated for streaming reads only; the guide says to calibrate other shapes on a known
        # byte count.  tools/fetch_calibrate.py did (profiles/r03_fetch_calibrate.log, librdx probe kernel, one record per lane
        # at a random index): streaming 16 B/lane reports 0.50 of the requested bytes (the guide's case), per-lane gathers of
        # aligned 64-B records report 1.00 of them (HBM- and Infinity-Cache-resident tables alike; an L2-resident table reports
        # ~0: the counter is L2-miss traffic), gathers of 48-B records report 80 B per record (the 32-B sectors they touch).
        # This kernel's reads are such gathers (64-B wide nodes, 48-B triangle records), so its traffic is the counter at face
        # value; the x 2 figure is kept as the upper bound it would be if every read were a wide coalesced one.
        rd_b = f["FETCH_SIZE"] * 1024.0
        wr_b = w["WRITE_SIZE"] * 1024.0
        out["traffic"] = int(rd_b + wr_b)
        out["achieved"] = (rd_b + wr_b) / avg_s / 1e9
        out["traffic_detail"] = {"FETCH_SIZE_KiB_per_launch_raw": round(f["FETCH_SIZE"], 1), "WRITE_SIZE_KiB_per_launch": round(w["WRITE_SIZE"], 1),
                                 "read_bytes": int(rd_b), "write_bytes": int(wr_b),
                                 "read_bytes_if_all_reads_were_wide_coalesced_x2": int(2 * rd_b),
                                 "fetch_size_calibration": "per-lane 64-B record gathers: counter = requested bytes x 1.00; 48-B records: 80 B each; "
                                                           "streaming 16 B/lane: x 0.50 (profiles/r03_fetch_calibrate.log)",
                                 "launches_profiled": int(f["launches"]),
                                 "source": "rocprofv3 --pmc passes run by this bench.py invocation (child processes, after the timed region)"}
    q = res.get("issue")
    if q and avg_s > 0:
        gui = q.get("GRBM_GUI_ACTIVE", 0.0)
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3; the PMC pass serialises dispatches, so its cycle count is
        # that pass's, and only instruction COUNTS are carried over to the timed region's durations
        clk_ghz = None
        valu, salu = q.get("SQ_INSTS_VALU", 0.0), q.get("SQ_INSTS_SALU", 0.0)
        # issue roofline with the timed region's duration and the clock the PMC pass observed per XCD
        pass_cycles = gui / 8.0 if gui else 0.0
        out["issue"] = {
            "valu_wave_insts_per_launch": int(valu), "salu_wave_insts_per_launch": int(salu),
            "lds_wave_insts_per_launch": int(q.get("SQ_INSTS_LDS", 0.0)), "vmem_rd_wave_insts_per_launch": int(q.get("SQ_INSTS_VMEM_RD", 0.0)),
            "active_lanes_per_valu_inst": round(q.get("SQ_THREAD_CYCLES_VALU", 0.0) / valu, 2) if valu else None,
            "wave_cycles_parked_frac": round(q.get("SQ_WAIT_ANY", 0.0) / q["SQ_WAVE_CYCLES"], 3) if q.get("SQ_WAVE_CYCLES") else None,
            "gui_active_cycles_per_launch_in_pmc_pass": int(pass_cycles),
            # VALU issue slots used / available: a wave64 VALU instruction takes 2 cycles of its SIMD-32; 1024 SIMDs.
            # Cycles = duration in the TIMED region x 2.4 GHz (the clock is not observable without PMC; an upper bound on the
            # cycles, so a lower bound on the fraction) and, beside it, the same with the PMC pass's own cycle count
            "valu_issue_frac_at_2p4GHz": round(valu * 2.0 / (N_SIMD * avg_s * 2.4e9), 4),
            "valu_issue_frac_pmc_pass_cycles": round(valu * 2.0 / (N_SIMD * pass_cycles), 4) if pass_cycles else None,
            "salu_per_valu": round(salu / valu, 3) if valu else None,
            "launches_profiled": int(q["launches"]),
        }
    return out


def reference_on_gpu(scene, tlas_blob, product_ms):
    """the reference's own OpenCL megakernel on this GPU (informational): one frame, launched with local_work_size 1 as
    radiance/src/radiance.cpp:250-259 does, and with 64"""
    try:
        import refgpu_bind as rg
        if not rg.available("d"):
            return None
        ref = rg.RefGpu("d")
        out = {"build": "reference samples/shader.cl + radiance/shader/*.cl, ROCm clang OpenCL C for gfx950, clang's OpenCL defaults "
                        "(what clBuildProgram(\"-g -I...\") would get), ROCm OpenCL builtin library"}
        for local in (64, 1):
            rs = rg.RefScene(ref, scene, tlas_blob)
            rs.frame(local)                                   # warm-up (code object load, caches)
            rs.set_rtprop(totalSamples=0)
            ms = rs.frame(local)
            out["ms_per_frame_local%d" % local] = round(ms, 2)
        # the two floating-point contracts of the same reference source, on a small frame of this scene: the product is bit-identical
        # to build p (the parity contract); build d is what clBuildProgram("-g -I...") would run (tests/test_gpu_reference.py)
        try:
            import test_gpu_reference as tgr
            from radiance_ray_tracing_amd import rd as _rd, scenes as _sc
            r_prod, r_pd, same = tgr.rmse_vs_default_build(_rd, _sc, scene.name)
            out["radiance_rmse_160x90"] = {"product_vs_build_d": float("%.3g" % r_prod), "build_p_vs_build_d": float("%.3g" % r_pd),
                                           "product_bit_identical_to_build_p": same,
                                           "note": "contract p = -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt; RMSE < 1e-4 (0) holds under it"}
        except Exception as e:
            out["radiance_rmse_160x90"] = {"error": str(e)[:160]}
        out["product_speedup_vs_local1"] = round(out["ms_per_frame_local1"] / product_ms, 1)
        out["product_speedup_vs_local64"] = round(out["ms_per_frame_local64"] / product_ms, 1)
        return out
    except Exception as e:      # informational leg: never fails the bench
        return {"error": str(e)[:200]}


def cpu_baseline(scene, budget_s=20.0):
    """oracle (CPU port of the reference megakernel) on a strided pixel sample of the same frame"""
    import numpy as np
    import oracle_bind as ob
    osc = ob.OracleScene(scene)
    n = scene.width * scene.height
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    nproc = os.cpu_count() or cores
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cores = max(1, min(cores, 16))           # a 1-GPU box's CPU share is 16 cores, whatever nproc says (256 on the pool's hosts)
    rng = np.random.default_rng(0)
    probe = rng.choice(n, 4096, replace=False).astype(np.uint32)
    osc.render(nthreads=cores, pixels=probe[:256])
    t = time.time(); osc.render(nthreads=cores, pixels=probe); dt = max(time.time() - t, 1e-3)
    rate = 4096 / dt
    m = int(min(n, max(4096, rate * budget_s)))
    px = rng.choice(n, m, replace=False).astype(np.uint32)
    rays, dt, reps = 0, 0.0, 0
    while dt < 10.0 and reps < 64:
        osc.scratch[:] = 0
        t = time.time(); c = osc.render(nthreads=cores, counters=True, pixels=px); dt += time.time() - t
        d = c.as_dict()
        rays += d["rays"][0] + d["rays"][1]
        reps += 1
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "nproc": nproc, "cpu_model": model, "threads": cores,
            "sample": "%d random pixels of the same frame (%d spp, depth %d) x %d passes, %.1f s, %d reference-algorithm rays "
                      "(incl. the reference's duplicate re-trace after a primary miss)"
                      % (m, int(scene.rtprop["batchSize"]), int(scene.rtprop["depth"]), reps, dt, rays)}


STAGE_KEYS = ("ms_extend", "ms_shadow", "ms_shade", "ms_generate", "ms_accumulate", "ms_total", "ms_fused", "ms_path", "ms_sort")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="sponza", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="N > 1: strong (default) = the metric's own frame (1920x1080, --spp samples per pixel) is split N ways; weak = every "
                         "GPU keeps the rays of the N = 1 frame (the frame has N x --spp samples per pixel: BASELINE configs 3 / 4 are such frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 PMC child passes (roofline.traffic / achieved become null)")
    ap.add_argument("--no-reference", action="store_true", help="skip the informational run of the reference's own kernel")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--fuse", type=int, default=-1, help="-1 auto, 0/1: shadow(d)+extend(d+1) in one launch")
    ap.add_argument("--pipeline", type=int, default=-1, help="-1 library default, 0 staged wavefront, 1 whole paths in one persistent launch")
    ap.add_argument("--kernel", type=int, default=-1, help="-1 library default; 2 cooperative, 3 cooperative with a shared node pool, 1 / 0 per-lane")
    ap.add_argument("--top-flat", type=int, default=-1, help="-1 library default; 0/1: evaluate small top-level trees all at once (pool engine)")
    ap.add_argument("--groups", type=int, default=-1, help="-1 library default; 1..4 sample groups of a chunk on their own streams")
    ap.add_argument("--cull", type=int, default=-1, help="-1 library default (automatic); 0 = exhaustive walk, 1 = culled walk (pool engine)")
    ap.add_argument("--sort", type=int, default=-1, help="-1 library default (automatic); 0 / 1: per-bounce ray sort off / on")
    ap.add_argument("--opt", default=None, help="comma list name=value of library options (rdx_set_option) for experiments")
    ap.add_argument("--also", default=None, help="comma list of extra workloads to time (reported under 'also'); default at N=1: the other two")
    args = ap.parse_args()

    if args.pmc_child:          # (the parent has built everything; no compiler runs inside the profiled process)
        pmc_child(args)
        return

    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.build()
    import rrt_amd
    from radiance_ray_tracing_amd import dist as rdist, rd, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ray-tracing core has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("RDX_DIST_BACKEND", "nccl")      # "gloo": rehearsal with ranks sharing a GPU
    if local_rank >= ndev:
        if backend == "nccl":
            raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev))
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as tdist
        if backend == "nccl":
            tdist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            tdist.init_process_group(backend)
    plt = rd.Platform.GetPlatform(local_rank)
    apply_options(rd, args)

    # weak scaling: per-GPU work is that of the N = 1 frame -- the frame is rendered with N x spp samples per pixel and its 64x64
    # tiles are dealt to the ranks (what BASELINE configs 3 and 4 do: 64 / 256 spp over 8 GPUs); strong: the N = 1 frame split N ways
    spp_main = args.spp * world if (world > 1 and args.scaling == "weak") else args.spp

    def run_workload(key, steps, warmup, want_roofline, spp=None):
        cfg, _ = WORKLOADS[key]
        scene = scenes.CONFIGS[cfg](args.width, args.height, spp_main if spp is None else spp, args.depth)
        dev = scenes.DeviceScene(scene, plt)
        sharder = rdist.FrameSharder(rd, plt, args.width, args.height, rank, world, 64, 64, torch.device("cuda", local_rank))

        def frame():
            dev.set_rtprop(totalSamples=0)
            rd.TraceRays(plt, 0, 0, 0, args.width, args.height)
            sharder.gather_image(dev.rdImage)

        def sync():
            torch.cuda.synchronize()
            if world > 1:
                tdist.barrier()
            torch.cuda.synchronize()

        # untimed visit-count pass: reference-walk bytes of this frame
        visits = None
        if want_roofline:
            rd.SetOption("count_visits", 1)
            frame()
            st = rd.GetTraceStats()
            visits = {k: [int(getattr(st, k)[0]), int(getattr(st, k)[1])] for k in
                      ("visit_top_nodes", "visit_instances", "visit_bot_nodes", "visit_triangles")}
            visits["profile"] = rd.GetVisitProfile(args.depth).astype(np.float64)     # [bounce][class][kind]
            visits["counts"] = rd.GetBounceCounts(args.depth + 1).astype(np.float64)  # rays per bounce
            rd.SetOption("count_visits", 0)
        for _ in range(warmup):
            frame()
        # per-stage HIP events (the roofline's kernel durations) ride along in the timed region at N = 1; with N > 1
        # a rank's frame is a few ms and the event records between its ~36 dispatches cost ~5 % of it, so there the
        # timed region runs without them and the stage times come from extra, untimed frames of the same workload
        prof_inline = world == 1
        rd.SetProfiling(prof_inline)
        acc = dict(primary=0, bounce=0, shadow=0, hits=0, ms_extend=0.0, ms_shadow=0.0, ms_shade=0.0, ms_generate=0.0,
                   ms_accumulate=0.0, ms_total=0.0, ms_fused=0.0, ms_path=0.0, ms_sort=0.0, launches_extend=0)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            frame()
            st = rd.GetTraceStats()
            acc["primary"] += st.rays_primary; acc["bounce"] += st.rays_bounce; acc["shadow"] += st.rays_shadow
            acc["hits"] += st.closest_hits
            if prof_inline:
                for k in STAGE_KEYS:
                    acc[k] += getattr(st, k)
            acc["launches_extend"] += st.launches_extend
            acc["groups"] = int(st.groups)
        sync()
        dt = time.perf_counter() - t0
        if not prof_inline:
            rd.SetProfiling(True)
            nprof = min(steps, 3)
            for _ in range(nprof):
                frame()
                st = rd.GetTraceStats()
                for k in STAGE_KEYS:
                    acc[k] += getattr(st, k) * steps / nprof
            sync()
        rd.SetProfiling(False)
        return scene, dev, acc, dt, visits

    scene, dev, acc, dt, visits = run_workload(args.workload, args.steps, args.warmup, True)

    # aggregate over ranks: rays summed, time = max
    rays_local = acc["primary"] + acc["bounce"] + acc["shadow"]
    if world > 1:
        import torch.distributed as tdist
        t = torch.tensor([float(rays_local), float(acc["primary"] + acc["bounce"])], dtype=torch.float64, device="cuda")
        tdist.all_reduce(t)
        tm = torch.tensor([dt], dtype=torch.float64, device="cuda")
        tdist.all_reduce(tm, op=tdist.ReduceOp.MAX)
        rays_total, rays_pb, dt = float(t[0]), float(t[1]), float(tm[0])
    else:
        rays_total, rays_pb = float(rays_local), float(acc["primary"] + acc["bounce"])

    tlas_blob = None
    if rank == 0 and world == 1 and not args.no_reference and not under_profiler():
        tlas_blob = rd.ReadBuffer(dev.plt, dev.topAccelStruct, dev.topAccelStruct.size).tobytes()
    del dev                                                      # frees nothing on the device (the API has no release), drops the host side

    also = {}
    if world > 1:
        # the other scaling mode on the same ranks, for the record.  strong headline -> also the weak figure (N x spp samples per
        # pixel, every GPU keeps the rays of the N = 1 frame); weak headline -> also the metric's own frame split N ways
        import torch.distributed as tdist
        other_weak = args.scaling == "strong"
        spp_o = args.spp * world if other_weak else args.spp
        st_s = max(3, min(args.steps // 2, 20))
        _, dev_s, a_s, dt_s, _ = run_workload(args.workload, st_s, 1, False, spp=spp_o)
        del dev_s
        t_s = torch.tensor([float(a_s["primary"] + a_s["bounce"] + a_s["shadow"])], dtype=torch.float64, device="cuda")
        tdist.all_reduce(t_s)
        tm_s = torch.tensor([dt_s], dtype=torch.float64, device="cuda")
        tdist.all_reduce(tm_s, op=tdist.ReduceOp.MAX)
        also["weak_scaling_n_x_spp" if other_weak else "strong_scaling_of_the_n1_frame"] = {
            "workload": workload_label(args.workload, args, spp_o), "Mrays_per_s": round(float(t_s[0]) / float(tm_s[0]) / 1e6, 2),
            "ms_per_frame": round(1e3 * float(tm_s[0]) / st_s, 3), "steps": st_s}
    also_keys = [k for k in (args.also.split(",") if args.also is not None else
                             ([w for w in ("sample1", "sponza", "sanmiguel", "sponza400") if w != args.workload] if world == 1 else [])) if k]
    for key in also_keys:
        sc2, _, a2, dt2, v2 = run_workload(key, max(2, args.steps // 2), 1, True)
        r2 = a2["primary"] + a2["bounce"] + a2["shadow"]
        st2 = max(2, args.steps // 2)
        kn2, b2, t2, l2 = traversal_roofline(a2, v2, st2, args.depth)
        also[key] = {"workload": workload_label(key, args), "Mrays_per_s": round(r2 / dt2 / 1e6, 2), "ms_per_frame": round(1e3 * dt2 / st2, 3),
                     "dominant_kernel": kn2, "avg_launch_ms": round(1e3 * t2 / l2, 4) if l2 else None,
                     "reference_walk_equiv_GBps": round(b2 * st2 / t2 / 1e9, 1) if t2 else None,
                     # what the REFERENCE's exhaustive walk visits for this workload's rays (SURVEY 8(d) byte model): the work a scene
                     # asks for, whatever engine walks it -- compare workloads on it before comparing their times
                     "reference_walk_bytes_per_frame": int(b2),
                     "note": "fabric traffic / issue counters of this workload: profiles/ (tools/gpu_profile.sh)"}

    if rank != 0:
        return
    steps = args.steps
    ms_per_step = 1e3 * dt / steps
    rays_extend_per_frame = (acc["primary"] + acc["bounce"]) / steps
    bytes_extend_frame = algorithmic_bytes(visits["visit_top_nodes"][0], visits["visit_instances"][0],
                                           visits["visit_bot_nodes"][0], visits["visit_triangles"][0], rays_extend_per_frame)
    bytes_shadow_frame = algorithmic_bytes(visits["visit_top_nodes"][1], visits["visit_instances"][1],
                                           visits["visit_bot_nodes"][1], visits["visit_triangles"][1], acc["shadow"] / steps)
    kernel_name, roof_bytes, trav_s, launches = traversal_roofline(acc, visits, steps, args.depth)
    ref_equiv = roof_bytes * steps / trav_s / 1e9 if trav_s > 0 else 0.0
    pixels = args.width * args.height if world == 1 else None
    frame_bytes = bytes_extend_frame + bytes_shadow_frame + 184 * acc["hits"] / steps + (20 * pixels if pixels else 0)
    meas = {"traffic": None, "achieved": None, "issue": None}
    if world == 1 and not args.no_pmc and not under_profiler():
        meas = measured_roofline(kernel_name, trav_s, launches, option_args(args))
    achieved = meas["achieved"]
    out = {
        "metric": "Mrays/sec (primary+secondary) and ms/frame at 1920x1080, 4 spp, depth 8",
        "value": round(rays_total / dt / 1e6, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload_label(args.workload, args, spp_main), "width": args.width, "height": args.height, "spp": spp_main, "depth": args.depth,
                   "spp_per_gpu_equivalent": args.spp,
                   "sharding": "none" if world == 1 else "64x64 image tiles interleaved over %d ranks + RGBA8 gather" % world,
                   "traversal": "pool engine, exhaustive walk: every node whose ancestors' boxes the ray passes is visited, as in the reference "
                                "(the culled walk -- docs/CULLED_WALK.md -- is automatic only for scenes of >= 1 M inner BVH nodes, i.e. the "
                                "10.4 M-triangle workload under `also`; option cull forces it); verified bit-identical to the reference's own "
                                "device code (tests/test_gpu_reference.py, tests/test_gpu_cull.py)",
                   # sample groups traced concurrently on their own streams (library rule: 2 for chunks of <= 4.7 M paths);
                   # with more than one, the per-launch durations behind `roofline` overlap in time
                   "sample_groups": acc.get("groups", 1)},
        "rays_per_frame": {"primary": acc["primary"] // steps, "bounce": acc["bounce"] // steps, "shadow": acc["shadow"] // steps,
                           "note": "rank 0 share" if world > 1 else "whole frame"},
        "Mrays_per_s_primary_plus_bounce": round(rays_pb / dt / 1e6, 3),
        "roofline": {
            "bound": "hbm", "kernel": (meas.get("kernel_profiled") or kernel_name).replace("rdx::", ""),
            "achieved": round(achieved, 2) if achieved is not None else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved is not None else None,
            "traffic": meas["traffic"],
            "traffic_detail": meas.get("traffic_detail"),
            "avg_launch_ms": round(1e3 * trav_s / launches, 4),
            "launches": launches,
            "issue": meas["issue"],
            # SURVEY 8(d) byte model, kept as a normalisation: what the REFERENCE's exhaustive, cache-less walk would have
            # to read for the same rays.  No peak, no fraction: the product neither makes those visits nor misses the caches.
            "reference_walk_bytes_per_launch": int(roof_bytes * steps / launches),
            "reference_walk_equiv_GBps": round(ref_equiv, 2),
            "note": "achieved = fabric bytes per launch (FETCH_SIZE + WRITE_SIZE, KiB, PMC child passes of this run; FETCH_SIZE at face "
                    "value as calibrated for this kernel's 64-B / 48-B record gathers, see traffic_detail) / average "
                    "launch duration (HIP events in the timed region).  The scene (BVH 25 MB) lives in L2 + Infinity Cache, so "
                    "the kernel is bound by instruction issue, not by HBM: see `issue` and DESIGN.md section 5",
        },
        "roofline_frame": {"reference_walk_equiv_GBps": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 2) if world == 1 else None,
                           "bytes_per_frame": int(frame_bytes)},
        "stage_timing": "HIP events inside the timed region" if world == 1 else
                        "HIP events on 3 extra untimed frames per rank (the timed region runs without per-stage events)",
        "stage_ms_per_frame": {k[3:]: round(acc[k] / steps, 4) for k in ("ms_generate", "ms_extend", "ms_shade", "ms_sort", "ms_shadow", "ms_fused", "ms_path", "ms_accumulate", "ms_total")},
        "device": rd.Platform.device_name(),
    }
    if also:
        out["also"] = also
    if tlas_blob is not None:
        r = reference_on_gpu(scene, tlas_blob, ms_per_step)
        if r:
            out["reference_on_gpu"] = r
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as _d
        if _d.is_initialized():
            _d.destroy_process_group()

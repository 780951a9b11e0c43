#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (primary + bounce + shadow) and ms/frame at 1920x1080,
4 spp, depth 8 (BASELINE.json).

A step = one RD::TraceRays frame (RayTraceProperties{totalSamples 0, batchSize 4, depth 8}) of the
workload, scene and accumulators already resident in HBM.  N > 1: one process per GPU
(torch.distributed / RCCL), the frame is sharded by interleaved 64x64 image tiles, no collective
while rendering, one RGBA8 gather to rank 0 at frame end -- inside the timed region.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline      dominant kernel (the traversal launches: k_fused_pool = shadow rays of bounce d + closest-hit rays of
                bounce d+1 per launch; k_extend_pool when launches are not fused): algorithmic bytes of the launches
                in the timed region / their HIP-event time, against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (a port of the reference algorithm) on a bounded pixel sample of the
                same workload, on this box's host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/
    *_pmc_hbm.json, written by tools/prof_summary.py from separate FETCH_SIZE / WRITE_SIZE passes of this
    same command); None if there is none for this workload."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json"))):
        try:
            d = json.load(open(p))
        except Exception:
            continue
        if d.get("workload", "sample1") != workload:
            continue
        k = d.get("kernels", {}).get(kernel)
        if k:
            best = (int(k["hbm_bytes_per_launch"]), os.path.basename(p))
    return best

WORKLOADS = {
    # BASELINE.json configs[1]: "sample1.cpp scene, 1920x1080, 4 spp, depth 8, 1xMI355X"
    "sample1": ("c1_cornell", "sample1 scene (procedural Cornell stand-in, 20.5k tris, 8 instances), 1920x1080, 4 spp, depth 8"),
    # BASELINE.json configs[2]: "Sponza via assimp (~260k tris) ..." -- no asset offline, procedural atrium
    "sponza": ("c2_atrium", "Sponza-class procedural atrium (262k tris, 25 instances), 1920x1080, 4 spp, depth 8"),
    # BASELINE.json configs[4] geometry ("San-Miguel-scale ~10M tris"): BVH (446 MB) beyond L2 + Infinity Cache
    "sanmiguel": ("c4_atrium_10m", "San-Miguel-scale procedural atrium (10.4M tris, 25 instances), 1920x1080, 4 spp, depth 8"),
}


def algorithmic_bytes(top, inst, bot, tri, rays):
    """B = 16/ray + 48/top node + (80+16)/instance visit + 48/bottom node + 64/triangle,
    every visit of the REFERENCE algorithm's exhaustive walk charged as an uncached read."""
    return 16 * rays + 48 * top + 96 * inst + 48 * bot + 64 * tri


ENGINE = "pool"          # suffix of the traversal kernels' names: k_fused_pool (library default, kernel option 3) / k_fused_coop (2)


def traversal_roofline(acc, visits, steps, depth):
    """(kernel name, algorithmic bytes of its launches per frame, seconds of those launches over the timed region,
    launches) of the dominant traversal kernel"""
    if acc.get("ms_path", 0.0) > 0.0:
        # whole paths in one launch per chunk (k_path_coop): every closest-hit and shadow walk of the frame + the
        # closest-hit shading reads (184 B per hit)
        b = sum(algorithmic_bytes(visits["visit_top_nodes"][c], visits["visit_instances"][c], visits["visit_bot_nodes"][c],
                                  visits["visit_triangles"][c], r)
                for c, r in ((0, (acc["primary"] + acc["bounce"]) / steps), (1, acc["shadow"] / steps)))
        return ("k_path_coop", b + 184 * acc["hits"] / steps, acc["ms_path"] * 1e-3, max(1, acc["launches_extend"]))
    if acc["ms_fused"] > 0.0:
        # shadow(d) and extend(d+1) share one launch (k_fused_coop, D-1 launches per frame): its algorithmic bytes are
        # those of the shadow rays of bounces 0..D-2 plus the closest-hit rays of bounces 1..D-1
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
    cores = max(1, min(cores, 16))           # a 1-GPU box's CPU share is 16 cores
    # calibrate on a small sample (after a warm-up that starts the thread pool), then size the timed
    # sample for ~budget_s of CPU work
    rng = np.random.default_rng(0)
    probe = rng.choice(n, 4096, replace=False).astype(np.uint32)
    osc.render(nthreads=cores, pixels=probe[:256])
    t = time.time(); osc.render(nthreads=cores, pixels=probe); dt = max(time.time() - t, 1e-3)
    rate = 4096 / dt
    m = int(min(n, max(4096, rate * budget_s)))
    px = rng.choice(n, m, replace=False).astype(np.uint32)
    # the sample is rendered again (same pixels, same rays) until ~10 s of CPU work have been timed
    rays, dt, reps = 0, 0.0, 0
    while dt < 10.0 and reps < 64:
        osc.scratch[:] = 0
        t = time.time(); c = osc.render(nthreads=cores, counters=True, pixels=px); dt += time.time() - t
        d = c.as_dict()
        rays += d["rays"][0] + d["rays"][1]
        reps += 1
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d random pixels of the same frame (%d spp, depth %d) x %d passes, %.1f s, %d reference-algorithm rays "
                      "(incl. the reference's duplicate re-trace after a primary miss)"
                      % (m, int(scene.rtprop["batchSize"]), int(scene.rtprop["depth"]), reps, dt, rays)}


STAGE_KEYS = ("ms_extend", "ms_shadow", "ms_shade", "ms_generate", "ms_accumulate", "ms_total", "ms_fused", "ms_path")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="sample1", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse", type=int, default=-1, help="-1 auto, 0/1: shadow(d)+extend(d+1) in one launch")
    ap.add_argument("--pipeline", type=int, default=-1, help="-1 library default, 0 staged wavefront, 1 whole paths in one persistent launch")
    ap.add_argument("--kernel", type=int, default=-1, help="-1 library default; 2 cooperative, 3 cooperative with a shared node pool, 1 / 0 per-lane")
    ap.add_argument("--top-flat", type=int, default=-1, help="-1 library default; 0/1: evaluate small top-level trees all at once (pool engine)")
    ap.add_argument("--groups", type=int, default=-1, help="-1 library default; 1..4 sample groups of a chunk on their own streams")
    ap.add_argument("--cull", type=int, default=-1, help="-1 library default (on); 0 = exhaustive walk, 1 = culled walk (pool engine)")
    ap.add_argument("--also", default="", help="comma list of extra workloads to time (reported under 'also')")
    args = ap.parse_args()

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
    rd.SetOption("fuse", args.fuse)
    if args.pipeline >= 0:
        rd.SetOption("pipeline", args.pipeline)
    if args.top_flat >= 0:
        rd.SetOption("top_flat", args.top_flat)
    if args.groups >= 1:
        rd.SetOption("groups", args.groups)
    if args.cull >= 0:
        rd.SetOption("cull", args.cull)
    if args.kernel >= 0:
        rd.SetOption("kernel", args.kernel)
        global ENGINE
        ENGINE = "pool" if args.kernel == 3 else "coop"

    def run_workload(key, steps, warmup, want_roofline):
        cfg, label = WORKLOADS[key]
        scene = scenes.CONFIGS[cfg](args.width, args.height, args.spp, args.depth)
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

        # untimed visit-count pass: algorithmic bytes of the reference's exhaustive walk for this frame
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
                   ms_accumulate=0.0, ms_total=0.0, ms_fused=0.0, ms_path=0.0, launches_extend=0)
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
        return scene, dev, acc, dt, visits, label

    scene, dev, acc, dt, visits, label = run_workload(args.workload, args.steps, args.warmup, True)

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

    also = {}
    for key in [k for k in args.also.split(",") if k]:
        sc2, _, a2, dt2, v2, lab2 = run_workload(key, max(2, args.steps // 2), 1, True)
        r2 = a2["primary"] + a2["bounce"] + a2["shadow"]
        st2 = max(2, args.steps // 2)
        kn2, b2, t2, _ = traversal_roofline(a2, v2, st2, args.depth)
        also[key] = {"workload": lab2, "Mrays_per_s": round(r2 / dt2 / 1e6, 2), "ms_per_frame": round(1e3 * dt2 / st2, 3),
                     "roofline_kernel": kn2, "roofline_achieved_GBps": round(b2 * st2 / t2 / 1e9, 1) if t2 else None,
                     "roofline_frac": round(b2 * st2 / t2 / 1e9 / HBM_PEAK_GBS, 4) if t2 else None}

    if rank != 0:
        return
    steps = args.steps
    ms_per_step = 1e3 * dt / steps
    # roofline of the dominant kernel: the traversal launches of the timed region
    rays_extend_per_frame = (acc["primary"] + acc["bounce"]) / steps
    bytes_extend_frame = algorithmic_bytes(visits["visit_top_nodes"][0], visits["visit_instances"][0],
                                           visits["visit_bot_nodes"][0], visits["visit_triangles"][0], rays_extend_per_frame)
    bytes_shadow_frame = algorithmic_bytes(visits["visit_top_nodes"][1], visits["visit_instances"][1],
                                           visits["visit_bot_nodes"][1], visits["visit_triangles"][1], acc["shadow"] / steps)
    fused = acc["ms_fused"] > 0.0
    kernel_name, roof_bytes, trav_s, launches = traversal_roofline(acc, visits, steps, args.depth)
    achieved = roof_bytes * steps / trav_s / 1e9 if trav_s > 0 else 0.0
    pixels = args.width * args.height if world == 1 else None
    frame_bytes = bytes_extend_frame + bytes_shadow_frame + 184 * acc["hits"] / steps + (20 * pixels if pixels else 0)
    out = {
        "metric": "Mrays/sec (primary+secondary) and ms/frame at 1920x1080, 4 spp, depth 8",
        "value": round(rays_total / dt / 1e6, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": label, "width": args.width, "height": args.height, "spp": args.spp, "depth": args.depth,
                   "sharding": "none" if world == 1 else "64x64 image tiles interleaved over %d ranks + RGBA8 gather" % world,
                   "traversal": "exact (reference visit set and order)",
                   # sample groups traced concurrently on their own streams (library rule: 2 for chunks of <= 4.7 M paths);
                   # with more than one, the per-launch durations behind `roofline` overlap in time: `achieved` is a lower bound
                   "sample_groups": acc.get("groups", 1)},
        "rays_per_frame": {"primary": acc["primary"] // steps, "bounce": acc["bounce"] // steps, "shadow": acc["shadow"] // steps,
                           "note": "rank 0 share" if world > 1 else "whole frame"},
        "Mrays_per_s_primary_plus_bounce": round(rays_pb / dt / 1e6, 3),
        "roofline": {
            "bound": "hbm", "kernel": kernel_name,
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            # HBM bytes per launch from the committed PMC passes of the 1-GPU, full-frame run of this workload (a rank of
            # an N-GPU run launches on 1/N of the frame: no PMC figure for that)
            "traffic": (pmc_traffic("rdx::" + kernel_name, args.workload) or (None, None))[0]
                       if world == 1 and (args.width, args.height, args.spp, args.depth) == (1920, 1080, 4, 8) else None,
            "traffic_source": (pmc_traffic("rdx::" + kernel_name, args.workload) or (None, None))[1]
                              if world == 1 and (args.width, args.height, args.spp, args.depth) == (1920, 1080, 4, 8) else None,
            "algorithmic_bytes_per_launch": int(roof_bytes * steps / launches),
            "avg_launch_ms": round(1e3 * trav_s / launches, 4),
            "launches": launches,
            "note": "achieved = reference-walk bytes (16/ray + 48/node + 96/instance visit + 64/triangle) of the traversal "
                    "launches (fused: shadow rays of bounce d + closest-hit rays of bounce d+1 per launch) in the timed region / "
                    "their HIP-event time; traffic (PMC) see profiles/",
        },
        "roofline_frame": {"algorithmic_GBps": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 2) if world == 1 else None,
                           "frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if world == 1 else None,
                           "bytes_per_frame": int(frame_bytes)},
        "stage_timing": "HIP events inside the timed region" if world == 1 else
                        "HIP events on 3 extra untimed frames per rank (the timed region runs without per-stage events)",
        "stage_ms_per_frame": {k[3:]: round(acc[k] / steps, 4) for k in ("ms_generate", "ms_extend", "ms_shade", "ms_shadow", "ms_fused", "ms_path", "ms_accumulate", "ms_total")},
        "device": rd.Platform.device_name(),
    }
    if also:
        out["also"] = also
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as _d
        if _d.is_initialized():
            _d.destroy_process_group()

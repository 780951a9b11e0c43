#!/bin/bash
# gpu_n2.sh -- rehearsal of the N > 1 bench path on a 1-GPU box: two ranks sharing the GPU over gloo (transport staged
# through host memory), and what the per-stage events cost a 1/8 frame (profiling on / off around TraceRays)
mkdir -p gpurun_out
RDX_DIST_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_n2_gloo.log 2>&1
grep -o '{"metric.*' gpurun_out/bench_n2_gloo.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('n2 gloo', d['value'], d['ms_per_step'], d['stage_timing'][:40], d['roofline']['frac'], d['stage_ms_per_frame'])"
timeout -k 10 200 python - <<'PY'
import importlib, time, torch
rd = importlib.import_module("radiance-ray-tracing_amd.rd"); scenes = importlib.import_module("radiance-ray-tracing_amd.scenes")
plt = rd.Platform.GetPlatform(0)
for name in ("c1_cornell", "c2_atrium"):
    s = scenes.CONFIGS[name](680, 381, 4, 8); dev = scenes.DeviceScene(s, plt)
    for prof in (True, False, True, False):
        rd.SetProfiling(prof)
        for _ in range(3): rd.TraceRays(plt, 0, 0, 0, 680, 381)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): rd.TraceRays(plt, 0, 0, 0, 680, 381)
        torch.cuda.synchronize(); print(name, "profiling", prof, "%.3f ms" % ((time.perf_counter() - t0) * 50))
rd.SetProfiling(False)
PY

#!/bin/bash
# gpu_pool_var.sh -- kernel 3 (shared node pool) on every experiment build, default build with kernel 2 first
mkdir -p gpurun_out
run() { for wh in "1920 1080 sample1" "1920 1080 sponza" "680 381 sample1"; do set -- $wh
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --width $1 --height $2 --workload $3 --kernel $K --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('kernel $K $3 $1x$2', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shd', s['shadow'], 'fused', s['fused'])"
  done; }
true
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null); do
  export RDX_LIB=$PWD/$lib; K=3; echo "=== $lib"; run
done

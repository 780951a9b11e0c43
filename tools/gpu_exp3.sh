#!/bin/bash
# gpu_exp3.sh -- the three full-size workloads + the 1/8 frame on every experiment build
mkdir -p gpurun_out
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null); do
  export RDX_LIB=$PWD/$lib
  echo "=== $lib"
  for wh in "1920 1080 sample1" "1920 1080 sponza" "1920 1080 sanmiguel" "680 381 sample1" "680 381 sponza"; do set -- $wh
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --width $1 --height $2 --workload $3 --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$3 $1x$2', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shade', s['shade'], 'shd', s['shadow'], 'fused', s['fused'])"
  done
done

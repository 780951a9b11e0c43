#!/bin/bash
# gpu_small.sh -- 1/8-frame (one rank of eight) and 1/2-frame timings on every experiment build
mkdir -p gpurun_out
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null); do
  export RDX_LIB=$PWD/$lib
  echo "=== $lib"
  for wh in "680 381 sample1" "680 381 sponza" "1358 764 sample1"; do set -- $wh
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --width $1 --height $2 --workload $3 --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$3 $1x$2', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shade', s['shade'], 'shd', s['shadow'], 'fused', s['fused'])"
  done
done

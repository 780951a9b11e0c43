#!/bin/bash
# gpu_exp.sh -- bench every experiment build radiance-ray-tracing_amd/librdx_*.so next to the default one (no parity run:
# timing experiments may compute wrong results on purpose)
mkdir -p gpurun_out
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null); do
  export RDX_LIB=$PWD/$lib
  echo "=== $lib"
  for wh in "1920 1080 sample1" "1920 1080 sponza"; do set -- $wh
    timeout -k 10 200 python bench.py --steps 6 --warmup 2 --width $1 --height $2 --workload $3 --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$3 $1x$2', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shade', s['shade'], 'shd', s['shadow'], 'fused', s['fused'])"
  done
done

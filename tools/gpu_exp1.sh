#!/bin/bash
# gpu_exp1.sh -- depth-1 frames (generate, extend, one shade, one shadow) on every experiment build: isolates the first shade launch
mkdir -p gpurun_out
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null); do
  export RDX_LIB=$PWD/$lib
  echo "=== $lib"
  for wl in sample1 sponza; do
    timeout -k 10 200 python bench.py --steps 6 --warmup 2 --depth 1 --workload $wl --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$wl depth1', d['ms_per_step'], s)"
  done
done

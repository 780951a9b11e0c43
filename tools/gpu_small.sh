#!/bin/bash
# gpu_small.sh -- 1/8-frame (one rank of eight), 1/2-frame and full-frame timings on every experiment build
# (radiance-ray-tracing_amd/librdx_*.so beside the default), two passes
mkdir -p gpurun_out
for rep in 1 2; do
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null); do
  export RDX_LIB=$PWD/$lib
  line="$(basename $lib)"
  for wh in "680 381 sample1" "680 381 sponza" "1358 764 sample1" "1920 1080 sample1" "1920 1080 sponza"; do set -- $wh
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --width $1 --height $2 --workload $3 --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    line="$line | $3 $1: $(python -c "
import json; d=json.load(open('gpurun_out/bv.json')); print(d['ms_per_step'])")"
  done
  echo "$line"
done
done

#!/bin/bash
# gpu_ab.sh -- frame times of the default build and of every experiment build radiance-ray-tracing_amd/librdx_*.so
mkdir -p gpurun_out
run() { for wl in sample1 sponza sanmiguel; do
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --workload $wl --also= --no-cpu-baseline --no-pmc --no-reference > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$wl', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shd', s['shadow'], 'fused', s['fused'], 'shade', s['shade'])"
  done; }
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null); do
  export RDX_LIB=$PWD/$lib; echo "=== $lib"; run
done

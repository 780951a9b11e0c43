#!/bin/bash
# gpu_ab.sh [workloads...] -- frame times of the default build and of every experiment build radiance-ray-tracing_amd/librdx_*.so
# (RDX_DEFINES="-DFOO" RDX_LIB_NAME=librdx_foo.so python radiance-ray-tracing_amd/build.py --force); the two-table test variant is skipped
mkdir -p gpurun_out
WLS=${@:-sample1 sponza sanmiguel}
run() { for wl in $WLS; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --workload $wl --also= --no-cpu-baseline --no-pmc --no-reference $BENCH_OPTS > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$wl', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shd', s['shadow'], 'fused', s['fused'], 'shade', s['shade'])"
  done; }
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null | grep -v "sbt2\|wip"); do
  export RDX_LIB=$PWD/$lib; echo "=== $lib"; run
done

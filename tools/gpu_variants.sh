#!/bin/bash
# gpu_variants.sh -- parity suite on the default build, then the same bench / launch-size probes on every experiment
# build radiance-ray-tracing_amd/librdx_*.so (RDX_LIB selects the library; ge.build() is skipped for them)
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for lib in radiance-ray-tracing_amd/librdx.so $(ls radiance-ray-tracing_amd/librdx_*.so 2>/dev/null); do
  export RDX_LIB=$PWD/$lib
  echo "=== $lib"; RDX_VERBOSE=1 python -c "import sys; sys.path[:0]=[\".\",\"tests\"]; import rrt_amd; from radiance_ray_tracing_amd import rd, scenes; [scenes.DeviceScene(scenes.CONFIGS[c](64,64,1,1)).render() for c in (\"c1_cornell\",\"c2_atrium\")]" 2>&1 | grep "\[rdx\]" | head -2
  for wh in "1920 1080 sample1" "680 381 sample1" "1920 1080 sponza" "680 381 sponza"; do set -- $wh
    timeout -k 10 200 python bench.py --steps 6 --warmup 2 --width $1 --height $2 --workload $3 --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$3 $1x$2', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shade', s['shade'], 'shd', s['shadow'], 'fused', s['fused'], 'frac', d['roofline']['frac'])"
  done
  [ -n "$SKIP_SCALE" ] || timeout -k 10 200 python tools/trav_scale.py c1_cornell 2>&1 | grep "kernel=2" | awk '{printf "%s %s; ", $2" "$3, $5}'; echo
done
unset RDX_LIB
for wl in sample1 sponza; do
timeout -k 10 200 python bench.py --steps 6 --warmup 2 --workload $wl --pipeline 1 --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err && python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('pipeline 1 $wl', d['value'], d['ms_per_step'], 'path', s['path'])"
done

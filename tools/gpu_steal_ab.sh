#!/bin/bash
# gpu_steal_ab.sh -- parity suite, bench at full / 1/8 frame size, launch-size scaling of the traversal kernel
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for wh in "1920 1080 sample1" "680 381 sample1" "1920 1080 sponza" "680 381 sponza"; do set -- $wh
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --width $1 --height $2 --workload $3 --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$3 $1x$2', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shade', s['shade'], 'shd', s['shadow'], 'fused', s['fused'], 'frac', d['roofline']['frac'])"
done
timeout -k 10 200 python tools/trav_scale.py c1_cornell 2>&1 | grep "kernel=2"
timeout -k 10 200 python tools/trav_scale.py c2_atrium 2>&1 | grep "kernel=2"

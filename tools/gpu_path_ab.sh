#!/bin/bash
# gpu_path_ab.sh -- parity suite, then staged vs whole-path pipeline at full and 1/8 frame size
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest exit $rc"; tail -5 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for pl in 0 1; do
for wh in "1920 1080 sample1" "680 381 sample1" "1920 1080 sponza" "680 381 sponza"; do set -- $wh
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --width $1 --height $2 --workload $3 --pipeline $pl --no-cpu-baseline > gpurun_out/bv_$pl.json 2>gpurun_out/bv_$pl.err || { echo "bench failed"; tail -5 gpurun_out/bv_$pl.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bv_$pl.json')); s=d['stage_ms_per_frame']; print('pipeline $pl $3 $1x$2', d['value'], d['ms_per_step'], s, 'frac', d['roofline']['frac'])"
done
done

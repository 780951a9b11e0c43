#!/bin/bash
# gpu_ab.sh -- quick A/B on the GPU box: parity suite, then stage timings at full and at 1/8 frame size
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
for wh in "1920 1080 sample1" "680 381 sample1" "1920 1080 sponza" "680 381 sponza"; do set -- $wh
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --width $1 --height $2 --workload $3 --no-cpu-baseline > gpurun_out/bv.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$3 $1x$2', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shade', s['shade'], 'shd', s['shadow'], 'frac', d['roofline']['frac'])"
done

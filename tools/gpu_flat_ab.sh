#!/bin/bash
# gpu_flat_ab.sh -- pool engine with / without the flat evaluation of small top-level trees
mkdir -p gpurun_out
for f in 0 1; do
  for wh in "1920 1080 sample1" "1920 1080 sponza" "1920 1080 sanmiguel" "680 381 sample1" "680 381 sponza"; do set -- $wh
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --width $1 --height $2 --workload $3 --top-flat $f --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('top_flat $f $3 $1x$2', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shd', s['shadow'], 'fused', s['fused'])"
  done
done

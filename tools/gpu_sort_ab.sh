#!/bin/bash
# gpu_sort_ab.sh -- frame times with and without the per-bounce ray sort, full frame and 1/8-size frame
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for wl in sample1 sponza sanmiguel; do
  for srt in 0 1; do
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --workload $wl --also= --no-cpu-baseline --no-pmc --no-reference --sort $srt > gpurun_out/bs.json 2>gpurun_out/bs.err || { echo "$wl sort=$srt failed"; tail -3 gpurun_out/bs.err; continue; }
    python -c "
import json; d=json.load(open('gpurun_out/bs.json')); s=d['stage_ms_per_frame']; print('$wl sort=$srt', d['value'], d['ms_per_step'], 'fused', s['fused'], 'sort', s['sort'], 'shade', s['shade'], 'ext', s['extend'])"
  done
done

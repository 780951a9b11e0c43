#!/bin/bash
# gpu_groups.sh -- sample groups on their own streams with a divided persistent grid, at 1/8, 1/4, 1/2 of a 1080p frame
mkdir -p gpurun_out
run() { # groups div w h workload
  RDX_COOP_GRID_DIV=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --groups $1 --width $3 --height $4 --workload $5 --no-cpu-baseline > gpurun_out/bg.json 2>gpurun_out/bg.err || { echo "bench failed"; tail -5 gpurun_out/bg.err; return; }
  python -c "
import json; d=json.load(open('gpurun_out/bg.json')); print('groups $1 div $2 $5 $3x$4', d['value'], d['ms_per_step'])"
}
for wl in sample1 sponza; do
  for wh in "680 381" "960 540" "1358 764"; do
    for gd in "1 1" "2 2" "2 1"; do set -- $gd
      run $1 $2 $wh $wl
    done
  done
done

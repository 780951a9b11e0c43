#!/bin/bash
# occupancy_probe.sh -- frame time of the traversal kernels as a function of resident waves per CU
# (extra LDS per wave lowers the residency the LDS allows; 128-thread blocks).  LDS per wave = (need + 18) * 256 B.
mkdir -p gpurun_out
for cfg in "0 0" "128 0" "128 900" "128 2048" "128 3500" "128 5400"; do set -- $cfg
  for wl in sample1 sponza; do
  RDX_COOP_THREADS=$1 RDX_COOP_LDS_PAD=$2 timeout -k 10 200 python bench.py --steps 6 --warmup 2 --workload $wl --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('threads $1 pad $2 $wl', d['value'], d['ms_per_step'], 'ext', s['extend'], 'fused', s['fused'])"
  done
done

#!/bin/bash
# occupancy_probe.sh -- frame time of the traversal kernels as a function of resident waves per CU
# (64-thread blocks; extra LDS per wave lowers the residency the LDS allows).  sample1: 11520 B LDS per wave.
mkdir -p gpurun_out
for cfg in "256 0" "128 0" "64 0" "64 1024" "64 2200" "64 4900" "64 9000" "64 15800" "64 29500"; do set -- $cfg
  for wl in sample1 sponza; do
  RDX_COOP_THREADS=$1 RDX_COOP_LDS_PAD=$2 timeout -k 10 200 python bench.py --steps 6 --warmup 2 --workload $wl --no-cpu-baseline > gpurun_out/bv.json 2>gpurun_out/bv.err || { echo "bench failed"; tail -5 gpurun_out/bv.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('threads $1 pad $2 $wl', d['value'], d['ms_per_step'], 'ext', s['extend'], 'fused', s['fused'])"
  done
done

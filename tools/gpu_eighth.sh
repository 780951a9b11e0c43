#!/bin/bash
# gpu_eighth.sh -- per-dispatch timeline (queue, start, duration) of the last frame of a 1/8-frame
# run (one rank of eight) next to the full frame, from rocprofv3 --kernel-trace
set -e
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/eighth; mkdir -p $OUT; cd /tmp
for cfg in "680 381 sample1" "1920 1080 sample1" "680 381 sponza"; do set -- $cfg
  D=$OUT/$3_$1; rm -rf $D
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -o r1 -- python3 $R/bench.py --workload $3 --width $1 --height $2 --steps 3 --warmup 2 --no-cpu-baseline > $D.log 2>&1
  python3 - $D $3_$1 <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/r1_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'rdx::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last frame = the dispatches between the last two k_accumulate (with sample groups a frame has several k_generate)
acc = [i for i, r in enumerate(rows) if 'k_accumulate' in r['Kernel_Name']]
fr = rows[acc[-2] + 1:acc[-1] + 1]
t0 = int(fr[0]['Start_Timestamp'])
print('==', sys.argv[2], 'frame total %.3f ms' % ((max(int(r['End_Timestamp']) for r in fr) - t0) / 1e6))
busy = 0
for r in fr:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rdx::', '')
    print('  q%-2s %-24s start %8.1f us  dur %8.1f us  grid %s wg %s' % (r['Queue_Id'], name[:24], (s - t0) / 1e3, (e - s) / 1e3, r['Grid_Size_X'], r['Workgroup_Size_X']))
    busy += e - s
print('  sum of durations %.3f ms' % (busy / 1e6))
PY
done > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt

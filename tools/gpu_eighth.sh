#!/bin/bash
# gpu_eighth.sh -- per-dispatch timeline (duration + gap to the previous dispatch) of the last frame of a 1/8-frame
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
# last frame = from the last k_generate on
last = max(i for i, r in enumerate(rows) if 'k_generate' in r['Kernel_Name'])
fr = rows[last:]
t0 = int(fr[0]['Start_Timestamp']); prev_end = t0
print('==', sys.argv[2], 'frame total %.3f ms' % ((int(fr[-1]['End_Timestamp']) - t0) / 1e6))
busy = 0
for r in fr:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rdx::', '')
    print('  %-28s dur %8.1f us  gap %6.1f us  grid %s wg %s' % (name[:28], (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))))
    busy += e - s; prev_end = e
print('  busy %.3f ms' % (busy / 1e6))
PY
done > $OUT/timeline.txt 2>&1
cat $OUT/timeline.txt

#!/bin/bash
# gpu_shard_timeline.sh [opt=value ...] -- per-dispatch timeline of the last frame of rank 0 of 8 (1080p x 4 spp x depth 8), both scenes
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/shard_tl; mkdir -p $OUT; cd /tmp
for cfg in c2_atrium c1_cornell; do
  D=$OUT/$cfg; rm -rf $D
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $D -o r1 -- python3 $R/tools/shard_frame.py $cfg 0 8 "$@" > $D.log 2>&1
  python3 - $D $cfg <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/r1_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'rdx::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acc = [i for i, r in enumerate(rows) if 'k_accumulate' in r['Kernel_Name']]
fr = rows[acc[-2] + 1:acc[-1] + 1]
t0 = int(fr[0]['Start_Timestamp'])
print('==', sys.argv[2], 'rank 0 of 8, frame total %.3f ms' % ((max(int(r['End_Timestamp']) for r in fr) - t0) / 1e6))
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

#!/bin/bash
# gpu_sort_stats.sh [workload] -- on the GPU box: rocprofv3 kernel stats of a frame with the per-bounce ray sort forced on
WL=${1:-sanmiguel}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/sortstats_$WL; mkdir -p $OUT; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r1 -- python3 $R/bench.py --workload $WL --also= --no-cpu-baseline --no-pmc --no-reference --sort 1 --steps 4 --warmup 1 > $OUT.log 2>&1
cd $R; grep -h "k_sort\|Memset\|fillBuffer" $(find $OUT -name "r1_kernel_stats.csv") | cut -d, -f1-8 | sed 's/(rdx::PathStreams[^"]*"/"/' | cut -c1-160

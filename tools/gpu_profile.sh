#!/bin/bash
# gpu_profile.sh <tag> [workload] -- on the GPU box: rocprofv3 kernel stats + HBM PMC passes of bench.py, then the bench line
set -e
TAG=${1:-r01x}; WL=${2:-sample1}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r1 -- python3 $R/bench.py --workload $WL --steps 5 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o r1 -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o r1 -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
cd $R
tail -1 $OUT/stats.log | cut -c1-400

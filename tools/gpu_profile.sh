#!/bin/bash
# gpu_profile.sh <tag> [workload] -- on the GPU box: rocprofv3 kernel stats + PMC passes (fabric traffic, issue counters) of one
# workload through bench.py's own child mode, condensed into gpurun_out/prof_<tag>/ for tools/prof_summary.py
set -e
TAG=${1:-r02x}; WL=${2:-sponza}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp
B="python3 $R/bench.py --workload $WL --also= --no-cpu-baseline --no-pmc --no-reference"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r1 -- $B --steps 5 --warmup 1 > $OUT/stats.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o r1 -- $B --steps 2 --warmup 1 > $OUT/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o r1 -- $B --steps 2 --warmup 1 > $OUT/write.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/issue -o r1 -- $B --steps 2 --warmup 1 > $OUT/issue.log 2>&1
timeout -k 10 400 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/cache -o r1 -- $B --steps 2 --warmup 1 > $OUT/cache.log 2>&1 || true
cd $R
tail -1 $OUT/stats.log | cut -c1-300

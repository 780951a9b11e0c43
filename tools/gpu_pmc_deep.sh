#!/bin/bash
# gpu_pmc_deep.sh [workload] -- on the GPU box: where do the traversal waves wait?  LDS / VMEM / scalar issue counters in
# separate rocprofv3 --pmc passes, printed as per-launch averages of the pool kernels (tools/prof_summary.py per_kernel_all).
WL=${1:-sponza}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/deep_$WL; mkdir -p $OUT; cd /tmp
B="python3 $R/bench.py --workload $WL --also= --no-cpu-baseline --no-pmc --no-reference --steps 2 --warmup 1"
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o r1 -- $B > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
pass a SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass b SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
pass c SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_VALU SQ_WAVE_CYCLES
pass d SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
pass e SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_WAVE_CYCLES
pass f SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES
cd $R
python3 - <<PY
import sys, json; sys.path.insert(0, "tools")
from prof_summary import per_kernel_all
for n in "abcdef":
    d = per_kernel_all("$OUT/%s/r1_counter_collection.csv" % n)
    for k, v in d.items():
        if "pool" in k: print(n, k, json.dumps({c: round(x) for c, x in v.items()}))
PY

#!/bin/bash
# gpu_pmc_mem.sh [workload] -- on the GPU box: is the vector memory path (TA address unit, L1 / TCP, TD) the limit of the
# traversal kernels?  rocprofv3 --pmc passes, per-launch averages of the pool kernels.
WL=${1:-sponza}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mem_$WL; mkdir -p $OUT; cd /tmp
B="python3 $R/bench.py --workload $WL --also= --no-cpu-baseline --no-pmc --no-reference --steps 2 --warmup 1"
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o r1 -- $B > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
# (the TA_* counters -- TA_TA_BUSY, TA_ADDR_STALLED_BY_TC_CYCLES, TA_*_WAVEFRONTS -- do not come back on this pool: rocprofv3 sat
#  until the 300 s limit on both TA passes in round 2, so they are not collected)
pass c TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE
pass d TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass e TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE
pass f TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TD_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass g TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass h TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
cd $R
python3 - <<PY
import sys, json; sys.path.insert(0, "tools")
from prof_summary import per_kernel_all
for n in "cdefgh":
    d = per_kernel_all("$OUT/%s/r1_counter_collection.csv" % n)
    for k, v in d.items():
        if "fused_pool" in k: print(n, k, json.dumps({c: round(x) for c, x in v.items()}))
PY

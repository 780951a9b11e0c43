#!/bin/bash
# gpu_pmc.sh <workload> -- issue / stall / cache counters of the traversal and shade kernels (separate PMC passes)
WL=${1:-sample1}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_$WL; mkdir -p $OUT; cd /tmp
run() { timeout -k 10 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -o r1 -- python3 $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$1.log 2>&1 || { echo "pass $1 failed"; tail -3 $OUT/$1.log; exit 1; }; }
run sq1 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
run sq2 "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run tc1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
cd $R
python3 - <<PY
import csv, collections, glob
for p in ("sq1","sq2","tc1"):
    f = glob.glob("$OUT/%s/**/r1_counter_collection.csv" % p, recursive=True)
    if not f: print(p, "no csv"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ","")
        if not k.startswith("rdx::"): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print(p, k, {a: "%.4g" % b for a, b in v.items()})
PY

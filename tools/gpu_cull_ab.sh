#!/bin/bash
# gpu_cull_ab.sh -- on the GPU box: the GPU test suite, then frame times with the culled and the exhaustive walk
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cull; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for wl in sample1 sponza sanmiguel; do
  for c in 1 0; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 --workload $wl --cull $c > $O/${wl}_c$c.json 2> $O/${wl}_c$c.err
    python - <<PY
import json
try:
    d=json.load(open("$O/${wl}_c$c.json")); print("$wl cull=$c", d["value"], "Mrays/s", d["ms_per_step"], "ms", {k:v for k,v in d["stage_ms_per_frame"].items() if v})
except Exception as e: print("$wl cull=$c failed", e)
PY
  done
done

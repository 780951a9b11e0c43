#!/bin/bash
# gpu_stats_ab.sh [cfg,...] -- step statistics (tools/coop_stats.py) of every -DCOOP_STATS build radiance-ray-tracing_amd/librdx_stats*.so
CFG=${1:-c2_atrium}
for lib in $(ls radiance-ray-tracing_amd/librdx_stats*.so 2>/dev/null); do
  echo "=== $lib"; RDX_STATS_CFG=$CFG RDX_LIB=$PWD/$lib timeout -k 10 300 python tools/coop_stats.py 2>&1
done

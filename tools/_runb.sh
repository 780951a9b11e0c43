mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_gpu.log
for w in sample1 sponza; do timeout -k 10 200 python bench.py --steps 4 --warmup 1 --workload $w --no-cpu-baseline > gpurun_out/bv.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bv.json')); s=d['stage_ms_per_frame']; print('$w', d['value'], d['ms_per_step'], 'ext', s['extend'], 'shade', s['shade'], 'shd', s['shadow'], 'frac', d['roofline']['frac'])"; done

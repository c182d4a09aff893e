#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_link.py tests/test_gpu_pipeline.py tests/test_gpu_scale.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_seconds_per_step']; m=d['roofline']['match']; print(d['value'], d['ms_per_step'], 'match ms', m['device_ms_per_step'], 'frac', m['frac'], 'dist', m['distances_per_step'], {k:round(s[k],3) for k in ('extract','link','host_cpu_load_link','relax')})"
done

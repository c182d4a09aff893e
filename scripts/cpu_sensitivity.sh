#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python -m pytest tests/test_gpu_extract.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -1
for f in 0 1 0 1; do
if [ $f = 1 ]; then export OCHIP_AKAZE_COPY_PER_IMAGE=1; else unset OCHIP_AKAZE_COPY_PER_IMAGE; fi
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_seconds_per_step']; print('per-image copies $f:', d['value'], d['ms_per_step'], d['roofline']['staged']['stage_seconds']['extract'], {k:round(s[k],3) for k in ('extract','link','host_cpu_load_link','relax')})"
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_link.py tests/test_gpu_pipeline.py tests/test_gpu_refit.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -2
OCHIP_LINK_VERBOSE=1 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/tmp/err.log | tail -1 > gpurun_out/line.json
grep "\[link\]" /tmp/err.log | tail -2
python3 -c "
import json; d=json.load(open('gpurun_out/line.json')); print(d['value'], d['ms_per_step']); print(d['roofline']['staged']['stage_seconds']); print(d['stage_seconds_per_step'])"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-250

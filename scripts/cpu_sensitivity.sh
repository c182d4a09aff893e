#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for f in 1 0 1 0 1 0; do
OCHIP_DET_FAST=$f python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_seconds_per_step']; print('det fast $f:', d['value'], d['ms_per_step'], d['roofline']['staged']['stage_seconds']['extract'], {k:round(s[k],3) for k in ('extract','link','relax')})"
done

#!/bin/bash
# A/B sweep of the pipeline's environment knobs on the default bench (images/s, ms per step); STEPS / WARMUP from the environment
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "$1: $(env $1 python3 bench.py --steps ${STEPS:-10} --warmup ${WARMUP:-3} --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_seconds_per_step']; print(d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], {k:round(s[k],3) for k in ('extract','link','relax')})")"; }
for k in "$@"; do run "$k"; done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
which perf 2>/dev/null; 
OCHIP_EXTRACT_VERBOSE=1 OCHIP_BENCH_VERBOSE=1 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | grep -v "^{" | tail -40 | cut -c1-400

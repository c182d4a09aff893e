#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OCHIP_DESCRIBE_PROFILE=1 OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "\[describe\]" | tail -1
bash scripts/quick_extract_ab.sh 2>&1 | grep -v "copyBuffer\|render_views\|gather_kernel\|hamming\|ransac\|chol_tiles\|dense_match"

#!/bin/bash
# Kernel times of the link stage (match + RANSAC): two steps of the bench with the stages one after the other (one launch
# sequence, one link runner: a 9 000-pair launch each).  usage: r5_link_trace.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/linktrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0 OCHIP_BENCH_EXTRAS=0 OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/run.log 2>&1
grep '^{' $OUT/run.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value', d['value'], 'ms_per_step', d['ms_per_step'])"
f=$(ls -t $OUT/t/*/*_kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("ransac", "hamming", "refit", "chol_tiles", "relax_pair", "edge_lists", "decompose")):
        print("%-60s calls %4s avg %9.1f us" % (n.replace("(anonymous namespace)::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $OUT/t

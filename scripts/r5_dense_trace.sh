#!/bin/bash
# Kernel times of densifyMesh on the bench's survey (scripts/probe_dense.py C3).  usage: r5_dense_trace.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/densetrace_${1:-x}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/t -- python3 $R/scripts/probe_dense.py C3 > $OUT/run.log 2>&1
tail -3 $OUT/run.log
f=$(ls -t $OUT/t/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "dense" in r["Name"]:
        print("%-50s calls %4s total %9.2f ms avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:50], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -rf $OUT/t

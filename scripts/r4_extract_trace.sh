#!/bin/bash
# Per-kernel microseconds of the extract stage alone (100 views of the C2 grid, one launch sequence, 3 repeats).
# usage: r4_extract_trace.sh <tag>   (environment switches of the caller apply)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/xtrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0 OCHIP_EXTRACT_STREAMS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/scripts/extract_only.py 100 3 > $OUT/run.log 2>&1
cat $OUT/run.log | tail -4
f=$(ls -t $OUT/t/*/*_kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "render" not in r["Name"])
print("kernel time per image (render excluded): %.1f us" % (tot / 300 / 1e3))
for r in rows[:22]:
    print("%-50s calls %5s  %8.2f us/image" % (r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:50], r["Calls"], float(r["TotalDurationNs"]) / 300 / 1e3))
PY
rm -rf $OUT/t

#!/bin/bash
# kernel-time shares of one overlapped and one single-stream C3 bench (per-kernel totals, top 25)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/quick_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ovl -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/ovl.log 2>&1
f=$(ls -t $OUT/ovl/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:26]:
    print("%-70s calls %6s total %9.2f ms  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
grep -a '^{' $OUT/ovl.log | cut -c1-200

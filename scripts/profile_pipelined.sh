#!/bin/bash
# kernel-trace stats of the pipelined bench (relax in the shadow of the next survey's load + link): relax kernels only
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_pipe
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, os
f = max(glob.glob("gpurun_out/prof_pipe/trace/*/*_kernel_stats.csv"), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    nm = r["Name"].replace("(anonymous namespace)::", "")
    if any(k in nm for k in ("relax", "chol", "lm_", "back_solve", "plane")):
        print(f"{nm[:60]:60s} calls {int(r['Calls']):6d} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY

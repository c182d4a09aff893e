#!/bin/bash
# kernel-trace stats of the relax flavours on a C-sized survey (scripts/probe_relax_mesh.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_relax_mesh
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/probe_relax_mesh.py ${1:-C3} > $OUT/probe.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, os
f = max(glob.glob("gpurun_out/prof_relax_mesh/trace/*/*_kernel_stats.csv"), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:22]:
    nm = r['Name'].replace('(anonymous namespace)::', '')[:60]
    print(f"{nm:60s} calls {int(r['Calls']):6d} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
grep "^mesh\|^ground" $OUT/probe.log | tail -6

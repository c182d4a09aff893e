#!/bin/bash
# Per (kernel, grid) microseconds of the extract stage alone: which pyramid level a launch works on shows in its grid.
# usage: r5_level_trace.sh <tag> [filter]   (environment switches of the caller apply)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}
OUT=$R/gpurun_out/ltrace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0 OCHIP_EXTRACT_STREAMS=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/scripts/extract_only.py 100 3 > $OUT/run.log 2>&1
f=$(ls -t $OUT/t/*/*_kernel_trace.csv | head -1)
python3 - "$f" "${2:-}" <<'PY' | tee $OUT/levels.txt
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    if "render" in n: continue
    key = (n, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(a[1] for a in acc.values())
print("kernel time per image: %.1f us" % (tot / 300))
for (n, g), (c, t) in acc.items():
    if sys.argv[2] and sys.argv[2] not in n: continue
    print("%-44s grid %6d  calls %3d  %8.1f us/launch  %6.2f us/image" % (n[:44], g, c, t / c, t / 300))
PY
rm -rf $OUT/t

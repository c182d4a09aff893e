#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun).  Three separate rocprofv3 passes:
# kernel trace + stats, then FETCH_SIZE, then WRITE_SIZE (they do not fit one pass on gfx950).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.log 2>&1
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
for f in $(find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv"); do echo "== $f"; head -3 $f; python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as fh:
    for row in csv.DictReader(fh):
        k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
        acc[k][0] += 1
        acc[k][1] += float(row.get("Counter_Value", 0))
for k, v in acc.items():
    print(k, "dispatches", v[0], "sum", v[1], "per dispatch", v[1] / max(v[0], 1))
PY
done
tail -2 $OUT/bench_trace.log

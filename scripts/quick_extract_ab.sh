#!/bin/bash
# Quick A/B of the extract stage on the GPU box: extract parity tests, then a single-stream kernel trace of the C2
# bench (per-kernel microseconds without concurrent sequences) and the default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/quick_ab
rm -rf $OUT; mkdir -p $OUT
cd $R && timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_single -- python3 $R/bench.py --config C2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace_single.log 2>&1
f=$(ls -t $OUT/trace_single/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-60s calls %6s total %9.3f ms avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
cd $R && python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200

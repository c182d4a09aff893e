#!/bin/bash
# Round 4 A/B of the descriptor: extract parity tests, then a single-stream kernel trace of the C2 bench per variant
# (per-kernel microseconds without concurrent sequences).  usage: r4_extract_ab.sh [variants...]  (v2 default, v1 = OCHIP_DESCRIBE_V1)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r4_ab
rm -rf $OUT; mkdir -p $OUT
cd $R && timeout 900 python -m pytest tests/test_gpu_extract.py tests/test_gpu_extract_tail_device.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
for v in ${@:-v2 v1}; do
  if [ $v = v1 ]; then export OCHIP_DESCRIBE_V1=1; else unset OCHIP_DESCRIBE_V1; fi
  OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/bench.py --config C2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace_$v.log 2>&1
  f=$(ls -t $OUT/trace_$v/*/*_kernel_stats.csv | head -1)
  cp $f $OUT/kernel_stats_$v.csv
  echo "== $v"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("%-56s calls %5s total %9.3f ms avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:56], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
  rm -rf $OUT/trace_$v
done
unset OCHIP_DESCRIBE_V1
cd $R && python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300

#!/bin/bash
# kernel-trace stats of one C2 bench run (quick look at per-kernel times); extra env via "VAR=val ..." arguments
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r1b
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config ${BENCH_CONFIG:-C2} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
cd $R && python3 scripts/kstats.py 600 ${KSTATS_ROWS:-12}

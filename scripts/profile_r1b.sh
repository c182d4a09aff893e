#!/bin/bash
# Round-1 profile of the image path (extract + link + relax): kernel trace + stats on C2.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r1b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config ${1:-C2} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
tail -1 $OUT/bench_trace.log | cut -c1-1500

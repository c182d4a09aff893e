#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/det_exp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in 0 1 2 4 6 7; do
OCHIP_DET_DBG=$e OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$e -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/log$e 2>&1
f=$(ls -t $OUT/t$e/*/*_kernel_stats.csv | head -1)
echo "DBG $e: $(grep -a 'det_maxima_kernel<3>' $f | awk -F'",' '{print $2}' | cut -d, -f1-3)"
done

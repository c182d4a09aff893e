#!/bin/bash
# Round-1 profile of the image path (extract + link + relax), run on the GPU box through gpurun.
# Three separate rocprofv3 passes (kernel trace + stats; FETCH_SIZE; WRITE_SIZE do not fit one pass on gfx950).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r1c
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
OCHIP_PIPELINE_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_staged -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace_staged.log 2>&1
OCHIP_PIPELINE_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
OCHIP_PIPELINE_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.log 2>&1
python3 $R/scripts/summarise_profile.py $OUT $R/gpurun_out/prof_r1c_summary
grep -a '^{' $OUT/bench_trace.log | cut -c1-400

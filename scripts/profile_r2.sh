#!/bin/bash
# Round-2 profile of the image path (extract + link + relax), run on the GPU box through gpurun: kernel trace + stats
# of the default bench (overlapped), the staged variant, a single-stream variant (per-kernel microseconds without
# concurrent sequences), and FETCH_SIZE / WRITE_SIZE in passes of their own (no trace domain combined with --pmc).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
OCHIP_PIPELINE_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_staged -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace_staged.log 2>&1
OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_single -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trace_single.log 2>&1
OCHIP_PIPELINE_OVERLAP=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
OCHIP_PIPELINE_OVERLAP=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_write.log 2>&1
python3 $R/scripts/summarise_profile.py $OUT $R/gpurun_out/r02_e2e
f=$(ls $OUT/trace_single/*/*_kernel_stats.csv | tail -1); [ -n "$f" ] && cp $f $R/gpurun_out/r02_e2e_single_stream_kernel_stats.csv
grep -a '^{' $OUT/bench_trace.log | cut -c1-300
ls -la $R/gpurun_out/r02_e2e*

# Round 3, final code: per-kernel microseconds of the default bench configuration with one launch sequence and one link
# runner at a time (the r03_e2e_single_stream_kernel_stats.csv of profiles/README.md), and the same with the default knobs
# (overlapped stages).  Program directly after "--"; csv output (the database output's post-processing hangs).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r3_final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0 OCHIP_BENCH_EXTRAS=0
OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -- python3 $R/bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/single.log 2>&1
cp $(ls $OUT/single/*/*_kernel_stats.csv | head -1) $R/gpurun_out/r03_final_e2e_single_stream_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/overlap -- python3 $R/bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/overlap.log 2>&1
cp $(ls $OUT/overlap/*/*_kernel_stats.csv | head -1) $R/gpurun_out/r03_final_e2e_kernel_stats.csv
grep "\"metric\"" $OUT/single.log | tail -n 1 | cut -c1-300
grep "\"metric\"" $OUT/overlap.log | tail -n 1 > $R/gpurun_out/r03_final_e2e_bench_line.json
head -5 $R/gpurun_out/r03_final_e2e_kernel_stats.csv | cut -c1-160

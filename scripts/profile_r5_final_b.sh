R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_final_b
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ovl -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r05_e2e_bench_line.json 2> $OUT/ovl.err
cp $(ls -t $OUT/ovl/*/*_kernel_stats.csv | head -1) $OUT/r05_e2e_kernel_stats.csv; rm -rf $OUT/ovl
bash $R/scripts/r5_dense_trace.sh final > $OUT/r05_dense_kernel_trace.txt 2>&1
grep -i "densify" $R/gpurun_out/densetrace_final/run.log >> $OUT/r05_dense_kernel_trace.txt
tail -8 $OUT/r05_dense_kernel_trace.txt
head -c 200 $OUT/r05_e2e_bench_line.json

#!/bin/bash
# Round-3 counter evidence for the extract kernels on the CURRENT code: LDS bank-conflict cycles
# against LDS-active cycles, VALU / wave cycles, L1 -> L2 requests, HBM bytes - per kernel over one C2 step (200 images),
# one launch sequence at a time.  One rocprofv3 --pmc pass per counter group (no trace domains combined with --pmc),
# program directly after "--".
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r3_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
export OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 OCHIP_BENCH_EXTRAS=0
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/g$i.log 2>&1
  echo "group $i [$grp] rc=$?" >> $OUT/groups.txt
done
python3 $R/scripts/summarise_pmc.py $OUT $R/gpurun_out/r03_extract_pmc_counters.json > $OUT/summary.txt 2>&1
# and the per-kernel microseconds of the same configuration
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
cp $(ls $OUT/trace/*/*_kernel_stats.csv | head -1) $R/gpurun_out/r03_e2e_single_stream_kernel_stats.csv
head -30 $R/gpurun_out/r03_e2e_single_stream_kernel_stats.csv
cat $OUT/groups.txt

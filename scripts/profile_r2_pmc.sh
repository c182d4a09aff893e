#!/bin/bash
# Round-2 counter evidence for the extract kernels: L2 hit/miss, L1->L2 requests, TA stalls, VALU
# instruction counts and wave cycles per kernel over one C2 step (200 images).  One rocprofv3 --pmc pass per counter group
# (no trace domains combined with --pmc), program directly after "--".
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r2_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# read by the HIP runtime when rocprofv3's preloaded library initialises it, i.e. before python starts
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1
export OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/g$i.log 2>&1
  echo "group $i [$grp] rc=$?" >> $OUT/groups.txt
done
python3 $R/scripts/summarise_pmc.py $OUT $R/gpurun_out/r02_extract_pmc_counters.json

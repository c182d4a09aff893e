#!/bin/bash
# Round-4 counter evidence for the descriptor kernels (one C2 step = 200 images, one launch sequence at a time):
# instruction mix, wait/active split, LDS activity, L1 (TCP) accesses and L1 -> L2 requests, texture-addresser stalls,
# HBM bytes.  One rocprofv3 --pmc pass per counter group; nothing but the program after "--".
# usage: profile_r4_pmc.sh <tag> [kernel regex]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4}
RE=${2:-describe|orient|kp_frame}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
export OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 OCHIP_BENCH_EXTRAS=0
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-include-regex "$RE" --output-format csv -d $OUT/g$i -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/g$i.log 2>&1
  echo "group $i [$grp] rc=$?" >> $OUT/groups.txt
done
python3 $R/scripts/summarise_pmc.py $OUT $R/gpurun_out/${TAG}_pmc_counters.json > $OUT/summary.txt 2>&1
cat $OUT/groups.txt
python3 - $R/gpurun_out/${TAG}_pmc_counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in d.items():
    print("==", k)
    for c in sorted(v):
        print("   %-40s %16.0f" % (c, v[c]))
PY
rm -rf $OUT/g*/

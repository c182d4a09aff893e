#!/bin/bash
# Counter passes over ransac_homography_kernel (one C2 step of the bench, one link runner, no overlap).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04r
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0 OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 OCHIP_BENCH_EXTRAS=0
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-include-regex "ransac_homography" --output-format csv -d $OUT/g$i -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/g$i.log 2>&1
  echo "group $i rc=$?" >> $OUT/groups.txt
done
python3 $R/scripts/summarise_pmc.py $OUT $R/gpurun_out/r04r_pmc_counters.json > $OUT/summary.txt 2>&1
python3 - $R/gpurun_out/r04r_pmc_counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in d.items():
    print("==", k)
    w = max(v.get("SQ_WAVES", 0), 1)
    for c in sorted(v):
        print("   %-28s %16.0f   per wave %12.1f" % (c, v[c], v[c] / w))
PY
rm -rf $OUT/g*/

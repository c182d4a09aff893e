#!/bin/bash
# Counter passes over the extract stage alone (scripts/extract_only.py: 100 views of the C2 grid, extracted twice).
# usage: profile_r4_extract_pmc.sh <tag> <kernel regex> [counter groups, ';'-separated]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4x}
RE=${2:-describe}
GROUPS_ARG=${3:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY;SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA;TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0 OCHIP_EXTRACT_STREAMS=1
i=0
IFS=';' read -ra GRPS <<< "$GROUPS_ARG"
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "$RE" --output-format csv -d $OUT/g$i -- python3 $R/scripts/extract_only.py 100 1 > $OUT/g$i.log 2>&1
  echo "group $i [$grp] rc=$?" >> $OUT/groups.txt
done
python3 $R/scripts/summarise_pmc.py $OUT $R/gpurun_out/${TAG}_pmc_counters.json > $OUT/summary.txt 2>&1
cat $OUT/groups.txt
python3 - $R/gpurun_out/${TAG}_pmc_counters.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in d.items():
    print("==", k)
    w = max(v.get("SQ_WAVES", 0), 1)
    for c in sorted(v):
        print("   %-36s %16.0f   per wave %10.1f" % (c, v[c], v[c] / w))
PY
rm -rf $OUT/g*/

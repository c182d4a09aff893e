#!/bin/bash
# Counter passes + a kernel trace of the matcher alone (scripts/match_only.py).  usage: profile_r4_match_pmc.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4m}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
python3 $R/scripts/match_only.py 128 3500 5 4 2>&1 | tail -4
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-include-regex "mfma" --output-format csv -d $OUT/g$i -- python3 $R/scripts/match_only.py 128 3500 5 1 > $OUT/g$i.log 2>&1
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

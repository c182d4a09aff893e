#!/bin/bash
# Round-5 evidence for profiles/ (final code): per-kernel times of the extract stage alone, HBM bytes and vector instructions
# of the extract sequence (separate counter passes, no trace domain), counters of the strip kernels and the descriptor, the
# kernel trace of the default bench with the line it printed.  Every rocprofv3 call has the program right after "--".
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05_final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
# 1. extract alone, one sequence at a time
bash $R/scripts/r4_extract_trace.sh final > $OUT/extract_only_trace.txt 2>&1
cp $R/gpurun_out/xtrace_final/kernel_stats.csv $OUT/r05_extract_only_kernel_stats.csv
# 2. HBM bytes and vector instructions of the extract sequence
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  OCHIP_EXTRACT_STREAMS=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/hbm/g_$tag -- python3 $R/scripts/extract_only.py 100 1 > $OUT/pmc_$tag.log 2>&1
done
python3 $R/scripts/summarise_pmc.py $OUT/hbm $OUT/hbm_counters.json > /dev/null 2>&1
python3 $R/scripts/summarise_r4_hbm.py $OUT/hbm_counters.json $OUT/r05_e2e_pmc_hbm.json
python3 $R/scripts/summarise_r5_valu.py $OUT/hbm_counters.json $OUT/r05_extract_valu.json
rm -rf $OUT/hbm
# 3. counters of the round's kernels
bash $R/scripts/r5_pmc.sh r05s "level_strip_kernel|det_strip_kernel|describe3" > $OUT/strip_pmc.txt 2>&1
cp $R/gpurun_out/r05s_pmc_counters.json $OUT/r05_strip_describe_pmc.json
# 4. the default bench under the tracer, and one launch sequence / one link runner / one survey at a time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ovl -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r05_e2e_bench_line.json 2> $OUT/ovl.err
cp $(ls -t $OUT/ovl/*/*_kernel_stats.csv | head -1) $OUT/r05_e2e_kernel_stats.csv; rm -rf $OUT/ovl
OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 OCHIP_BENCH_EXTRAS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/single.json 2> $OUT/single.err
cp $(ls -t $OUT/single/*/*_kernel_stats.csv | head -1) $OUT/r05_e2e_single_stream_kernel_stats.csv; rm -rf $OUT/single
# 5. the RANSAC kernel's counters (one C2 step of the bench, one link runner, no overlap)
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 OCHIP_BENCH_EXTRAS=0 timeout 600 rocprofv3 --pmc $grp --kernel-include-regex "ransac_homography" --output-format csv -d $OUT/rpmc/g$i -- python3 $R/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/rpmc_g$i.log 2>&1
done
python3 $R/scripts/summarise_pmc.py $OUT/rpmc $OUT/r05_ransac_pmc.json > /dev/null 2>&1
rm -rf $OUT/rpmc
tail -30 $OUT/extract_only_trace.txt
python3 - $OUT/r05_ransac_pmc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in d.items():
    print("==", k[:60], "SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.2f" % (v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)), "HBM read+write bytes", v.get("FETCH_SIZE"), v.get("WRITE_SIZE"))
PY
python3 -c "
import json; d=json.load(open('$OUT/r05_e2e_pmc_hbm.json')); print('HBM bytes per image', d['extract_hbm_bytes_per_image'], d['calibration'])
d=json.load(open('$OUT/r05_extract_valu.json')); print('VALU wave-instructions per image', d['extract_valu_wave_instructions_per_image'], 'issue us', d['extract_issue_us_per_image_at_2.4GHz'])"
head -c 300 $OUT/r05_e2e_bench_line.json

#!/bin/bash
# Round-6 evidence for profiles/ (final code): per-kernel times of the extract stage alone, HBM bytes of the extract sequence
# (separate counter passes, no trace domain), the kernel trace of the default bench with the line it printed, and the kernel
# trace of INITIAL_PROCESSING in the reference's schedule (the resident bootstrap launch, csrc/relax_chain.hip).  Every
# rocprofv3 call has the program right after "--".
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
# 1. extract alone, one sequence at a time
bash $R/scripts/r4_extract_trace.sh r06 > $OUT/extract_only_trace.txt 2>&1
cp $R/gpurun_out/xtrace_r06/kernel_stats.csv $OUT/r06_extract_only_kernel_stats.csv
# 2. HBM bytes and vector instructions of the extract sequence
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  OCHIP_EXTRACT_STREAMS=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/hbm/g_$tag -- python3 $R/scripts/extract_only.py 100 1 > $OUT/pmc_$tag.log 2>&1
done
python3 $R/scripts/summarise_pmc.py $OUT/hbm $OUT/hbm_counters.json > /dev/null 2>&1
python3 $R/scripts/summarise_r4_hbm.py $OUT/hbm_counters.json $OUT/r06_e2e_pmc_hbm.json
python3 $R/scripts/summarise_r5_valu.py $OUT/hbm_counters.json $OUT/r06_extract_valu.json
rm -rf $OUT/hbm
# 3. the default bench under the tracer (durations stretched by the concurrency) and one launch sequence / one link runner / one
#    survey at a time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ovl -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r06_e2e_bench_line.json 2> $OUT/ovl.err
cp $(ls -t $OUT/ovl/*/*_kernel_stats.csv | head -1) $OUT/r06_e2e_kernel_stats.csv; rm -rf $OUT/ovl
OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 OCHIP_BENCH_EXTRAS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/single.json 2> $OUT/single.err
cp $(ls -t $OUT/single/*/*_kernel_stats.csv | head -1) $OUT/r06_e2e_single_stream_kernel_stats.csv; rm -rf $OUT/single
# 4. INITIAL_PROCESSING in the reference's schedule: the stepper's three runs under the tracer
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ip -- python3 $R/scripts/probe_initial_processing.py C3 100 > $OUT/r06_initial_processing_under_rocprof.txt 2> $OUT/ip.err
cp $(ls -t $OUT/ip/*/*_kernel_stats.csv | head -1) $OUT/r06_initial_processing_kernel_stats.csv; rm -rf $OUT/ip
# 5. the same without the tracer, with the chain's own phase clock (OCHIP_VERBOSE=relax on one batch of each kind)
python3 $R/scripts/probe_initial_processing.py C3 100 > $OUT/r06_initial_processing.txt 2>&1
python3 $R/scripts/probe_incremental.py C3 100 2 1 > $OUT/chain_batch2.log 2>&1
python3 $R/scripts/probe_incremental.py C3 100 6 1 > $OUT/chain_batch6.log 2>&1
(echo "== a batch bootstrapped with the group (batch 2 of 10):"; grep "relax chain" $OUT/chain_batch2.log; echo "== a batch bootstrapped one camera at a time (batch 6 of 10):"; grep "relax chain" $OUT/chain_batch6.log) > $OUT/r06_chain_phase_clock.txt
# 6. the from-host sort fix: the e2e trace's sort kernels
grep -i "sort_" $OUT/r06_e2e_kernel_stats.csv | head -8 > $OUT/sort_lines.txt
tail -28 $OUT/extract_only_trace.txt
head -c 400 $OUT/r06_e2e_bench_line.json; echo
tail -3 $OUT/r06_initial_processing.txt
cat $OUT/r06_chain_phase_clock.txt
grep -i "plane_chain" $OUT/r06_initial_processing_kernel_stats.csv | head -3
python3 -c "
import json; d=json.load(open('$OUT/r06_e2e_pmc_hbm.json')); print('HBM bytes per image', d['extract_hbm_bytes_per_image'], d['calibration'])"
# 7. the suppression's rounds, workgroup times and waiting points per (pass, level) (profiles/r06_suppression_rounds.txt quotes them)
OCHIP_EXTRACT_STREAMS=1 OCHIP_VERBOSE=extract python3 $R/scripts/extract_only.py 100 1 2>&1 | grep -i "suppression\|repeat" > $OUT/r06_suppression_levels.txt
grep -i "suppress" $OUT/r06_extract_only_kernel_stats.csv | cut -c1-200 >> $OUT/r06_suppression_levels.txt

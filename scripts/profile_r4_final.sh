#!/bin/bash
# Round-4 evidence for profiles/ (final code): per-kernel times of the extract stage alone, HBM bytes of the extract sequence
# (FETCH_SIZE / WRITE_SIZE passes), counters of the descriptor and of the matcher, kernel traces of the default bench
# (overlapped, and one launch sequence at a time) with the line it printed.  Every rocprofv3 call has the program right
# after "--"; counter passes carry no trace domain.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16 HSA_ENABLE_IPC_MODE_LEGACY=0
# 1. extract alone, one sequence at a time
bash $R/scripts/r4_extract_trace.sh final > $OUT/extract_only_trace.txt 2>&1
cp $R/gpurun_out/xtrace_final/kernel_stats.csv $OUT/r04_extract_only_kernel_stats.csv
# 2. HBM bytes of the extract sequence
for c in FETCH_SIZE WRITE_SIZE; do
  OCHIP_EXTRACT_STREAMS=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/hbm/g_$c -- python3 $R/scripts/extract_only.py 100 1 > $OUT/hbm_$c.log 2>&1
done
python3 $R/scripts/summarise_pmc.py $OUT/hbm $OUT/hbm_counters.json > /dev/null 2>&1
python3 $R/scripts/summarise_r4_hbm.py $OUT/hbm_counters.json $OUT/r04_e2e_pmc_hbm.json
rm -rf $OUT/hbm
# 3. descriptor and matcher counters
bash $R/scripts/profile_r4_extract_pmc.sh r04d describe > $OUT/describe_pmc.txt 2>&1
cp $R/gpurun_out/r04d_pmc_counters.json $OUT/r04_describe_pmc.json
bash $R/scripts/profile_r4_match_pmc.sh r04m > $OUT/match_pmc.txt 2>&1
cp $R/gpurun_out/r04m_pmc_counters.json $OUT/r04_match_pmc.json
# 4. the default bench under the tracer: overlapped, then one launch sequence / one link runner / one survey at a time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ovl -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/r04_e2e_bench_line.json 2> $OUT/ovl.err
cp $(ls -t $OUT/ovl/*/*_kernel_stats.csv | head -1) $OUT/r04_e2e_kernel_stats.csv; rm -rf $OUT/ovl
OCHIP_PIPELINE_OVERLAP=0 OCHIP_EXTRACT_STREAMS=1 OCHIP_LINK_RUNNERS=1 OCHIP_BENCH_EXTRAS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/single.json 2> $OUT/single.err
cp $(ls -t $OUT/single/*/*_kernel_stats.csv | head -1) $OUT/r04_e2e_single_stream_kernel_stats.csv; rm -rf $OUT/single
tail -30 $OUT/extract_only_trace.txt
tail -12 $OUT/describe_pmc.txt
head -c 300 $OUT/r04_e2e_bench_line.json

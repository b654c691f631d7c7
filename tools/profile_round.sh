#!/bin/bash
# Round profile: per-kernel time table of the bench workload + the PMC passes (separate passes, counters only
# with --kernel-trace, as MI355X_MICROARCH.md prescribes).  Run on the GPU box through gpurun:
#   gpurun -- 'bash tools/profile_round.sh r02_c'
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=${1:-r02}
out=gpurun_out
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
B="python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-recall --no-live-traffic --preheat-seconds 2"
rocprofv3 --kernel-trace --stats -d $out/${tag}_prof -o r -- $B > $out/${tag}_prof.log 2>&1
db=$(find $out/${tag}_prof -name "*_results.db" | head -1)
python tools/rocpd_stats.py $db --csv $out/${tag}_kernel_stats.csv > /dev/null
# MFMA utilisation inside the step
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out/${tag}_pmc_mfma -o r -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-recall --no-live-traffic --preheat-seconds 0.5 > $out/${tag}_pmc_mfma.log 2>&1
db=$(find $out/${tag}_pmc_mfma -name "*_results.db" | head -1)
python tools/pmc_mfma_util.py $db > $out/${tag}_pmc_mfma_util.csv
# HBM traffic: GEMM shapes and gather / scatter kernels, FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_gemm_$c -o r -- python tools/gemm_pmc.py > $out/${tag}_pmc_gemm_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_gather_$c -o r -- python tools/gather_pmc.py > $out/${tag}_pmc_gather_$c.log 2>&1
done
python tools/pmc_table.py $(find $out/${tag}_pmc_gemm_FETCH_SIZE $out/${tag}_pmc_gemm_WRITE_SIZE -name "*counter_collection.csv") > $out/${tag}_pmc_gemm_fetch_write.csv
python tools/pmc_table.py $(find $out/${tag}_pmc_gather_FETCH_SIZE $out/${tag}_pmc_gather_WRITE_SIZE -name "*counter_collection.csv") > $out/${tag}_pmc_gather_fetch_write.csv
head -30 $out/${tag}_kernel_stats.csv
cat $out/${tag}_pmc_mfma_util.csv | head -30
cat $out/${tag}_pmc_gemm_fetch_write.csv
cat $out/${tag}_pmc_gather_fetch_write.csv

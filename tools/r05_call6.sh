set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for tag in plain budget; do
  flag=""; [ $tag = budget ] && flag="--budget"
  rocprofv3 --kernel-trace --stats -d gpurun_out/occ_$tag -o r -- python tools/occupier_curve.py --steps 30 --warmup 5 --only 16 $flag --out gpurun_out/r05_f_occ16_$tag.json > gpurun_out/r05_f_occ_$tag.log 2>&1
  db=$(find gpurun_out/occ_$tag -name "*_results.db" | head -1)
  python tools/rocpd_stats.py $db --csv gpurun_out/r05_f_occ16_${tag}_kernel_stats.csv > /dev/null
  rm -rf gpurun_out/occ_$tag
  head -22 gpurun_out/r05_f_occ16_${tag}_kernel_stats.csv | cut -c1-140
done

# per-kernel times of the two priors alone (1024 x 300 frames, 5 runs) for one or more builds: VARIANTS="new qkv" (tools/_lib_<v>.so)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-new}; do
  rm -rf /tmp/prof_pri_$v
  GLAMR_LIB_PATH=$R/tools/_lib_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pri_$v -- python $R/tools/priors_ab.py > /dev/null 2>&1
  f=$(ls /tmp/prof_pri_$v/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r05_priors_kernel_stats_$v.csv
  echo "== $v"; head -12 $f | cut -c1-150
done

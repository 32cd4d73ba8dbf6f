cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GLAMR_LIB_PATH=$R/tools/_lib_new.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pri -- python $R/tools/priors_ab.py > /dev/null 2>&1
f=$(ls /tmp/prof_pri/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r05_priors_kernel_stats.csv; head -16 $f | cut -c1-140

# A/B of two builds of the library on ONE box (tools/_lib_base.so = round 4's kernel, tools/_lib_new.so = the working tree): bit fingerprints of
# every output of the optimiser stage over seven cases + launch time of 1 / 1024 scenes x 200 iterations, alternating the builds twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_ab.log
: > $L
if [ -x tools/_scan_probe_dpp ]; then timeout 120 tools/_scan_probe_dpp 2>&1 | tail -4 >> $L; fi
for v in base new; do
  echo "== bits $v" >> $L
  GLAMR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lib_$v.so timeout 600 python tools/stage_bits.py 2>&1 | grep "^bits\|Error\|error" >> $L
done
for rep in 1 2; do
  for v in base new; do
    echo "== time $v (rep $rep)" >> $L
    GLAMR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lib_$v.so timeout 300 python tools/stage_bits.py --time --cases=dyn120 2>&1 | grep "^time" >> $L
  done
done
cat $L

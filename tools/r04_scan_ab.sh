# DPP row-shift scans (tools/_lib_new.so = the product build) against the ds_bpermute shuffle scans of rounds 1-3 (tools/_lib_base.so) on ONE box:
# the stage kernel alone, the co-scheduled bench, and the K-step parity states of both builds
cd $GRAFT_REPO_ROOT
L=gpurun_out/r04_scan_dpp_ab.log
echo "base = shuffle scans (rounds 1-3), new = DPP scans (round 4)" > $L
bash tools/r04_ab.sh 2>&1 | grep "grecon stage\|==" >> $L
bash tools/r04_ab_bench.sh 2>&1 | grep "rep" >> $L
for v in base new; do
  echo "== K-step parity states, $v" >> $L
  GLAMR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lib_$v.so timeout 600 python -m pytest tests/test_grecon_gpu.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "K-step|passed|failed" | sed 's/^\.//' >> $L
done
cat $L

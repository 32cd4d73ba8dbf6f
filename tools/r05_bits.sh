# bit fingerprints of several builds on one box: bash tools/r05_bits.sh "<variants>" "<cases>" [--time]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r05_bits.log
: > $L
for v in $1; do
  echo "== $v" >> $L
  GLAMR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lib_$v.so timeout 600 python tools/stage_bits.py --cases=$2 $3 2>&1 | grep "^bits\|^time\|Error\|error" >> $L
done
cat $L

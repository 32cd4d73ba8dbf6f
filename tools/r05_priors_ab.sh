# A/B of two builds of the library on one box: the two priors on 1024 x 300 frames (time + output fingerprints), alternating twice
cd $GRAFT_REPO_ROOT
L=gpurun_out/r05_priors_ab.log
: > $L
for rep in 1 2; do
  for v in ${VARIANTS:-new qkv}; do
    echo "== $v (rep $rep)" >> $L
    GLAMR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lib_$v.so timeout 300 python tools/priors_ab.py 2>&1 | grep "^priors\|^bits\|rror" >> $L
  done
done
cat $L

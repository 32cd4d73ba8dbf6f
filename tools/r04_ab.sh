# A/B of two builds of the library on ONE box: stage kernel alone (1, 256, 1024 scenes x 200 iterations), alternating the builds twice
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in base new; do
    echo "== $v (rep $rep)"
    GLAMR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lib_$v.so GLAMR_MB_FRAMES=300 GLAMR_MB_SCENES=1,256,1024 python tools/microbench.py 2>&1 | grep "grecon stage"
  done
done

cd $GRAFT_REPO_ROOT
for T in 192 256 260 300; do
  GLAMR_MB_FRAMES=$T GLAMR_MB_SCENES=1,256 python tools/microbench.py 2>&1 | grep "grecon stage"
done

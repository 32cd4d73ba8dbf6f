# bench.py --steps 20 for several builds of the library, alternated REPS times on one box: VARIANTS="new x y" REPS=3
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${REPS:-3}); do
  for v in ${VARIANTS:-new}; do
    GLAMR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lib_$v.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-lines --no-strong-line 2>/tmp/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step'],3), d['pipeline']['stage_launch_ms_beside_the_priors'])" || tail -5 /tmp/err_$v.log
  done
done

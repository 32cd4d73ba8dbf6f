# A/B of the gate position relative to the per-person preparation (GLAMR_GATE_PREP=late: the whole batch waits; early: only the priors do)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in late early; do
    GLAMR_GATE_PREP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-lines --no-strong-line 2>/tmp/err_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step'],3), d['pipeline']['stage_launch_ms_beside_the_priors'])" || tail -5 /tmp/err_$v.log
  done
done

# A/B of two builds of the library on ONE box through the co-scheduled bench (headline configuration), alternating twice
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in base new; do
    GLAMR_LIB_PATH=$GRAFT_REPO_ROOT/tools/_lib_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-lines --no-host-stream --no-strong-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v rep $rep: %.2f ms/step, %d seq/s, stage alone %.2f ms, beside the priors %s' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['pipeline']['stage_launch_ms_beside_the_priors']))"
  done
done

# the headline pipeline at other batch sizes (sequences per step and GPU)
cd $GRAFT_REPO_ROOT
for b in 256 512 1024 2048; do
  python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-lines --no-host-stream --no-strong-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('batch $b: %.2f ms/step, %d seq/s, stage alone %.2f ms, beside the priors %s' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['pipeline']['stage_launch_ms_beside_the_priors']))"
done

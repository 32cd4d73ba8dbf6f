#!/bin/bash
# development aid (GPU): A/B of stage-kernel variants on ONE box -- tools/stage_ab.sh <name> [<name> ...]   (tools/_lib_<name>.so from tools/build_variant.sh;
# "base" = the library in the package).  Each variant twice, alternating: optimiser-stage launch alone (ms per 1024 scenes x 500 iterations) and the pipelined step.
mkdir -p gpurun_out/ab
for rep in 1 2; do for v in "$@"; do
  if [ $v = base ]; then unset GLAMR_LIB_PATH; else export GLAMR_LIB_PATH=tools/_lib_$v.so; fi
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-lines --no-strong-line > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err
  python -c "
import json; d=json.loads(open('gpurun_out/ab/$v.json').read().strip().splitlines()[-1]); r=d['roofline']; print('%-12s stage alone %s ms = %.2f us per scene-iteration | step %.2f ms %d seq/s | stage beside the priors %s' % ('$v', r['launch_ms_each'], r['us_per_scene_iteration'], d['ms_per_step'], d['value'], d.get('pipeline',{}).get('stage_launch_ms_beside_the_priors')))" || tail -3 gpurun_out/ab/$v.err
done; done

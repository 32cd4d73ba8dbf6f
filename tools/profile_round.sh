#!/bin/bash
# End-of-round profile set on one GPU box: GPU tests, kernel stats of the single-stream and the co-scheduled bench, PMC passes of the stage kernel and
# of the priors, the pipeline's gap trace, the default bench line and the two-graph cut.  Everything lands in gpurun_out/<tag>_*; copy what is to be
# judged into profiles/.   usage: GLAMR_ROUND_TAG=r06 bash tools/profile_round.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${GLAMR_ROUND_TAG:-r06}
O=$R/gpurun_out
mkdir -p $O
python -m pytest $R/tests -m gpu -q -rA --durations=15 -p no:cacheprovider > $O/${T}_gputest.log 2>&1      # (whole log: RCCL prints a banner at exit, after pytest's summary line)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --streams 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-lines --no-host-stream --no-strong-line > $O/${T}_bench_b1024_under_rocprof.json 2>/dev/null
cp $(ls /tmp/prof_bench/*/*kernel_stats.csv | head -1) $O/${T}_bench_b1024_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cos -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-lines --no-strong-line > $O/${T}_bench_coscheduled_under_rocprof.json 2>/dev/null < /dev/null
f=$(ls /tmp/prof_cos/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then cp "$f" $O/${T}_bench_coscheduled_kernel_stats.csv; fi
f=$(ls /tmp/prof_cos/*/*kernel_trace.csv 2>/dev/null | head -1); if [ -n "$f" ]; then python $R/tools/gap_trace.py $f > $O/${T}_gap_trace.log 2>&1; fi
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_smpl -- python $R/tools/smpl_profile.py > /dev/null 2>&1
cp $(ls /tmp/prof_smpl/*/*kernel_stats.csv | head -1) $O/${T}_smpl_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lat -- python $R/tools/latent_nodes.py 20 > /dev/null 2>&1
f=$(ls /tmp/prof_lat/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then cp "$f" $O/${T}_latent_kernel_stats.csv; fi
cd $R && python tools/smpl_ab.py shipped > $O/${T}_smpl_ab.log 2>&1; python tools/latent_time.py shipped >> $O/${T}_smpl_ab.log 2>&1; python tools/configs3_time.py shipped >> $O/${T}_smpl_ab.log 2>&1
cd $R && GLAMR_ROUND_TAG=$T python tools/collect_pmc.py > $O/${T}_collect_pmc.log 2>&1
GLAMR_ROUND_TAG=$T python tools/collect_pmc.py priors > $O/${T}_collect_pmc_priors.log 2>&1
GLAMR_NETS_FREE=1 GLAMR_ROUND_TAG=${T}cos python tools/collect_pmc.py priors > $O/${T}cos_collect_pmc_priors.log 2>&1
cp $O/${T}_pmc_stage_kernel.json $R/profiles/ 2>/dev/null      # (bench.py reads the newest profiles/rNN_pmc_stage_kernel.json: the lines below use this round's)
python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err
python bench.py --no-early-prep --no-cpu-baseline --no-kernel-lines --no-strong-line > $O/${T}_bench_no_early_prep.json 2> $O/${T}_bench_no_early_prep.err
grep -E " passed| failed" $O/${T}_gputest.log | tail -2; head -8 $O/${T}_bench_b1024_kernel_stats.csv | cut -c1-160
python - <<PY
import json
for n in ('default', 'no_early_prep'):
    try:
        d = json.loads(open('$O/${T}_bench_%s.json' % n).read().strip().splitlines()[-1])
        r = d['roofline']
        print(n, round(d['value']), round(d['ms_per_step'], 2), 'frac', round(r['frac'], 4), 'stage ms', r['launch_ms_each'], d['replay_check'])
    except Exception as e:
        print(n, 'failed', e)
PY

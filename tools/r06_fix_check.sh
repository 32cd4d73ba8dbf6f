#!/bin/bash
mkdir -p gpurun_out/fix
echo "== race_mini, whole priors beside the skinning (library without packed-fp32 instructions)"
timeout 300 python tools/race_mini.py 0 36 nets > gpurun_out/fix/mini.log 2>&1; grep SUMMARY gpurun_out/fix/mini.log
echo "== race_probe, old launch order + three-graph cut (failed 8 of 8 pairs before)"
GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early timeout 400 python tools/race_probe.py 1024 12 > gpurun_out/fix/probe_old_early.log 2>&1; grep SUMMARY gpurun_out/fix/probe_old_early.log
echo "== race_probe, old launch order + two-graph cut"
GLAMR_SKIN_AFTER_PRIORS=1 timeout 400 python tools/race_probe.py 1024 12 > gpurun_out/fix/probe_old_late.log 2>&1; grep SUMMARY gpurun_out/fix/probe_old_late.log
echo "== race_probe, current order + three-graph cut"
GLAMR_GATE_PREP=early timeout 400 python tools/race_probe.py 1024 12 > gpurun_out/fix/probe_new_early.log 2>&1; grep SUMMARY gpurun_out/fix/probe_new_early.log
echo "== 48 sequences, three-graph cut (the case that failed about 1 in 6)"
GLAMR_GATE_PREP=early timeout 400 python tools/race_probe.py 48 40 > gpurun_out/fix/probe_48_early.log 2>&1; grep SUMMARY gpurun_out/fix/probe_48_early.log
GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early timeout 400 python tools/race_probe.py 48 40 > gpurun_out/fix/probe_48_old_early.log 2>&1; grep SUMMARY gpurun_out/fix/probe_48_old_early.log
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/fix/bench.json 2> gpurun_out/fix/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/fix/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('replay_check'), d['roofline'])"
tail -3 gpurun_out/fix/bench.err

#!/bin/bash
mkdir -p gpurun_out/race10
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-8} > gpurun_out/race10/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY" gpurun_out/race10/$name.log; grep "blocks of smpl_prep" gpurun_out/race10/$name.log | sed 's/.*blocks of smpl_prep_kernel//' | cut -c1-40 | head -6 | tr '\n' ' '; echo; tail -2 gpurun_out/race10/$name.log | grep -i "error\|Traceback" ; }
run b_none GLAMR_PROBE_B=none
run b_sleep GLAMR_PROBE_B=sleep
run b_mm GLAMR_PROBE_B=mm
run b_add GLAMR_PROBE_B=add
run b_nets GLAMR_PROBE_B=nets

#!/bin/bash
mkdir -p gpurun_out/race11
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early GLAMR_PROBE_B=nets GLAMR_LIB_PATH=tools/_lib_spin.so
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-6} > gpurun_out/race11/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY" gpurun_out/race11/$name.log; grep "blocks of smpl_prep" gpurun_out/race11/$name.log | sed 's/.*blocks of smpl_prep_kernel//' | cut -c1-40 | head -6 | tr '\n' ' '; echo; tail -3 gpurun_out/race11/$name.log | grep -i "error\|Traceback" ; }
run full X=1
for p in 1 2 3 5 6 7 4; do run stop$p GLAMR_PROBE_B_ENV=GLAMR_NETS_PROBE_STOP=$p; done

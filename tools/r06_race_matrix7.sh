#!/bin/bash
mkdir -p gpurun_out/race7
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-10} > gpurun_out/race7/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY" gpurun_out/race7/$name.log; grep -c "j_local rows" gpurun_out/race7/$name.log; grep "probe records" gpurun_out/race7/$name.log | sort | uniq -c | sort -rn | head -5; }
run lvl128 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=128
run lvl128b GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=128
run lvl132 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=132
run lvl0 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=0

#!/bin/bash
mkdir -p gpurun_out/race5
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-12} > gpurun_out/race5/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY" gpurun_out/race5/$name.log; grep "probe records" gpurun_out/race5/$name.log | sort | uniq -c | sort -rn | head -4; }
run lvl64 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=64
run lvl68 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=68
run lvl64snap GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=64 GLAMR_PROBE_SNAPSHOT=1

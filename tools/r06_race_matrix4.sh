#!/bin/bash
mkdir -p gpurun_out/race4
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-8} > gpurun_out/race4/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY|CONSUMER STARTED" gpurun_out/race4/$name.log | sort | uniq -c | tail -3; grep -c "kind 4" gpurun_out/race4/$name.log; }
run lvl32 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=32
run lvl32b GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=32 GLAMR_PROBE_SNAPSHOT=1
run lvl36 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=36
run lvl32snap GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=32 GLAMR_PROBE_SNAPSHOT=1

#!/bin/bash
mkdir -p gpurun_out/race2
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-6} > gpurun_out/race2/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY|CONSUMER STARTED" gpurun_out/race2/$name.log | sort | uniq -c | tail -3; }
run two_inputs GLAMR_PROBE_TWO_INPUTS=1
run lvl0 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=0
run lvl1 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=1
run lvl2 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=2
run lvl3 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=3
run lvl7 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=7
run lvl19 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=19
run lvl8 GLAMR_LIB_PATH=tools/_lib_race.so GLAMR_PROBE_LEVEL=8
run plain_pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run plain_dynq0 DEBUG_HIP_DYNAMIC_QUEUES=0
run plain_optflush0 AMD_OPT_FLUSH=0
run plain_graphq1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run plain_hwq2 GPU_MAX_HW_QUEUES=2
run plain_skiprel0 DEBUG_CLR_SKIP_RELEASE_SCOPE=0

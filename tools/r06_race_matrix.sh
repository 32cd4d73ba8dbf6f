#!/bin/bash
# round 6: the pipeline corruption under runtime knobs (old launch order + three-graph cut = the configuration that failed 6 of 6 in round 5)
mkdir -p gpurun_out/race
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-8} > gpurun_out/race/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY|CONSUMER STARTED" gpurun_out/race/$name.log | sort | uniq -c | tail -3; }
run base_plainlib X=1
run base GLAMR_LIB_PATH=tools/_lib_race.so
run pktcap0 GLAMR_LIB_PATH=tools/_lib_race.so DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run skiprel0 GLAMR_LIB_PATH=tools/_lib_race.so DEBUG_CLR_SKIP_RELEASE_SCOPE=0
run skiprel1 GLAMR_LIB_PATH=tools/_lib_race.so DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run dynq0 GLAMR_LIB_PATH=tools/_lib_race.so DEBUG_HIP_DYNAMIC_QUEUES=0
run optflush0 GLAMR_LIB_PATH=tools/_lib_race.so AMD_OPT_FLUSH=0
run graphq1 GLAMR_LIB_PATH=tools/_lib_race.so DEBUG_HIP_FORCE_GRAPH_QUEUES=1

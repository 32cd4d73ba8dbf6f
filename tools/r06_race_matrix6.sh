#!/bin/bash
mkdir -p gpurun_out/race6
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-8} > gpurun_out/race6/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY" gpurun_out/race6/$name.log; grep -c "j_local rows" gpurun_out/race6/$name.log; grep "blocks of smpl_prep" gpurun_out/race6/$name.log | sed 's/.*blocks of smpl_prep_kernel//' | cut -c1-60 | head -3; }
run base X=1
run hostwait GLAMR_PROBE_HOST_WAIT=1
run spin300 GLAMR_PROBE_SPIN_US=300
run spin1000 GLAMR_PROBE_SPIN_US=1000
run spin3000 GLAMR_PROBE_SPIN_US=3000
run hwq3 GPU_MAX_HW_QUEUES=3

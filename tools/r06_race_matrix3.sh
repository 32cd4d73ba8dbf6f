#!/bin/bash
mkdir -p gpurun_out/race3
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-6} > gpurun_out/race3/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY|CONSUMER STARTED" gpurun_out/race3/$name.log | sort | uniq -c | tail -3; }
run snap GLAMR_PROBE_SNAPSHOT=1
run snap_fp32blend GLAMR_PROBE_SNAPSHOT=1 GLAMR_SMPL_FP32_BLEND=1
run plain_traj_free GLAMR_NETS_TRAJ_LDS=0
run plain_lensmemcpy GLAMR_NETS_LENS_MEMCPY=1

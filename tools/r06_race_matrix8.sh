#!/bin/bash
mkdir -p gpurun_out/race8
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early GLAMR_LIB_PATH=tools/_lib_spin.so
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-6} > gpurun_out/race8/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY" gpurun_out/race8/$name.log; grep "blocks of smpl_prep" gpurun_out/race8/$name.log | sed 's/.*blocks of smpl_prep_kernel//' | cut -c1-50 | head -6 | tr '\n' ' '; echo; }
run nospin X=1
for p in 1 2 3 5 6 7 4; do run pos$p GLAMR_NETS_PROBE_SPIN=$p:1500; done

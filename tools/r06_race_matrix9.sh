#!/bin/bash
mkdir -p gpurun_out/race9
export GLAMR_SKIN_AFTER_PRIORS=1 GLAMR_GATE_PREP=early
run() { name=$1; shift; echo "== $name"; env "$@" timeout 400 python tools/race_probe.py 1024 ${N:-6} > gpurun_out/race9/$name.log 2>&1; echo "rc=$?"; grep -E "SUMMARY" gpurun_out/race9/$name.log; grep "blocks of smpl_prep" gpurun_out/race9/$name.log | sed 's/.*blocks of smpl_prep_kernel//' | cut -c1-40 | head -6 | tr '\n' ' '; echo; }
run plain X=1
run k6generic GLAMR_LIB_PATH=tools/_lib_spin.so GLAMR_NETS_PROBE_GENERIC_K=6
run k6generic_b GLAMR_LIB_PATH=tools/_lib_spin.so GLAMR_NETS_PROBE_GENERIC_K=6
run prepw4 GLAMR_LIB_PATH=tools/_lib_prepw4.so
run prepw4_b GLAMR_LIB_PATH=tools/_lib_prepw4.so
run nofree GLAMR_NETS_FREE=0

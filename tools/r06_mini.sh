#!/bin/bash
mkdir -p gpurun_out/mini
run() { name=$1; shift; echo "== $name"; env "$@" timeout 300 python tools/race_mini.py 0 24 nets > gpurun_out/mini/$name.log 2>&1; echo rc=$?; grep SUMMARY gpurun_out/mini/$name.log; tail -2 gpurun_out/mini/$name.log | grep -i "error"; }
run plainlib X=1
run noslp GLAMR_LIB_PATH=tools/_lib_noslp.so
run noslp2 GLAMR_LIB_PATH=tools/_lib_noslp.so

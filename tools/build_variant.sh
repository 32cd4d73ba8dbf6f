#!/bin/bash
# development aid: tools/build_variant.sh <name> [extra hipcc flags for grecon.hip...] -> tools/_lib_<name>.so (the other objects come from the last
# full build in glamr_amd/csrc/build)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
B=glamr_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -c glamr_amd/csrc/grecon.hip -o /tmp/grecon_$name.o
objs=$(ls $B/*.o | grep -v "/grecon.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/grecon_$name.o -o tools/_lib_$name.so
echo built tools/_lib_$name.so

#!/bin/bash
# development aid: tools/build_variant_nets.sh <name> [extra hipcc flags for nets.hip...] -> tools/_lib_<name>.so (other objects from the last full build)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
B=glamr_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${GLAMR_VARIANT_PACKED:--Xclang -target-feature -Xclang -packed-fp32-ops} "$@" -c glamr_amd/csrc/nets.hip -o /tmp/nets_$name.o
objs=$(ls $B/*.o | grep -v "/nets.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/nets_$name.o -o tools/_lib_$name.so
echo built tools/_lib_$name.so

#!/bin/bash
# development aid: tools/build_variant_init.sh <name> [flags] -> tools/_lib_<name>.so with this tree's init.hip (other objects from the last full build)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
B=glamr_amd/csrc/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${GLAMR_VARIANT_PACKED:--Xclang -target-feature -Xclang -packed-fp32-ops} -ffp-contract=off "$@" -c glamr_amd/csrc/init.hip -o /tmp/init_$name.o
objs=$(ls $B/*.o | grep -v "/init.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/init_$name.o -o tools/_lib_$name.so
echo built tools/_lib_$name.so

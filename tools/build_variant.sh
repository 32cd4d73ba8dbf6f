#!/bin/bash
# development aid: tools/build_variant.sh <name> [extra hipcc flags for grecon.hip...] -> tools/_lib_<name>.so (the other objects come from the last
# full build in glamr_amd/csrc/build).  GLAMR_VARIANT_ALGO=<file>: compile with that copy of grecon_algo.hpp (in a scratch copy of csrc/).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
B=glamr_amd/csrc/build
SRC=glamr_amd/csrc
if [ -n "$GLAMR_VARIANT_ALGO" ]; then
  rm -rf /tmp/variant_$name; mkdir -p /tmp/variant_$name/glamr_amd /tmp/variant_$name/include
  cp -r glamr_amd/csrc /tmp/variant_$name/glamr_amd/csrc; cp include/*.h /tmp/variant_$name/include/
  cp "$GLAMR_VARIANT_ALGO" /tmp/variant_$name/glamr_amd/csrc/grecon_algo.hpp
  SRC=/tmp/variant_$name/glamr_amd/csrc
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${GLAMR_VARIANT_PACKED:--Xclang -target-feature -Xclang -packed-fp32-ops} -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -c $SRC/grecon.hip -o /tmp/grecon_$name.o
objs=$(ls $B/*.o | grep -v "/grecon.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/grecon_$name.o -o tools/_lib_$name.so
echo built tools/_lib_$name.so

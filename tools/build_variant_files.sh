#!/bin/bash
# development aid: tools/build_variant_files.sh <name> "<file.hip ...>" [extra hipcc flags for those files] -> tools/_lib_<name>.so
# (the other objects come from the last full build in glamr_amd/csrc/build; per-file flags as in glamr_amd/build.py)
set -e
cd "$(dirname "$0")/.."
name=$1; files=$2; shift; shift
B=glamr_amd/csrc/build
objs=""
skip=""
for f in $files; do
  extra=""
  case $f in
    grecon.hip|grecon_wide.hip) extra="-fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize";;
    init.hip) extra="-ffp-contract=off";;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${GLAMR_VARIANT_PACKED:--Xclang -target-feature -Xclang -packed-fp32-ops} $extra "$@" -c glamr_amd/csrc/$f -o /tmp/${f}_$name.o
  objs="$objs /tmp/${f}_$name.o"
  skip="$skip|/$f.o"
done
rest=$(ls $B/*.o | grep -v -E "${skip:1}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $rest $objs -o tools/_lib_$name.so
echo built tools/_lib_$name.so

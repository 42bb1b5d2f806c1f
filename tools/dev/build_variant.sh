#!/bin/bash
# Build a variant of libpcfa_hip.so with extra -D flags for ONE source file:  build_variant.sh <name> <file.hip> <flags...>
# -> pcfa_amd/lib/libpcfa_hip_<name>.so (select with PCFA_HIP_LIB).  Objects of the other files are reused.
set -e
cd "$(dirname "$0")/../../pcfa_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../lib/obj_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fvisibility=hidden "$@" -c $src -o ../lib/obj_$name/${src%.hip}.o
objs=""
for f in ../lib/obj/*.o; do
  b=$(basename $f)
  if [ "$b" == "${src%.hip}.o" ]; then objs="$objs ../lib/obj_$name/$b"; else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpcfa_hip_$name.so $objs
echo built pcfa_amd/lib/libpcfa_hip_$name.so

#!/bin/bash
# Build an instrumented / experimental copy of libforge_hip.so: tools/debug/build_variant_lib.sh NAME "-DFLAG1 -DFLAG2"  ->  tools/debug/libforge_hip_NAME.so
# (loaded through FORGE_AMD_LIB by the probes under tools/debug; never the product library)
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
NAME=$1; FLAGS=$2
OUT=$ROOT/tools/debug/_obj_$NAME
mkdir -p $OUT
for f in $ROOT/forge_amd/csrc/*.hip $ROOT/forge_amd/csrc/*.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DNDEBUG $FLAGS -x hip -c $f -o $OUT/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/debug/libforge_hip_$NAME.so $OUT/*.o
rm -rf $OUT
echo built $ROOT/tools/debug/libforge_hip_$NAME.so

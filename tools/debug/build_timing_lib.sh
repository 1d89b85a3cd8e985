#!/bin/bash
# instrumented build of libforge_hip.so (FORGE_CONV_TIMING: per-workgroup clock stamps in conv_igemm_kernel) -> tools/debug/libforge_hip_timing.so
set -e
cd "$(dirname "$0")/../.."
OBJ=forge_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DNDEBUG -DFORGE_CONV_TIMING -x hip -c forge_amd/csrc/conv_igemm.hip -o /tmp/conv_igemm_timing.o
objs=$(ls $OBJ/*.o | grep -v conv_igemm.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/debug/libforge_hip_timing.so /tmp/conv_igemm_timing.o $objs
ls -la tools/debug/libforge_hip_timing.so

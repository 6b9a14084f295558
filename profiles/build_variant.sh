#!/bin/bash
# Build a variant of the library with extra flags for the band / whole-device kernels (e.g. -DMWF_SYS_TIMING -DMWF_BAND_DEV)
# into profiles/_<name>_libmwf_hip.so; the other objects come from the regular build (miniwfa_amd/csrc/build).
# Usage: profiles/build_variant.sh <name> <flags...>     then run with MWF_HIP_LIB=profiles/_<name>_libmwf_hip.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
C=miniwfa_amd/csrc
[ -f $C/build/mwf_engine.cpp.o ] || python miniwfa_amd/build.py > /dev/null
mkdir -p /tmp/mwf_variant_$NAME
KERNELS="mwf_band2.hip mwf_sys.hip mwf_mid.hip mwf_lane.hip"
OBJS=""
for f in $KERNELS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -I include -I $C "$@" -c $C/$f -o /tmp/mwf_variant_$NAME/$f.o &
  OBJS="$OBJS /tmp/mwf_variant_$NAME/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS \
  $C/build/mwf_kernels.hip.o $C/build/mwf_engine.cpp.o $C/build/mwf_memory.cpp.o $C/build/mwf_plan.cpp.o $C/build/mwf_chain.cpp.o $C/build/mwf_async.cpp.o $C/build/kalloc.cpp.o $C/build/mwf_dbg.cpp.o -o profiles/_${NAME}_libmwf_hip.so -lpthread
echo profiles/_${NAME}_libmwf_hip.so

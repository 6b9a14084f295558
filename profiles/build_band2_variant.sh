#!/bin/bash
# Build a variant of the library in which only the packed band kernel (mwf_band2.hip) is compiled with extra flags; every other object comes
# from the regular build (miniwfa_amd/csrc/build).  Usage: profiles/build_band2_variant.sh <name> <flags...>  -> profiles/_<name>_libmwf_hip.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
C=miniwfa_amd/csrc
[ -f $C/build/mwf_engine.cpp.o ] || python miniwfa_amd/build.py > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -I include -I $C "$@" -c $C/mwf_band2.hip -o /tmp/mwf_b2_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared /tmp/mwf_b2_$NAME.o $C/build/mwf_sys.hip.o $C/build/mwf_mid.hip.o $C/build/mwf_lane.hip.o \
  $C/build/mwf_kernels.hip.o $C/build/mwf_engine.cpp.o $C/build/mwf_memory.cpp.o $C/build/mwf_plan.cpp.o $C/build/mwf_chain.cpp.o $C/build/mwf_async.cpp.o $C/build/kalloc.cpp.o $C/build/mwf_dbg.cpp.o -o profiles/_${NAME}_libmwf_hip.so -lpthread
echo profiles/_${NAME}_libmwf_hip.so

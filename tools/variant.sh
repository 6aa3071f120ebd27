#!/bin/bash
# tools/variant.sh NAME "-DFLAG ..." : builds zpaq-v_amd/lib/libzpaq_hip_NAME.so with extra compile flags for
# zpq_chain.hip only (kernel experiments; select at run time with ZPQ_LIB_PATH=zpaq-v_amd/lib/libzpaq_hip_NAME.so).
set -e
cd "$(dirname "$0")/../zpaq-v_amd/csrc"
name=$1; shift
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c -x hip zpq_chain.hip -o build_var/zpq_chain_$name.o
objs=$(ls build/*.o | grep -v zpq_chain.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libzpaq_hip_$name.so $objs build_var/zpq_chain_$name.o
echo built ../lib/libzpaq_hip_$name.so

#!/bin/bash
# tools/variant.sh NAME "-DFLAG ..." : builds zpaq-v_amd/lib/libzpaq_hip_NAME.so with extra compile flags for one
# kernel source only (SRC=zpq_chain by default; SRC=zpq_pipe ...) -- kernel experiments; select at run time with
# ZPQ_LIB_PATH=zpaq-v_amd/lib/libzpaq_hip_NAME.so.  Delete the variant libraries afterwards.
set -e
cd "$(dirname "$0")/../zpaq-v_amd/csrc"
name=$1; shift
src=${SRC:-zpq_chain}
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c -x hip $src.hip -o build_var/${src}_$name.o
objs=$(ls build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libzpaq_hip_$name.so $objs build_var/${src}_$name.o
echo built ../lib/libzpaq_hip_$name.so

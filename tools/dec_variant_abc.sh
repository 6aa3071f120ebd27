#!/bin/bash
# Level-2 decode of variant libraries in the order given (one process each; consecutive processes on a box differ by 2-4 %
# whatever they run -- EXPERIMENTS R4.11 -- so give every arm the same number of odd and even places: A B C C B A).
# Usage: tools/dec_variant_abc.sh <outfile> <variant|shipped> ...
out=$1; shift
for v in "$@"; do
  echo "== lib=$v" >> $out
  if [ "$v" != shipped ]; then export ZPQ_LIB_PATH=zpaq-v_amd/lib/libzpaq_hip_$v.so; else unset ZPQ_LIB_PATH; fi
  python tools/quickbench.py --blocks 8192 --level 2 --reps 3 2>&1 | grep -E "^rep2|^status" | cut -c1-170 >> $out || exit 1
done

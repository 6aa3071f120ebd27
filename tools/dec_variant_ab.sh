#!/bin/bash
# Level-2 decode of variant libraries (tools/variant.sh) against the shipped one, alternating processes on one box.
# Usage: tools/dec_variant_ab.sh <outfile> <variant> [<variant> ...]   ("" = shipped)
out=$1; shift
for rep in 1 2; do
  for v in "" "$@"; do
    echo "== lib=${v:-shipped}" >> $out
    if [ -n "$v" ]; then export ZPQ_LIB_PATH=zpaq-v_amd/lib/libzpaq_hip_$v.so; else unset ZPQ_LIB_PATH; fi
    python tools/quickbench.py --blocks 8192 --level 2 --reps 3 2>&1 | grep -E "^rep2|^status" | cut -c1-170 >> $out || exit 1
  done
done

#!/bin/bash
# Timing-only floors of the level-2 encoder (WRONG output by construction): no hash-row traffic / no coder work, whole stages
# (k_pipe) against paired ISSE waves (k_pipe2 "60cd1").  Usage: tools/enc_pair_floors.sh <outfile>
out=$1
for v in "" norows nocoder norowscoder; do
  for o in 0 60cd1; do
    echo "== lib=${v:-shipped} ZPQ_ENC_SPLIT=$o" >> $out
    if [ -n "$v" ]; then export ZPQ_LIB_PATH=zpaq-v_amd/lib/libzpaq_hip_$v.so; else unset ZPQ_LIB_PATH; fi
    ZPQ_ENC_SPLIT=$o python tools/quickbench.py --blocks 8192 --level 2 --reps 2 --check 0 2>&1 | grep "^rep1" | sed -e 's/  dec.*//' >> $out || exit 1
  done
done

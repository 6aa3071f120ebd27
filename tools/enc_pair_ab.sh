#!/bin/bash
# A/B of the level-2 encoder's wave orders on one box, alternating: k_pipe (whole stages) against k_pipe2 with both ISSEs
# paired on one wave (c = H1+H2, d = P1+P2).  Usage: tools/enc_pair_ab.sh <outfile> [orders...]
out=$1; shift
orders=${@:-"0 60cd1 0 60cd1 610cd 6cd01 60dc1"}
for o in $orders; do
  echo "== ZPQ_ENC_SPLIT=$o" >> $out
  ZPQ_ENC_SPLIT=$o python tools/quickbench.py --blocks 8192 --level 2 --reps 3 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done

#!/bin/bash
# bench.py's headline step with the level-2 encoder on whole stages (ZPQ_ENC_SPLIT=0, k_pipe) against the paired default
# (k_pipe2 "60cd1"), alternating on one box.  Usage: tools/enc_pair_bench_ab.sh <outfile>
out=$1
for o in 0 60cd1 0 60cd1; do
  echo "== ZPQ_ENC_SPLIT=$o" >> $out
  ZPQ_ENC_SPLIT=$o python bench.py --no-secondary --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms'])" >> $out || exit 1
done

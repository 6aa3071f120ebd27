#!/bin/bash
# Does the speed of a process depend on WHERE its slot pool lands?  Consecutive processes on one box, some with a block of
# device memory allocated (and held) before the context.  Usage: tools/placement_ab.sh <outfile> <hold GiB> ...
out=$1; shift
for h in "$@"; do
  echo "== hold=$h GiB" >> $out
  python tools/quickbench.py --blocks 8192 --level 2 --reps 3 --check 0 --hold $h 2>&1 | grep -E "^rep2" | cut -c1-170 >> $out || exit 1
done

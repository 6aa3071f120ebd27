#!/bin/bash
# tools/profile_round.sh OUT COMMIT: the round's profile set on the GPU box -- rocprofv3 kernel trace + stats of the
# headline bench command, then counter passes (FETCH_SIZE, WRITE_SIZE, two SQ groups; counters in their own runs,
# never combined with tracing domains), aggregated by tools/pmc_collect.py.  The program itself follows `--`.
out=$1; export ZPQ_COMMIT=$2
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --no-secondary --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- $B --steps 5 --warmup 1 > $out/kt.log 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for p in "FETCH_SIZE" "WRITE_SIZE" "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $p -d $out/pmc$i -o pmc$i --output-format csv -- $B --steps 1 --warmup 0 > $out/pmc$i.log 2>&1
done
python3 tools/pmc_collect.py $out $out/${ROUND:-r03} > $out/collect.log 2>&1
tail -5 $out/kt.log; cat $out/collect.log | head -40

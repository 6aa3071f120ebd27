#!/bin/bash
# tools/pmc_passes.sh OUTDIR [prof_run.py args]: SQ counter passes (8 SQ slots per pass) + kernel trace for one
# encode + decode launch; counters only, no tracing domains combined with --pmc (gpurun refuses that).
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVES"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
i=0
for p in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $p -d $out/p$i -o p$i --output-format csv -- python3 ${PROF_TARGET:-tools/prof_run.py} "$@" > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_chain" not in k and "k_lanes" not in k and "k_rows" not in k: continue
        role = ("decode" if ("<true" in k or "ILb1" in k) else "encode")
        agg[role][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump(agg, open(os.path.join(sys.argv[1], "sq.json"), "w"), indent=1)
bits = 65537 * 8
for role, c in agg.items():
    w = c.get("SQ_WAVES", 1024.0)
    print(role, "waves", w)
    for k in sorted(c):
        print("   %-28s %14.0f  per wave-bit %8.2f" % (k, c[k], c[k] / w / bits))
PY

#!/bin/bash
# tools/pmc_c4b.sh OUT: HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass, no trace domains) of the general-model
# kernels on C4b, 16 384 blocks: k_gpipe (batched), k_gpipe (bit-serial stages), k_rows encode, and the default decoder
# k_gdec in each pass (k_rows<decode> only when ZPQ_DEC_GPIPE=0 is set by the caller).
out=$1; mkdir -p $out; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/${c}_batch -o p --output-format csv -- python3 tools/prof_c4b.py 16384 > $out/${c}_batch.log 2>&1 || exit 1
  ZPQ_GPIPE_BATCH=0 rocprofv3 --pmc $c -d $out/${c}_serial -o p --output-format csv -- python3 tools/prof_c4b.py 16384 > $out/${c}_serial.log 2>&1 || exit 1
  ZPQ_ENC_GPIPE=0 rocprofv3 --pmc $c -d $out/${c}_rows -o p --output-format csv -- python3 tools/prof_c4b.py 16384 > $out/${c}_rows.log 2>&1 || exit 1
done
python3 - $out <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(float)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    tag = f.split("/")[-3] if "/p_" not in f else f.split("/")[-2]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gpipe" in k or "k_rows" in k or "k_gdec" in k or "k_lanes" in k:
            agg[(f.split(out)[1].split("/")[1], k[:60], r["Counter_Name"])] += float(r["Counter_Value"])
for k in sorted(agg):
    print(k, "%.1f GB" % (agg[k] * 1024 / 1e9))
P

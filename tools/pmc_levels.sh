#!/bin/bash
# tools/pmc_levels.sh OUT: HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the level-3 and level-4 kernels at the
# bench shapes (4096 x 64 KiB, line store), sixteen-lane two-hypothesis decoders against the eight-lane ones (ZPQ_DEC_HYP16=0).
out=$1; mkdir -p $out; export TMPDIR=/tmp
for lv in 3 4; do for h in 1 0; do for c in FETCH_SIZE WRITE_SIZE; do
  ZPQ_DEC_HYP16=$h rocprofv3 --pmc $c -d $out/l${lv}_h${h}_$c -o p --output-format csv -- python3 tools/quickbench.py --level $lv --blocks 4096 --reps 1 --check 0 > $out/l${lv}_h${h}_$c.log 2>&1 || exit 1
done; done; done
python3 - $out <<'P'
import csv, glob, sys, re, collections
out = sys.argv[1]
agg = collections.defaultdict(float); ms = {}
for f in glob.glob(out + "/l*_h*_*/**/*counter_collection.csv", recursive=True):
    m = re.search(r"/l(\d)_h(\d)_", f)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        role = "dec" if "k_chain<true" in kn else "enc" if ("k_pipe" in kn or "k_chain<false" in kn) else None
        if role:
            agg[(int(m.group(1)), int(m.group(2)), role, r["Counter_Name"])] += float(r["Counter_Value"]) * 1024.0
for f in glob.glob(out + "/l*_h*_FETCH_SIZE.log"):
    m = re.search(r"/l(\d)_h(\d)_", f)
    for line in open(f):
        mm = re.search(r"kernel ([\d.]+) ms (\S+)\)\s+dec .*kernel ([\d.]+) ms", line)
        if mm:
            ms[(int(m.group(1)), int(m.group(2)))] = (float(mm.group(1)), float(mm.group(3)))
print("levels 3-4, 4096 x 64 KiB, line store: HBM traffic per launch (rocprofv3 --pmc, separate passes), lines = bytes / 64 (reads) and / 32-64 (writes: sectors)")
print("%-34s %8s %9s %9s %12s %14s" % ("kernel", "ms", "read GB", "write GB", "B per in-byte", "G read-lines/s"))
for lv in (3, 4):
    for h in (1, 0):
        for role in ("enc", "dec"):
            if role == "enc" and h == 0:
                continue
            rd, wr = agg[(lv, h, role, "FETCH_SIZE")], agg[(lv, h, role, "WRITE_SIZE")]
            t = ms.get((lv, h), (0, 0))[0 if role == "enc" else 1]
            name = "level %d %s" % (lv, "k_pipe<encode>" if role == "enc" else ("decode, 16 lanes, two copies" if h else "decode, 8 lanes"))
            print("%-34s %8.1f %9.1f %9.1f %12.1f %14.1f" % (name, t, rd / 1e9, wr / 1e9, (rd + wr) / (4096 * 65536.0), rd / 64 / (t * 1e-3) / 1e9 if t else 0))
P

#!/bin/bash
# tools/pmc_c4b_split.sh OUT [blocks]: HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass, no trace domains) of k_gpipe and
# k_gdec on the first K = 2..9 components of C4b (tools/prof_c4b_prefix.py): the difference between K and K - 1 is component K's
# share (a per-component-type split of C4b's 16.8x / 24.6x algorithmic traffic).
out=$1; nb=${2:-16384}; mkdir -p $out; export TMPDIR=/tmp
for k in 2 3 4 5 6 7 8 9; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/k${k}_$c -o p --output-format csv -- python3 tools/prof_c4b_prefix.py $k $nb > $out/k${k}_$c.log 2>&1 || exit 1
  done
  grep PREFIX $out/k${k}_FETCH_SIZE.log
done
python3 - $out $nb <<'P'
import csv, glob, sys, collections, re
out, nb = sys.argv[1], int(sys.argv[2])
NAMES = ["CONST", "CM", "ICM", "MATCH", "AVG", "MIX2", "ISSE", "MIX", "SSE"]
agg = collections.defaultdict(float)
for f in glob.glob(out + "/k*_*/**/*counter_collection.csv", recursive=True):
    k = int(re.search(r"/k(\d)_", f).group(1))
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        role = "enc" if "k_gpipe" in kn else "dec" if "k_gdec" in kn else None
        if role:
            agg[(k, role, r["Counter_Name"])] += float(r["Counter_Value"]) * 1024.0      # KB -> bytes
ms = {}
for k in range(2, 10):
    for line in open("%s/k%d_FETCH_SIZE.log" % (out, k)):
        m = re.search(r"enc ([\d.]+) ms .* dec ([\d.]+) ms", line)
        if m and line.startswith("PREFIX"):
            ms[k] = (float(m.group(1)), float(m.group(2)))
bits = nb * 65537 * 8.0
print("C4b prefixes, %d x 64 KiB blocks: HBM bytes per coded bit and block (FETCH + WRITE), and what each component adds" % nb)
print("%-22s %9s %9s %9s %9s | %9s %9s %9s %9s" % ("components", "enc ms", "enc rd", "enc wr", "enc +B/bit", "dec ms", "dec rd", "dec wr", "dec +B/bit"))
prev = {"enc": 0.0, "dec": 0.0}
for k in range(2, 10):
    row = []
    for role in ("enc", "dec"):
        rd, wr = agg[(k, role, "FETCH_SIZE")] / bits, agg[(k, role, "WRITE_SIZE")] / bits
        row += [ms.get(k, (0, 0))[0 if role == "enc" else 1], rd, wr, rd + wr - prev[role]]
        prev[role] = rd + wr
    print("%-22s %9.1f %9.2f %9.2f %9.2f | %9.1f %9.2f %9.2f %9.2f" % ("1..%d (+%s)" % (k, NAMES[k - 1]), *row))
print("(64-byte lines per bit = bytes / 64; algorithmic bytes of the full model: 212.6 B per input byte = 26.6 B per bit)")
P

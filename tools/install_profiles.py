"""Copy what tools/profile_round.sh left under gpurun_out/<dir> into profiles/ (tracked): the bench line, the rocprofv3 kernel
stats, the PMC / SQ summaries (stamped with the commit: the GPU box has no .git) and, of the raw counter CSVs, only the rows of
this repo's kernels.  Usage (in the repo, after the gpurun call has merged its files back):
    python tools/install_profiles.py gpurun_out/r3x r03 [bench.json]"""
import csv
import json
import os
import shutil
import subprocess
import sys

src, rnd = sys.argv[1], sys.argv[2]
bench = sys.argv[3] if len(sys.argv) > 3 else os.path.join(src, "bench.json")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(src, "prof")
dst = os.path.join(ROOT, "profiles")
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
for f in ("%s_pmc.json" % rnd, "%s_sq_counters.json" % rnd):
    j = json.load(open(os.path.join(prof, f)))
    j["_commit"] = commit + " (kernel sources; profile taken on the GPU box, which has no .git)"
    json.dump(j, open(os.path.join(dst, f), "w"), indent=1)
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(dst, "%s_bench_final.json" % rnd))
shutil.copy(os.path.join(prof, "kt", "kt_kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats.csv" % rnd))
for i in range(1, 5):
    f = os.path.join(prof, "pmc%d" % i, "pmc%d_counter_collection.csv" % i)
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_chain", "k_pipe", "k_dpipe", "k_gpipe", "k_gdec", "k_rows", "k_lanes", "k_generic", "k_sha1", "k_gather"))]
    w = csv.DictWriter(open(os.path.join(dst, "%s_pmc_pass%d.csv" % (rnd, i)), "w", newline=""), fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)
p = json.load(open(os.path.join(dst, "%s_pmc.json" % rnd)))
for k, v in p.items():
    if isinstance(v, dict):
        print(k, {a: round(b / 1e9, 1) for a, b in v.items() if a.startswith("hbm")})
print("kernel source", p.get("_kernel_src_sha"), "commit", commit)

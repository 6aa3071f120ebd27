"""Aggregate rocprofv3 --pmc counter CSVs (one pass per counter group, as the gfx950 guide prescribes) into the
JSON files bench.py and DESIGN.md cite.  Usage on the GPU box, from /tmp:
  rocprofv3 --pmc FETCH_SIZE -d OUT/fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE -d OUT/write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --pmc SQ_... (two passes)
  python3 tools/pmc_collect.py OUT profiles/r01
bench.py --steps 1 --warmup 0 launches every kernel twice (timed step + per-kernel timing step)."""
import collections, csv, glob, hashlib, json, os, subprocess, sys
src, dst = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_src_sha():                      # the same fingerprint bench.py computes: a profile is only cited for the source it was taken on
    h = hashlib.sha256()
    for f in ("zpq_chain.hip", "zpq_pipe.hip", "zpq_chain_cfg.h", "zpq_common.h", "zpq_vm.h"):
        h.update(open(os.path.join(ROOT, "zpaq-v_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("ZPQ_COMMIT", "")
    except Exception:
        return os.environ.get("ZPQ_COMMIT", "")
LAUNCHES = 2
enc_name = "k_chain<encode>"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_chain" not in k and "k_pipe" not in k:
            continue
        role = "decode" if "k_chain<true" in k else "encode"
        if role == "encode":
            enc_name = ("k_pipe2<encode>" if "k_pipe2" in k else "k_pipe<encode>") if "k_pipe" in k else "k_chain<encode>"
        agg[role][r["Counter_Name"]] += float(r["Counter_Value"]) / LAUNCHES
pmc = {}
for role in ("encode", "decode"):
    c = agg[role]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        pmc[enc_name if role == "encode" else "k_chain<decode>"] = {
            "FETCH_SIZE_KB_per_launch": c["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": c["WRITE_SIZE"],
            "hbm_bytes_per_launch": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
            "hbm_bytes_per_launch_fetch_x2": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)}
if pmc:
    pmc["_blocks_per_launch"] = 8192
    pmc["_kernel_src_sha"] = kernel_src_sha()
    pmc["_commit"] = commit()
    pmc["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 --warmup 0 "
                    "--no-cpu-baseline` (level 2, 8192 x 64 KiB blocks per launch), aggregated by tools/pmc_collect.py. Unit: KB (x1024). "
                    "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE halves WIDE COALESCED 16 B/lane streams (128-B requests tallied at 64 B); "
                    "this kernel's reads are random 16-B rows out of 64-B lines, an uncalibrated pattern, so both the raw sum "
                    "(hbm_bytes_per_launch, used as roofline.traffic) and the doubled-fetch bound are recorded. Writes include 103.8 GB of "
                    "table zero-fill (16 B/lane streaming stores, counted exactly) + row write-backs + output.")
    json.dump(pmc, open(dst + "_pmc.json", "w"), indent=1)
sq = {role: {k: v for k, v in agg[role].items() if k.startswith("SQ_")} for role in ("encode", "decode")}
if any(sq.values()):
    sq["_kernel_src_sha"] = kernel_src_sha()
    sq["_commit"] = commit()
    sq["_encode_kernel"] = enc_name
    sq["_note"] = "per launch, summed over the 1024 waves; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"
    json.dump(sq, open(dst + "_sq_counters.json", "w"), indent=1)
print(json.dumps({"pmc": {k: v for k, v in pmc.items() if not k.startswith("_note")}, "sq_keys": sorted(k for k in sq.get("decode", {}))}, indent=1)[:1500])

"""Balanced A/B of library variants on one GPU box (EXPERIMENTS R4.11).

Consecutive processes on one box differ by 2-4 % whatever they run (it goes with where the slot pool lands), often every other
process -- so an A/B that alternates two libraries hands one of them all the fast places.  This tool
  * runs the arms in a balanced order (A B B A A B ... / A B C C B A ...), one process per sample (tools/quickbench.py),
  * takes the kernel that is THE SAME in every arm (--ref enc|dec) as the process's reference,
  * and compares the arms at equal reference time: samples are split at the reference's median into the box's fast and slow places,
    every arm's kernel-under-test is averaged per place.
Arms are names of variant libraries built by tools/variant.sh (zpaq-v_amd/lib/libzpaq_hip_<name>.so); "shipped" is the default one.
    python tools/ab_balanced.py --arms shipped,nohyp4 --rounds 3 --ref enc          # the decoder is under test
    python tools/ab_balanced.py --replay gpurun_out/p8/ab.txt --ref enc             # re-read a recorded run (tools/dec_variant_abc.sh)
"""
import argparse
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(r"enc .*?kernel ([0-9.]+) ms ([^)]*)\)\s+dec .*?kernel ([0-9.]+) ms")


def balanced_order(arms, rounds):
    """A B C | C B A | A B C ...: every arm gets as many odd as even places over two rounds."""
    order = []
    for r in range(rounds):
        order += arms if r % 2 == 0 else arms[::-1]
    return order


def parse(text):
    """[(arm, enc_ms, dec_ms)] from '== lib=<arm>' headers followed by quickbench's rep lines (the LAST rep of a process counts)."""
    out, arm, last = [], None, None
    for ln in text.splitlines():
        if ln.startswith("== lib="):
            if arm is not None and last:
                out.append((arm,) + last)
            arm, last = ln[len("== lib="):].strip(), None
            continue
        m = LINE.search(ln)
        if m and ln.startswith("rep"):
            last = (float(m.group(1)), float(m.group(3)))
    if arm is not None and last:
        out.append((arm,) + last)
    return out


def report(samples, ref):
    ri, ti = (1, 2) if ref == "enc" else (2, 1)
    refs = [s[ri] for s in samples]
    cut = statistics.median(refs)
    lines = ["%-14s %8s %8s   place" % ("arm", "enc ms", "dec ms")]
    for s in samples:
        lines.append("%-14s %8.1f %8.1f   %s" % (s[0], s[1], s[2], "fast" if s[ri] <= cut else "slow"))
    lines.append("reference = %s kernel (same code in every arm), median %.1f ms; kernel under test = %s" % (ref, cut, "dec" if ref == "enc" else "enc"))
    arms = []
    for s in samples:
        if s[0] not in arms:
            arms.append(s[0])
    res = {}
    for a in arms:
        row = {}
        for place, sel in (("fast", lambda v: v <= cut), ("slow", lambda v: v > cut)):
            xs = [s[ti] for s in samples if s[0] == a and sel(s[ri])]
            row[place] = (statistics.mean(xs), len(xs)) if xs else (None, 0)
        res[a] = row
        lines.append("%-14s fast places: %s   slow places: %s" % (a, *("%.1f ms (n=%d)" % row[p] if row[p][1] else "-" for p in ("fast", "slow"))))
    return res, "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", default="shipped")
    ap.add_argument("--rounds", type=int, default=4, help="passes over the arms (even: every arm in odd and even places)")
    ap.add_argument("--ref", choices=("enc", "dec"), default="enc", help="the kernel that is the same in every arm")
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--blocks", type=int, default=8192)
    ap.add_argument("--replay", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.replay:
        text = open(a.replay).read()
    else:
        text = ""
        for arm in balanced_order(a.arms.split(","), a.rounds):
            env = dict(os.environ)
            env.pop("ZPQ_LIB_PATH", None)
            if arm != "shipped":
                env["ZPQ_LIB_PATH"] = os.path.join(ROOT, "zpaq-v_amd", "lib", "libzpaq_hip_%s.so" % arm)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "quickbench.py"), "--blocks", str(a.blocks), "--level", str(a.level),
                                "--reps", "3", "--check", "0"], env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:                                   # stop at the first failure: no GPU step after a failed one
                sys.stderr.write(r.stdout[-2000:] + r.stderr[-2000:])
                return 1
            text += "== lib=%s\n%s" % (arm, "".join(ln + "\n" for ln in r.stdout.splitlines() if ln.startswith("rep")))
            print("== lib=%s" % arm, [ln for ln in r.stdout.splitlines() if ln.startswith("rep")][-1][:150], flush=True)
    samples = parse(text)
    if not samples:
        print("no samples")
        return 1
    _, rep = report(samples, a.ref)
    print(rep)
    if a.out:
        open(a.out, "w").write(text + "\n" + rep + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())

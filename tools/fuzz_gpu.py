"""Randomised parity soak on the GPU box: ragged batches of random shape through the batch entry points (whatever
kernels the plan picks: k_pipe / k_chain / k_rows / k_generic, dense tables or the line store, one launch or several
rounds of slot reuse), every coded stream compared with the CPU oracle, every block decoded back.
Usage: python tools/fuzz_gpu.py [--seconds 300] [--seed 1]; prints one line per batch and a summary; exit code 1 on
the first mismatch (the offending batch is described so that it can be replayed with --seed/--only)."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
import oracle_lib as O  # noqa: E402
import workload as W  # noqa: E402
from inputs import C4B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--only", type=int, default=-1, help="run only batch number N of the seed's sequence")
ap.add_argument("--general", action="store_true", help="general models: random mixes of the nine component types (inputs = earlier "
                "components, hash-chain program), C4b, levels 2-3 as general models -- the wave-per-component encoder k_gpipe in its "
                "byte-batched and bit-serial forms, decoded back by k_rows")
ap.add_argument("--levels", default="", help="comma-separated levels to draw from instead of the default mix (e.g. 3,4: the round-4 decoders)")
a = ap.parse_args()

z = ge.load()
ctx = z.Context(0)
L = z.lib()
rnd = random.Random(a.seed)
t_end = time.time() + a.seconds
names = {}
nbatch = nblocks = nbytes = 0


HC = [74, 18, 104, 95, 0] + [59, 112, 25] * 7 + [59, 112, 56, 0]


def random_model(r):
    """A header with 1..12 components of random type whose inputs are earlier components."""
    n = r.randint(1, 12)
    comps = []
    for i in range(n):
        kinds = [1, 2, 3, 4] + ([5, 6, 7, 8, 9] * 2 if i else [])
        t = r.choice(kinds)
        j = r.randrange(i) if i else 0
        k = r.randrange(i) if i else 0
        if t == 1:
            c = [1, r.randrange(256)]
        elif t == 2:
            c = [2, r.randint(6, 16), r.randint(1, 255)]
        elif t == 3:
            c = [3, r.randint(5, 14)]
        elif t == 4:
            c = [4, r.randint(6, 14), r.randint(1, 14)]
        elif t == 5:
            c = [5, j, k, r.randrange(256)]
        elif t == 6:
            c = [6, r.randint(0, 10), j, k, r.randint(1, 30), r.choice([0, 255, 15, 3])]
        elif t == 7:
            m = r.randint(1, min(8, i))
            c = [7, r.randint(0, 8), r.randint(0, i - m), m, r.randint(1, 30), r.choice([0, 255, 15])]
        elif t == 8:
            c = [8, r.randint(5, 14), j]
        else:
            c = [9, r.randint(1, 8), j, r.randint(1, 32), r.randint(1, 255)]
        comps.append(c)
    b = [4, 16, 0, 0, n]
    for c in comps:
        b += c
    return bytes(b + [0] + HC), comps


def make(kind, n, r):
    if kind == 0:
        return bytes(n)
    if kind == 1:
        return bytes(r.getrandbits(8) for _ in range(n))
    if kind == 2:
        return bytes(r.choice(b"etaoin shrdlu\n,.THE") for _ in range(n))
    if kind == 3:
        per = bytes(r.getrandbits(8) for _ in range(r.randint(1, 300)))
        return (per * (n // len(per) + 1))[:n]
    if kind == 4:                                            # long runs with rare changes
        out = bytearray()
        while len(out) < n:
            out += bytes([r.getrandbits(8)]) * r.randint(1, 400)
        return bytes(out[:n])
    return bytes(W.make_block(r.randrange(1 << 20), n))      # the bench generator's classes


while time.time() < t_end:
    r = random.Random(rnd.getrandbits(64))
    nbatch += 1
    if a.only >= 0 and nbatch - 1 != a.only:
        if nbatch - 1 > a.only:
            break
        continue
    which = r.choice([1, 1, 2, 2, 2, 3, 3, 4, 5, "c4b"])
    if a.levels:
        which = r.choice([int(x) for x in a.levels.split(",")])
    lanes_flag = 0
    if a.general:
        which = r.choice(["rnd", "rnd", "rnd", "c4b", "c4b", 2, 3])
        lanes_flag = z.FLAG_LANES if which in (2, 3) else 0
    if which == "rnd":
        header, comps = random_model(r)
        which = "rnd%s" % comps
    else:
        header = C4B if which == "c4b" else O.level_header(which)
    model = z.Model(header=header)
    nb = r.choice([1, 2, 5, 11, 12, 13, 16, 17, 31, 33, 64, 100, 150])
    if which in (4, 5, "c4b") and not a.general:
        nb = min(nb, 33)
    maxlen = r.choice([0, 1, 7, 64, 300, 2000, 9000])
    blocks = [make(r.randrange(6), r.randint(0, maxlen) if r.random() < 0.8 else maxlen, r) for _ in range(nb)]
    pp = r.random() < 0.7
    flags = (z.FLAG_PP if pp else 0) | lanes_flag
    env = {}
    if a.general:
        nb = r.choice([1, 2, 5, 33, 64, 65, 100, 150, 200])
        blocks = [make(r.randrange(6), r.randint(0, maxlen) if r.random() < 0.8 else maxlen, r) for _ in range(nb)]
        if r.random() < 0.3:
            env["ZPQ_GPIPE_BATCH"] = "0"
    if which in (1, 2, 3, 4) and r.random() < 0.3:
        env["ZPQ_SPARSE_FORCE_LOG2"] = str(r.choice([12, 13, 15]))
    elif which in (1, 2, 3, 4) and r.random() < 0.3:
        env["ZPQ_SPARSE_MODE"] = r.choice(["never", "always"])
    if which in (3, 4) and r.random() < 0.15:
        env["ZPQ_DEC_HYP16"] = "0"                          # (the eight-lane decoders stay covered)
    if which in (1, 2, 3) and r.random() < 0.15:
        env["ZPQ_ENC_PIPE"] = "0"
    budget = None
    if r.random() < 0.4:                                     # fewer slots than blocks: rounds inside the kernel
        budget = int(model.state_bytes * r.choice([1.5, 3.2, 7.7, 19.3])) + 4096
    for k, v in env.items():
        os.environ[k] = v
    if budget:
        L.zpq_ctx_set_state_budget(ctx.h, budget)
    desc = "batch %d: model %s nb %d maxlen %d pp %s env %s budget %s" % (nbatch - 1, which, nb, maxlen, pp, env, budget)
    try:
        coded, status, _ = ctx.encode_blocks(model, blocks, flags=flags)
        ename = ctx.last_kernel_name
        slots = ctx.last_slots
        big = [i for i in range(nb) if status[i] == -4]      # a forced small line store may refuse a block: allowed
        assert all(status[i] == 0 for i in range(nb) if i not in big), (desc, list(status))
        assert not big or "ZPQ_SPARSE_FORCE_LOG2" in env, (desc, list(status))
        want = O.encode_blocks(header, blocks, pp=pp, nthreads=8)
        for i in range(nb):
            if i not in big:
                assert coded[i] == want[i], (desc, "block", i, len(blocks[i]), len(coded[i]), len(want[i]))
        dec, dstatus, consumed, _, first = ctx.decode_blocks(model, want, cap=maxlen + 16, flags=flags)
        dname = ctx.last_kernel_name
        bigd = [i for i in range(nb) if dstatus[i] == -4]
        assert all(dstatus[i] == 0 for i in range(nb) if i not in bigd) and (not bigd or "ZPQ_SPARSE_FORCE_LOG2" in env), (desc, list(dstatus))
        for i in range(nb):
            if i not in bigd:
                assert dec[i] == blocks[i] and int(consumed[i]) == len(want[i]), (desc, "decode of block", i)
    except AssertionError as e:
        print("MISMATCH", e, flush=True)
        sys.exit(1)
    finally:
        for k in env:
            del os.environ[k]
        if budget:
            L.zpq_ctx_set_state_budget(ctx.h, 150 << 30)
    names[(ename, dname)] = names.get((ename, dname), 0) + 1
    nblocks += nb
    nbytes += sum(len(b) for b in blocks)
    if nbatch % 20 == 0:
        print("%d batches, %d blocks, %.1f MB ok; last: %s -> %s/%s slots %d" % (nbatch, nblocks, nbytes / 1e6, desc, ename, dname, slots), flush=True)
print("DONE: %d batches, %d blocks, %.1f MB, all equal to the oracle; kernels used: %s" % (nbatch, nblocks, nbytes / 1e6, names))

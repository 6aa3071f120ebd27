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
a = ap.parse_args()

z = ge.load()
ctx = z.Context(0)
L = z.lib()
rnd = random.Random(a.seed)
t_end = time.time() + a.seconds
names = {}
nbatch = nblocks = nbytes = 0


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
    header = C4B if which == "c4b" else O.level_header(which)
    model = z.Model(header=header)
    nb = r.choice([1, 2, 5, 11, 12, 13, 16, 17, 31, 33, 64, 100, 150])
    if which in (4, 5, "c4b"):
        nb = min(nb, 33)
    maxlen = r.choice([0, 1, 7, 64, 300, 2000, 9000])
    blocks = [make(r.randrange(6), r.randint(0, maxlen) if r.random() < 0.8 else maxlen, r) for _ in range(nb)]
    pp = r.random() < 0.7
    flags = z.FLAG_PP if pp else 0
    env = {}
    if which in (1, 2, 3) and r.random() < 0.3:
        env["ZPQ_SPARSE_FORCE_LOG2"] = str(r.choice([12, 13, 15]))
    elif which in (1, 2, 3) and r.random() < 0.3:
        env["ZPQ_SPARSE_MODE"] = r.choice(["never", "always"])
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

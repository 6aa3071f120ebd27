"""Compare the wave-pipelined encoder with the lane-per-component one, block by block (debug aid)."""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import __graft_entry__ as ge
z = ge.load(); ctx = z.Context(0)
level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = z.Model(level=level)
rnd = np.random.default_rng(1)
blocks = [bytes(n) for n in (1, 2, 3, 16, 100, 4096)] + [rnd.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (1, 2, 5, 64, 1000)] + [b"abcabcabd" * 50, b""]
for flags in (z.FLAG_PP, 0):
    os.environ["ZPQ_ENC_PIPE"] = "1"
    a, st, _ = ctx.encode_blocks(model, blocks, flags=flags); na = ctx.last_kernel_name
    os.environ["ZPQ_ENC_PIPE"] = "0"
    b, st2, _ = ctx.encode_blocks(model, blocks, flags=flags); nb = ctx.last_kernel_name
    print("flags", flags, na, nb, list(st), list(st2))
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            k = next((j for j in range(min(len(x), len(y))) if x[j] != y[j]), min(len(x), len(y)))
            print("  block", i, "len(in)", len(blocks[i]), "pipe", len(x), "chain", len(y), "first diff at", k, x[:12].hex(), y[:12].hex())
        else:
            print("  block", i, "len(in)", len(blocks[i]), "equal", len(x))

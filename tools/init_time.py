"""How long do the level-2 kernels spend clearing their blocks' tables (12 MiB per block)?  8192 blocks of 1 byte each: the kernel
time is the slot initialisation plus launch overhead."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as ge
import bench as BM
z = ge.load(); ctx = z.Context(0)
for level in (2, 1, 3):
    model = z.Model(level=level)
    nb, size = (8192 if level < 3 else 4096), 1
    dev = torch.device('cuda:0')
    d_in = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
    b = BM.ResidentBatch(z, ctx, torch, dev, nb, size)
    for rep in range(3):
        b.step(model, d_in, z.FLAG_PP, record=True)
    print("level %d, %d blocks of %d byte: encode %.2f ms (%s), decode %.2f ms (%s), slots %d x %.1f MiB, ok %s" %
          (level, nb, size, min(b.enc_ms), b.enc_name, min(b.dec_ms), b.dec_name, ctx.last_slots, model.state_bytes / 2**20, b.ok(d_in)), flush=True)

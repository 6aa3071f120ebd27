"""Does a process's speed depend on WHERE the slot pool lands?  ONE process, several contexts one after the other (the previous one kept
alive, so every pool is different memory; the one before that closed, so the next may get its memory back): the level-2 kernels on the
same 8192 blocks through each.  Usage: python tools/placement_pools.py [--pools 4]"""
import sys, argparse
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
ap = argparse.ArgumentParser(); ap.add_argument('--pools', type=int, default=4); ap.add_argument('--blocks', type=int, default=8192)
a = ap.parse_args()
z = ge.load(); model = z.Model(level=2)
nb = a.blocks; size = 65536
dev = torch.device('cuda:0')
d_in = torch.from_numpy(W.make_blocks_fast(nb, size).reshape(-1)).to(dev)
cap = size + size // 8 + 1024
in_off = torch.arange(nb + 1, dtype=torch.int64, device=dev) * size
out_off = torch.arange(nb + 1, dtype=torch.int64, device=dev) * cap
d_out = torch.zeros(nb * cap, dtype=torch.uint8, device=dev)
i32 = lambda: torch.zeros(nb, dtype=torch.int32, device=dev)
d_len, d_st, d_dlen, d_cons, d_code, d_first, d_dst = i32(), i32(), i32(), i32(), i32(), i32(), i32()
d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
flags = z.FLAG_PP
live = []
for p in range(a.pools):
    ctx = z.Context(0)
    for rep in range(3):
        ctx.encode_blocks_dev(model, nb, d_in.data_ptr(), in_off.data_ptr(), flags, d_out.data_ptr(), out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
        ctx.sync(); ems = ctx.last_kernel_ms
        ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), flags, d_dec.data_ptr(), in_off.data_ptr(), d_dlen.data_ptr(), d_cons.data_ptr(), d_code.data_ptr(), d_first.data_ptr(), d_dst.data_ptr())
        ctx.sync(); dms = ctx.last_kernel_ms
    free, total = torch.cuda.mem_get_info()
    print(f"pool {p}: enc {ems:.1f} ms  dec {dms:.1f} ms  slots {ctx.last_slots}  free HBM after {free / 2**30:.0f} GiB", flush=True)
    live.append(ctx)
    if len(live) > 1:
        live.pop(0).close()
assert bool((d_st == 0).all()) and bool((d_dst == 0).all()) and bool(torch.equal(d_dec, d_in))

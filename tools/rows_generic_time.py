"""k_rows (four blocks per wave) against k_lanes (one block per wave) on a model whose HCOMP program is NOT the shipped hash
chain -- it runs through the interpreter (round 3: on the first lane of every row).  Device-resident round trip."""
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
z = ge.load(); ctx = z.Context(0)
# CM + ICM + ISSE with the order-1/2 contexts computed by a program that is deliberately not the recognised shape
prog = [74, 18, 104, 95, 0, 59, 112, 25, 59, 59, 112, 25, 59, 112, 56, 0]     # b=c c-- *c=a d=0 hash *d=a d++ hash hash *d=a d++ hash *d=a halt
header = bytes([3, 6, 0, 0, 3, 2, 16, 40, 3, 16, 8, 16, 1, 0]) + bytes(prog)
model = z.Model(header=header)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
size = 65536
arr = W.make_blocks_fast(nb, size)
dev = torch.device('cuda:0')
d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
cap = size * 2 + 1024
i64 = dict(dtype=torch.int64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
in_off = torch.arange(nb + 1, **i64) * size; out_off = torch.arange(nb + 1, **i64) * cap
d_out = torch.zeros(nb * cap, dtype=torch.uint8, device=dev); d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
d_len, d_st, d_dlen, d_cons, d_code, d_first, d_dst = (torch.zeros(nb, **i32) for _ in range(7))
torch.cuda.synchronize()
for rows in ("", "h"):
    if rows: os.environ["ZPQ_LANES_ROWS"] = rows
    else: os.environ.pop("ZPQ_LANES_ROWS", None)
    for rep in range(2):
        ctx.encode_blocks_dev(model, nb, d_in.data_ptr(), in_off.data_ptr(), z.FLAG_PP, d_out.data_ptr(), out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
        ctx.sync(); e = ctx.last_kernel_ms; en = ctx.last_kernel_name
        ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), z.FLAG_PP, d_dec.data_ptr(), in_off.data_ptr(), d_dlen.data_ptr(), d_cons.data_ptr(), d_code.data_ptr(), d_first.data_ptr(), d_dst.data_ptr())
        ctx.sync(); d = ctx.last_kernel_ms; dn = ctx.last_kernel_name
    ok = bool((d_st == 0).all()) and bool((d_dst == 0).all()) and bool(torch.equal(d_dec, d_in))
    print("%s / %s: %d blocks, slots %d: encode %.0f ms, decode %.0f ms, round trip %.1f MB/s, ok %s" % (en, dn, nb, ctx.last_slots, e, d, nb * size / (e + d) / 1e3, ok), flush=True)

"""One encode + one decode launch of the first K components of the all-nine-types model (C4b) -- CONST, CM, ICM, MATCH, AVG,
MIX2, ISSE, MIX, SSE in that order, the same HCOMP program -- at a block count: the profiling target of tools/pmc_c4b_split.sh.
The difference between K and K - 1 is what component K adds (traffic, time).  Usage: prof_c4b_prefix.py K [blocks]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
COMPS = [[1, 160], [2, 16, 255], [3, 16], [4, 16, 16], [5, 1, 2, 128], [6, 8, 3, 4, 24, 255], [8, 16, 5], [7, 8, 0, 7, 24, 255], [9, 8, 7, 32, 255]]
NAMES = ["CONST", "CM", "ICM", "MATCH", "AVG", "MIX2", "ISSE", "MIX", "SSE"]
k = int(sys.argv[1])
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
hdr = bytes([4, 16, 0, 0, k] + [b for c in COMPS[:k] for b in c] + [0, 74, 18, 104, 95, 0] + [59, 112, 25] * 7 + [59, 112, 56, 0])
z = ge.load(); ctx = z.Context(0); model = z.Model(header=hdr)
size = 65536
arr = W.make_blocks_fast(nb, size)
dev = torch.device("cuda:0")
d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
cap = size * 6 + 1024
i64 = dict(dtype=torch.int64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
in_off = torch.arange(nb + 1, **i64) * size; out_off = torch.arange(nb + 1, **i64) * cap
d_out = torch.zeros(nb * cap, dtype=torch.uint8, device=dev); d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
d_len, d_st, d_dlen, d_cons, d_code, d_first, d_dst = (torch.zeros(nb, **i32) for _ in range(7))
torch.cuda.synchronize()
ctx.encode_blocks_dev(model, nb, d_in.data_ptr(), in_off.data_ptr(), 1, d_out.data_ptr(), out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
ctx.sync(); e = ctx.last_kernel_ms; en = ctx.last_kernel_name; slots = ctx.last_slots
ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), 1, d_dec.data_ptr(), in_off.data_ptr(), d_dlen.data_ptr(),
                      d_cons.data_ptr(), d_code.data_ptr(), d_first.data_ptr(), d_dst.data_ptr())
ctx.sync(); d = ctx.last_kernel_ms
print("PREFIX %d (+%s) %d blocks slots %d: enc %.1f ms (%s) dec %.1f ms (%s) ok=%s ratio %.4f slot_bytes %d" % (
    k, NAMES[k - 1], nb, slots, e, en, d, ctx.last_kernel_name, bool(torch.equal(d_dec, d_in)), float(d_len.sum()) / (nb * size), model.state_bytes), flush=True)
ctx.close()

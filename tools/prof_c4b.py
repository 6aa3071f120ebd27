"""One encode + one decode launch of the all-nine-types model (C4b) at a block count (profiling target)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
from inputs import C4B
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
z = ge.load(); ctx = z.Context(0); model = z.Model(header=C4B)
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
ctx.sync(); e = ctx.last_kernel_ms
ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), 1, d_dec.data_ptr(), in_off.data_ptr(), d_dlen.data_ptr(),
                      d_cons.data_ptr(), d_code.data_ptr(), d_first.data_ptr(), d_dst.data_ptr())
ctx.sync(); d = ctx.last_kernel_ms
print("C4b %d blocks: enc %.1f ms dec %.1f ms ok=%s (%s)" % (nb, e, d, bool(torch.equal(d_dec, d_in)), ctx.last_kernel_name), flush=True)
ctx.close()

"""The wave-per-component encoder (k_gpipe) against the lane-per-component one (k_rows) on C4b (all nine component types) or a
shipped level run as a general model: device-resident encode at resident capacity, checked by decoding back and by comparing
the two encoders' streams.   python tools/gpipe_time.py [c4b|2|3|4|5] [blocks]"""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
from inputs import C4B
z = ge.load(); ctx = z.Context(0)
which = sys.argv[1] if len(sys.argv) > 1 else "c4b"
model = z.Model(header=C4B) if which == "c4b" else z.Model(level=int(which))
flags = z.FLAG_PP | (0 if which == "c4b" else z.FLAG_LANES)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
size = 65536
dev = torch.device('cuda:0')
res = {}
n = nb or ctx.resident_capacity(model, flags)
for gp in ("1", "0"):
    os.environ["ZPQ_ENC_GPIPE"] = gp
    os.environ["ZPQ_DEC_GPIPE"] = gp
    arr = W.make_blocks_fast(n, size)
    d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
    cap = size * (6 if which == "c4b" else 2) + 1024
    i64 = dict(dtype=torch.int64, device=dev); i32 = dict(dtype=torch.int32, device=dev)
    in_off = torch.arange(n + 1, **i64) * size; out_off = torch.arange(n + 1, **i64) * cap
    d_out = torch.zeros(n * cap, dtype=torch.uint8, device=dev); d_dec = torch.zeros(n * size, dtype=torch.uint8, device=dev)
    d_len, d_st, d_dlen, d_cons, d_code, d_first, d_dst = (torch.zeros(n, **i32) for _ in range(7))
    torch.cuda.synchronize()
    for rep in range(2):
        ctx.encode_blocks_dev(model, n, d_in.data_ptr(), in_off.data_ptr(), flags, d_out.data_ptr(), out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
        ctx.sync(); e = ctx.last_kernel_ms; en = ctx.last_kernel_name; sl = ctx.last_slots
    ctx.decode_blocks_dev(model, n, d_out.data_ptr(), out_off.data_ptr(), flags, d_dec.data_ptr(), in_off.data_ptr(), d_dlen.data_ptr(), d_cons.data_ptr(), d_code.data_ptr(), d_first.data_ptr(), d_dst.data_ptr())
    ctx.sync(); d = ctx.last_kernel_ms; dn = ctx.last_kernel_name
    ok = bool((d_st == 0).all()) and bool((d_dst == 0).all()) and bool(torch.equal(d_dec, d_in))
    lens = d_len.cpu().numpy().astype(np.int64)
    import hashlib
    out = d_out.cpu().numpy()
    hh = hashlib.sha256()
    for i in range(0, n, max(1, n // 64)):
        hh.update(out[i * cap:i * cap + int(lens[i])].tobytes())
    res[gp] = (int(lens.sum()), hh.hexdigest())
    print("%s: %s / %s: %d blocks, slots %d: encode %.1f ms (%.1f MB/s), decode %.1f ms, round trip %.1f MB/s, decoded back %s" %
          (which, en, dn, n, sl, e, n * size / e / 1e3, d, n * size / (e + d) / 1e3, ok), flush=True)
    del d_in, d_out, d_dec
    torch.cuda.empty_cache()
print("streams of the two encoders identical (total bytes, sha256 over a 64-block sample):", res["1"] == res["0"])

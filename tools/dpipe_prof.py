"""Where a wave-split decoder's waves spend their cycles (timing build: SRC=zpq_dpipe tools/variant.sh prof -DZPD_PROF).
Per wave of workgroup 0 and bit position: cycles working (barrier release -> next arrival) and cycles inside the barrier."""
import ctypes as C, os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ.setdefault("ZPQ_LIB_PATH", "/root/repo/zpaq-v_amd/lib/libzpaq_hip_prof.so")
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
z = ge.load(); ctx = z.Context(0); model = z.Model(level=level); L = z.lib()
size = 65536
arr = W.make_blocks_fast(nb, size)
blocks = None
dev = torch.device('cuda:0')
d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
cap = size + size // 8 + 1024
in_off = torch.arange(nb + 1, dtype=torch.int64, device=dev) * size
out_off = torch.arange(nb + 1, dtype=torch.int64, device=dev) * cap
d_out = torch.zeros(nb * cap, dtype=torch.uint8, device=dev)
i32 = lambda: torch.zeros(nb, dtype=torch.int32, device=dev)
d_len, d_st, d_dlen, d_cons, d_code, d_first, d_dst = (i32() for _ in range(7))
d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
ctx.encode_blocks_dev(model, nb, d_in.data_ptr(), in_off.data_ptr(), z.FLAG_PP, d_out.data_ptr(), out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
ctx.sync()
buf = (C.c_ulonglong * (16 * 8 * 2))()
L.zpq_debug_dpipe_prof(None, 1)
ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), z.FLAG_PP, d_dec.data_ptr(), in_off.data_ptr(), d_dlen.data_ptr(), d_cons.data_ptr(), d_code.data_ptr(), d_first.data_ptr(), d_dst.data_ptr())
ctx.sync()
print("kernel", ctx.last_kernel_name, "%.1f ms" % ctx.last_kernel_ms, "ok", bool(torch.equal(d_dec, d_in)))
L.zpq_debug_dpipe_prof(buf, 0)
a = np.frombuffer(buf, dtype=np.uint64).reshape(16, 8, 2).astype(np.float64) / (size + 2)
nw = {1: 3, 2: 4, 3: 6}[level]
print("cycles per byte position (work | in barrier), per wave; last wave = decoder")
for w in range(nw):
    print("wave %d: " % w + "  ".join("%5.0f|%-5.0f" % (a[w, k, 0], a[w, k, 1]) for k in range(8)) + "   sum %6.0f" % a[w].sum())

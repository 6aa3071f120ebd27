"""Experiment: what does clearing the NEXT call's table slots (level 2: 12 MiB per block, 96 GiB for 8192 blocks) cost when it runs
on a side stream beside a coding kernel, instead of inside the next kernel?  Times (a) the clear alone, (b) encode / decode
alone, (c) encode / decode with the clear of a second 96 GiB buffer running beside it."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
import bench as BM
z = ge.load(); ctx = z.Context(0)
z.lib().zpq_ctx_set_state_budget(ctx.h, 110 << 30)
model = z.Model(level=2)
nb, size = 8192, 65536
dev = torch.device('cuda:0')
arr = W.make_blocks_fast(nb, size)
d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
b = BM.ResidentBatch(z, ctx, torch, dev, nb, size)
flags = z.FLAG_PP
other = torch.empty(nb * model.state_bytes, dtype=torch.uint8, device=dev)     # the second slot set
side = torch.cuda.Stream()
torch.cuda.synchronize()
# (a) the clear alone
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(); other.zero_(); e1.record()
    torch.cuda.synchronize()
    print("clear of %.1f GiB alone: %.1f ms = %.2f TB/s" % (other.numel() / 2**30, e0.elapsed_time(e1), other.numel() / e0.elapsed_time(e1) / 1e9), flush=True)
# (b) alone
b.step(model, d_in, flags, record=True); b.step(model, d_in, flags, record=True)
print("alone: encode %.1f ms, decode %.1f ms (%s / %s)" % (b.enc_ms[-1], b.dec_ms[-1], b.enc_name, b.dec_name), flush=True)
# (c) with the clear beside it
c = ctx
for rep in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c.encode_blocks_dev(model, nb, d_in.data_ptr(), b.in_off.data_ptr(), flags, b.d_out.data_ptr(), b.out_off.data_ptr(), b.d_len.data_ptr(), b.d_st.data_ptr())
    with torch.cuda.stream(side):
        e0.record(); other.zero_(); e1.record()
    c.sync(); torch.cuda.synchronize()
    em, cm1 = c.last_kernel_ms, e0.elapsed_time(e1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c.decode_blocks_dev(model, nb, b.d_out.data_ptr(), b.out_off.data_ptr(), flags, b.d_dec.data_ptr(), b.in_off.data_ptr(), b.d_dlen.data_ptr(), b.d_cons.data_ptr(), b.d_code.data_ptr(), b.d_first.data_ptr(), b.d_dst.data_ptr())
    with torch.cuda.stream(side):
        e0.record(); other.zero_(); e1.record()
    c.sync(); torch.cuda.synchronize()
    dm, cm2 = c.last_kernel_ms, e0.elapsed_time(e1)
    print("with a 96 GiB clear beside it: encode %.1f ms (clear took %.1f), decode %.1f ms (clear took %.1f); round trip ok %s" % (em, cm1, dm, cm2, b.ok(d_in)), flush=True)

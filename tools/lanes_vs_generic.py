import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as ge
import workload as W
z = ge.load(); ctx = z.Context(0)
blocks = [bytes(W.make_block(b, 65536)) for b in range(64)]
for level in (2, 5):
    model = z.Model(level=level)
    for nm, fl in (("lanes", z.FLAG_PP | z.FLAG_LANES), ("generic", z.FLAG_PP | z.FLAG_GENERIC)):
        t0 = time.time(); coded, st, _ = ctx.encode_blocks(model, blocks, flags=fl); t1 = time.time()
        print(level, nm, ctx.last_kernel_name, round(t1 - t0, 2), "s for 64 blocks x 64 KiB", (st == 0).all())

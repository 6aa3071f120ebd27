"""Sequential front end (Compressor / Decompresser, one segment per block) timing: first segment on the fast kernels
vs forced generic kernel (what every segment used before lazy state materialisation)."""
import sys, time, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as ge
import workload as W
z = ge.load(); ctx = z.Context(0)
data = bytes(W.make_block(2, 262144))
res = {}
for level in (2, 5):
    model = z.Model(level=level)
    for mode, flags in (("fast first segment", z.FLAG_PP), ("generic kernel", z.FLAG_PP | z.FLAG_GENERIC)):
        b = z.Block(ctx, model); t0 = time.time(); c = b.encode_segment(data, flags=flags); t1 = time.time(); b.close()
        d = z.Block(ctx, model); out, *_ = d.decode_segment(c, cap=len(data) + 16, flags=flags); t2 = time.time(); d.close()
        assert out == data
        res["level %d, %s" % (level, mode)] = dict(encode_s=round(t1 - t0, 2), decode_s=round(t2 - t1, 2), kernel=ctx.last_kernel_name)
print(json.dumps(res, indent=1))

import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import __graft_entry__ as ge
import oracle_lib as O
z = ge.load(); ctx = z.Context(0)
model = z.Model(level=2)
blocks = [bytes(4), b"abcd"]
a, st, _ = ctx.encode_blocks(model, blocks, flags=z.FLAG_PP)
print(ctx.last_kernel_name)
for blk, out in zip(blocks, a):
    c = O.Codec(model.header)
    coded, tr = c.encode(blk, pp=True, ntrace=64)
    vals = [out[2*i] | (out[2*i+1] << 8) for i in range(min(40, len(out)//2))]
    print("pipe :", [(v & 0x7fff, v >> 15) for v in vals])
    print("oracle:", [(t[0], t[1]) for t in tr[:45]])

import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as ge, random
import oracle_lib as O
z=ge.load(); ctx=z.Context(0)
model=z.Model(level=2)
data=bytes(random.Random(5).getrandbits(8) for _ in range(2000))
for cap in (100, 2029, 2030, 2031, 5000):
    coded,status,out_len=ctx.encode_blocks(model,[data],flags=z.FLAG_PP|z.FLAG_GENERIC,cap=cap)
    print(cap,status,out_len,len(coded[0]), len(O.Codec(model.header).encode(data)))

import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import __graft_entry__ as ge
import oracle_lib as O
from inputs import INPUTS, C4B
z=ge.load(); ctx=z.Context(0)
HC=[74,18,104,95,0]+[59,112,25]*7+[59,112,56,0]
def hdr(comps):
    n=len(comps); b=[4,16,0,0,n]
    for c in comps: b+=c
    return bytes(b+[0]+HC)
models={
 'const':[[1,160]],
 'cm':[[2,16,255]],
 'icm':[[3,16]],
 'match':[[3,16],[4,16,16]],
 'matchonly':[[4,16,16]],
 'avg':[[3,16],[2,16,255],[5,0,1,128]],
 'mix2':[[3,16],[2,16,255],[6,8,0,1,24,255]],
 'isse':[[3,16],[8,16,0]],
 'mix':[[3,16],[2,16,255],[7,8,0,2,24,255]],
 'sse':[[3,16],[9,8,0,32,255]],
 'c4b':None,
}
data=INPUTS['lcg4k']
for name,comps in models.items():
    h = C4B if comps is None else hdr(comps)
    m=z.Model(header=h)
    coded,tr=ctx.debug_encode_trace(m,data,ntrace=40000)
    want,otr=O.Codec(h).encode(data,pp=True,ntrace=40000)
    ot=[t[0] for t in otr]
    bad=[i for i,(a,b) in enumerate(zip(ot,tr.tolist())) if a!=b]
    print(name, 'OK' if coded==want else 'DIFF', 'first bad bit', bad[:1], (ot[bad[0]],int(tr[bad[0]])) if bad else '')

"""Timing + parity spot-check for the non-headline BASELINE configs (C2, C4a, C4b)."""
import sys, time, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tests/golden')
import numpy as np, torch
import __graft_entry__ as ge
import workload as W, oracle_lib as O
from inputs import C4B
z=ge.load(); ctx=z.Context(0); dev=torch.device('cuda:0')
def run(name, model, nb, size, flags, check=2, capmul=1.125):
    arr=W.make_blocks_fast(nb,size)
    d_in=torch.from_numpy(arr.reshape(-1)).to(dev)
    cap=int(size*capmul)+1024
    i64=dict(dtype=torch.int64,device=dev); i32=dict(dtype=torch.int32,device=dev)
    in_off=torch.arange(nb+1,**i64)*size; out_off=torch.arange(nb+1,**i64)*cap; dec_off=torch.arange(nb+1,**i64)*size
    d_out=torch.zeros(nb*cap,dtype=torch.uint8,device=dev); d_dec=torch.zeros(nb*size,dtype=torch.uint8,device=dev)
    d_len,d_st,d_dlen,d_cons,d_code,d_first,d_dst=(torch.zeros(nb,**i32) for _ in range(7))
    res={}
    for rep in range(2):
        t0=time.time(); ctx.encode_blocks_dev(model,nb,d_in.data_ptr(),in_off.data_ptr(),flags,d_out.data_ptr(),out_off.data_ptr(),d_len.data_ptr(),d_st.data_ptr()); ctx.sync(); t1=time.time()
        ems=ctx.last_kernel_ms; ename=ctx.last_kernel_name
        ctx.decode_blocks_dev(model,nb,d_out.data_ptr(),out_off.data_ptr(),flags,d_dec.data_ptr(),dec_off.data_ptr(),d_dlen.data_ptr(),d_cons.data_ptr(),d_code.data_ptr(),d_first.data_ptr(),d_dst.data_ptr()); ctx.sync(); t2=time.time()
        dms=ctx.last_kernel_ms
    B=nb*size
    ok=bool((d_st==0).all()) and bool((d_dst==0).all()) and bool(torch.equal(d_dec,d_in))
    blocks=[arr[i].tobytes() for i in range(min(check,nb))]
    want=O.encode_blocks(model.header,blocks,nthreads=4,slack=cap)
    outc=d_out[:check*cap].cpu().numpy(); lens=d_len.cpu().numpy()
    par=all(outc[i*cap:i*cap+int(lens[i])].tobytes()==w for i,w in enumerate(want))
    r=dict(config=name,blocks=nb,size=size,kernel=ename,slots=ctx.last_slots,enc_ms=round(ems,1),dec_ms=round(dms,1),comp_MBps=round(B/ems/1e3,1),decomp_MBps=round(B/dms/1e3,1),roundtrip_MBps=round(B/(ems+dms)/1e3,1),ratio=round(float(d_len.sum())/B,4),roundtrip_ok=ok,oracle_parity_first_blocks=par)
    print(json.dumps(r),flush=True)
which=sys.argv[1:] or ['C2','C4b','C3l3','C4a']
if 'C2' in which: run('C2 level1 4096x64KiB', z.Model(level=1), 4096, 65536, z.FLAG_PP)
if 'C3l3' in which: run('level3 2048x64KiB', z.Model(level=3), 2048, 65536, z.FLAG_PP)
if 'C4b' in which: run('C4b all-9-types 4096x64KiB', z.Model(header=C4B), 4096, 65536, z.FLAG_PP, capmul=6)
if 'C4b16k' in which: run('C4b all-9-types at capacity', z.Model(header=C4B), ctx.resident_capacity(z.Model(header=C4B)), 65536, z.FLAG_PP, capmul=6)
if 'C4a' in which: run('C4a level5 1024x64KiB', z.Model(level=5), 1024, 65536, z.FLAG_PP)
if 'L4' in which: run('level4 1536x64KiB', z.Model(level=4), 1536, 65536, z.FLAG_PP)

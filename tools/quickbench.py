"""Quick timing of encode+decode for N synthetic blocks with buffers resident in HBM."""
import sys, time, argparse
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
ap=argparse.ArgumentParser(); ap.add_argument('--blocks',type=int,default=256); ap.add_argument('--level',type=int,default=2)
ap.add_argument('--size',type=int,default=65536); ap.add_argument('--reps',type=int,default=2); ap.add_argument('--generic',action='store_true')
ap.add_argument('--check',type=int,default=4); ap.add_argument('--cls',type=int,default=-1)
ap.add_argument('--hold',type=float,default=0.0,help='GiB of device memory allocated (and kept) BEFORE the context: shifts where the slot pool lands')
a=ap.parse_args()
z=ge.load()
hold=torch.empty(int(a.hold*(1<<30)),dtype=torch.uint8,device='cuda:0') if a.hold>0 else None
ctx=z.Context(0); model=z.Model(level=a.level)
nb=a.blocks; size=a.size
arr=W.make_blocks_fast(nb,size)
if a.cls>=0:
    arr=np.stack([W.make_block(4*i+a.cls if a.cls!=2 else 2,size) for i in range(nb)])
dev=torch.device('cuda:0')
d_in=torch.from_numpy(arr.reshape(-1)).to(dev)
cap=size+size//8+1024
in_off=torch.arange(nb+1,dtype=torch.int64,device=dev)*size
out_off=torch.arange(nb+1,dtype=torch.int64,device=dev)*cap
d_out=torch.zeros(nb*cap,dtype=torch.uint8,device=dev)
d_len=torch.zeros(nb,dtype=torch.int32,device=dev); d_st=torch.zeros(nb,dtype=torch.int32,device=dev)
dec_off=torch.arange(nb+1,dtype=torch.int64,device=dev)*size
d_dec=torch.zeros(nb*size,dtype=torch.uint8,device=dev)
d_dlen=torch.zeros(nb,dtype=torch.int32,device=dev); d_cons=torch.zeros(nb,dtype=torch.int32,device=dev)
d_code=torch.zeros(nb,dtype=torch.int32,device=dev); d_first=torch.zeros(nb,dtype=torch.int32,device=dev); d_dst=torch.zeros(nb,dtype=torch.int32,device=dev)
torch.cuda.synchronize()
flags=z.FLAG_PP|(z.FLAG_GENERIC if a.generic else 0)
for rep in range(a.reps):
    t0=time.time()
    ctx.encode_blocks_dev(model,nb,d_in.data_ptr(),in_off.data_ptr(),flags,d_out.data_ptr(),out_off.data_ptr(),d_len.data_ptr(),d_st.data_ptr())
    ctx.sync(); t1=time.time(); ems=ctx.last_kernel_ms; ename=ctx.last_kernel_name
    # coded segments are decoded in place from the encode slabs (in_off = out_off, lengths ignored: decoder stops at EOF)
    ctx.decode_blocks_dev(model,nb,d_out.data_ptr(),out_off.data_ptr(),flags,d_dec.data_ptr(),dec_off.data_ptr(),d_dlen.data_ptr(),d_cons.data_ptr(),d_code.data_ptr(),d_first.data_ptr(),d_dst.data_ptr())
    ctx.sync(); t2=time.time(); dms=ctx.last_kernel_ms
    B=nb*size
    print(f"rep{rep} L{a.level} {nb}x{size}: enc {t1-t0:.3f}s ({B/(t1-t0)/1e6:.1f} MB/s, kernel {ems:.1f} ms {ename})  dec {t2-t1:.3f}s ({B/(t2-t1)/1e6:.1f} MB/s, kernel {dms:.1f} ms)  slots {ctx.last_slots}",flush=True)
ok=bool((d_st==0).all()) and bool((d_dst==0).all()) and bool(torch.equal(d_dec,d_in)) and bool((d_dlen==size).all())
ratio=float(d_len.sum())/(nb*size)
print("status ok/roundtrip equal:",ok," ratio %.4f"%ratio, " bits/cycle-ish: cycles/bit @2.1GHz enc %.0f dec %.0f"%(ems*1e-3*2.1e9/((size+1)*8), dms*1e-3*2.1e9/((size+1)*8)))
if a.check:
    import oracle_lib as O
    blocks=[arr[i].tobytes() for i in range(min(a.check,nb))]
    want=O.encode_blocks(model.header,blocks,nthreads=4,slack=cap)
    outc=d_out.cpu().numpy(); lens=d_len.cpu().numpy()
    for i,w in enumerate(want):
        got=outc[i*cap:i*cap+int(lens[i])].tobytes()
        assert got==w,("oracle mismatch block",i,len(got),len(w))
    print("oracle parity on first",len(blocks),"blocks: OK")

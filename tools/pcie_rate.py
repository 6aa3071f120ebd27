"""PCIe-inclusive rate: the host-pointer entry points (H2D + kernel + D2H per call)."""
import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import __graft_entry__ as ge, workload as W
z=ge.load(); ctx=z.Context(0); model=z.Model(level=2)
nb,size=8192,65536
arr=W.make_blocks_fast(nb,size)
L=z.lib()
cap=size+size//8+1024
in_off=(np.arange(nb+1,dtype=np.uint64)*size); out_off=(np.arange(nb+1,dtype=np.uint64)*cap); dec_off=in_off.copy()
src=np.ascontiguousarray(arr.reshape(-1)); out=np.zeros(nb*cap,dtype=np.uint8); dec=np.zeros(nb*size,dtype=np.uint8)
olen=np.zeros(nb,dtype=np.uint32); st=np.zeros(nb,dtype=np.int32); dlen=np.zeros(nb,dtype=np.uint32); dst=np.zeros(nb,dtype=np.int32)
for rep in range(3):
    t0=time.time()
    rc=L.zpq_encode_blocks(ctx.h,model.h,nb,src.ctypes.data,in_off.ctypes.data,1,out.ctypes.data,out_off.ctypes.data,olen.ctypes.data,st.ctypes.data); assert rc==0
    t1=time.time()
    rc=L.zpq_decode_blocks(ctx.h,model.h,nb,out.ctypes.data,out_off.ctypes.data,1,dec.ctypes.data,dec_off.ctypes.data,dlen.ctypes.data,None,None,None,dst.ctypes.data); assert rc==0
    t2=time.time()
    B=nb*size
    print(f"rep{rep}: host-pointer encode {t1-t0:.3f}s ({B/(t1-t0)/1e6:.0f} MB/s)  decode {t2-t1:.3f}s ({B/(t2-t1)/1e6:.0f} MB/s)  round trip {B/(t2-t0)/1e6:.0f} MB/s  ok={bool((st==0).all() and (dst==0).all() and np.array_equal(dec,src))}",flush=True)

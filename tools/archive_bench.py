"""End-to-end archive pipeline timing: zpaq::archive_add / archive_extract (host buffers in, archive
bytes out, PCIe + framing + SHA-1 included) and the SHA-1 side kernel alone."""
import sys, time, json, hashlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as ge
import workload as W
z = ge.load(); ctx = z.Context(0); dev = torch.device('cuda:0')
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
level = int(sys.argv[2]) if len(sys.argv) > 2 else 2
size = 65536
arr = W.make_blocks_fast(nb, size)
files = [("f%05d" % i, "%d bytes" % size, arr[i].tobytes()) for i in range(nb)]
B = nb * size
# SHA-1 kernel alone, device-resident
d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
off = torch.arange(nb + 1, dtype=torch.int64, device=dev) * size
d_sha = torch.zeros(nb * 20, dtype=torch.uint8, device=dev)
st = torch.cuda.ExternalStream(ctx.stream)
for rep in range(2):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record(); ctx.sha1_blocks_dev(nb, d_in.data_ptr(), off.data_ptr(), d_sha.data_ptr()); e1.record()
    ctx.sync(); sha_ms = e0.elapsed_time(e1)
got = d_sha.cpu().numpy().reshape(nb, 20)
ok_sha = all(got[i].tobytes() == hashlib.sha1(files[i][2]).digest() for i in range(0, nb, max(1, nb // 64)))
t0 = time.time(); hs = [hashlib.sha1(f[2]).digest() for f in files]; host_sha_s = time.time() - t0
res = {}
for rep in range(2):
    t0 = time.time(); arc = z.archive_add(ctx, level, files); t1 = time.time()
    out = z.archive_extract(ctx, arc); t2 = time.time()
    lst = z.archive_extract(ctx, arc, want_data=False); t3 = time.time()
ok = all(o["data"] == f[2] and o["sha1_ok"] and o["status"] == 0 for o, f in zip(out, files)) and len(out) == nb
print(json.dumps(dict(workload="%d files x 64 KiB, level %d" % (nb, level), archive_bytes=len(arc), ratio=round(len(arc) / B, 4),
                      add_s=round(t1 - t0, 3), add_MBps=round(B / (t1 - t0) / 1e6, 1),
                      extract_s=round(t2 - t1, 3), extract_MBps=round(B / (t2 - t1) / 1e6, 1),
                      list_s=round(t3 - t2, 3), roundtrip_ok=ok,
                      sha1_kernel_ms=round(sha_ms, 3), sha1_kernel_GBps=round(B / sha_ms / 1e6, 1), sha1_ok=ok_sha,
                      host_hashlib_sha1_s=round(host_sha_s, 3))))
# one big file cut into 64 KiB blocks (-fragment 6): the case the reference's one-block-per-file layout cannot parallelise
one = [("big.bin", "%d bytes" % B, arr.tobytes())]
u0 = time.time(); arc1 = z.archive_add(ctx, level, one, fragment_bytes=size); u1 = time.time()
out1 = z.archive_extract(ctx, arc1, join_unnamed=True); u2 = time.time()
ok1 = len(out1) == 1 and out1[0]["data"] == one[0][2] and out1[0]["sha1_ok"]
print(json.dumps(dict(workload="1 file of %d MiB, -fragment 6, level %d" % (B >> 20, level), add_s=round(u1 - u0, 3),
                      add_MBps=round(B / (u1 - u0) / 1e6, 1), extract_s=round(u2 - u1, 3), extract_MBps=round(B / (u2 - u1) / 1e6, 1),
                      roundtrip_ok=ok1)))

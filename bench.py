#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ZPAQ block codec.

A "step" = one pass of the hot path over one batch: level-2 compress of
`--blocks` synthetic 64 KiB blocks per GPU followed by decompress of the coded
streams (both through the C ABI's device-pointer entry points, inputs already
resident in HBM).  value = uncompressed MB / (t_comp + t_decomp), whole job.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started WITHOUT a launcher and with --gpus N > 1, it starts the N ranks itself (fresh child processes through
torch.distributed.run, before anything here has touched a GPU), relays rank 0's JSON line and exits with the
launcher's code.  Inside a rank --gpus must equal WORLD_SIZE, or the run stops.

One rank per GPU; block b of the global batch goes to rank b mod N (static
round-robin, no data-path collective); timing is barrier + synchronize on both
sides, MAX over ranks; rank 0 prints ONE JSON line.

After the timed headline, at N = 1 only and outside the timed region, the same
line gains:
  per_class        the headline workload one block class at a time
  value_incl_pcie  the same batch through the host-pointer entry points
                   (H2D + kernel + D2H inside the call)
  secondary        C2 (level 1, 4096 blocks), levels 3-4, C4a (level 5 at its
                   resident capacity) and C4b (all nine component types), each
                   with its own roofline
  cpu_baseline     the C oracle on this host at nproc threads (value), at 16 threads and at 1 thread
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# A(L) = 1 + r + 64*K_h + 2 + 32*[MIX2]; the table holds A(L) - r
ALG_BYTES_PER_INPUT_BYTE_L = {1: 131, 2: 195, 3: 323, 4: 419, 5: 547}
ALG_BYTES_C4B = 211
RANDOM_LINE_CEILING_G = 48.2       # G random 64-byte lines per second, MI355X, measured (tools/micro/linerate.hip)
HBM_PEAK_GBS = 8000.0                                                  # MI355X_MICROARCH.md: 8.0 TB/s spec
LEVEL_NAMES = {1: "ICM16+ISSE19", 2: "ICM16+ISSE16+ISSE16", 3: "ICM18+4xISSE18", 4: "ICM20+5xISSE20+MIX2",
               5: "ICM22+7xISSE22+MIX2"}


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(level, size, host, per_thread_blocks, one_thread_blocks):
    """The C oracle (a restatement of the V CPU path -- V itself cannot be built
    here) timed on this host: compress+decompress of a bounded sample of the same
    synthetic blocks (`host`: the headline batch), blocks divided statically over
    `cores` pthreads: at every hardware thread this process may use (SURVEY.md 8(d):
    nproc), at 16 threads and at one thread."""
    import oracle_lib as O
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    # a container may see every hardware thread of the host and still be scheduled on a few (cgroup CPU quota): the
    # threads that can actually run at once are what "nproc" means for a baseline
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            f = open(path).read().split()
            if path.endswith("cpu.max"):
                if f[0] != "max":
                    quota = max(1, int(round(int(f[0]) / int(f[1]))))
            else:
                q = int(f[0])
                if q > 0:
                    quota = max(1, int(round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    avail = min(visible, quota) if quota else visible
    hdr = O.level_header(level)

    def timed(nb, nthreads):
        nb = max(1, min(nb, len(host)))
        blocks = [host[i].tobytes() for i in range(nb)]
        t0 = time.time()
        coded = O.encode_blocks(hdr, blocks, nthreads=nthreads, slack=size + size // 8 + 1024)
        t1 = time.time()
        dec = O.decode_blocks(hdr, coded, cap=size + 16, nthreads=nthreads)
        t2 = time.time()
        assert dec == [b"\0" + b for b in blocks]
        B = nb * size
        return {"value": round(B / (t2 - t0) / 1e6, 3), "comp_MBps": round(B / (t1 - t0) / 1e6, 3),
                "decomp_MBps": round(B / (t2 - t1) / 1e6, 3), "cores": nthreads, "blocks": nb, "seconds": round(t2 - t0, 2)}

    # one thread codes a 64 KiB block both ways in ~30 ms: 32 blocks per thread keep every leg at a few seconds
    full = timed(per_thread_blocks * avail if avail <= 64 else 8 * avail, avail)
    t16 = timed(per_thread_blocks * 4 * min(avail, 16), min(avail, 16))
    one = timed(one_thread_blocks, 1)
    return {
        "value": full["value"], "unit": "MB/s", "cores": avail, "kind": "port",
        "comp_MBps": full["comp_MBps"], "decomp_MBps": full["decomp_MBps"], "seconds": full["seconds"],
        "blocks": full["blocks"],
        "sixteen_threads": t16, "one_thread": one,
        "nproc": os.cpu_count(), "nproc_visible": visible, "cgroup_cpu_quota": quota, "nproc_available": avail,
        "cpu_model": cpu_model_string(),
        "sample": "the first %d blocks of the headline batch (64 KiB, classes b mod 4), level %d, C oracle "
                  "compress+decompress on %d pthreads = every hardware thread this process may run on (blocks divided "
                  "statically; per-block table alloc+zero-fill included); sixteen_threads / one_thread = the first %d / %d "
                  "of those blocks on 16 / 1 threads.  A real V build adds bounds checks and interface dispatch per byte, "
                  "so it would be slower than this port." % (full["blocks"], level, avail, t16["blocks"], one["blocks"]),
    }


def incl_pcie_run(z, ctx, model, flags, size, cap, nblk, arr2d, what):
    """One batch through the host-pointer entry points (what a V front end holding host buffers calls): H2D + kernel + D2H
    inside each call.  Buffers come from zpq_host_alloc (pinned); the decoder is handed the coded streams packed back to
    back, as an archive holds them.  Returns the second of two calls."""
    import numpy as np
    L = z.lib()
    nbytes = nblk * size
    in_off = np.arange(nblk + 1, dtype=np.uint64) * np.uint64(size)
    out_off = np.arange(nblk + 1, dtype=np.uint64) * np.uint64(cap)
    p_src, p_out, p_dec = z.PinnedArray(nbytes), z.PinnedArray(nblk * cap), z.PinnedArray(nbytes)
    p_src.array[:] = arr2d.reshape(-1)
    olen = np.zeros(nblk, dtype=np.uint32); st = np.zeros(nblk, dtype=np.int32)
    dlen = np.zeros(nblk, dtype=np.uint32); dst = np.zeros(nblk, dtype=np.int32)
    r = {}
    p_cod = None
    for rep in range(2):
        t0 = time.time()
        rc1 = L.zpq_encode_blocks(ctx.h, model.h, nblk, p_src.array.ctypes.data, in_off.ctypes.data, flags,
                                  p_out.array.ctypes.data, out_off.ctypes.data, olen.ctypes.data, st.ctypes.data)
        t1 = time.time()
        # pack the coded streams (outside the timed calls: an archive already holds them like this)
        c_off = np.zeros(nblk + 1, dtype=np.uint64)
        c_off[1:] = np.cumsum(olen.astype(np.uint64))
        if rep == 0:
            p_cod = z.PinnedArray(int(c_off[-1]) + 16)
            for i in range(nblk):
                p_cod.array[int(c_off[i]):int(c_off[i + 1])] = p_out.array[i * cap:i * cap + int(olen[i])]
        t2 = time.time()
        rc2 = L.zpq_decode_blocks(ctx.h, model.h, nblk, p_cod.array.ctypes.data, c_off.ctypes.data, flags,
                                  p_dec.array.ctypes.data, in_off.ctypes.data, dlen.ctypes.data, None, None, None,
                                  dst.ctypes.data)
        t3 = time.time()
        tt = (t1 - t0) + (t3 - t2)
        r = {"value": round(nbytes / tt / 1e6, 1), "unit": "MB/s", "blocks": nblk, "rounds": -(-nblk // ctx.last_slots),
             "comp_MBps": round(nbytes / (t1 - t0) / 1e6, 1), "decomp_MBps": round(nbytes / (t3 - t2) / 1e6, 1),
             "ok": bool(rc1 == 0 and rc2 == 0 and (st == 0).all() and (dst == 0).all()
                        and np.array_equal(p_dec.array, p_src.array)),
             "what": what}
    for pa in (p_src, p_out, p_dec, p_cod):
        pa.free()
    return r


class ResidentBatch:
    """nb blocks of `size` bytes resident in HBM plus every output buffer a round trip needs."""

    def __init__(self, z, ctx, torch, dev, nb, size, capmul=1.125):
        self.z, self.ctx, self.torch, self.nb, self.size = z, ctx, torch, nb, size
        self.cap = int(size * capmul) + 1024
        i64 = dict(dtype=torch.int64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.in_off = torch.arange(nb + 1, **i64) * size
        self.out_off = torch.arange(nb + 1, **i64) * self.cap
        self.d_out = torch.zeros(nb * self.cap, dtype=torch.uint8, device=dev)
        self.d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
        self.d_len, self.d_st, self.d_dlen, self.d_cons, self.d_code, self.d_first, self.d_dst = (
            torch.zeros(nb, **i32) for _ in range(7))
        self.enc_ms, self.dec_ms = [], []
        self.enc_name = self.dec_name = ""
        torch.cuda.synchronize()          # the ctx stream is non-blocking: order torch's fills before its kernels

    def step(self, model, d_in, flags, record=False):
        c = self.ctx
        c.encode_blocks_dev(model, self.nb, d_in.data_ptr(), self.in_off.data_ptr(), flags, self.d_out.data_ptr(),
                            self.out_off.data_ptr(), self.d_len.data_ptr(), self.d_st.data_ptr())
        if record:
            c.sync()
            self.enc_ms.append(c.last_kernel_ms)       # HIP events on the ctx stream around the kernel
            self.enc_name = c.last_kernel_name
        c.decode_blocks_dev(model, self.nb, self.d_out.data_ptr(), self.out_off.data_ptr(), flags, self.d_dec.data_ptr(),
                            self.in_off.data_ptr(), self.d_dlen.data_ptr(), self.d_cons.data_ptr(), self.d_code.data_ptr(),
                            self.d_first.data_ptr(), self.d_dst.data_ptr())
        if record:
            c.sync()
            self.dec_ms.append(c.last_kernel_ms)
            self.dec_name = c.last_kernel_name

    def ok(self, d_in):
        t = self.torch
        return bool((self.d_st == 0).all()) and bool((self.d_dst == 0).all()) and bool((self.d_dlen == self.size).all()) \
            and bool(t.equal(self.d_dec, d_in)) and bool((self.d_first == 0).all())

    def coded_bytes(self):
        return float(self.d_len.sum().item())


def roofline_of(A_minus_r, ratio, nb, size, enc_ms, dec_ms, names):
    A = A_minus_r + ratio
    dom_ms, dom = (dec_ms, names[1]) if dec_ms >= enc_ms else (enc_ms, names[0])
    achieved = A * nb * size / (dom_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_input_byte": round(A, 3),
            "encode_frac": round(A * nb * size / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "decode_frac": round(A * nb * size / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}


def kernel_src_sha():
    h = hashlib.sha256()
    for f in ("zpq_chain.hip", "zpq_pipe.hip", "zpq_chain_cfg.h", "zpq_common.h", "zpq_vm.h"):
        h.update(open(os.path.join(ROOT, "zpaq-v_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(dom_name, nb):
    """HBM bytes per launch of the dominant kernel from the round's rocprofv3 --pmc passes
    (profiles/r03_pmc.json, written by tools/pmc_collect.py).  Only reported when that profile was
    taken on exactly the kernel source this run executes; otherwise null, with the reason."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")), reverse=True)   # newest round first
    if not cands:
        return None, "no PMC profile in profiles/"
    why = []
    for pmc in cands:
        name = os.path.relpath(pmc, ROOT)
        try:
            pj = json.load(open(pmc))
            if pj.get("_kernel_src_sha") != kernel_src_sha():
                why.append("%s was taken on another kernel source (%s)" % (name, pj.get("_kernel_src_sha")))
                continue
            t = pj.get(dom_name, {}).get("hbm_bytes_per_launch")
            if t is None:
                why.append("%s: kernel not in profile" % name)
                continue
            return int(t * nb / pj.get("_blocks_per_launch", nb)), "%s: rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE passes, %s" % (name, pj.get("_note", ""))
        except Exception as e:                                      # noqa: BLE001
            why.append("%s unreadable: %r" % (name, e))
    return None, "; ".join(why[:2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=8192, help="blocks per GPU (weak scaling; C5 = 65536 blocks over 8 GPUs)")
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--size", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip per_class / value_incl_pcie / secondary (N=1 extras)")
    ap.add_argument("--cpu-blocks-per-thread", type=int, default=32)
    ap.add_argument("--cpu-one-thread-blocks", type=int, default=48)
    ap.add_argument("--backend", default="gloo",
                    help="process group for the barrier and the two scalar reductions of an N > 1 run: gloo (default: CPU tensors; the "
                         "data path has no collective and the north star says no RCCL -- this is also the path the rehearsals ran) or nccl (= RCCL)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--state-budget-gib", type=float, default=0.0, help="cap the per-ctx state pool (rehearsals sharing one GPU)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain invocation: be the launcher.  Nothing in this process has touched a GPU (torch is not even imported), the
        # ranks are fresh children, rank 0's JSON line goes to our stdout through the inherited descriptor.
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import workload as W

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert a.gpus == world, "--gpus %d but WORLD_SIZE is %d: the line would be mislabelled" % (a.gpus, world)
    if a.single_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            # one node: the ranks meet over the loopback interface (gloo would otherwise resolve the container's host name, which
            # need not resolve on these boxes)
            if os.path.isdir("/sys/class/net/lo"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group(backend=a.backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    z = ge.load()
    from zpaq_v_amd.sharding import shard_indices
    ctx = z.Context(local_rank)
    if a.state_budget_gib > 0:
        z.lib().zpq_ctx_set_state_budget(ctx.h, int(a.state_budget_gib * (1 << 30)))
    model = z.Model(level=a.level)
    nb, size = a.blocks, a.size
    total_blocks = nb * world
    mine = shard_indices(total_blocks, rank, world)          # block b -> rank b mod world
    assert len(mine) == nb
    # synthetic data of the global batch, this rank's round-robin share
    # (SURVEY.md 8(d)'s generator, one block per global index b; host threads only to build the batch in seconds)
    host = np.empty((nb, size), dtype=np.uint8)

    def _fill(i):
        host[i] = W.make_block(mine[i], size)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(_fill, range(nb), chunksize=64))
    d_in = torch.from_numpy(host.reshape(-1)).to(dev)
    flags = z.FLAG_PP
    rb = ResidentBatch(z, ctx, torch, dev, nb, size)

    def sync():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        rb.step(model, d_in, flags)
    sync()
    t0 = time.time()
    for _ in range(a.steps):
        rb.step(model, d_in, flags)
    sync()
    dt = time.time() - t0
    # separate, untimed pass for per-kernel durations (event queries would add host syncs to the timed region)
    rb.step(model, d_in, flags, record=True)
    sync()

    ok = rb.ok(d_in)
    coded_bytes = rb.coded_bytes()
    rdev = dev if a.backend == "nccl" else torch.device("cpu")   # the only cross-rank traffic: two tiny reductions
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    stats = torch.tensor([coded_bytes, 1.0 if ok else 0.0], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    all_ok = stats[1].item() == world
    # which physical GPU each rank ran on (so that a reader of the N-GPU line can see N distinct devices)
    props = torch.cuda.get_device_properties(local_rank)
    me = {"rank": rank, "device": local_rank, "name": props.name,
          "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
          "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid()}
    devices = [me]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, me)

    res = None
    if rank == 0:
        B_total = total_blocks * size
        ms_per_step = dt / a.steps * 1e3
        value = B_total / (dt / a.steps) / 1e6
        ratio = stats[0].item() / B_total
        enc_ms, dec_ms = rb.enc_ms[-1], rb.dec_ms[-1]
        roof = roofline_of(ALG_BYTES_PER_INPUT_BYTE_L.get(a.level, 195), ratio, nb, size, enc_ms, dec_ms,
                           (rb.enc_name, rb.dec_name))
        traffic, tnote = measured_traffic(roof["kernel"], nb)
        roof["traffic"] = traffic
        roof["traffic_source"] = tnote
        if traffic:
            # What the memory system gives THIS access pattern: the traffic is random 64-byte lines (16 bytes used of each) plus
            # the streaming clear of the tables.  tools/micro/linerate.hip on MI355X (profiles/r03_linerate.txt): 48.2-48.7 G
            # random lines/s read-only (3.1 TB/s), 25.3 G lines/s read + written back (50.6 G transfers/s), flat over
            # occupancy and loads in flight; the clear runs at 6.9 TB/s (tools/init_time.py: 14.7-15.3 ms per launch).
            clear_bytes = float(ctx.last_slots) * model.state_bytes if not ctx.last_line_store else 0.0
            clear_ms = clear_bytes / 6.9e12 * 1e3
            dom_ms = max(enc_ms, dec_ms)
            glines = (traffic - clear_bytes) / 64.0 / ((dom_ms - clear_ms) * 1e-3) / 1e9
            roof["random_lines"] = {"ceiling_G_per_s": RANDOM_LINE_CEILING_G, "achieved_G_per_s": round(glines, 2),
                                    "frac": round(glines / RANDOM_LINE_CEILING_G, 4), "clear_bytes": int(clear_bytes),
                                    "clear_ms_at_6.9TBps": round(clear_ms, 2),
                                    "what": "(traffic - table clear) / 64 B over (kernel time - clear time), against the chip's measured "
                                            "rate for random 64-byte lines (profiles/r03_linerate.txt)"}
        roof["cycles_per_coded_bit_at_2.4GHz"] = round(max(enc_ms, dec_ms) * 1e-3 * 2.4e9 / ((size + 1) * 8), 1)
        res = {
            "metric": "MB/s (comp+decomp) on level-2 64KiB blocks", "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "level %d (%s), %d x %d B blocks per GPU%s, one segment per "
                                   "block, classes b mod 4 = zeros/uniform/markov-text/periodic (SURVEY 8(d) generator, "
                                   "block b of the global batch on rank b mod N)" % (
                                       a.level, LEVEL_NAMES.get(a.level, "?"), nb, size,
                                       " (C5's per-GPU share of 65536 blocks over 8 GPUs)" if nb == 8192 else ""),
                       "blocks_per_gpu": nb, "block_bytes": size, "level": a.level,
                       "parallelism": "block b -> gpu b mod %d, no collective" % world},
            "value_definition": "device-resident: inputs and outputs in HBM when the timed region starts and ends; value_incl_pcie "
                                "(N = 1) is SURVEY 8(d)'s t, with H2D/D2H inside the calls",
            "backend": a.backend if world > 1 else None,
            "roundtrip_bit_exact": all_ok, "ratio": round(ratio, 4),
            "coded_bytes": int(stats[0].item()),
            "comp_MBps": round(nb * size / (enc_ms * 1e-3) / 1e6, 1),
            "decomp_MBps": round(nb * size / (dec_ms * 1e-3) / 1e6, 1),
            "kernel_ms": {rb.enc_name: round(enc_ms, 3), rb.dec_name: round(dec_ms, 3)},
            "resident_blocks": ctx.last_slots,
            "devices": devices,
            "roofline": roof,
        }

    if rank == 0 and world == 1 and not a.no_secondary:
        # Every extra below is optional: a failure there (say, no room for the pinned buffers) is recorded in the line,
        # it never costs the headline.
        extras_failed = {}
        cap = rb.cap

        def _per_class():
            # ---- per class: the headline workload, one block class at a time (each class's blocks of the headline
            #      batch repeated to the same launch size, so residency and launch shape are the headline's)
            per_class = {}
            names = ["zeros", "uniform", "markov_text", "periodic"]
            for cls in range(4):
                idx = torch.arange(cls, nb, 4, device=dev)
                reps = (nb + idx.numel() - 1) // idx.numel()
                sel = idx.repeat(reps)[:nb]
                d_cls = d_in.view(nb, size).index_select(0, sel).reshape(-1).contiguous()
                torch.cuda.synchronize()
                rb.enc_ms.clear(); rb.dec_ms.clear()
                rb.step(model, d_cls, flags)
                rb.step(model, d_cls, flags, record=True)
                okc = rb.ok(d_cls)
                e, d = rb.enc_ms[-1], rb.dec_ms[-1]
                per_class[names[cls]] = {"comp_MBps": round(nb * size / e / 1e3, 1), "decomp_MBps": round(nb * size / d / 1e3, 1),
                                         "roundtrip_MBps": round(nb * size / (e + d) / 1e3, 1),
                                         "ratio": round(rb.coded_bytes() / (nb * size), 4), "roundtrip_bit_exact": okc}
                del d_cls
            res["per_class"] = per_class


        def _incl_pcie():
            # ---- PCIe-inclusive: the same batch through the host-pointer entry points (what a V front end holding
            #      host buffers calls): H2D + kernel + D2H inside each call.  Buffers come from zpq_host_alloc (pinned);
            #      the decoder is handed the coded streams packed back to back, as an archive holds them.
            L = z.lib()

            def incl_pcie(nblk, arr2d, what):
                return incl_pcie_run(z, ctx, model, flags, size, cap, nblk, arr2d, what)

            pc = incl_pcie(nb, host, "zpq_encode_blocks + zpq_decode_blocks on pinned host buffers (zpq_host_alloc), the headline batch in "
                                     "one round: striped upload / download beside the kernel (host_pipeline, HIO kernels); second of two calls")
            res["value_incl_pcie"] = pc["value"]
            res["incl_pcie"] = pc
            res["incl_pcie"]["fraction_of_device_resident"] = round(pc["value"] / res["value"], 4)
            host2 = np.concatenate([host, host, host, host])
            ps = incl_pcie(4 * nb, host2, "the same calls on a batch of four times the resident capacity: four rounds, the upload of round "
                                          "r+1 and the download of round r-1 overlap the coding of round r (host_pipeline)")
            res["incl_pcie_streamed"] = ps
            res["incl_pcie_streamed"]["fraction_of_device_resident"] = round(ps["value"] / res["value"], 4)
            del host2


        def _secondary():
            # ---- secondary configs, device-resident like the headline
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
            from inputs import C4B
            secondary = {}

            def run_cfg(key, what, mdl, want_nb, A_minus_r, capmul, names_, pcie=False):
                cap_res = ctx.resident_capacity(mdl, flags)
                n = min(want_nb, cap_res) if want_nb else cap_res
                arr = host[:n] if n <= nb else np.concatenate([host] * ((n + nb - 1) // nb))[:n]
                d = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1)).to(dev)
                b = ResidentBatch(z, ctx, torch, dev, n, size, capmul)
                b.step(mdl, d, flags, record=True)
                b.step(mdl, d, flags, record=True)
                e, dd = min(b.enc_ms), min(b.dec_ms)
                r = b.coded_bytes() / (n * size)
                secondary[key] = {
                    "workload": what, "blocks": n, "resident_blocks": ctx.last_slots, "resident_capacity": cap_res,
                    "comp_MBps": round(n * size / e / 1e3, 1), "decomp_MBps": round(n * size / dd / 1e3, 1),
                    "roundtrip_MBps": round(n * size / (e + dd) / 1e3, 1), "ratio": round(r, 4),
                    "kernel_ms": {b.enc_name: round(e, 2), b.dec_name: round(dd, 2)},
                    "roundtrip_bit_exact": b.ok(d),
                    "roofline": roofline_of(A_minus_r, r, n, size, e, dd, (b.enc_name, b.dec_name))}
                cap_ = b.cap
                del b, d
                torch.cuda.empty_cache()
                if pcie:
                    # the same batch through the host-pointer entry points (pinned buffers; the line-store kernels have no
                    # striped transfers: upload, kernel, download in turn)
                    pc = incl_pcie_run(z, ctx, mdl, flags, size, cap_, n, np.ascontiguousarray(arr), "zpq_encode_blocks + zpq_decode_blocks on "
                                       "pinned host buffers, one round, second of two calls")
                    pc["fraction_of_device_resident"] = round(pc["value"] / secondary[key]["roundtrip_MBps"], 4)
                    secondary[key]["incl_pcie"] = pc

            run_cfg("C2_level1", "level 1 (%s, levels.v:53-92), 4096 x 64 KiB" % LEVEL_NAMES[1], z.Model(level=1), 4096,
                    ALG_BYTES_PER_INPUT_BYTE_L[1], 1.125, None)
            run_cfg("level3", "level 3 (%s), 64 KiB blocks at resident capacity" % LEVEL_NAMES[3], z.Model(level=3), 0,
                    ALG_BYTES_PER_INPUT_BYTE_L[3], 1.125, None, pcie=True)
            run_cfg("level4", "level 4 (%s), 64 KiB blocks at resident capacity" % LEVEL_NAMES[4], z.Model(level=4), 0,
                    ALG_BYTES_PER_INPUT_BYTE_L[4], 1.125, None)
            run_cfg("C4a_level5", "level 5 as shipped (%s, levels.v:294-335), 64 KiB blocks at resident capacity" % LEVEL_NAMES[5],
                    z.Model(level=5), 0, ALG_BYTES_PER_INPUT_BYTE_L[5], 1.125, None, pcie=True)
            run_cfg("C4b_all_nine_types", "synthetic header with all nine component types (SURVEY 8(d) C4b), 64 KiB blocks at resident "
                                          "capacity (encode k_gpipe: a wave per component, lane = block, a pipeline over bytes; decode k_gdec: the same waves, bit-synchronous)", z.Model(header=C4B), 0, ALG_BYTES_C4B, 6.0, None)
            res["secondary"] = secondary

        for _name, _fn in (("per_class", _per_class), ("incl_pcie", _incl_pcie), ("secondary", _secondary)):
            try:
                _fn()
            except Exception as e:                                   # noqa: BLE001
                extras_failed[_name] = repr(e)[:300]
            if _name == "per_class":                                 # the headline's device buffers are not needed any more
                rb = None
                torch.cuda.empty_cache()
                try:
                    torch.cuda.empty_cache()
                except Exception:                                    # noqa: BLE001
                    pass
        if extras_failed:
            res["extras_failed"] = extras_failed

    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.level, size, host, a.cpu_blocks_per_thread, a.cpu_one_thread_blocks)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if not all_ok:
        sys.exit(3)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ZPAQ block codec.

A "step" = one pass of the hot path over one batch: level-2 compress of
`--blocks` synthetic 64 KiB blocks per GPU followed by decompress of the coded
streams (both through the C ABI's device-pointer entry points, inputs already
resident in HBM).  value = uncompressed MB / (t_comp + t_decomp), whole job.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One rank per GPU; block b of the global batch goes to rank b mod N (static
round-robin, no data-path collective); timing is barrier + synchronize on both
sides, MAX over ranks; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALG_BYTES_PER_INPUT_BYTE_L = {1: 131, 2: 195, 3: 323, 4: 419, 5: 547}   # A(L) = this + r
HBM_PEAK_GBS = 8000.0                                                  # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(level, size, per_thread_blocks=24):
    """The C oracle (a restatement of the V CPU path -- V itself cannot be built
    here) timed on this host: compress+decompress of a bounded sample of the same
    synthetic blocks, blocks divided statically over `cores` pthreads."""
    import numpy as np
    import oracle_lib as O
    import workload as W
    cores = min(os.cpu_count() or 1, 16)
    nb = per_thread_blocks * cores
    arr = W.make_blocks_fast(nb, size)
    blocks = [arr[i].tobytes() for i in range(nb)]
    hdr = O.level_header(level)
    t0 = time.time()
    coded = O.encode_blocks(hdr, blocks, nthreads=cores, slack=size + size // 8 + 1024)
    t1 = time.time()
    dec = O.decode_blocks(hdr, coded, cap=size + 16, nthreads=cores)
    t2 = time.time()
    assert dec == [b"\0" + b for b in blocks]
    B = nb * size
    return {
        "value": round(B / (t2 - t0) / 1e6, 3), "unit": "MB/s", "cores": cores, "kind": "port",
        "comp_MBps": round(B / (t1 - t0) / 1e6, 3), "decomp_MBps": round(B / (t2 - t1) / 1e6, 3),
        "sample": "%d synthetic 64 KiB blocks (same generator, classes b mod 4), level %d, C oracle "
                  "compress+decompress, %d pthreads, per-block table alloc+zero-fill included" % (nb, level, cores),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=8192, help="blocks per GPU (weak scaling; C5 = 65536 blocks over 8 GPUs)")
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--size", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (with --backend gloo)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import workload as W

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.single_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=a.backend)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    z = ge.load()
    from zpaq_v_amd.sharding import shard_indices
    ctx = z.Context(local_rank)
    model = z.Model(level=a.level)
    nb, size = a.blocks, a.size
    total_blocks = nb * world
    mine = shard_indices(total_blocks, rank, world)          # block b -> rank b mod world
    assert len(mine) == nb
    # synthetic data of the global batch, this rank's round-robin share
    host = np.empty((nb, size), dtype=np.uint8)
    pool = {}
    for i, b in enumerate(mine):
        if b % 4 == 2:
            key = 2 + 4 * ((b // 4) % 64)
            if key not in pool:
                pool[key] = W.make_block(key, size)
            host[i] = pool[key]
        else:
            host[i] = W.make_block(b, size)
    d_in = torch.from_numpy(host.reshape(-1)).to(dev)
    cap = size + size // 8 + 1024
    i64 = dict(dtype=torch.int64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    in_off = torch.arange(nb + 1, **i64) * size
    out_off = torch.arange(nb + 1, **i64) * cap
    dec_off = torch.arange(nb + 1, **i64) * size
    d_out = torch.zeros(nb * cap, dtype=torch.uint8, device=dev)
    d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
    d_len, d_st = torch.zeros(nb, **i32), torch.zeros(nb, **i32)
    d_dlen, d_cons, d_code, d_first, d_dst = (torch.zeros(nb, **i32) for _ in range(5))
    flags = z.FLAG_PP

    def sync():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    enc_ms, dec_ms = [], []

    def step(record):
        ctx.encode_blocks_dev(model, nb, d_in.data_ptr(), in_off.data_ptr(), flags, d_out.data_ptr(),
                              out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
        if record:
            ctx.sync()
            enc_ms.append(ctx.last_kernel_ms)      # HIP events on the ctx stream around the kernel
        ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), flags, d_dec.data_ptr(),
                              dec_off.data_ptr(), d_dlen.data_ptr(), d_cons.data_ptr(), d_code.data_ptr(),
                              d_first.data_ptr(), d_dst.data_ptr())
        if record:
            ctx.sync()
            dec_ms.append(ctx.last_kernel_ms)

    for _ in range(a.warmup):
        step(False)
    sync()
    t0 = time.time()
    for _ in range(a.steps):
        step(False)
    sync()
    dt = time.time() - t0
    # separate, untimed pass for per-kernel durations (event queries would add host syncs to the timed region)
    step(True)
    sync()

    ok = bool((d_st == 0).all()) and bool((d_dst == 0).all()) and bool((d_dlen == size).all()) \
        and bool(torch.equal(d_dec, d_in)) and bool((d_first == 0).all())
    coded_bytes = float(d_len.sum().item())
    rdev = dev if a.backend == "nccl" else torch.device("cpu")   # the only cross-rank traffic: two tiny reductions
    tmax = torch.tensor([dt], dtype=torch.float64, device=rdev)
    stats = torch.tensor([coded_bytes, 1.0 if ok else 0.0], dtype=torch.float64, device=rdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    all_ok = stats[1].item() == world

    if rank == 0:
        B_total = total_blocks * size
        ms_per_step = dt / a.steps * 1e3
        value = B_total / (dt / a.steps) / 1e6
        ratio = stats[0].item() / B_total
        A = ALG_BYTES_PER_INPUT_BYTE_L.get(a.level, 195) + ratio
        dom_ms, dom_name = (dec_ms[-1], "k_chain<decode>") if dec_ms[-1] >= enc_ms[-1] else (enc_ms[-1], "k_chain<encode>")
        # PMC traffic was collected at 4096 blocks per launch; scale to this launch size
        pmc_blocks = 4096
        launch_bytes = A * nb * size                          # algorithmic bytes one launch moves
        achieved = launch_bytes / (dom_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get(dom_name, {}).get("hbm_bytes_per_launch")
                pmc_blocks = pj.get("_blocks_per_launch", 4096)
                if traffic is not None:
                    traffic = int(traffic * nb / pmc_blocks)
            except Exception:
                traffic = None
        res = {
            "metric": "MB/s (comp+decomp) on level-2 64KiB blocks", "value": round(value, 2), "unit": "MB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "level %d (ICM16+ISSE16+ISSE16), %d x %d B blocks per GPU%s, one segment per "
                                   "block, classes b mod 4 = zeros/uniform/markov-text/periodic (text blocks drawn "
                                   "from 64 distinct generated blocks)" % (
                                       a.level, nb, size,
                                       " (C5's per-GPU share of 65536 blocks over 8 GPUs)" if nb == 8192 else ""),
                       "blocks_per_gpu": nb, "block_bytes": size, "level": a.level,
                       "parallelism": "block b -> gpu b mod %d, no collective" % world},
            "roundtrip_bit_exact": all_ok, "ratio": round(ratio, 4),
            "comp_MBps": round(nb * size / (enc_ms[-1] * 1e-3) / 1e6, 1),
            "decomp_MBps": round(nb * size / (dec_ms[-1] * 1e-3) / 1e6, 1),
            "kernel_ms": {"k_chain<encode>": round(enc_ms[-1], 3), "k_chain<decode>": round(dec_ms[-1], 3)},
            "resident_blocks": ctx.last_slots,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_input_byte": round(A, 3),
                         "cycles_per_coded_bit_at_2.4GHz": round(dom_ms * 1e-3 * 2.4e9 / ((size + 1) * 8), 1)},
        }
        if not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.level, size)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    if not all_ok:
        sys.exit(3)


if __name__ == "__main__":
    main()

"""N>1 path on CPU: two gloo ranks shard a batch round-robin (block b -> rank b mod
2), code their shares independently (the ORACLE stands in for the GPU codec here --
this test is about the partitioning/merge plumbing bench.py and a multi-GPU host
use), and the merged result equals the single-rank result bit for bit."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as ge
    import oracle_lib as O
    import workload as W
    ge.load()
    from zpaq_v_amd.sharding import shard_indices
    total = 10
    mine = shard_indices(total, rank, world)
    hdr = O.level_header(2)
    blocks = [W.make_block(b, 2048).tobytes() for b in mine]
    coded = O.encode_blocks(hdr, blocks)
    # the only cross-rank traffic: a max-reduced time and summed byte counts, as in bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    nbytes = torch.tensor([float(sum(len(c) for c in coded))], dtype=torch.float64)
    dist.all_reduce(nbytes, op=dist.ReduceOp.SUM)
    q.put((rank, coded, t.item(), nbytes.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_two_ranks_matches_single_rank():
    import oracle_lib as O
    import workload as W
    import __graft_entry__ as ge
    ge.load()
    from zpaq_v_amd.sharding import merge_in_block_order, shard_indices
    world, total = 2, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, coded, tmax, nbytes = q.get(timeout=120)
        got[r] = (coded, tmax, nbytes)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    merged = merge_in_block_order([got[r][0] for r in range(world)], total)
    single = O.encode_blocks(O.level_header(2), [W.make_block(b, 2048).tobytes() for b in range(total)])
    assert merged == single
    assert got[0][1] == got[1][1] == 2.0
    assert got[0][2] == got[1][2] == float(sum(len(c) for c in single))
    assert shard_indices(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((shard_indices(11, r, 3) for r in range(3)), [])) == list(range(11))


def test_bench_refuses_a_gpus_flag_that_is_not_the_world_size():
    """bench.py --gpus N inside a launcher's rank must see WORLD_SIZE == N; a mislabelled line is worse than none
    (`--gpus` used to be parsed and never read).  Fails before anything touches a GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE is 1" in r.stderr, r.stderr[-1500:]


def test_markov_generator_forms_agree():
    """The vectorised order-1 Markov text generator (one prefix sum per block) equals SURVEY 8(d)'s symbol-by-symbol recurrence."""
    import numpy as np
    import workload as W
    for b in (2, 6, 4094, 123458):
        for size in (1, 5, 4096, 65536):
            r = W._splitmix_stream((W.SEED0 + b) & W.MASK, size)
            fav = ((r >> np.uint64(8)) & np.uint64(3)).astype(np.int64)
            uni = ((r >> np.uint64(16)) & np.uint64(63)).astype(np.int64)
            uf = (r & np.uint64(3)) != 0
            assert np.array_equal(W._markov_text(uf, fav, uni), W._markov_text_loop(uf, fav, uni)), (b, size)

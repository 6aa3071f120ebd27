"""GPU parity: the LDS-resident chain kernel (ICM + ISSE chain [+ MIX2], every
shipped level) against the CPU oracle and the generic kernel, through the C ABI."""
import hashlib
import json
import os
import random
import sys

import numpy as np
import pytest

import oracle_lib as O
import workload as W

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from inputs import C4B, INPUTS  # noqa: E402

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


def mixed_blocks(rnd, count, sizes):
    blocks = []
    for i in range(count):
        n = rnd.choice(sizes)
        kind = i % 4
        if kind == 0:
            b = bytes(n)
        elif kind == 1:
            b = bytes(rnd.getrandbits(8) for _ in range(n))
        elif kind == 2:
            b = bytes(rnd.choice(b"etaoin shrdlu\n") for _ in range(n))
        else:
            per = bytes(rnd.getrandbits(8) for _ in range(rnd.randint(1, 40)))
            b = (per * (n // len(per) + 1))[:n]
        blocks.append(b)
    return blocks


@pytest.mark.parametrize("level", [1, 2, 3, 4, 5])
def test_chain_matches_golden_streams(zpq, gpu_ctx, level):
    model = zpq.Model(level=level)
    assert model.has_fast_path
    for mode in ("pp", "raw"):
        ks = [k for k in sorted(G["streams"]) if k.startswith("%d/" % level) and k.endswith(mode)]
        blocks = [INPUTS[k.split("/")[1]] for k in ks]
        flags = zpq.FLAG_PP if mode == "pp" else 0
        coded, status, _ = gpu_ctx.encode_blocks(model, blocks, flags=flags)
        pipe = "k_pipe2<encode>" if level == 1 else "k_pipe<encode>"     # (level 1, dense: every stage split into two waves)
        assert gpu_ctx.last_kernel_name == (pipe if len(blocks) >= 12 else "k_chain<encode>")
        assert (status == 0).all()
        for k, c in zip(ks, coded):
            assert hashlib.sha256(c).hexdigest() == G["streams"][k]["sha256"], k
        dec, status, consumed, code, first = gpu_ctx.decode_blocks(model, coded, cap=8192, flags=flags)
        assert gpu_ctx.last_kernel_name == "k_chain<decode>"
        assert (status == 0).all() and dec == blocks
        assert [int(c) for c in consumed] == [G["streams"][k]["consumed"] for k in ks]


@pytest.mark.parametrize("level", [1, 2, 3])
def test_chain_ragged_batch_vs_oracle(zpq, gpu_ctx, level):
    """Ragged sizes (0, 1, odd, > one nibble row reuse), more blocks than one workgroup holds."""
    rnd = random.Random(1000 + level)
    hdr = O.level_header(level)
    model = zpq.Model(level=level)
    blocks = mixed_blocks(rnd, 70, [0, 1, 2, 3, 15, 16, 17, 255, 256, 257, 1000, 2500])
    coded, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert (status == 0).all()
    want = O.encode_blocks(hdr, blocks, nthreads=4)
    assert coded == want
    dec, status, consumed, code, first = gpu_ctx.decode_blocks(model, coded, cap=4096)
    assert (status == 0).all() and dec == blocks and (first == 0).all()
    assert [int(c) for c in consumed] == [len(c) for c in coded]
    # generic kernel agrees too (two independent device implementations)
    coded_g, status, _ = gpu_ctx.encode_blocks(model, blocks, flags=zpq.FLAG_PP | zpq.FLAG_GENERIC)
    assert coded_g == coded


def test_chain_level2_64k_blocks_all_classes(zpq, gpu_ctx):
    """The bench workload's four block classes at full 64 KiB size, level 2."""
    hdr = O.level_header(2)
    model = zpq.Model(level=2)
    arr = W.make_blocks(8, 65536)
    blocks = [arr[i].tobytes() for i in range(8)]
    coded, status, _ = gpu_ctx.encode_blocks(model, blocks, cap=80000)
    assert (status == 0).all()
    want = O.encode_blocks(hdr, blocks, nthreads=8, slack=80000)
    assert coded == want
    dec, status, *_ = gpu_ctx.decode_blocks(model, coded, cap=65536)
    assert (status == 0).all() and dec == blocks


def test_chain_slot_reuse_more_blocks_than_slots(zpq, gpu_ctx):
    """Persistent groups re-initialise their slot between blocks: restrict the
    state budget so that 100 blocks share 37 slots (one full workgroup + a partial one)."""
    model = zpq.Model(level=2)
    zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 37 * model.state_bytes + 1000)
    try:
        rnd = random.Random(77)
        blocks = mixed_blocks(rnd, 100, [300, 1200, 2048])
        coded, status, _ = gpu_ctx.encode_blocks(model, blocks)
        assert gpu_ctx.last_slots == 37
        assert (status == 0).all()
        assert coded == O.encode_blocks(O.level_header(2), blocks, nthreads=4)
        dec, status, *_ = gpu_ctx.decode_blocks(model, coded, cap=4096)
        assert (status == 0).all() and dec == blocks
    finally:
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)


def test_chain_generic_vm_program(zpq, gpu_ctx):
    """A chain model whose HCOMP is NOT one of the recognised shapes goes through
    the in-kernel ZPAQL interpreter."""
    hdr = bytes([3, 8, 0, 0, 2, 3, 16, 8, 16, 0, 0,
                 104, 17, 95, 0, 59, 135, 7, 112, 25, 60, 59, 112, 56, 0])
    offs = O.scan_header(hdr)
    model = zpq.Model(header=hdr)
    assert model.has_fast_path
    rnd = random.Random(5)
    blocks = mixed_blocks(rnd, 12, [100, 700, 1500])
    coded, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert gpu_ctx.last_kernel_name == "k_chain<encode>"
    assert (status == 0).all()
    assert coded == O.encode_blocks(hdr, blocks, nthreads=4)
    dec, status, *_ = gpu_ctx.decode_blocks(model, coded, cap=2048)
    assert (status == 0).all() and dec == blocks


def test_chain_overflow_status(zpq, gpu_ctx):
    model = zpq.Model(level=2)
    rnd = random.Random(5)
    data = bytes(rnd.getrandbits(8) for _ in range(2000))
    coded, status, out_len = gpu_ctx.encode_blocks(model, [data], cap=100)
    assert status[0] == -7 and int(out_len[0]) == len(O.Codec(model.header).encode(data))
    good, status, _ = gpu_ctx.encode_blocks(model, [data])
    dec, status, *_ = gpu_ctx.decode_blocks(model, good, cap=100)
    assert status[0] == -7


def test_chain_full_size_batch_properties(zpq, gpu_ctx):
    """BASELINE size (level 2, 8192 x 64 KiB, buffers resident in HBM): size-independent
    properties -- encode -> decode is the identity on every block, every status is OK, the
    decoder consumed exactly the bytes the encoder produced -- plus byte parity with the
    CPU oracle on EVERY block and a checksum of checksums over all coded
    streams that must not depend on how blocks were grouped into launches."""
    import torch
    nb, size = 8192, 65536
    arr = W.make_blocks_fast(nb, size)
    dev = torch.device("cuda:0")
    model = zpq.Model(level=2)
    cap = size + size // 8 + 1024
    d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
    i64 = dict(dtype=torch.int64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    in_off = torch.arange(nb + 1, **i64) * size
    out_off = torch.arange(nb + 1, **i64) * cap
    d_out = torch.zeros(nb * cap, dtype=torch.uint8, device=dev)
    d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
    d_len, d_st, d_dlen, d_cons, d_code, d_first, d_dst = (torch.zeros(nb, **i32) for _ in range(7))
    torch.cuda.synchronize()                             # the _dev forms run on the ctx's own non-blocking stream
    gpu_ctx.encode_blocks_dev(model, nb, d_in.data_ptr(), in_off.data_ptr(), zpq.FLAG_PP, d_out.data_ptr(),
                              out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
    gpu_ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), zpq.FLAG_PP, d_dec.data_ptr(),
                              in_off.data_ptr(), d_dlen.data_ptr(), d_cons.data_ptr(), d_code.data_ptr(),
                              d_first.data_ptr(), d_dst.data_ptr())
    gpu_ctx.sync()
    assert gpu_ctx.last_slots == nb                      # all blocks resident: 32 per CU
    assert bool((d_st == 0).all()) and bool((d_dst == 0).all())
    assert bool((d_dlen == size).all()) and bool((d_first == 0).all())
    assert bool(torch.equal(d_dec, d_in))
    assert bool(torch.equal(d_cons, d_len))              # decoder pulled exactly what the encoder put
    assert bool((d_code == -1).all())                    # Decoder.code after EOF = the 4 flush bytes FF FF FF FF
    lens = d_len.cpu().numpy()
    out = d_out.cpu().numpy()
    # EVERY coded stream of the headline batch against the oracle (a 24-block sample covered 0.3 %):
    # the oracle codes 8192 x 64 KiB in about 8 s on the box's 16 host threads, in chunks that keep its slabs small
    nthreads = min(16, os.cpu_count() or 1)
    for c0 in range(0, nb, 1024):
        idx = range(c0, min(nb, c0 + 1024))
        want = O.encode_blocks(model.header, [arr[i].tobytes() for i in idx], nthreads=nthreads, slack=cap)
        for i, w in zip(idx, want):
            assert int(lens[i]) == len(w) and out[i * cap:i * cap + len(w)].tobytes() == w, i
    # checksum of checksums: same 512 blocks coded in a separate, smaller launch
    sub = list(range(0, nb, 16))
    coded_sub, status, _ = gpu_ctx.encode_blocks(model, [arr[i].tobytes() for i in sub], cap=cap)
    assert (status == 0).all()
    h_big, h_small = hashlib.sha256(), hashlib.sha256()
    for i, c in zip(sub, coded_sub):
        h_big.update(hashlib.sha256(out[i * cap:i * cap + int(lens[i])].tobytes()).digest())
        h_small.update(hashlib.sha256(c).digest())
    assert h_big.digest() == h_small.digest()


@pytest.mark.parametrize("level", [1, 2])
def test_chain_large_blocks(zpq, gpu_ctx, level):
    """Blocks much larger than 64 KiB: C1's 1 MiB of zeros (M array wraps 16x at level 2)
    and 256 KiB of mixed data, against the oracle."""
    hdr = O.level_header(level)
    model = zpq.Model(level=level)
    rnd = random.Random(31 + level)
    mixed = bytes(rnd.getrandbits(8) for _ in range(65536)) + bytes(65536) + \
        bytes(rnd.choice(b"abcdefgh\n") for _ in range(131072))
    blocks = [bytes(1 << 20), mixed]
    coded, status, _ = gpu_ctx.encode_blocks(model, blocks, cap=300000)
    assert (status == 0).all()
    assert coded == O.encode_blocks(hdr, blocks, nthreads=2, slack=300000)
    dec, status, consumed, *_ = gpu_ctx.decode_blocks(model, coded, cap=1 << 20)
    assert (status == 0).all() and dec == blocks
    assert [int(c) for c in consumed] == [len(c) for c in coded]


def test_decoder_on_garbage_terminates_with_status(zpq, gpu_ctx):
    """Corrupt or truncated coded input must end in a per-block status (or a short
    output), never a hang or a fault: both kernels, tiny output slabs."""
    rnd = random.Random(77)
    model = zpq.Model(level=2)
    good = gpu_ctx.encode_blocks(model, [bytes(rnd.getrandbits(8) for _ in range(3000))])[0][0]
    garbage = [bytes(rnd.getrandbits(8) for _ in range(2000)), bytes(2000), b"\xff" * 2000,
               good[:len(good) // 2], good[:3], b""]
    for flags in (zpq.FLAG_PP, zpq.FLAG_PP | zpq.FLAG_GENERIC):
        dec, status, consumed, code, first = gpu_ctx.decode_blocks(model, garbage, cap=4096, flags=flags)
        assert all(int(s) in (0, -7) for s in status)
        assert all(len(d) <= 4096 for d in dec)
        assert all(int(c) <= len(g) for c, g in zip(consumed, garbage))
    # both kernels agree on what garbage decodes to (same arithmetic, same EOF handling)
    d1 = gpu_ctx.decode_blocks(model, garbage, cap=4096, flags=zpq.FLAG_PP)
    d2 = gpu_ctx.decode_blocks(model, garbage, cap=4096, flags=zpq.FLAG_PP | zpq.FLAG_GENERIC)
    assert d1[0] == d2[0] and (d1[1] == d2[1]).all()
    # and with the oracle where the oracle terminates within the slab
    for g, d, s in zip(garbage, d1[0], d1[1]):
        if int(s) == 0:
            try:
                want, _ = O.Codec(model.header).decode(g, cap=4097)
            except OverflowError:
                continue
            assert want[1:] == d or (len(want) == 0 and d == b"")


def test_empty_batch_and_bad_arguments(zpq, gpu_ctx):
    model = zpq.Model(level=2)
    coded, status, _ = gpu_ctx.encode_blocks(model, [])
    assert coded == [] and len(status) == 0
    L = zpq.lib()
    assert L.zpq_encode_blocks(gpu_ctx.h, model.h, -1, None, None, 0, None, None, None, None) == -2
    assert L.zpq_encode_blocks(None, model.h, 1, None, None, 0, None, None, None, None) == -2
    off = np.array([0, 5], dtype=np.uint64)
    bad = np.array([5, 0], dtype=np.uint64)                 # decreasing offsets
    out = np.zeros(64, dtype=np.uint8); ol = np.zeros(1, dtype=np.uint32); st = np.zeros(1, dtype=np.int32)
    src = np.zeros(8, dtype=np.uint8)
    assert L.zpq_encode_blocks(gpu_ctx.h, model.h, 1, src.ctypes.data, bad.ctypes.data, 0, out.ctypes.data,
                               off.ctypes.data, ol.ctypes.data, st.ctypes.data) == -2


def test_compact_line_store_forced_on_small_model(zpq, gpu_ctx, monkeypatch):
    """The compact line store (dense line index -> slot, open addressing) must be invisible
    to the coder.  Force it onto level 2 with a tiny 8192-line store so that probing and
    collisions are exercised, and compare with the oracle and the dense run."""
    hdr = O.level_header(2)
    model = zpq.Model(level=2)
    rnd = random.Random(4242)
    blocks = mixed_blocks(rnd, 40, [0, 1, 17, 300, 1000, 1900])
    dense, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert (status == 0).all()
    monkeypatch.setenv("ZPQ_SPARSE_FORCE_LOG2", "13")
    sparse, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert (status == 0).all()
    assert sparse == dense == O.encode_blocks(hdr, blocks, nthreads=4)
    dec, status, *_ = gpu_ctx.decode_blocks(model, sparse, cap=4096)
    assert (status == 0).all() and dec == blocks
    # a block that needs more lines than the store holds is refused, not miscoded
    monkeypatch.setenv("ZPQ_SPARSE_FORCE_LOG2", "10")
    big = [bytes(rnd.getrandbits(8) for _ in range(3000))]
    _, status, _ = gpu_ctx.encode_blocks(model, big)
    assert status[0] == -4


def test_level5_uses_compact_store_and_matches_oracle(zpq, gpu_ctx):
    """Level 5 (ICM22 + 7 x ISSE22 + MIX2): 2 GiB of dense tables per block becomes ~140 MiB,
    so hundreds of blocks are resident; coded streams still equal the oracle's."""
    model = zpq.Model(level=5)
    hdr = O.level_header(5)
    arr = W.make_blocks(4, 65536, start=1)
    blocks = [arr[i].tobytes() for i in range(4)] + [b"", b"abc" * 50]
    blocks = blocks * 40                                   # 240 blocks: far more than 2 GiB slots would allow
    coded, status, _ = gpu_ctx.encode_blocks(model, blocks, cap=80000)
    assert (status == 0).all()
    assert gpu_ctx.last_slots == len(blocks)
    want = O.encode_blocks(hdr, blocks[:6], nthreads=3, slack=80000)
    assert coded[:6] == want and coded[6:12] == want       # same content -> same stream in any slot
    dec, status, *_ = gpu_ctx.decode_blocks(model, coded[:12], cap=65536)
    assert (status == 0).all() and dec == blocks[:12]


def test_back_to_back_launches_do_not_see_each_others_state(zpq, gpu_ctx):
    """The slot pool is reused by every launch and by every kernel family: results must not depend on what
    the pool held before (in-kernel re-initialisation), nor on another kernel having used it in between."""
    import workload as W
    model = zpq.Model(level=2)
    c4b = zpq.Model(header=C4B)
    for rnd_ in range(5):
        blocks = [bytes(W.make_block(64 * rnd_ + b, 3000 + 17 * b)) for b in range(40)]
        coded, status, _ = gpu_ctx.encode_blocks(model, blocks)
        assert gpu_ctx.last_kernel_name == "k_pipe<encode>"
        assert (status == 0).all()
        assert coded == O.encode_blocks(model.header, blocks, nthreads=4)
        if rnd_ == 2:                                       # another kernel works in pool 0 in between
            other, st2, _ = gpu_ctx.encode_blocks(c4b, blocks[:3])
            assert (st2 == 0).all() and other == O.encode_blocks(C4B, blocks[:3])
        dec, status, *_ = gpu_ctx.decode_blocks(model, coded, cap=4096)
        assert (status == 0).all() and dec == blocks


def _device_round_trip(zpq, ctx, model, arr, capmul, flags=None):
    """encode + decode of a resident batch; returns (coded slabs as numpy, lens, cap) after the property checks."""
    import torch
    flags = zpq.FLAG_PP if flags is None else flags
    nb, size = arr.shape
    dev = torch.device("cuda:0")
    cap = int(size * capmul) + 1024
    d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
    i64 = dict(dtype=torch.int64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    in_off = torch.arange(nb + 1, **i64) * size
    out_off = torch.arange(nb + 1, **i64) * cap
    d_out = torch.zeros(nb * cap, dtype=torch.uint8, device=dev)
    d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
    d_len, d_st, d_dlen, d_cons, d_code, d_first, d_dst = (torch.zeros(nb, **i32) for _ in range(7))
    torch.cuda.synchronize()                             # order torch's fills before the ctx stream's kernels
    ctx.encode_blocks_dev(model, nb, d_in.data_ptr(), in_off.data_ptr(), flags, d_out.data_ptr(),
                          out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
    ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), flags, d_dec.data_ptr(),
                          in_off.data_ptr(), d_dlen.data_ptr(), d_cons.data_ptr(), d_code.data_ptr(),
                          d_first.data_ptr(), d_dst.data_ptr())
    ctx.sync()
    assert bool((d_st == 0).all()) and bool((d_dst == 0).all())
    assert bool((d_dlen == size).all()) and bool((d_first == 0).all())
    assert bool(torch.equal(d_dec, d_in)) and bool(torch.equal(d_cons, d_len)) and bool((d_code == -1).all())
    return d_out.cpu().numpy(), d_len.cpu().numpy(), cap


@pytest.mark.parametrize("level,nb", [(1, 4096), (3, 1024), (4, 512)])
def test_other_levels_at_baseline_block_size(zpq, gpu_ctx, level, nb):
    """64 KiB blocks of all four classes at the levels the headline test does not cover: round-trip
    properties on every block, byte parity with the oracle on a sample.  Level 1 runs at C2's stated
    size (BASELINE.json configs[1]: 4096 x 64 KiB, levels.v:53-92), all blocks resident at once -- the
    launch shape that config ships (16 blocks per workgroup, two waves)."""
    arr = W.make_blocks_fast(nb, 65536)
    model = zpq.Model(level=level)
    out, lens, cap = _device_round_trip(zpq, gpu_ctx, model, arr, 1.125)
    assert gpu_ctx.last_kernel_name == "k_chain<decode>"
    if level == 1:
        assert gpu_ctx.last_slots == nb                  # C2: every block has its own 36 MiB slot
    sample = [0, 1, 2, 3, nb // 2 + 1, nb - 2] + ([5, 1026, 2051, 3000, 4093, 4095] if level == 1 else [])
    want = O.encode_blocks(model.header, [arr[i].tobytes() for i in sample], nthreads=6, slack=cap)
    for i, w in zip(sample, want):
        assert out[i * cap:i * cap + int(lens[i])].tobytes() == w, i


@pytest.mark.parametrize("level", [4, 5])
def test_mix2_levels_ragged_batch_with_slot_reuse(zpq, gpu_ctx, level):
    """Levels 4-5 keep the MIX2 weights out of the bit loop (encode: the byte's eight weights in registers,
    decode: the nibble's fifteen candidates in LDS).  Ragged batch of all data classes, fewer slots than blocks
    (that per-block state must start clean every time), empty and one-byte blocks."""
    model = zpq.Model(level=level)
    rnd = random.Random(90 + level)
    blocks = [bytes(W.make_block(3 * b + level, rnd.choice([0, 1, 2, 15, 16, 17, 300, 1500, 4000]))) for b in range(14)]
    want = O.encode_blocks(model.header, blocks, nthreads=4)
    zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 300 << 20)      # 3-5 slots of the compact layout (about 58 / 77 MiB)
    try:
        coded, status, _ = gpu_ctx.encode_blocks(model, blocks)
        assert gpu_ctx.last_kernel_name == "k_chain<encode>" and 2 <= gpu_ctx.last_slots <= 5
        assert (status == 0).all() and coded == want
        dec, status, consumed, _, first = gpu_ctx.decode_blocks(model, coded, cap=8192)
        assert gpu_ctx.last_kernel_name == "k_chain<decode>"
        assert (status == 0).all() and dec == blocks and (first == 0).all()
    finally:
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)


@pytest.mark.parametrize("level", [4, 5])
def test_mix2_weight_aliasing_stress(zpq, gpu_ctx, monkeypatch, level):
    """The MIX2 weight index is a hash: consecutive nibbles may reach the same weight under different slots (and so,
    in the decoder, under different lanes).  Shrinking the level's MIX2 table to 256 entries makes that happen all
    the time; the decoder must then forward the trained value instead of re-reading memory."""
    h = bytearray(O.level_header(level))
    sz = [0, 2, 3, 2, 3, 4, 6, 6, 3, 5]
    p = 5
    for _ in range(h[4]):
        if h[p] == 6:
            h[p + 1] = 8                      # sizebits: 256 weights (still >= 256, mask stays 255)
        p += sz[h[p]]
    header = bytes(h)
    model = zpq.Model(header=header)
    assert model.has_fast_path
    blocks = [bytes(W.make_block(b, 3000 + 500 * (b % 3))) for b in range(12)]
    want = O.encode_blocks(header, blocks, nthreads=4)
    coded, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert gpu_ctx.last_kernel_name == "k_pipe<encode>" and (status == 0).all() and coded == want
    monkeypatch.setenv("ZPQ_ENC_PIPE", "0")                 # the lane-per-component encoder (weights in registers per byte)
    coded2, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert gpu_ctx.last_kernel_name == "k_chain<encode>" and (status == 0).all() and coded2 == want
    dec, status, consumed, _, first = gpu_ctx.decode_blocks(model, coded, cap=8192)
    assert (status == 0).all() and dec == blocks


@pytest.mark.parametrize("level", [1, 3])
def test_line_store_on_levels_1_and_3(zpq, gpu_ctx, level, monkeypatch):
    """Levels 1 and 3 switch from dense tables to the compact line store only when that saves a round of resident
    blocks (thousands of blocks); force the store here so that their store-carrying kernels (k_chain<.., SP = true>)
    are compared with the oracle and with the dense kernels at test size: ragged blocks, displaced lines (a store
    that a block fills to 45 %), slot reuse."""
    model = zpq.Model(level=level)
    rnd = random.Random(300 + level)
    blocks = mixed_blocks(rnd, 48, [0, 1, 2, 17, 255, 256, 257, 1500, 3000, 7000])
    want = O.encode_blocks(model.header, blocks, nthreads=4)
    dense, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert (status == 0).all() and dense == want
    monkeypatch.setenv("ZPQ_SPARSE_MODE", "always")
    zpq.lib().zpq_ctx_set_max_block_bytes(gpu_ctx.h, 7000)          # store sized for these blocks: it really fills up
    zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 40 << 20)         # a handful of slots for 48 blocks
    try:
        sparse, status, _ = gpu_ctx.encode_blocks(model, blocks)
        assert (status == 0).all() and sparse == want and gpu_ctx.last_slots < len(blocks)
        dec, status, consumed, _, first = gpu_ctx.decode_blocks(model, sparse, cap=8192)
        assert (status == 0).all() and dec == blocks and (first == 0).all()
        # a block larger than promised is refused, not miscoded
        big = [bytes(rnd.getrandbits(8) for _ in range(20000))]
        _, status, _ = gpu_ctx.encode_blocks(model, big)
        assert status[0] == -4
    finally:
        zpq.lib().zpq_ctx_set_max_block_bytes(gpu_ctx.h, 65536)
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)

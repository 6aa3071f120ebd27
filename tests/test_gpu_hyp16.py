"""GPU parity of the sixteen-lane two-hypothesis decoders (zpq_chain.hip, HYP16: a block = one DPP row, lanes 2c and
2c + 1 = the two copies of component c, each assuming one outcome of the bit being decoded; round 4) against the CPU
oracle's coded streams and against the eight-lane decoder of the same model (ZPQ_DEC_HYP16=0), through the C ABI.
What is new in these instantiations and therefore aimed at here: the copies over a compact LINE STORE (only the copy that
guessed a nibble's last bit right may probe, claim and reload; the store's bookkeeping must stay equal on both), the
broadcast of the decoded bit from lanes 8 / 9 (level 3) of the row, four blocks per wave, a MIX2 whose
weights the right copy trains (level 4)."""
import os
import random
import sys

import pytest

import oracle_lib as O
import workload as W

sys.path.insert(0, os.path.dirname(__file__))
from test_gpu_chain import mixed_blocks  # noqa: E402
from test_gpu_pipe import small_table_header  # noqa: E402

pytestmark = pytest.mark.gpu
LEVELS = [3, 4]


def decode_both(zpq, gpu_ctx, monkeypatch, model, coded, cap, flags=None):
    """Decode with the sixteen-lane and with the eight-lane decoder; everything the ABI returns must agree (a refused
    block's bytes are unspecified)."""
    kw = {} if flags is None else {"flags": flags}
    monkeypatch.delenv("ZPQ_DEC_HYP16", raising=False)
    a = gpu_ctx.decode_blocks(model, coded, cap=cap, **kw)
    assert gpu_ctx.last_kernel_name == "k_chain<decode>"
    monkeypatch.setenv("ZPQ_DEC_HYP16", "0")
    b = gpu_ctx.decode_blocks(model, coded, cap=cap, **kw)
    monkeypatch.delenv("ZPQ_DEC_HYP16", raising=False)
    assert list(a[1]) == list(b[1]), (list(a[1]), list(b[1]))
    for i in range(len(coded)):
        if a[1][i] == 0:
            assert a[0][i] == b[0][i], i
            assert int(a[2][i]) == int(b[2][i]) and int(a[3][i]) == int(b[3][i]) and int(a[4][i]) == int(b[4][i]), i
    return a


@pytest.mark.parametrize("level", LEVELS)
@pytest.mark.parametrize("store", ["dense", "store"])
def test_inverts_the_oracles_streams(zpq, gpu_ctx, monkeypatch, level, store):
    """Ragged batch (empty, one byte, sizes around a dword and a nibble row), with and without the PP byte, more blocks
    than one workgroup holds; the coded streams come from the CPU oracle."""
    rnd = random.Random(7400 + level)
    model = zpq.Model(level=level)
    if store == "store":
        monkeypatch.setenv("ZPQ_SPARSE_FORCE_LOG2", "13")          # an 8192-line store: probing, displaced lines
    else:
        monkeypatch.setenv("ZPQ_SPARSE_MODE", "never")
    blocks = mixed_blocks(rnd, 75, [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 1000, 3000])
    for flags, pp in ((zpq.FLAG_PP, True), (0, False)):
        coded = O.encode_blocks(model.header, blocks, pp=pp, nthreads=4)
        dec, status, consumed, code, first = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 4096, flags=flags)
        assert bool(gpu_ctx.last_line_store) == (store == "store")
        assert (status == 0).all() and dec == blocks
        assert [int(c) for c in consumed] == [len(c) for c in coded]
        assert (code == 0xFFFFFFFF).all()
        if pp:
            assert (first == 0).all()


@pytest.mark.parametrize("level,bits", [(3, 0), (3, 1), (3, 2), (3, 4), (4, 0), (4, 1), (4, 3)])
def test_rows_under_heavy_aliasing(zpq, gpu_ctx, monkeypatch, level, bits):
    """Every hash table shrunk to 64 << bits bytes: the row being finished is, all the time, one of the three candidates
    of the request in flight, for one copy's outcome of the bit or for both."""
    header = small_table_header(level, bits)
    model = zpq.Model(header=header)
    assert model.has_fast_path
    rnd = random.Random(23 * level + bits)
    blocks = [bytes(3000), b"a" * 2500, b"ab" * 1500, b"abc" * 1000, b"abcd" * 700, bytes(range(256)) * 8,
              bytes(rnd.getrandbits(8) for _ in range(3000)), bytes(rnd.choice(b"01") for _ in range(3000)),
              b"\x00\x10" * 1200, b"\x0f\xf0\x00" * 900, b"", b"x"]
    coded = O.encode_blocks(header, blocks, nthreads=4)
    dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 4096)
    assert (status == 0).all() and dec == blocks


@pytest.mark.parametrize("level", LEVELS)
@pytest.mark.parametrize("store", ["dense", "store"])
def test_rounds_and_partial_workgroups(zpq, gpu_ctx, monkeypatch, level, store):
    """Fewer slots than blocks: every row of lanes decodes several blocks one after the other (tables, store, coder state
    and the copies' bookkeeping must start clean), the last round and the last workgroup are partly idle; blocks of one
    wave end at very different times."""
    model = zpq.Model(level=level)
    rnd = random.Random(299 + level)
    blocks = mixed_blocks(rnd, 83, [0, 1, 300, 1200, 2048, 6000])
    coded = O.encode_blocks(model.header, blocks, nthreads=4)
    if store == "store":
        monkeypatch.setenv("ZPQ_SPARSE_MODE", "always")
        zpq.lib().zpq_ctx_set_max_block_bytes(gpu_ctx.h, 6000)      # the store really fills up (a block of 6000 bytes: ~45 %)
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 60 << 20)
    else:
        monkeypatch.setenv("ZPQ_SPARSE_MODE", "never")
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 19 * model.state_bytes + 1000)
    try:
        dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 6100)
        assert gpu_ctx.last_slots < len(blocks)
        assert (status == 0).all() and dec == blocks
    finally:
        zpq.lib().zpq_ctx_set_max_block_bytes(gpu_ctx.h, 65536)
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)


@pytest.mark.parametrize("level", LEVELS)
def test_store_too_small_is_a_status_for_both_copies(zpq, gpu_ctx, monkeypatch, level):
    """A block that needs more lines than the store holds is refused (ZPQ_E_TOOBIG) by the decoder as well -- the claim
    count lives on both copies of a component and only one of them claims -- and the blocks beside it decode."""
    model = zpq.Model(level=level)
    rnd = random.Random(77 + level)
    blocks = [bytes(rnd.getrandbits(8) for _ in range(3000))] + [bytes(100 + i) for i in range(12)]
    coded = O.encode_blocks(model.header, blocks, nthreads=4)
    monkeypatch.setenv("ZPQ_SPARSE_FORCE_LOG2", "10")               # 1024 lines: 3000 random bytes need ~6000 per table
    dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 4096)
    assert status[0] == -4 and (status[1:] == 0).all() and dec[1:] == blocks[1:]
    monkeypatch.setenv("ZPQ_SPARSE_FORCE_LOG2", "14")               # room for it: the very same streams decode
    dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 4096)
    assert (status == 0).all() and dec == blocks


@pytest.mark.parametrize("level", LEVELS)
def test_output_overflow_and_garbage_input(zpq, gpu_ctx, monkeypatch, level):
    """A slab that is too small is a per-block status; random bytes in place of a coded stream end with a status or at
    the slab's end -- both decoders alike, nothing hangs."""
    model = zpq.Model(level=level)
    rnd = random.Random(5 + level)
    data = [bytes(rnd.getrandbits(8) for _ in range(2000))] + [bytes(50 + i) for i in range(12)]
    coded = O.encode_blocks(model.header, data)
    dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 100)
    assert status[0] == -7 and (status[1:] == 0).all() and dec[1:] == data[1:]
    junk = [bytes(rnd.getrandbits(8) for _ in range(n)) for n in (0, 1, 3, 4, 5, 100, 1000, 3000, 7, 64, 65, 2000, 12, 13)]
    decode_both(zpq, gpu_ctx, monkeypatch, model, junk, 1500)


@pytest.mark.parametrize("level", LEVELS)
def test_64k_blocks_all_classes_both_table_forms(zpq, gpu_ctx, monkeypatch, level):
    """Sixty-four 64 KiB blocks of the bench generator's four classes, dense tables and the line store sized as bench.py's
    ctx sizes it (1.12 x the touched-line bound), against the oracle's streams."""
    model = zpq.Model(level=level)
    nb = 64 if level == 3 else 32                                   # (level 4 dense: 385 MiB of tables per block)
    arr = W.make_blocks(nb, 65536)
    blocks = [arr[i].tobytes() for i in range(nb)]
    coded = O.encode_blocks(model.header, blocks, nthreads=min(16, os.cpu_count() or 1), slack=80000)
    for mode in ("never", "always"):
        monkeypatch.setenv("ZPQ_SPARSE_MODE", mode)
        dec, status, consumed, code, first = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 65536)
        assert bool(gpu_ctx.last_line_store) == (mode == "always")
        assert (status == 0).all() and dec == blocks and (first == 0).all()
        assert [int(c) for c in consumed] == [len(c) for c in coded]


def test_mix2_weights_alias_between_nibbles(zpq, gpu_ctx, monkeypatch):
    """Level 4 with its MIX2 weight table shrunk to 256 entries: a candidate weight of the next nibble is all the time an
    entry the nibble that just ended trained under another lane (forwarded from LDS), for both copies' training."""
    h = bytearray(O.level_header(4))
    sz = [0, 2, 3, 2, 3, 4, 6, 6, 3, 5]
    p = 5
    for _ in range(h[4]):
        if h[p] == 6:
            h[p + 1] = 8
        p += sz[h[p]]
    header = bytes(h)
    model = zpq.Model(header=header)
    assert model.has_fast_path
    rnd = random.Random(4004)
    blocks = mixed_blocks(rnd, 30, [0, 1, 17, 300, 1000, 3000, 5000])
    coded = O.encode_blocks(header, blocks, nthreads=4)
    for mode in ("never", "always"):
        monkeypatch.setenv("ZPQ_SPARSE_MODE", mode)
        dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 5100)
        assert (status == 0).all() and dec == blocks

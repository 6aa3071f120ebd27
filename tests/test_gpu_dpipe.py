"""GPU parity of the wave-split decoder (zpq_dpipe.hip: one wave per component, a block = a pair of lanes holding the
two outcomes of the bit being decoded, one wave for the arithmetic decoder) against the CPU oracle's coded streams and
against the lane-per-component decoder (zpq_chain.hip, ZPQ_DEC_PIPE=0), through the C ABI."""
import os
import random
import sys

import numpy as np
import pytest

import oracle_lib as O
import workload as W

sys.path.insert(0, os.path.dirname(__file__))
from test_gpu_chain import mixed_blocks  # noqa: E402
from test_gpu_pipe import small_table_header  # noqa: E402

pytestmark = pytest.mark.gpu


def decode_both(zpq, gpu_ctx, monkeypatch, model, coded, cap, flags=None):
    """Decode with both decoders; everything the ABI returns must agree (a refused block's bytes are unspecified)."""
    kw = {} if flags is None else {"flags": flags}
    assert len(coded) >= 12                      # (smaller batches stay with the lane-per-component decoder)
    monkeypatch.setenv("ZPQ_DEC_PIPE", "1")      # (opt-in: the lane-per-component decoder measured faster, EXPERIMENTS.md 4.5)
    a = gpu_ctx.decode_blocks(model, coded, cap=cap, **kw)
    assert gpu_ctx.last_kernel_name == "k_dpipe<decode>"
    monkeypatch.delenv("ZPQ_DEC_PIPE", raising=False)
    b = gpu_ctx.decode_blocks(model, coded, cap=cap, **kw)
    assert gpu_ctx.last_kernel_name == "k_chain<decode>"
    assert list(a[1]) == list(b[1])
    for i in range(len(coded)):
        if a[1][i] == 0:
            assert a[0][i] == b[0][i], i
            assert int(a[2][i]) == int(b[2][i]) and int(a[3][i]) == int(b[3][i]) and int(a[4][i]) == int(b[4][i]), i
    return a


@pytest.mark.parametrize("level", [1, 2, 3])
def test_both_decoders_invert_the_oracles_streams(zpq, gpu_ctx, monkeypatch, level):
    """Ragged batch (empty, one byte, sizes around a dword and a nibble row), with and without the PP byte, more blocks
    than one workgroup holds; the coded streams come from the CPU oracle, not from the GPU encoder."""
    rnd = random.Random(7000 + level)
    model = zpq.Model(level=level)
    blocks = mixed_blocks(rnd, 75, [0, 1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 1000, 3000])
    for flags, pp in ((zpq.FLAG_PP, True), (0, False)):
        coded = O.encode_blocks(model.header, blocks, pp=pp, nthreads=4)
        dec, status, consumed, code, first = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 4096, flags=flags)
        assert (status == 0).all() and dec == blocks
        assert [int(c) for c in consumed] == [len(c) for c in coded]
        assert (code == 0xFFFFFFFF).all()
        if pp:
            assert (first == 0).all()


@pytest.mark.parametrize("level,bits", [(1, 0), (1, 2), (2, 0), (2, 1), (2, 3), (3, 0), (3, 2)])
def test_rows_under_heavy_aliasing(zpq, gpu_ctx, monkeypatch, level, bits):
    """Every hash table shrunk to 64 << bits bytes: the row being finished is, all the time, one of the three candidates
    of the request in flight, for one outcome of the bit or for both."""
    header = small_table_header(level, bits)
    model = zpq.Model(header=header)
    assert model.has_fast_path
    rnd = random.Random(19 * level + bits)
    blocks = [bytes(3000), b"a" * 2500, b"ab" * 1500, b"abc" * 1000, b"abcd" * 700, bytes(range(256)) * 8,
              bytes(rnd.getrandbits(8) for _ in range(3000)), bytes(rnd.choice(b"01") for _ in range(3000)),
              b"\x00\x10" * 1200, b"\x0f\xf0\x00" * 900, b"", b"x"]
    coded = O.encode_blocks(header, blocks, nthreads=4)
    dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 4096)
    assert (status == 0).all() and dec == blocks


@pytest.mark.parametrize("level", [1, 2, 3])
def test_rounds_and_partial_workgroups(zpq, gpu_ctx, monkeypatch, level):
    """Fewer slots than blocks: every lane pair decodes several blocks one after the other (tables, published entries and
    decoder state must start clean), the last round and the last workgroup are partly idle; blocks of one workgroup end
    at very different times."""
    model = zpq.Model(level=level)
    rnd = random.Random(199 + level)
    nslots = 19
    blocks = mixed_blocks(rnd, 83, [0, 1, 300, 1200, 2048])
    coded = O.encode_blocks(model.header, blocks, nthreads=4)
    monkeypatch.setenv("ZPQ_SPARSE_MODE", "never")
    monkeypatch.setenv("ZPQ_DEC_PIPE", "1")
    zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, nslots * model.state_bytes + 1000)
    try:
        dec, status, *_ = gpu_ctx.decode_blocks(model, coded, cap=2100)
        assert gpu_ctx.last_kernel_name == "k_dpipe<decode>" and gpu_ctx.last_slots == nslots
        assert (status == 0).all() and dec == blocks
    finally:
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)


def test_output_overflow_and_garbage_input(zpq, gpu_ctx, monkeypatch):
    """A slab that is too small is a per-block status (and no byte lands beyond it); random bytes in place of a coded
    stream end with a status or at the slab's end -- both decoders alike, nothing hangs."""
    model = zpq.Model(level=2)
    rnd = random.Random(5)
    data = [bytes(rnd.getrandbits(8) for _ in range(2000))] + [bytes(50 + i) for i in range(12)]
    coded = O.encode_blocks(model.header, data)
    dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, 100)
    assert status[0] == -7 and (status[1:] == 0).all() and dec[1:] == data[1:]
    junk = [bytes(rnd.getrandbits(8) for _ in range(n)) for n in (0, 1, 3, 4, 5, 100, 1000, 3000, 7, 64, 65, 2000, 12, 13)]
    decode_both(zpq, gpu_ctx, monkeypatch, model, junk, 1500)


def test_small_batches_stay_with_the_lane_per_component_decoder(zpq, gpu_ctx, monkeypatch):
    monkeypatch.setenv("ZPQ_DEC_PIPE", "1")
    model = zpq.Model(level=2)
    for n, name in ((1, "k_chain<decode>"), (11, "k_chain<decode>"), (12, "k_dpipe<decode>"), (37, "k_dpipe<decode>")):
        blocks = [bytes(W.make_block(b, 700 + 13 * b)) for b in range(n)]
        coded = O.encode_blocks(model.header, blocks)
        dec, status, *_ = gpu_ctx.decode_blocks(model, coded, cap=2000)
        assert gpu_ctx.last_kernel_name == name, (n, gpu_ctx.last_kernel_name)
        assert (status == 0).all() and dec == blocks


@pytest.mark.parametrize("level", [1, 2])
def test_blocks_larger_than_64k(zpq, gpu_ctx, monkeypatch, level):
    """C1's 1 MiB of zeros (level 2's M array wraps 16 times) and 256 KiB of mixed data among small blocks."""
    model = zpq.Model(level=level)
    rnd = random.Random(31 + level)
    mixed = bytes(rnd.getrandbits(8) for _ in range(65536)) + bytes(65536) + bytes(rnd.choice(b"abcdefgh\n") for _ in range(131072))
    blocks = [bytes(1 << 20), mixed] + [bytes(100 + i) for i in range(11)]
    coded = O.encode_blocks(model.header, blocks, nthreads=4)
    dec, status, *_ = decode_both(zpq, gpu_ctx, monkeypatch, model, coded, (1 << 20) + 16)
    assert (status == 0).all() and dec == blocks

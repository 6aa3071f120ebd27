"""The host-pointer batch entry points as a pipeline (pinned staging, upload / coding / download of
different rounds overlapped; several contexts behind one call).  What is checked is that NOTHING about the result
depends on how a batch was cut into rounds, on the kind of host memory, or on how many contexts shared it: every
variant must equal the one-call pageable result, which the other GPU tests hold to the oracle."""
import ctypes as C
import random

import numpy as np
import pytest

import oracle_lib as O
import workload as W

pytestmark = pytest.mark.gpu


def _blocks(n, seed):
    rnd = random.Random(seed)
    return [bytes(W.make_block(5 * b + seed, rnd.choice([0, 1, 3, 100, 2000, 4096, 9000]))) for b in range(n)]


def _layout(blocks, caps):
    in_off = np.zeros(len(blocks) + 1, dtype=np.uint64)
    in_off[1:] = np.cumsum([len(b) for b in blocks])
    out_off = np.zeros(len(blocks) + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum(caps)
    return in_off, out_off


def _encode_raw(zpq, ctxs, model, blocks, caps, pinned):
    """zpq_encode_blocks / _multi on explicit buffers; returns the coded streams."""
    L = zpq.lib()
    nb = len(blocks)
    in_off, out_off = _layout(blocks, caps)
    src_b = b"".join(blocks) + b"\0"
    if pinned:
        pin_in, pin_out = zpq.PinnedArray(len(src_b)), zpq.PinnedArray(int(out_off[-1]) + 1)
        src, out = pin_in.array, pin_out.array
        src[:] = np.frombuffer(src_b, dtype=np.uint8)
        out[:] = 0xEE
    else:
        src = np.frombuffer(src_b, dtype=np.uint8).copy()
        out = np.full(int(out_off[-1]) + 1, 0xEE, dtype=np.uint8)
    olen = np.zeros(nb, dtype=np.uint32); st = np.full(nb, -99, dtype=np.int32)
    if len(ctxs) == 1:
        rc = L.zpq_encode_blocks(ctxs[0].h, model.h, nb, src.ctypes.data, in_off.ctypes.data, zpq.FLAG_PP, out.ctypes.data,
                                 out_off.ctypes.data, olen.ctypes.data, st.ctypes.data)
    else:
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        rc = L.zpq_encode_blocks_multi(arr, len(ctxs), model.h, nb, src.ctypes.data, in_off.ctypes.data, zpq.FLAG_PP,
                                       out.ctypes.data, out_off.ctypes.data, olen.ctypes.data, st.ctypes.data)
    assert rc == 0 and (st == 0).all(), (rc, st)
    coded = [out[int(out_off[i]):int(out_off[i]) + int(olen[i])].tobytes() for i in range(nb)]
    if pinned:                                               # bytes behind a block's produced length were never touched
        for i in range(nb):
            tail = out[int(out_off[i]) + int(olen[i]):int(out_off[i + 1])]
            assert (tail == 0xEE).all()
    return coded


def test_rounds_pinned_and_multi_context_give_identical_streams(zpq, gpu_ctx, monkeypatch):
    monkeypatch.setenv("ZPQ_PIPE_MIN_BYTES", "0")          # rounds even for this small batch (default: only above 64 MiB)
    model = zpq.Model(level=2)
    blocks = _blocks(61, 7)
    caps = [len(b) * 2 + 4096 for b in blocks]
    base = _encode_raw(zpq, [gpu_ctx], model, blocks, caps, pinned=False)
    assert base == O.encode_blocks(model.header, blocks, nthreads=4)
    assert _encode_raw(zpq, [gpu_ctx], model, blocks, caps, pinned=True) == base
    other = zpq.Context(0)
    try:
        # two contexts on the one GPU, and a state budget that forces rounds (7 slots for 30 / 31 blocks each)
        for c in (gpu_ctx, other):
            zpq.lib().zpq_ctx_set_state_budget(c.h, 7 * model.state_bytes + 1000)
        assert _encode_raw(zpq, [gpu_ctx, other], model, blocks, caps, pinned=True) == base
        assert gpu_ctx.last_slots == 30 % 7                      # its share (30 blocks) went through in rounds of 7; the last one held 2
        assert _encode_raw(zpq, [gpu_ctx, other, gpu_ctx], model, blocks, caps, pinned=False) == base
        assert _encode_raw(zpq, [gpu_ctx], model, blocks, caps, pinned=True) == base        # 9 rounds through one context
        # decode through the same pipeline, rounds and all
        dec, status, consumed, code, first = gpu_ctx.decode_blocks(model, base, cap=9100)
        assert (status == 0).all() and dec == blocks and (first == 0).all()
        assert [int(c) for c in consumed] == [len(c) for c in base]
    finally:
        for c in (gpu_ctx, other):
            zpq.lib().zpq_ctx_set_state_budget(c.h, 150 << 30)
        other.close()


def test_decode_multi_matches_single(zpq, gpu_ctx):
    model = zpq.Model(level=3)
    blocks = _blocks(23, 11)
    coded, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert (status == 0).all()
    L = zpq.lib()
    nb = len(coded)
    in_off, out_off = _layout(coded, [9100] * nb)
    pin_in, pin_out = zpq.PinnedArray(int(in_off[-1]) + 1), zpq.PinnedArray(int(out_off[-1]) + 1)
    pin_in.array[:int(in_off[-1])] = np.frombuffer(b"".join(coded), dtype=np.uint8)
    olen = np.zeros(nb, dtype=np.uint32); cons = np.zeros(nb, dtype=np.uint32); code = np.zeros(nb, dtype=np.uint32)
    first = np.zeros(nb, dtype=np.uint32); st = np.full(nb, -99, dtype=np.int32)
    other = zpq.Context(0)
    try:
        arr = (C.c_void_p * 2)(gpu_ctx.h, other.h)
        rc = L.zpq_decode_blocks_multi(arr, 2, model.h, nb, pin_in.array.ctypes.data, in_off.ctypes.data, zpq.FLAG_PP,
                                       pin_out.array.ctypes.data, out_off.ctypes.data, olen.ctypes.data, cons.ctypes.data,
                                       code.ctypes.data, first.ctypes.data, st.ctypes.data)
        assert rc == 0 and (st == 0).all() and (first == 0).all()
        got = [pin_out.array[int(out_off[i]):int(out_off[i]) + int(olen[i])].tobytes() for i in range(nb)]
        assert got == blocks and [int(c) for c in cons] == [len(c) for c in coded]
    finally:
        other.close()


@pytest.mark.parametrize("level", [2, 1])
def test_striped_upload_equals_plain_upload(zpq, gpu_ctx, monkeypatch, level):
    """(Level 1: the encoder with split stages, k_pipe2, in its striped-upload form.)  A single round of equally long blocks from pinned memory is uploaded in two stripes: the first 8 KiB of every block,
    then -- beside the running encoder -- the rest, with an in-kernel gate on a pinned flag.  Same streams as the plain
    upload and as the oracle; block contents differ in BOTH stripes, so a lane that read past the gate too early, or
    stale bytes, would show."""
    model = zpq.Model(level=level)
    nb, size = 160, 40960
    blocks = [bytes(W.make_block(b, size)) for b in range(nb)]
    caps = [size + size // 8 + 1024] * nb
    striped = _encode_raw(zpq, [gpu_ctx], model, blocks, caps, pinned=True)
    monkeypatch.setenv("ZPQ_NO_STRIPE", "1")
    plain = _encode_raw(zpq, [gpu_ctx], model, blocks, caps, pinned=True)
    assert striped == plain
    sample = [0, 1, 2, 3, 77, 158, 159]
    assert [striped[i] for i in sample] == O.encode_blocks(model.header, [blocks[i] for i in sample], nthreads=4)
    dec, status, *_ = gpu_ctx.decode_blocks(model, striped, cap=size)
    assert (status == 0).all() and dec == blocks
    # and the mirror image: the decoder's output leaves in two stripes, the first one beside the running kernel once
    # every block has reported its first 3/4 stored (release at system scope + a counter in pinned memory)
    monkeypatch.delenv("ZPQ_NO_STRIPE")
    L = zpq.lib()
    in_off, out_off = _layout(striped, [size] * nb)
    pin_in, pin_out = zpq.PinnedArray(int(in_off[-1]) + 16), zpq.PinnedArray(nb * size)
    pin_in.array[:int(in_off[-1])] = np.frombuffer(b"".join(striped), dtype=np.uint8)
    for rep in range(2):
        pin_out.array[:] = 0xEE
        olen = np.zeros(nb, dtype=np.uint32); st = np.full(nb, -99, dtype=np.int32); first = np.zeros(nb, dtype=np.uint32)
        rc = L.zpq_decode_blocks(gpu_ctx.h, model.h, nb, pin_in.array.ctypes.data, in_off.ctypes.data, zpq.FLAG_PP,
                                 pin_out.array.ctypes.data, out_off.ctypes.data, olen.ctypes.data, None, None, first.ctypes.data,
                                 st.ctypes.data)
        assert rc == 0 and (st == 0).all() and (olen == size).all() and (first == 0).all()
        assert pin_out.array.tobytes() == b"".join(blocks)


def test_c5_whole_workload_through_one_context(zpq, gpu_ctx):
    """BASELINE.json's C5 at its stated TOTAL size -- level 2, 65 536 x 64 KiB = 4 GiB -- on the one GPU a test has:
    eight rounds of the resident capacity through the host-pointer calls (pinned buffers, transfers of neighbouring
    rounds overlapped with coding).  Size-independent properties: every block decodes back to its input; a block's coded
    bytes do not depend on where in the batch (which round, which slot, which workgroup) it was coded -- the generator
    repeats each text block every 256 positions; sampled blocks of every class equal the oracle's stream."""
    L = zpq.lib()
    model = zpq.Model(level=2)
    nb, size = 65536, 65536
    cap = size + size // 8 + 1024
    p_src = zpq.PinnedArray(nb * size)
    src2d = p_src.array.reshape(nb, size)
    for b0 in range(0, nb, 4096):
        src2d[b0:b0 + 4096] = W.make_blocks_fast(4096, size, start=b0)
    in_off = np.arange(nb + 1, dtype=np.uint64) * np.uint64(size)
    out_off = np.arange(nb + 1, dtype=np.uint64) * np.uint64(cap)
    p_out = zpq.PinnedArray(nb * cap)
    olen = np.zeros(nb, dtype=np.uint32); st = np.full(nb, -99, dtype=np.int32)
    rc = L.zpq_encode_blocks(gpu_ctx.h, model.h, nb, p_src.array.ctypes.data, in_off.ctypes.data, zpq.FLAG_PP,
                             p_out.array.ctypes.data, out_off.ctypes.data, olen.ctypes.data, st.ctypes.data)
    assert rc == 0 and (st == 0).all()
    assert gpu_ctx.last_slots == 8192                      # eight rounds of one resident batch each
    out = p_out.array

    def coded(i):
        return out[i * cap:i * cap + int(olen[i])].tobytes()

    # position independence: text block b is pool entry (b // 4) % 64, i.e. identical every 256 blocks
    for b in (2, 6, 250):
        ref = coded(b)
        for k in range(1, nb // 256, 7):
            assert coded(b + 256 * k) == ref, (b, k)
    rnd = random.Random(65536)
    sample = sorted({0, 1, 2, 3, 8191, 8192, nb - 4, nb - 3, nb - 2, nb - 1} | {rnd.randrange(nb) for _ in range(22)})
    want = O.encode_blocks(model.header, [src2d[i].tobytes() for i in sample], nthreads=8, slack=cap)
    for i, w in zip(sample, want):
        assert coded(i) == w, i
    # decode from the packed streams, as an archive holds them
    c_off = np.zeros(nb + 1, dtype=np.uint64)
    c_off[1:] = np.cumsum(olen.astype(np.uint64))
    p_cod = zpq.PinnedArray(int(c_off[-1]) + 16)
    for i in range(nb):
        p_cod.array[int(c_off[i]):int(c_off[i + 1])] = out[i * cap:i * cap + int(olen[i])]
    p_out.free()
    p_dec = zpq.PinnedArray(nb * size)
    dlen = np.zeros(nb, dtype=np.uint32); dst = np.full(nb, -99, dtype=np.int32)
    rc = L.zpq_decode_blocks(gpu_ctx.h, model.h, nb, p_cod.array.ctypes.data, c_off.ctypes.data, zpq.FLAG_PP,
                             p_dec.array.ctypes.data, in_off.ctypes.data, dlen.ctypes.data, None, None, None, dst.ctypes.data)
    assert rc == 0 and (dst == 0).all() and (dlen == size).all()
    assert np.array_equal(p_dec.array, p_src.array)
    for pa in (p_src, p_cod, p_dec):
        pa.free()

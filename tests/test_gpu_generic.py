"""GPU parity: the generic (all nine component types + full ZPAQL) HIP kernel
against the CPU oracle and the committed golden vectors, through the C ABI."""
import hashlib
import json
import os
import random
import sys

import numpy as np
import pytest

import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from inputs import C4B, INPUTS  # noqa: E402

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


def hdr_of(name):
    return C4B if name == "c4b" else O.level_header(int(name))


@pytest.mark.parametrize("name", ["1", "2", "3", "c4b"])
def test_generic_matches_golden_streams(zpq, gpu_ctx, name):
    model = zpq.Model(header=hdr_of(name))
    keys = [k for k in sorted(G["streams"]) if k.split("/")[0] == name]
    for mode in ("pp", "raw"):
        ks = [k for k in keys if k.endswith(mode)]
        blocks = [INPUTS[k.split("/")[1]] for k in ks]
        flags = zpq.FLAG_GENERIC | (zpq.FLAG_PP if mode == "pp" else 0)
        coded, status, _ = gpu_ctx.encode_blocks(model, blocks, flags=flags)
        assert (status == 0).all()
        for k, c in zip(ks, coded):
            ref = G["streams"][k]
            assert len(c) == ref["len"], k
            assert hashlib.sha256(c).hexdigest() == ref["sha256"], k
        dec, status, consumed, code, first = gpu_ctx.decode_blocks(model, coded, cap=8192, flags=flags)
        assert (status == 0).all()
        assert dec == blocks
        for k, n in zip(ks, consumed):
            assert int(n) == G["streams"][k]["consumed"]
        if mode == "pp":
            assert (first == 0).all()


@pytest.mark.parametrize("name", ["2", "c4b", "4"])
def test_generic_bit_trace_matches_oracle(zpq, gpu_ctx, name):
    hdr = hdr_of(name)
    model = zpq.Model(header=hdr)
    data = INPUTS["text2k"][:300]
    coded, tr = gpu_ctx.debug_encode_trace(model, data, ntrace=2000)
    want, otr = O.Codec(hdr).encode(data, pp=True, ntrace=2000)
    assert [t[0] for t in otr] == tr.tolist()
    assert coded == want


def test_generic_vm_contexts_level_programs(zpq, gpu_ctx):
    """Device ZPAQL against the oracle VM on the shipped HCOMP programs."""
    L = O.lib()
    data = INPUTS["lcg4k"][:600] + INPUTS["text2k"][:600]
    for name in ["1", "2", "5", "c4b"]:
        hdr = hdr_of(name)
        model = zpq.Model(header=hdr)
        got = gpu_ctx.debug_contexts(model, data)
        cend, hbegin, hend = O.scan_header(hdr)
        z = L.zo_vm_new(hdr, len(hdr), cend, hbegin, hend)
        n = model.ncomp
        for i, b in enumerate(data):
            L.zo_vm_run(z, b)
            want = [L.zo_vm_h(z, k) if k < L.zo_vm_hlen(z) else 0 for k in range(n)]
            assert got[i].tolist() == want, (name, i)
        L.zo_vm_free(z)


def test_generic_vm_every_opcode_group(zpq, gpu_ctx):
    """The golden VM program (every opcode group, jumps, R, LJ-free) on the device."""
    v = G["vm"]
    prog = bytes.fromhex(v["header"])[6:]
    # same program behind a 1-component header so that contexts are reported
    hdr = bytes([3, 4, 0, 0, 1, 1, 128, 0]) + prog
    model = zpq.Model(header=hdr, offsets=(7, 8, len(hdr) - 1))
    L = O.lib()
    z = L.zo_vm_new(hdr, len(hdr), 7, 8, len(hdr) - 1)
    data = bytes([0, 1, 65, 255, 200, 13, 77] * 20)
    got = gpu_ctx.debug_contexts(model, data)
    for i, b in enumerate(data):
        L.zo_vm_run(z, b)
        assert int(got[i][0]) == L.zo_vm_h(z, 0), i
    L.zo_vm_free(z)


def test_generic_random_programs(zpq, gpu_ctx):
    """Fuzz: random straight-line + short-jump HCOMP programs, device VM vs oracle VM."""
    L = O.lib()
    rnd = random.Random(1234)
    valid = [op for op in range(256) if op not in (56, 255) and not ((op & 7) in (5, 6) and op < 56)
             and op not in (58, 61, 62) and not (120 <= op < 128) and not (240 <= op < 255) and op != 57]
    for trial in range(12):
        prog = []
        for _ in range(rnd.randint(5, 40)):
            op = rnd.choice(valid)
            prog.append(op)
            if op & 7 == 7:
                if op in (39, 47, 63):
                    prog.append(rnd.choice([0, 1, 2, 3]))   # forward jumps only: always terminates
                else:
                    prog.append(rnd.choice([v for v in range(1, 255) if v not in (39, 47, 63)]))
        prog += [112, 56, 0]
        hdr = bytes([3, 5, 0, 0, 2, 1, 128, 1, 100, 0]) + bytes(prog)
        offs = (9, 10, len(hdr) - 1)
        model = zpq.Model(header=hdr, offsets=offs)
        data = bytes(rnd.getrandbits(8) for _ in range(200))
        got = gpu_ctx.debug_contexts(model, data)
        z = L.zo_vm_new(hdr, len(hdr), *offs)
        for i, b in enumerate(data):
            L.zo_vm_run(z, b)
            want = [L.zo_vm_h(z, k) for k in range(2)]
            assert got[i].tolist() == want, (trial, i, prog)
        L.zo_vm_free(z)


def test_generic_random_blocks_all_levels(zpq, gpu_ctx):
    rnd = random.Random(99)
    for name in ["1", "2", "3", "c4b"]:
        hdr = hdr_of(name)
        model = zpq.Model(header=hdr)
        blocks = []
        for i in range(24):
            n = rnd.choice([0, 1, 2, 17, 255, 256, 257, 1000, 3000])
            kind = i % 4
            if kind == 0:
                b = bytes(n)
            elif kind == 1:
                b = bytes(rnd.getrandbits(8) for _ in range(n))
            elif kind == 2:
                b = bytes(rnd.choice(b"abcdefgh ") for _ in range(n))
            else:
                per = bytes(rnd.getrandbits(8) for _ in range(rnd.randint(1, 40)))
                b = (per * (n // len(per) + 1))[:n]
            blocks.append(b)
        coded, status, _ = gpu_ctx.encode_blocks(model, blocks, flags=zpq.FLAG_PP | zpq.FLAG_GENERIC)
        assert (status == 0).all()
        want = O.encode_blocks(hdr, blocks, nthreads=4)
        assert coded == want, name
        dec, status, consumed, _, first = gpu_ctx.decode_blocks(model, coded, cap=4096,
                                                                flags=zpq.FLAG_PP | zpq.FLAG_GENERIC)
        assert (status == 0).all() and dec == blocks
        assert [int(c) for c in consumed] == [len(c) for c in coded]


def test_generic_overflow_and_status(zpq, gpu_ctx):
    model = zpq.Model(level=2)
    rnd = random.Random(5)
    data = bytes(rnd.getrandbits(8) for _ in range(2000))
    coded, status, out_len = gpu_ctx.encode_blocks(model, [data], flags=zpq.FLAG_PP | zpq.FLAG_GENERIC, cap=100)
    assert status[0] == -7 and int(out_len[0]) == len(O.Codec(model.header).encode(data))
    good, status, _ = gpu_ctx.encode_blocks(model, [data], flags=zpq.FLAG_PP | zpq.FLAG_GENERIC)
    dec, status, _, _, _ = gpu_ctx.decode_blocks(model, good, cap=100, flags=zpq.FLAG_PP | zpq.FLAG_GENERIC)
    assert status[0] == -7


def test_block_multisegment_state_carryover(zpq, gpu_ctx):
    """compressor.v:238-245: tables persist across segments, coder and c8/hmap4/h reset."""
    model = zpq.Model(level=2)
    blk = zpq.Block(gpu_ctx, model)
    segs = [blk.encode_segment(INPUTS["hello"]), blk.encode_segment(INPUTS["hello"]),
            blk.encode_segment(INPUTS["text2k"])]
    assert [segs[0].hex(), segs[1].hex(), hashlib.sha256(segs[2]).hexdigest()] == G["multiseg_level2"]
    blk.close()
    dblk = zpq.Block(gpu_ctx, model)
    for s, want in zip(segs, [INPUTS["hello"], INPUTS["hello"], INPUTS["text2k"]]):
        out, cons, code, first = dblk.decode_segment(s, cap=4096)
        assert out == want and first == 0 and cons == len(s)
    dblk.close()


def test_no_model_header(zpq, gpu_ctx):
    """zpaq_test.v:405-425: Encoder on a component-less predictor (p = 16384)."""
    model = zpq.Model(header=b"", offsets=(0, 0, 0))
    coded, status, _ = gpu_ctx.encode_blocks(model, [b"\x55", b"Hello"], flags=zpq.FLAG_GENERIC)
    assert (status == 0).all()
    assert coded[0] == O.Codec(b"", (0, 0, 0)).encode(b"\x55", pp=False)
    assert coded[1] == O.Codec(b"", (0, 0, 0)).encode(b"Hello", pp=False)


@pytest.mark.parametrize("name", ["2", "c4b"])
def test_first_segment_takes_the_fast_kernel_and_state_is_materialised_on_demand(zpq, gpu_ctx, name):
    """zpq_block_*: segment 1 of a block runs on the batch kernels; only when a second segment arrives is the
    persistent state rebuilt (replay of segment 1's symbols through the generic kernel).  Streams must equal
    the oracle's multi-segment coding in both directions, including a decoder that sees PP bytes != 0."""
    header = O.level_header(2) if name == "2" else C4B
    model = zpq.Model(header=header)
    fast = "k_chain" if name == "2" else "k_gdec"                 # (general models: a wave per component in both directions)
    fast_enc = fast if name == "2" else "k_gpipe"             # (one block: below k_pipe's minimum; general models: the wave pipeline)
    segs_in = [INPUTS["text2k"][:900], b"", INPUTS["lcg4k"][:700], INPUTS["text2k"][300:1300]]
    blk = zpq.Block(gpu_ctx, model)
    got = []
    for i, s in enumerate(segs_in):
        got.append(blk.encode_segment(s))
        assert gpu_ctx.last_kernel_name.startswith(fast_enc if i == 0 else "k_generic"), (i, gpu_ctx.last_kernel_name)
    blk.close()
    # the same segments through a block that never leaves the generic kernel
    ref = zpq.Block(gpu_ctx, model)
    assert [ref.encode_segment(s, flags=zpq.FLAG_PP | zpq.FLAG_GENERIC) for s in segs_in] == got
    ref.close()
    dblk = zpq.Block(gpu_ctx, model)
    for i, (c, s) in enumerate(zip(got, segs_in)):
        out, cons, code, first = dblk.decode_segment(c, cap=4096)
        assert gpu_ctx.last_kernel_name.startswith(fast if i == 0 else "k_generic")
        assert out == s and first == 0 and cons == len(c)
    dblk.close()
    # first segment whose PP byte is not 0 (a PROG-mode stream): the replay must feed that byte literally
    stream = bytes([1, 2, 0, 57, 56]) + b"payload payload"
    e = zpq.Block(gpu_ctx, model)
    c1, c2 = e.encode_segment(stream, flags=0), e.encode_segment(b"second", flags=zpq.FLAG_PP)
    e.close()
    d = zpq.Block(gpu_ctx, model)
    out, cons, code, first = d.decode_segment(c1, cap=4096)
    assert first == 1 and out == stream[1:]
    out2, *_ = d.decode_segment(c2, cap=4096)
    assert out2 == b"second"
    d.close()

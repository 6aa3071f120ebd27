"""CPU tests: the C oracle against (a) every known-answer value the reference's
own tests hold for this path and (b) golden vectors produced by the independent
Python restatement (tests/golden/make_golden.py).

Reference test being mirrored is cited per test (zpaq/zpaq_test.v:line).
Coded-stream parity is unpinned by the reference itself (it holds no golden
compressed bytes); it is pinned here to the second restatement.
"""
import ctypes as C
import hashlib
import json
import os
import sys

import pytest

import oracle_lib as O

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from inputs import INPUTS  # noqa: E402


def inputs():
    return INPUTS


def sha(b):
    return hashlib.sha256(b).hexdigest()


def test_tables_match_golden_fingerprints():
    L = O.lib()
    sq = (C.c_int32 * 4096)(); st = (C.c_int32 * 32768)(); dt = (C.c_int32 * 1024)()
    d2 = (C.c_int32 * 256)(); ns = (C.c_uint8 * 1024)()
    L.zo_tables(sq, st, dt, d2, ns)
    t = G["tables"]
    assert sha(bytes(sq)) == t["squash_sha256_i32le"]
    assert sha(bytes(st)) == t["stretch_sha256_i32le"]
    assert sha(bytes(dt)) == t["dt_sha256_i32le"]
    assert sha(bytes(d2)) == t["dt2k_sha256_i32le"]
    assert sha(bytes(ns)) == t["ns_sha256"]
    # SURVEY.md 8(a) Q1/Q2 fingerprints (computed independently at survey time)
    assert t["squash_sha256_i32le"] == "08fe9187c9ce866ac664f9c77b42c51c47a47fa0946d198f3d08f21b5b111f95"
    assert t["stretch_sha256_i32le"] == "c6d5b4adc0bcfddd990563b7699047f797473c272b0a0652f9890ecd37f5d893"
    for d, v in t["squash_spot"].items():
        assert L.zo_squash(int(d)) == v
    for p, v in t["stretch_spot"].items():
        assert L.zo_stretch(int(p)) == v


def test_statetable_kats():
    """zpaq_test.v:55-107 (exact values)."""
    L = O.lib()
    assert (L.zo_ns_n0(0), L.zo_ns_n1(0)) == (0, 0)
    assert (L.zo_ns_n0(1), L.zo_ns_n1(1)) == (1, 0)
    assert (L.zo_ns_n0(2), L.zo_ns_n1(2)) == (0, 1)
    assert L.zo_ns_next(0, 0) == 1
    assert L.zo_ns_next(0, 1) == 2
    assert L.zo_cminit(0) == 1 << 22
    assert L.zo_cminit(1) == (1 << 22) // 2
    assert L.zo_cminit(2) == (3 << 22) // 2
    # statetable.v:76-78,91-93 guards
    assert L.zo_ns_next(256, 0) == 0 and L.zo_ns_next(-1, 1) == 0
    assert L.zo_cminit(300) == 1 << 22


def test_oplen_iserr():
    """zpaq_test.v:266-278."""
    L = O.lib()
    assert L.zo_oplen(0) == 1 and L.zo_oplen(7) == 2 and L.zo_oplen(56) == 1 and L.zo_oplen(255) == 3
    assert L.zo_iserr(56) == 1 and L.zo_iserr(0) == 0 and L.zo_iserr(255) == 0
    assert [L.zo_compsize(i) for i in range(10)] == [0, 2, 3, 2, 3, 4, 6, 6, 3, 5]


def test_squash_stretch_ranges():
    """zpaq_test.v:281-292."""
    L = O.lib()
    assert 15000 <= L.zo_squash(0) <= 18000
    assert 50 <= L.zo_stretch(L.zo_squash(100)) <= 150


def test_levels_and_scan():
    """zpaq_test.v:387-402 + compressor.v:96-145 offsets."""
    expect = {0: (5, 6, 6), 1: (10, 11, 25), 2: (13, 14, 28), 3: (19, 20, 40), 4: (28, 29, 55), 5: (34, 35, 67)}
    for lv in range(6):
        h = O.level_header(lv)
        assert h.hex() == G["levels"][str(lv)]["header"]
        assert list(O.scan_header(h)) == G["levels"][str(lv)]["scan"] == list(expect[lv])
        assert len(O.lib().zo_level_name(lv)) > 0
    assert len(O.level_header(0)) == 7 and O.level_header(0)[4] == 0
    assert O.level_header(9) == O.level_header(1)  # levels.v:34
    c4b = bytes.fromhex(G["c4b"]["header"])
    assert list(O.scan_header(c4b)) == G["c4b"]["scan"] == [39, 40, 69]


def test_empty_predictor_cycle():
    """zpaq_test.v:339-361: no components -> predict() stays in 1..32767 (16384)."""
    L = O.lib()
    c = O.Codec(b"", (0, 0, 0))
    for y in [1] * 8 + [0] * 8:
        p = L.zo_pred_predict(c.h)
        assert p == 16384
        L.zo_pred_update(c.h, y)
    assert L.zo_pred_c8(c.h) == 1 and L.zo_pred_hmap4(c.h) == 1  # zpaq_test.v:295-299


def test_encoder_decoder_symmetry_empty_model():
    """zpaq_test.v:405-425: Encoder on an empty predictor produces output."""
    c = O.Codec(b"", (0, 0, 0))
    coded = c.encode(b"\x55", pp=False)
    assert len(coded) > 0
    d = O.Codec(b"", (0, 0, 0))
    assert d.decode(coded)[0] == b"\x55"


def test_codec_roundtrip_level1_hello():
    """zpaq_test.v:430-527: 'Hello World!' through raw Encoder/Decoder at level 1."""
    h = O.level_header(1)
    coded = O.Codec(h).encode(b"Hello World!", pp=False)
    assert len(coded) > 0
    dec, cons = O.Codec(h).decode(coded)
    assert dec == b"Hello World!"
    assert coded.hex() == G["streams"]["1/hello/raw"]["hex"]


@pytest.mark.parametrize("key", sorted(G["streams"].keys()))
def test_coded_streams_match_second_restatement(key):
    model, iname, mode = key.split("/")
    hdr = bytes.fromhex(G["c4b"]["header"]) if model == "c4b" else O.level_header(int(model))
    data = inputs()[iname]
    ref = G["streams"][key]
    coded = O.Codec(hdr).encode(data, pp=(mode == "pp"))
    assert len(coded) == ref["len"]
    assert sha(coded) == ref["sha256"]
    if ref["hex"]:
        assert coded.hex() == ref["hex"]
    dec, cons = O.Codec(hdr).decode(coded)
    assert dec == (b"\0" if mode == "pp" else b"") + data
    assert cons == ref["consumed"]


@pytest.mark.parametrize("name", ["1", "2", "c4b"])
def test_bit_traces(name):
    hdr = bytes.fromhex(G["c4b"]["header"]) if name == "c4b" else O.level_header(int(name))
    _, tr = O.Codec(hdr).encode(b"Hello World!", pp=True, ntrace=64)
    assert [list(t) for t in tr] == G["trace_" + name]


def test_multisegment_state_carryover():
    """compressor.v:238-245: new Encoder + pr.reset() per segment, tables persist."""
    c = O.Codec(O.level_header(2))
    s1 = c.encode(b"Hello World!")
    s2 = c.encode(b"Hello World!")
    s3 = c.encode(inputs()["text2k"])
    assert [s1.hex(), s2.hex(), sha(s3)] == G["multiseg_level2"]
    assert s1 != s2
    d = O.Codec(O.level_header(2))
    assert d.decode(s1)[0] == b"\0Hello World!"
    assert d.decode(s2)[0] == b"\0Hello World!"
    assert d.decode(s3)[0] == b"\0" + inputs()["text2k"]


def test_vm_known_answers():
    L = O.lib()
    v = G["vm"]
    hdr = bytes.fromhex(v["header"])
    z = L.zo_vm_new(hdr, len(hdr), v["cend"], v["hbegin"], v["hend"])
    assert z
    for run in v["runs"]:
        L.zo_vm_run(z, run["in"])
        got = {"a": L.zo_vm_reg(z, 0), "b": L.zo_vm_reg(z, 1), "c": L.zo_vm_reg(z, 2),
               "d": L.zo_vm_reg(z, 3), "f": L.zo_vm_reg(z, 4)}
        for k in got:
            assert got[k] == run[k], (k, run["in"])
        assert [L.zo_vm_h(z, i) for i in range(L.zo_vm_hlen(z))] == run["h"]
        assert [L.zo_vm_m(z, i) for i in range(L.zo_vm_mlen(z))] == run["m"]
        assert L.zo_vm_r(z, 9) == run["r9"]
    L.zo_vm_free(z)


def test_sha1_prefix_kats():
    """zpaq_test.v:5-27: first 4 bytes of SHA-1('') and SHA-1('abc')."""
    L = O.lib()
    out = C.create_string_buffer(20)
    L.zo_sha1(b"", 0, out)
    assert out.raw == hashlib.sha1(b"").digest() and out.raw[:4] == bytes([0xda, 0x39, 0xa3, 0xee])
    L.zo_sha1(b"abc", 3, out)
    assert out.raw == hashlib.sha1(b"abc").digest() and out.raw[:4] == bytes([0xa9, 0x99, 0x3e, 0x36])
    for n in (55, 56, 63, 64, 65, 1000):
        d = bytes(range(256)) * 4
        L.zo_sha1(d[:n], n, out)
        assert out.raw == hashlib.sha1(d[:n]).digest()


def test_c1_level1_one_mib_zeros_archive():
    """BASELINE C1: level 1, one block, one segment, 1 MiB zeros, CPU only:
    compress, decompress, compare, SHA-1 check (cmd/main.v:298-311,349-380)."""
    L = O.lib()
    data = bytes(1 << 20)
    out = C.create_string_buffer(1 << 16)
    n = L.zo_compress_archive(1, b"zeros.bin", b"1048576 bytes", data, len(data), 1, out, len(out))
    assert 0 < n < 4096
    arc = out.raw[:n]
    assert arc[:13] == bytes([0x37, 0x6b, 0x53, 0x74, 0xa0, 0x31, 0x83, 0xd3, 0x8c, 0xb2, 0x28, 0xb0, 0xd3])
    assert arc[13:16] == b"zPQ" and arc[16] == 1 and arc[17] == 1
    assert arc[-1] == 0xFF and arc[-22] == 253 and arc[-21:-1] == hashlib.sha1(data).digest()
    dec = C.create_string_buffer(len(data) + 8)
    pos = C.c_size_t(0); ok = C.c_int(0)
    fn = C.create_string_buffer(64); cm = C.create_string_buffer(64)
    m = L.zo_decompress_archive(arc, n, C.byref(pos), fn, 64, cm, 64, dec, len(dec), C.byref(ok))
    assert m == len(data) and dec.raw[:m] == data and ok.value == 1
    assert fn.value == b"zeros.bin" and cm.value == b"1048576 bytes"
    assert pos.value == n - 1  # only the end-of-block 0xFF is left


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_archive_roundtrip_small_files(level):
    """compress-decompress.yml:41-115 shape: text, repetitive, random, empty."""
    import random
    L = O.lib()
    rnd = random.Random(7)
    files = [b"hello zpaq\n", b"line of repetitive text\n" * 100,
             bytes(rnd.getrandbits(8) for _ in range(5 * 1024)), b""]
    for i, data in enumerate(files):
        out = C.create_string_buffer(len(data) * 2 + 70000)
        n = L.zo_compress_archive(level, b"f%d" % i, b"%d bytes" % len(data), data, len(data), 1, out, len(out))
        assert n > 0
        dec = C.create_string_buffer(len(data) + 8)
        pos = C.c_size_t(0); ok = C.c_int(0)
        m = L.zo_decompress_archive(out.raw[:n], n, C.byref(pos), None, 0, None, 0, dec, len(dec), C.byref(ok))
        assert m == len(data) and dec.raw[:m] == data and ok.value == 1


def test_batch_helpers_match_single_calls():
    h = O.level_header(2)
    blocks = [inputs()["hello"], inputs()["zeros256"], inputs()["lcg4k"], b""]
    coded = O.encode_blocks(h, blocks, nthreads=3)
    for b, c in zip(blocks, coded):
        assert c == O.Codec(h).encode(b)
    dec = O.decode_blocks(h, coded, cap=8192, nthreads=2)
    assert dec == [b"\0" + b for b in blocks]

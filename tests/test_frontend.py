"""Host front end (C++ Compressor/Decompresser over the C ABI).  CPU tests cover store
mode and framing (no GPU needed); GPU tests mirror the reference's own integration
flows: zpaq_test.v:364-384 (basic compression), cmd/main.v:298-311 (add) and
cmd/main.v:349-380 (extract), compress-decompress.yml:41-115 (5 small files), with
archives compared byte for byte against the oracle's framing writer."""
import ctypes as C
import hashlib
import random

import pytest

import oracle_lib as O


def oracle_archive(level, name, comment, data, called=1):
    out = C.create_string_buffer(len(data) * 17 + 70000)
    n = O.lib().zo_compress_archive(level, name.encode(), comment.encode(), data, len(data), called, out, len(out))
    assert n > 0
    return out.raw[:n]


def add_file(comp, level, name, data):
    """cmd/main.v:298-311."""
    comp.start_block(level)
    comp.start_segment(name, "%d bytes" % len(data))
    comp.set_input(data)
    while comp.compress(65536):
        pass
    comp.end_segment()
    comp.end_block()


def extract_all(zpq, ctx, archive, chunk=65536):
    """cmd/main.v:349-380."""
    files = []
    d = zpq.Decompresser(ctx)
    d.set_input(archive)
    done = 0
    while d.find_block():
        while d.find_filename():
            while d.decompress(chunk):
                pass
            d.read_segment_end()
            out = d.output_bytes()
            files.append((d.get_filename(), d.get_comment(), out[done:], d.get_sha1()))
            done = len(out)
    return files


def small_files():
    rnd = random.Random(11)
    return [("a.txt", b"hello zpaq\n"), ("rep.txt", b"line of repetitive text\n" * 100),
            ("rand.bin", bytes(rnd.getrandbits(8) for _ in range(5 * 1024))), ("empty", b""),
            ("nested.txt", b"nested file content\n" * 3)]


def test_store_mode_archive_matches_oracle_cpu(zpq):
    """Level 0 needs no GPU: framing, 64 KiB chunking with the PP byte, SHA-1 trailer."""
    rnd = random.Random(3)
    for data in (b"", b"x", bytes(rnd.getrandbits(8) for _ in range(70000)), bytes(65535), bytes(65536)):
        comp = zpq.Compressor(None)
        add_file(comp, 0, "f.bin", data)
        arc = comp.output_bytes()
        assert arc == oracle_archive(0, "f.bin", "%d bytes" % len(data), data)
        assert comp.get_sha1() == hashlib.sha1(data).digest()
        (name, comment, out, sha), = extract_all(zpq, None, arc)
        assert (name, comment, out) == ("f.bin", "%d bytes" % len(data), data)
        assert sha == hashlib.sha1(data).digest()


def test_state_machine_guards_cpu(zpq):
    """compressor.v:80,213,260,358,403: out-of-order calls are silently ignored."""
    comp = zpq.Compressor(None)
    comp.end_block(); comp.end_segment(); comp.start_segment("x", "")
    assert comp.compress(10) is False and comp.output_bytes() == b""
    comp.start_block(0)
    comp.start_block(0)                       # ignored: already in a block
    n1 = len(comp.output_bytes())
    comp.end_segment()                        # ignored: no segment open
    assert len(comp.output_bytes()) == n1
    comp.start_segment("n", "c")
    assert comp.compress(5) is False          # no input set
    comp.set_input(b"abc")
    assert comp.compress(2) is True and comp.compress(2) is False
    comp.end_segment(); comp.end_block()
    assert comp.output_bytes() == oracle_archive(0, "n", "c", b"abc")
    d = zpq.Decompresser(None)
    assert d.find_block() is False and d.find_filename() is False and d.decompress(1) is False
    d.set_input(b"no locator in here")
    assert d.find_block() is False


def test_modelled_level_without_gpu_reports_nodevice(zpq):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    comp = zpq.Compressor(None)
    comp.start_block(2)
    assert comp.last_error == -1              # ZPQ_E_NODEVICE: no CPU fallback


@pytest.mark.gpu
def test_basic_compression(zpq, gpu_ctx):
    """zpaq_test.v:364-384."""
    comp = zpq.Compressor(gpu_ctx)
    comp.set_input(bytes([0x41] * 4 + [0x42] * 4))
    comp.start_block(1)
    comp.start_segment("test", "")
    while comp.compress(8):
        pass
    comp.end_segment()
    comp.end_block()
    out = comp.output_bytes()
    assert len(out) > 0 and comp.last_error == 0
    assert out == oracle_archive(1, "test", "", bytes([0x41] * 4 + [0x42] * 4))


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_cli_roundtrip_small_files(zpq, gpu_ctx, level):
    """compress-decompress.yml:41-115: archive five small files, list, extract, compare."""
    comp = zpq.Compressor(gpu_ctx)
    want = b""
    for name, data in small_files():
        add_file(comp, level, name, data)
        want += oracle_archive(level, name, "%d bytes" % len(data), data)
    arc = comp.output_bytes()
    assert comp.last_error == 0
    assert arc == want                                   # byte-identical to the reference writer's layout
    got = extract_all(zpq, gpu_ctx, arc)
    assert [(n, c, o) for n, c, o, _ in got] == [(n, "%d bytes" % len(d), d) for n, d in small_files()]
    for (_, _, o, sha) in got:
        assert sha == hashlib.sha1(o).digest()


@pytest.mark.gpu
def test_partial_decompress_calls_and_return_values(zpq, gpu_ctx):
    """decompress(n) returns true after n bytes, false once the EOF marker is met
    (decompressor.v:484-514); n < 0 means all (:480-482)."""
    data = bytes(range(256)) * 10
    comp = zpq.Compressor(gpu_ctx)
    add_file(comp, 2, "d", data)
    d = zpq.Decompresser(gpu_ctx)
    d.set_input(comp.output_bytes())
    assert d.find_block() and d.find_filename()
    assert d.decompress(1000) is True and len(d.output_bytes()) == 1000
    assert d.decompress(1560) is True and len(d.output_bytes()) == 2560
    assert d.decompress(10) is False
    d.read_segment_end()
    assert d.output_bytes() == data and d.find_filename() is False and d.find_block() is False
    d2 = zpq.Decompresser(gpu_ctx)
    d2.set_input(comp.output_bytes())
    assert d2.find_block() and d2.find_filename()
    assert d2.decompress(-1) is False and d2.output_bytes() == data


@pytest.mark.gpu
def test_multi_segment_block_and_skipped_segment(zpq, gpu_ctx):
    """Two segments in one block share the model (compressor.v:238-245); a reader that
    skips straight to read_segment_end of an undecoded segment still finds the next one
    the way skip_segment does (cmd/main.v:407-410 decodes first; here both orders)."""
    a, b = b"first segment " * 50, b"second segment, same block " * 40
    comp = zpq.Compressor(gpu_ctx)
    comp.start_block(2)
    for name, data in (("a", a), ("b", b)):
        comp.start_segment(name, "")
        comp.set_input(data)
        while comp.compress(100):
            pass
        comp.end_segment()
    comp.end_block()
    arc = comp.output_bytes()
    # oracle: same two segments on one carried-over model
    c = O.Codec(O.level_header(2))
    sa, sb = c.encode(a), c.encode(b)
    assert sa in arc and sb in arc and arc.index(sa) < arc.index(sb)
    d = zpq.Decompresser(gpu_ctx)
    d.set_input(arc)
    assert d.find_block()
    names = []
    while d.find_filename():
        names.append(d.get_filename())
        while d.decompress(4096):
            pass
        d.read_segment_end()
    assert names == ["a", "b"] and d.output_bytes() == a + b


@pytest.mark.gpu
def test_empty_input_and_never_compressed_segment(zpq, gpu_ctx):
    """compress() called on empty input codes only the PP byte; a segment that never saw
    compress() has no PP byte at all (compressor.v:271-274)."""
    comp = zpq.Compressor(gpu_ctx)
    add_file(comp, 2, "e", b"")
    assert comp.output_bytes() == oracle_archive(2, "e", "0 bytes", b"", called=1)
    comp2 = zpq.Compressor(gpu_ctx)
    comp2.start_block(2); comp2.start_segment("never", ""); comp2.end_segment(); comp2.end_block()
    assert comp2.output_bytes() == oracle_archive(2, "never", "", b"", called=0)
    for arc in (comp.output_bytes(), comp2.output_bytes()):
        d = zpq.Decompresser(gpu_ctx)
        d.set_input(arc)
        assert d.find_block() and d.find_filename()
        assert d.decompress(-1) is False and d.output_bytes() == b""
        d.read_segment_end()
        assert d.find_filename() is False


@pytest.mark.gpu
def test_start_block_hcomp_quirk(zpq, gpu_ctx):
    """Q14: start_block_hcomp never sets cend, so every component stays type 0 and
    nothing of the header reaches the output (compressor.v:191-209)."""
    hdr = O.level_header(2)
    comp = zpq.Compressor(gpu_ctx)
    comp.start_block_hcomp(hdr)
    comp.start_segment("q", "")
    comp.set_input(b"quirk data " * 20)
    while comp.compress(64):
        pass
    comp.end_segment(); comp.end_block()
    out = comp.output_bytes()
    coded = O.Codec(hdr, (0, 0, 0)).encode(b"quirk data " * 20)
    assert out == b"\x01q\x00\x00\x00" + coded + bytes(4) + b"\xfd" + hashlib.sha1(b"quirk data " * 20).digest() + b"\xff"


@pytest.mark.gpu
def test_decompresser_on_archive_larger_than_64_mib(zpq, gpu_ctx):
    """the sequential Decompresser handed the whole rest of the archive to the coder and sized
    its output buffer at 64x that, so a modelled segment followed by more than ~64 MiB of archive failed with
    ZPQ_E_ARG (the reference streams, decompressor.v:443-515, and has no such limit).  The segment's input is now
    bounded by the next block locator and the output buffer grows on demand."""
    rnd = random.Random(2)
    head = bytes(rnd.choice(b"zpaq on mi355x\n") for _ in range(5000))
    tail = b"the end " * 300
    big = bytes(rnd.getrandbits(8) for _ in range(1 << 16)) * 1100          # 68.75 MiB, stored (level 0)
    comp = zpq.Compressor(gpu_ctx)
    add_file(comp, 2, "head.txt", head)
    add_file(comp, 0, "big.bin", big)
    add_file(comp, 1, "tail.txt", tail)
    arc = comp.output_bytes()
    assert comp.last_error == 0 and len(arc) > (68 << 20)
    assert arc.startswith(oracle_archive(2, "head.txt", "%d bytes" % len(head), head))
    assert arc.endswith(oracle_archive(1, "tail.txt", "%d bytes" % len(tail), tail))
    got = extract_all(zpq, gpu_ctx, arc)
    assert [(n, len(o)) for n, _, o, _ in got] == [("head.txt", len(head)), ("big.bin", len(big)), ("tail.txt", len(tail))]
    assert got[0][2] == head and got[2][2] == tail and hashlib.sha1(got[1][2]).digest() == hashlib.sha1(big).digest()
    for (_, _, o, sha) in got:
        assert sha == hashlib.sha1(o).digest()

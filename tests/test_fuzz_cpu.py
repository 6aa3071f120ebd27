"""Malformed input must come back as a status code (or an empty result), never as a crash: the host-side
parsers of the boundary -- header scan, model walk, block/segment framing reader -- fed with random and
mutated bytes.  CPU only (no ctx): store-mode archives exercise the whole extract path; modelled blocks
stop at 'no device', which is itself the documented behaviour."""
import ctypes as C
import os
import random
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(__file__))
import oracle_lib as O  # noqa: E402


def test_scan_and_model_walk_survive_random_headers(zpq):
    rnd = random.Random(1234)
    seen = {0: 0}
    for trial in range(3000):
        n = rnd.choice([0, 1, 4, 5, 6, 8, 12, 20, 40, 70, 200])
        if trial % 3 == 0:                                     # mutate a real header
            h = bytearray(zpq.level_header(rnd.randint(1, 5)))
            for _ in range(rnd.randint(1, 4)):
                if h:
                    h[rnd.randrange(len(h))] = rnd.getrandbits(8)
            h = bytes(h[:rnd.randint(0, len(h))]) if trial % 2 else bytes(h)
        else:
            h = bytes(rnd.getrandbits(8) for _ in range(n))
        offs = zpq.scan_header(h)
        assert offs == O.scan_header(h)                        # same quirks as the restated scanner
        try:
            m = zpq.Model(header=h, offsets=offs)
            seen[0] += 1
            assert 0 <= m.ncomp <= 255 and m.state_bytes >= 0
        except zpq.ZpqError as e:
            assert e.code < 0
            seen[e.code] = seen.get(e.code, 0) + 1
    assert seen[0] > 50 and len(seen) >= 2                     # both outcomes were exercised


def test_archive_reader_survives_garbage_and_mutations(zpq):
    rnd = random.Random(99)
    files = [("a.txt", "11 bytes", b"hello world"), ("b", "0 bytes", b""), ("c.bin", "300 bytes", bytes(range(256)) + bytes(44))]
    good = zpq.archive_add(None, 0, files)
    assert [(g["name"], g["data"]) for g in zpq.archive_extract(None, good)] == [(f[0], f[2]) for f in files]
    for trial in range(400):
        kind = trial % 4
        if kind == 0:
            arc = bytes(rnd.getrandbits(8) for _ in range(rnd.choice([0, 1, 15, 16, 17, 100, 1000])))
        elif kind == 1:
            arc = good[:rnd.randint(0, len(good))]                                    # truncation anywhere
        elif kind == 2:
            b = bytearray(good)
            for _ in range(rnd.randint(1, 6)):
                b[rnd.randrange(len(b))] = rnd.getrandbits(8)
            arc = bytes(b)
        else:                                                                          # a modelled block, but no device
            buf = C.create_string_buffer(4096)
            k = O.lib().zo_compress_archive(rnd.randint(1, 3), b"m", b"3 bytes", b"abc", 3, 1, buf, len(buf))
            arc = good[:rnd.choice([0, len(good)])] + buf.raw[:k] + good
        out = zpq.archive_extract(None, arc)
        assert isinstance(out, list)
        if kind == 3:                                          # the modelled block is reported, the store blocks still come out
            assert any(g["status"] == -1 for g in out) and [g["name"] for g in out if g["status"] == 0][-3:] == ["a.txt", "b", "c.bin"]
        for g in out:
            assert g["size"] == len(g["data"]) and g["size"] < 10 * len(good) + 1000


def test_cli_rejects_bad_usage_without_crashing(tmp_path):
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zpaq-v_amd", "bin", "zpaqv")
    for args, needle in (([], "Missing command"), (["q", "x"], "Unknown command"), (["a"], "Missing archive name"),
                         (["x", str(tmp_path / "nope")], "not found"), (["a", str(tmp_path / "arc"), "-bogus"], "Unknown option"),
                         (["a", str(tmp_path / "arc"), str(tmp_path / "missing-file")], "No files to add")):
        r = subprocess.run([cli] + args, capture_output=True, text=True)
        assert r.returncode == 1 and needle in r.stderr, (args, r.stderr)
    r = subprocess.run([cli, "help"], capture_output=True, text=True)
    assert r.returncode == 0 and "Usage" in r.stdout
    (tmp_path / "junk.zpaq").write_bytes(os.urandom(5000))
    r = subprocess.run([cli, "l", str(tmp_path / "junk")], capture_output=True, text=True)
    assert r.returncode == 0 and "Total files: 0" in r.stdout


def test_cli_filters_match_the_reference_known_answers():
    """cmd/main_test.v:5-68: the reference's own tables for matches_pattern and should_include (and -mN/-sN/-tN)."""
    cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zpaq-v_amd", "bin", "zpaqv")

    def match(s, pat):
        args = [cli, "__match"] + ([s] if s else []) + [pat]          # an empty argv string would be dropped by the parser
        return subprocess.run(args, capture_output=True, text=True).stdout.strip() == "true"

    def include(name, only, nots):
        args = [cli, "__include", name]
        for o in only:
            args += ["-only", o]
        for n in nots:
            args += ["-not", n]
        return subprocess.run(args, capture_output=True, text=True).stdout.strip() == "true"

    table = [("test.txt", "test.txt", True), ("test.txt", "test.doc", False), ("test.txt", "*.txt", True), ("file.txt", "*.txt", True),
             ("test.doc", "*.txt", False), ("test.txt", "test.*", True), ("test.doc", "test.*", True), ("file.txt", "test.*", False),
             ("abc", "*", True), ("", "*", True), ("test.txt", "tes?.txt", True), ("test.txt", "t?st.txt", True),
             ("test.txt", "????.txt", True), ("test.txt", "???.txt", False), ("a", "?", True), ("ab", "?", False),
             ("test123.txt", "test*.txt", True), ("test.txt", "test*.txt", True), ("testa.txt", "test?.txt", True),
             ("test12.txt", "test?.txt", False)]
    for s, pat, want in table:
        assert match(s, pat) == want, (s, pat)
    inc = [("file.txt", [], [], True), ("anything", [], [], True), ("file.txt", ["*.txt"], [], True), ("file.doc", ["*.txt"], [], False),
           ("file.txt", ["*.txt", "*.doc"], [], True), ("file.doc", ["*.txt", "*.doc"], [], True), ("file.pdf", ["*.txt", "*.doc"], [], False),
           ("file.txt", [], ["*.log"], True), ("file.log", [], ["*.log"], False), ("file.txt", [], ["*.log", "*.tmp"], True),
           ("file.tmp", [], ["*.log", "*.tmp"], False), ("file.txt", ["*.txt"], ["temp*"], True), ("temp.txt", ["*.txt"], ["temp*"], False),
           ("file.doc", ["*.txt"], ["temp*"], False)]
    for name, only, nots, want in inc:
        assert include(name, only, nots) == want, (name, only, nots)


def test_store_mode_archive_inside_a_store_mode_archive(zpq):
    """an archive added with -m0 keeps its locators verbatim inside the outer block's
    payload.  The reference's sequential Decompresser consumes a block before it searches for the next one
    (cmd/main.v:342-380, decompressor.v:518-587), so the inner archive is one ordinary member -- not extra
    top-level files, and not a truncated outer member."""
    inner_files = [("b.bin", "300 bytes", bytes(range(256)) + bytes(44)), ("a.txt", "11 bytes", b"hello world")]
    inner = zpq.archive_add(None, 0, inner_files)
    outer_files = [("in1", "5 bytes", b"first"), ("inner.zpaq", "%d bytes" % len(inner), inner), ("tail", "4 bytes", b"last")]
    outer = zpq.archive_add(None, 0, outer_files)
    got = zpq.archive_extract(None, outer)
    assert [(g["name"], g["data"], g["status"], g["sha1_ok"]) for g in got] == [(f[0], f[2], 0, True) for f in outer_files]
    # and the member really is the inner archive
    assert [(g["name"], g["data"]) for g in zpq.archive_extract(None, got[1]["data"])] == [(f[0], f[2]) for f in inner_files]
    # an inner archive that starts with a bare "zPQ" (no 13-byte locator), stored at the very start of a chunk
    bare = inner[13:]
    assert bare[:3] == b"zPQ" and [g["name"] for g in zpq.archive_extract(None, bare)] == ["b.bin", "a.txt"]
    outer2 = zpq.archive_add(None, 0, [("bare.zpaq", "", bare), ("z", "", b"zPQ")])
    assert [(g["name"], g["data"]) for g in zpq.archive_extract(None, outer2)] == [("bare.zpaq", bare), ("z", b"zPQ")]
    # two archives concatenated, the second without its locator: the reader's rolling hashes restart at every
    # find_block, so a bare "zPQ" right behind a block end is a block (decompressor.v:227-241)
    cat = outer + bare
    assert [g["name"] for g in zpq.archive_extract(None, cat)] == ["in1", "inner.zpaq", "tail", "b.bin", "a.txt"]
    # a store segment cut off inside a chunk, or without its 253/254 trailer, is not reported as OK
    cut = zpq.archive_extract(None, outer[:outer.index(b"hello world") + 3])     # inside inner.zpaq's chunk
    assert [g["name"] for g in cut] == ["in1", "inner.zpaq"] and cut[0]["status"] == 0 and cut[1]["status"] != 0
    notrailer = bytearray(outer)
    k = outer.index(b"first") + 5 + 4                     # in1's trailer byte (behind the zero length)
    assert notrailer[k] == 253
    notrailer[k] = 7
    assert zpq.archive_extract(None, bytes(notrailer))[0]["status"] != 0


def test_model_create_rejects_offsets_outside_the_header(zpq):
    """cend/hbegin/hend index the header on host and device; out-of-range values are
    refused (ZPQ_E_HEADER beyond the header, ZPQ_E_ARG when negative), hend < hbegin is an empty program."""
    h = zpq.level_header(2)
    cend, hbegin, hend = zpq.scan_header(h)
    for bad in ((len(h) + 1, hbegin, hend), (cend, len(h) + 5, hend), (cend, hbegin, len(h) + 1), (-1, hbegin, hend), (cend, -3, hend)):
        with pytest.raises(zpq.ZpqError) as e:
            zpq.Model(header=h, offsets=bad)
        assert e.value.code == (-2 if min(bad) < 0 else -3)
    m = zpq.Model(header=h, offsets=(cend, hbegin, hbegin - 4))     # empty program: accepted
    assert m.ncomp == 3

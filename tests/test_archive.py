"""Batch archive layer (zpaq::archive_add / archive_extract, include/zpaq_frontend.hpp) and the SHA-1
side kernel: the reference CLI's add / extract / list loops (cmd/main.v:239-470) with all files of a
call coded as one GPU batch.  Archives must be byte-identical to the per-file Compressor loop, which
the oracle's framing writer (zo_compress_archive, oracle/zpaq_oracle.c) restates."""
import ctypes as C
import hashlib
import os
import random
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import oracle_lib as O  # noqa: E402
from inputs import INPUTS  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "zpaq-v_amd", "bin", "zpaqv")


def file_set(seed=7, n=40):
    rnd = random.Random(seed)
    files = [("empty", b""), ("one", b"x"), ("hello.txt", b"Hello World!"), ("zeros", bytes(3000)),
             ("text", INPUTS["text2k"]), ("lcg", INPUTS["lcg4k"])]
    for i in range(n):
        k = rnd.choice([1, 17, 63, 64, 65, 500, 4096, 9000])
        kind = i % 3
        if kind == 0:
            d = bytes(rnd.getrandbits(8) for _ in range(k))
        elif kind == 1:
            d = (b"abcabcabd" * (k // 9 + 1))[:k]
        else:
            d = bytes((j * 7 + i) & 63 | 32 for j in range(k))
        files.append(("f%03d.bin" % i, d))
    return [(nm, "%d bytes" % len(d), d) for nm, d in files]


def oracle_one(level, name, comment, data):
    out = C.create_string_buffer(len(data) * 17 + 70000)
    n = O.lib().zo_compress_archive(level, name.encode(), comment.encode(), data, len(data), 1, out, len(out))
    assert n > 0
    return out.raw[:n]


def oracle_archive(level, files):
    return b"".join(oracle_one(level, nm, cm, d) for nm, cm, d in files)


# ---------------------------------------------------------------- CPU: store mode needs no GPU
def test_store_mode_batch_archive_equals_oracle_and_round_trips(zpq):
    files = file_set(n=6)
    arc = zpq.archive_add(None, 0, files)
    assert arc == oracle_archive(0, files)
    got = zpq.archive_extract(None, arc)
    assert [(g["name"], g["comment"], g["data"]) for g in got] == files
    assert all(g["sha1_ok"] and g["status"] == 0 for g in got)
    listed = zpq.archive_extract(None, arc, want_data=False)
    assert [(g["name"], g["size"]) for g in listed] == [(nm, len(d)) for nm, _, d in files]


def test_modelled_level_without_a_device_fails_loudly(zpq):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(zpq.ZpqError):
        zpq.archive_add(None, 2, file_set(n=1))


def test_cli_store_mode_round_trip(tmp_path):
    src = tmp_path / "in"
    (src / "sub").mkdir(parents=True)
    (src / "a.txt").write_bytes(b"hello world\n")
    (src / "sub" / "r.bin").write_bytes(INPUTS["lcg4k"])
    (src / "empty").write_bytes(b"")
    arc = str(tmp_path / "arc")
    r = subprocess.run([CLI, "a", arc, str(src), "-m0", "-s1"], capture_output=True, text=True)
    assert r.returncode == 0 and "Files added: 3" in r.stdout, r.stderr
    r = subprocess.run([CLI, "l", arc], capture_output=True, text=True)
    assert "r.bin (4096 bytes)" in r.stdout and "Total files: 3" in r.stdout
    out = tmp_path / "out"
    r = subprocess.run([CLI, "x", arc, "-to", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "Files extracted: 3" in r.stdout, r.stderr
    assert (out / "r.bin").read_bytes() == INPUTS["lcg4k"] and (out / "a.txt").read_bytes() == b"hello world\n"
    r = subprocess.run([CLI, "x", arc, "-to", str(out)], capture_output=True, text=True)      # no -force: skipped
    assert "Files extracted: 0" in r.stdout and "exists, skipping" in r.stderr
    r = subprocess.run([CLI, "x", arc, "-test", "-only", "*.txt"], capture_output=True, text=True)
    assert "Verified: a.txt" in r.stdout and "Files verified: 1" in r.stdout


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_sha1_side_kernel_matches_hashlib(gpu_ctx):
    rnd = random.Random(3)
    lens = [0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 129, 1000, 4095, 65536, 70001] + \
           [rnd.randint(0, 300) for _ in range(150)]
    blocks = [bytes(rnd.getrandbits(8) for _ in range(k)) for k in lens]      # contiguous: every alignment occurs
    got = gpu_ctx.sha1_blocks(blocks)
    assert got == [hashlib.sha1(b).digest() for b in blocks]


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 2, 3])
def test_batch_archive_equals_per_file_oracle_archive(zpq, gpu_ctx, level):
    files = file_set(seed=level)
    arc = zpq.archive_add(gpu_ctx, level, files)
    assert arc == oracle_archive(level, files)
    got = zpq.archive_extract(gpu_ctx, arc)
    assert [(g["name"], g["comment"], g["data"]) for g in got] == files
    assert all(g["sha1_ok"] and g["status"] == 0 for g in got)
    listed = zpq.archive_extract(gpu_ctx, arc, want_data=False)
    assert [(g["name"], g["size"]) for g in listed] == [(nm, len(d)) for nm, _, d in files]


@pytest.mark.gpu
def test_extract_mixed_archive_levels_store_and_multi_segment_blocks(zpq, gpu_ctx):
    files = file_set(seed=11, n=9)
    parts = [zpq.archive_add(gpu_ctx, 2, files[:5]), zpq.archive_add(None, 0, files[5:8]), zpq.archive_add(gpu_ctx, 1, files[8:12])]
    # one block holding two segments, written by the sequential front end
    c = zpq.Compressor(gpu_ctx)
    c.start_block(2)
    for nm, cm, d in files[12:14]:
        c.start_segment(nm, cm)
        c.set_input(d)
        while c.compress(65536):
            pass
        c.end_segment()
    c.end_block()
    parts.append(c.output_bytes())
    parts.append(zpq.archive_add(gpu_ctx, 2, files[14:]))
    arc = b"garbage before the first block" + b"".join(parts)
    got = zpq.archive_extract(gpu_ctx, arc)
    assert [(g["name"], g["comment"], g["data"]) for g in got] == files
    assert all(g["sha1_ok"] and g["status"] == 0 for g in got)


@pytest.mark.gpu
def test_block_found_without_the_locator_at_stream_start(zpq, gpu_ctx):
    """find_block's initial hash state stands for the 13 locator bytes (decompressor.v:227-236): a stream that
    starts with "zPQ" is a block; the same 3 bytes later in the stream are not."""
    files = file_set(seed=2, n=2)
    arc = zpq.archive_add(gpu_ctx, 2, files)
    assert arc[13:16] == b"zPQ"
    got = zpq.archive_extract(gpu_ctx, arc[13:])
    assert [(g["name"], g["data"]) for g in got] == [(nm, d) for nm, _, d in files]
    d = zpq.Decompresser(gpu_ctx)
    d.set_input(arc[13:])
    assert d.find_block()
    assert zpq.archive_extract(gpu_ctx, b"x" + arc[13:])[0]["name"] == files[1][0]     # first block is no longer found
    d = zpq.Decompresser(gpu_ctx)
    d.set_input(b"x" + arc[13:])
    assert d.find_block() and d.find_filename() and d.get_filename() == files[1][0]


@pytest.mark.gpu
def test_extract_reports_a_damaged_payload(zpq, gpu_ctx):
    files = [("a", "5000 bytes", INPUTS["lcg4k"] + bytes(904)), ("b", "12 bytes", b"Hello World!")]
    arc = bytearray(zpq.archive_add(gpu_ctx, 2, files))
    arc[200] ^= 0x40                                       # inside a's coded payload
    got = zpq.archive_extract(gpu_ctx, bytes(arc))
    by_name = {g["name"]: g for g in got}
    assert by_name["b"]["data"] == b"Hello World!" and by_name["b"]["sha1_ok"]
    assert "a" not in by_name or not by_name["a"]["sha1_ok"] or by_name["a"]["data"] != files[0][2] or by_name["a"]["status"] != 0


@pytest.mark.gpu
def test_wrong_size_comment_only_costs_a_retry(zpq, gpu_ctx):
    files = [("big", "3 bytes", INPUTS["text2k"] * 8), ("nohint", "", INPUTS["lcg4k"])]
    arc = zpq.archive_add(gpu_ctx, 1, files)
    got = zpq.archive_extract(gpu_ctx, arc)
    assert [(g["name"], g["data"]) for g in got] == [(nm, d) for nm, _, d in files]


@pytest.mark.gpu
def test_cli_level2_directory_round_trip_and_archive_bytes(tmp_path):
    files = file_set(seed=5, n=300)
    src = tmp_path / "in"
    src.mkdir()
    for nm, _, d in files:
        (src / nm).write_bytes(d)
    arc = str(tmp_path / "arc.zpaq")
    r = subprocess.run([CLI, "a", arc, str(src), "-m2"], capture_output=True, text=True)
    assert r.returncode == 0 and "Files added: %d" % len(files) in r.stdout, r.stderr
    by_name = {nm: (nm, cm, d) for nm, cm, d in files}
    listed = subprocess.run([CLI, "l", arc], capture_output=True, text=True).stdout.splitlines()
    order = [ln.split(" (")[0] for ln in listed[2:-2]]
    assert sorted(order) == sorted(by_name)
    assert open(arc, "rb").read() == oracle_archive(2, [by_name[nm] for nm in order])
    out = tmp_path / "out"
    r = subprocess.run([CLI, "x", arc, "-to", str(out)], capture_output=True, text=True)
    assert r.returncode == 0 and "Files extracted: %d" % len(files) in r.stdout, r.stderr
    for nm, _, d in files:
        assert (out / nm).read_bytes() == d


# ---------------------------------------------------------------- PostProcessor PROG mode (decompressor.v:14-167)
def _prog_archive(zpq, ctx, level, name, stream, expect):
    """An archive whose one segment decodes to `stream` (mode byte first): the framing of an ordinary block
    around the coded stream, with the SHA-1 of the post-processed output as libzpaq writers store it."""
    hdr = O.level_header(level)
    shell = zpq.archive_add(ctx, level, [(name, "", b"")])
    coded_empty = O.encode_blocks(hdr, [b""], pp=True)[0]
    prefix = shell[:len(shell) - (len(coded_empty) + 4 + 21 + 1)]
    coded = O.encode_blocks(hdr, [stream], pp=False)[0]
    return prefix + coded + bytes(4) + b"\xfd" + hashlib.sha1(expect).digest() + b"\xff"


def _jt_loop_program():
    """c=a  a=3  d=a  L: a=c out d-- a=d a>0 jt L halt  -- every input byte three times."""
    prog = [80, 71, 3, 88]
    loop = len(prog)
    prog += [66, 57, 26, 67, 239, 0]
    jt = len(prog)
    rel = loop - (jt + 2)                       # applied after the operand fetch, offset ((N+128)&255)-127 (quirk Q11)
    prog += [39, (rel - 1) & 255, 56]
    return prog


PROGS = {
    "identity": [57, 56],
    "plus_one": [1, 57, 56],
    "twice": [57, 57, 56],
    "delay_no_m": [80, 68, 57, 66, 96, 56],                       # psize < 250: the PCOMP VM has no M -> zeros come out
    "delay_with_m": [80, 68, 57, 66, 96, 56] + [0] * 250,         # (6 + psize) >> 8 == 1 -> M of 2 bytes exists
    "loop_r": _jt_loop_program(),
    "hashd_on_empty_h": [60, 64 + 6, 57, 56],                     # H is never allocated in the PostProcessor: *d reads 0
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(PROGS))
def test_postprocessor_prog_mode(zpq, gpu_ctx, name):
    sys.path.insert(0, os.path.join(ROOT, "oracle", "pyref"))
    import zpaq_pyref as P
    prog = PROGS[name]
    payload = INPUTS["text2k"][:700] + bytes(range(256))
    stream = bytes([1, len(prog) & 255, len(prog) >> 8]) + bytes(prog) + payload
    expect = P.postprocess(stream)
    assert len(expect) >= len(payload) and (name != "identity" or expect == payload)
    arc = _prog_archive(zpq, gpu_ctx, 2, "p.bin", stream, expect)
    got = zpq.archive_extract(gpu_ctx, arc)
    assert len(got) == 1 and got[0]["status"] == 0 and got[0]["name"] == "p.bin"
    assert got[0]["data"] == expect and got[0]["sha1_ok"]
    d = zpq.Decompresser(gpu_ctx)                                  # the sequential front end, 100 bytes at a time
    d.set_input(arc)
    assert d.find_block() and d.find_filename()
    while d.decompress(100):
        pass
    assert d.output_bytes() == expect and d.last_error == 0


@pytest.mark.gpu
def test_postprocessor_mode_byte_corner_cases(zpq, gpu_ctx):
    sys.path.insert(0, os.path.join(ROOT, "oracle", "pyref"))
    import zpaq_pyref as P
    cases = {
        "unknown mode is PASS": bytes([7]) + b"abcdef",
        "PROG with size 0 is PASS": bytes([1, 0, 0]) + b"abcdef",
        "stream ends inside the size": bytes([1, 5]),
        "stream ends inside the program": bytes([1, 9, 0, 57, 56]),
        "program only, no data": bytes([1, 2, 0, 57, 56]),
    }
    for what, stream in cases.items():
        expect = P.postprocess(stream)
        arc = _prog_archive(zpq, gpu_ctx, 1, "c", stream, expect)
        got = zpq.archive_extract(gpu_ctx, arc)
        assert [g["data"] for g in got] == [expect] and got[0]["status"] == 0, what
    # a PCOMP program that never halts: the reference would hang; here the file carries a status
    stream = bytes([1, 3, 0, 63, 0xFD, 56]) + b"x"               # jmp to itself: rel = ((0xFD + 128) & 255) - 127 = -2
    arc = _prog_archive(zpq, gpu_ctx, 1, "hang", stream, b"")
    got = zpq.archive_extract(gpu_ctx, arc)
    assert got[0]["status"] == -8


# ---------------------------------------------------------------- -fragment: one big file = many independent blocks
def test_fragmented_store_mode_archive_joins_back(zpq):
    files = [("big", "20000 bytes", bytes(range(256)) * 78 + bytes(32)), ("small", "5 bytes", b"hello"), ("edge", "4096 bytes", bytes(4096))]
    arc = zpq.archive_add(None, 0, files, fragment_bytes=4096)
    segs = zpq.archive_extract(None, arc)
    assert [s["name"] for s in segs] == ["big", "", "", "", "", "small", "edge"]
    assert [s["size"] for s in segs][:5] == [4096, 4096, 4096, 4096, 20000 - 4 * 4096]
    joined = zpq.archive_extract(None, arc, join_unnamed=True)
    assert [(j["name"], j["comment"], j["data"]) for j in joined] == files and all(j["sha1_ok"] for j in joined)
    assert zpq.archive_add(None, 0, files, fragment_bytes=1 << 20) == zpq.archive_add(None, 0, files)   # nothing to cut


@pytest.mark.gpu
def test_fragmented_archive_on_gpu(zpq, gpu_ctx, tmp_path):
    big = (INPUTS["text2k"] * 40)[:70001]
    files = [("a.txt", "70001 bytes", big), ("e", "0 bytes", b""), ("b.bin", "4096 bytes", INPUTS["lcg4k"])]
    arc = zpq.archive_add(gpu_ctx, 2, files, fragment_bytes=8192)
    segs = zpq.archive_extract(gpu_ctx, arc)
    assert len(segs) == 9 + 1 + 1 and all(s["status"] == 0 and s["sha1_ok"] for s in segs)
    # every fragment is an ordinary block of its own: the same bytes the per-file writer makes for that slice
    pieces = [("a.txt", "70001 bytes", big[:8192])] + [("", "", big[o:o + 8192]) for o in range(8192, len(big), 8192)] + files[1:]
    assert arc == oracle_archive(2, pieces)
    joined = zpq.archive_extract(gpu_ctx, arc, join_unnamed=True)
    assert [(j["name"], j["data"]) for j in joined] == [(nm, d) for nm, _, d in files]
    # CLI: -fragment 3 = 8 KiB blocks; list and extract join them again
    src = tmp_path / "in"
    src.mkdir()
    (src / "a.txt").write_bytes(big)
    a = str(tmp_path / "f.zpaq")
    r = subprocess.run([CLI, "a", a, str(src / "a.txt"), "-m2", "-fragment", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(a, "rb").read() == oracle_archive(2, pieces[:9])
    r = subprocess.run([CLI, "l", a], capture_output=True, text=True)
    assert "a.txt (70001 bytes)" in r.stdout and "Total files: 1" in r.stdout
    r = subprocess.run([CLI, "x", a, "-to", str(tmp_path / "out")], capture_output=True, text=True)
    assert r.returncode == 0 and (tmp_path / "out" / "a.txt").read_bytes() == big


@pytest.mark.gpu
def test_sharded_over_two_contexts_is_position_stable(zpq, gpu_ctx):
    """Several GPUs: block b -> context b mod G, a host thread per context, no collective.  Rehearsed with two
    contexts on the one GPU of this box: the archive and the extracted files must not depend on G."""
    other = zpq.Context(0)
    try:
        files = file_set(seed=21, n=30) + [("big", "50000 bytes", (INPUTS["text2k"] * 30)[:50000])]
        one = zpq.archive_add(gpu_ctx, 2, files, fragment_bytes=8192)
        two = zpq.archive_add([gpu_ctx, other], 2, files, fragment_bytes=8192)
        assert one == two
        a = zpq.archive_extract(gpu_ctx, one, join_unnamed=True)
        b = zpq.archive_extract([gpu_ctx, other], one, join_unnamed=True)
        assert a == b and [(x["name"], x["data"]) for x in b] == [(nm, d) for nm, _, d in files]
        mixed = zpq.archive_add(None, 0, files[:3]) + one               # store-mode blocks go through the replay path
        assert [x["name"] for x in zpq.archive_extract([gpu_ctx, other], mixed, join_unnamed=True)] == [f[0] for f in files[:3] + files]
    finally:
        other.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5])
def test_reference_ci_scenario_directory_round_trip(tmp_path, level):
    """The reference's own CLI acceptance run (.github/workflows/compress-decompress.yml:41-115): a directory
    with a short text, a repetitive text, 5 KiB of random bytes, an empty file and a nested file;
    `add archive dir/ -mN -s1`, `list`, `extract --to out -s1`, compare every file, then `-test`."""
    d = tmp_path / "testdir"
    (d / "subdir").mkdir(parents=True)
    content = {
        "simple.txt": b"Hello, ZPAQ World! This is a test file for compression.\n",
        "repetitive.txt": b"".join(b"Line %d with repetitive content for compression testing\n" % i for i in range(1, 101)),
        "random.bin": bytes(random.Random(level).getrandbits(8) for _ in range(5120)),
        "empty.txt": b"",
        "nested.txt": b"File in subdirectory\n",
    }
    for nm, data in content.items():
        ((d / "subdir" / nm) if nm == "nested.txt" else (d / nm)).write_bytes(data)
    arc = str(tmp_path / "test_archive.zpaq")
    r = subprocess.run([CLI, "add", arc, str(d) + "/", "-m%d" % level, "-s1"], capture_output=True, text=True)
    assert r.returncode == 0 and "Files added: 5" in r.stdout and r.stdout.count("Added: ") == 5, r.stderr
    r = subprocess.run([CLI, "list", arc], capture_output=True, text=True)
    assert r.returncode == 0 and "Total files: 5" in r.stdout and "repetitive.txt (%d bytes)" % len(content["repetitive.txt"]) in r.stdout
    order = [ln.split(" (")[0] for ln in r.stdout.splitlines()[2:-2]]
    assert open(arc, "rb").read() == oracle_archive(level, [(nm, "%d bytes" % len(content[nm]), content[nm]) for nm in order])
    out = tmp_path / "extracted"
    r = subprocess.run([CLI, "extract", arc, "--to", str(out), "-s1"], capture_output=True, text=True)
    assert r.returncode == 0 and "Files extracted: 5" in r.stdout, r.stderr
    for nm, data in content.items():
        assert (out / nm).read_bytes() == data
    r = subprocess.run([CLI, "extract", arc, "-test"], capture_output=True, text=True)
    assert r.returncode == 0 and "Files verified: 5" in r.stdout

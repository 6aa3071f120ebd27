"""CPU tests of the product's host side: libzpaq_hip.so loads, exports every symbol
include/zpaq_hip.h declares, builds models (no GPU needed), and FAILS LOUDLY without
a GPU (no CPU fallback).  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

import oracle_lib as O
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(zpq):
    L = zpq.lib()
    hdr = open(os.path.join(ROOT, "include", "zpaq_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(zpq_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    # the flat C surface of the C++ front end (include/zpaq_frontend.hpp, extern "C" block)
    fe = open(os.path.join(ROOT, "include", "zpaq_frontend.hpp")).read()
    fe = re.sub(r"/\*.*?\*/", "", fe[fe.index('extern "C" {'):], flags=re.S)
    fnames = sorted(set(re.findall(r"\b(zpqf_[a-z0-9_]+)\s*\(", fe)))
    assert len(fnames) >= 30
    for n in fnames:
        assert hasattr(L, n), "missing export: " + n


def test_tables_match_oracle(zpq):
    L = zpq.lib()
    sq = (C.c_int32 * 4096)(); st = (C.c_int32 * 32768)()
    assert L.zpq_tables(sq, st) == 0      # also runs the start-up fingerprint self-check
    osq = (C.c_int32 * 4096)(); ost = (C.c_int32 * 32768)()
    O.lib().zo_tables(osq, ost, None, None, None)
    assert bytes(sq) == bytes(osq) and bytes(st) == bytes(ost)


def test_level_headers_and_scan_match_oracle(zpq):
    for lv in range(-1, 8):
        assert zpq.level_header(lv) == O.level_header(lv)
        h = zpq.level_header(lv)
        assert zpq.scan_header(h) == O.scan_header(h)
    # scanner quirks: opcode 63 skips 2 operand bytes, 255 only 1 (compressor.v:130-137)
    for h in (bytes([1, 1, 0, 0, 0, 0, 63, 1, 2, 7, 0]), bytes([1, 1, 0, 0, 0, 0, 255, 9, 9, 0]), b"", b"\1\2"):
        assert zpq.scan_header(h) == O.scan_header(h)


def test_model_walk(zpq):
    sizes = {1: 36, 2: 12, 3: 80, 4: 385, 5: 2052}
    for lv, mib in sizes.items():
        m = zpq.Model(level=lv)
        assert m.has_fast_path and m.ncomp == {1: 2, 2: 3, 3: 5, 4: 7, 5: 9}[lv]
        assert mib <= m.state_bytes / 2**20 < mib + 1.1
    assert zpq.Model(level=0).ncomp == 0
    from inputs import C4B
    m = zpq.Model(header=C4B)
    assert m.ncomp == 9 and not m.has_fast_path and m.offsets == (39, 40, 69)
    with pytest.raises(zpq.ZpqError) as e:        # MIX with m == 0 (predictor.v:426 divides by m)
        zpq.Model(header=bytes([1, 1, 0, 0, 1, 7, 4, 0, 0, 24, 255, 0, 56, 0]))
    assert e.value.code == -5
    with pytest.raises(zpq.ZpqError) as e:        # truncated component record
        zpq.Model(header=bytes([1, 1, 0, 0, 1, 9, 4]), offsets=(8, 8, 8))
    assert e.value.code == -3
    with pytest.raises(zpq.ZpqError) as e:        # table too large
        zpq.Model(header=bytes([1, 1, 0, 0, 1, 3, 40, 0, 56, 0]))
    assert e.value.code == -4


def test_no_gpu_means_loud_failure_not_fallback(zpq):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(zpq.ZpqError) as e:
        zpq.Context(0)
    assert e.value.code == -1
    # nothing under the product tree may reference the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zpaq-v_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in src and "zpaq_oracle" not in src and "zo_" not in src, f


def test_encoder_choice_is_host_logic(zpq, monkeypatch):
    """Which encoder a chain model's batch gets is decided on the host (no GPU needed): the wave-pipelined one
    (zpq_pipe.hip) for the shipped levels' shapes from 12 resident blocks on, within the LDS capacity the chain layout
    allows; the lane-per-component one otherwise (zpq_pipe_applies, internal; a zpq_model starts with its DModel)."""
    L = zpq.lib()
    L.zpq_pipe_applies.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.zpq_pipe_applies.restype = C.c_int
    L.zpq_chain_blocks_per_wg.argtypes = [C.c_void_p]
    L.zpq_chain_blocks_per_wg.restype = C.c_int
    monkeypatch.delenv("ZPQ_ENC_PIPE", raising=False)
    for level, cap in ((1, 32), (2, 32), (3, 16), (4, 16), (5, 12)):
        m = zpq.Model(level=level)
        assert L.zpq_chain_blocks_per_wg(m.h) == cap
        assert L.zpq_pipe_applies(m.h, cap, 11) == 0            # a wave must not live on a handful of lanes
        assert L.zpq_pipe_applies(m.h, cap, 12) == 1
        assert L.zpq_pipe_applies(m.h, cap, 8192) == 1
        assert L.zpq_pipe_applies(m.h, cap + 1, 8192) == 0      # more blocks than the workgroup's LDS holds
        monkeypatch.setenv("ZPQ_ENC_PIPE", "0")
        assert L.zpq_pipe_applies(m.h, cap, 8192) == 0
        monkeypatch.delenv("ZPQ_ENC_PIPE")
    # a chain with another program goes through the runtime-loop kernel; a non-chain model has no chain layout at all
    hdr = bytes([3, 8, 0, 0, 2, 3, 16, 8, 16, 0, 0, 104, 17, 95, 0, 59, 135, 7, 112, 25, 60, 59, 112, 56, 0])
    m = zpq.Model(header=hdr)
    assert m.has_fast_path and L.zpq_pipe_applies(m.h, 8, 100) == 0
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from inputs import C4B
    m = zpq.Model(header=C4B)
    assert not m.has_fast_path and L.zpq_pipe_applies(m.h, 8, 100) == 0


def test_decoder_choice_is_host_logic(zpq, monkeypatch):
    """The wave-split decoder (zpq_dpipe.hip) is opt-in -- it measured slower than the lane-per-component one -- and only
    ever takes the dense chains of levels 1-3 from 12 resident blocks on, at most 32 per workgroup (a block is a lane pair);
    the 'touched bitmap' decode path exists only in timing builds (zpq_dpipe_applies, zpq_chain_touch_decode: internal)."""
    L = zpq.lib()
    for f in (L.zpq_dpipe_applies,):
        f.argtypes = [C.c_void_p, C.c_int, C.c_int]
        f.restype = C.c_int
    L.zpq_chain_touch_decode.argtypes = [C.c_void_p]
    L.zpq_chain_touch_decode.restype = C.c_int
    L.zpq_pipe_touch.restype = C.c_int
    monkeypatch.delenv("ZPQ_DEC_PIPE", raising=False)
    for level in (1, 2, 3, 4, 5):
        m = zpq.Model(level=level)
        assert L.zpq_dpipe_applies(m.h, 16, 8192) == 0           # not asked for
        assert L.zpq_chain_touch_decode(m.h) == 0 and L.zpq_pipe_touch() == 0
    monkeypatch.setenv("ZPQ_DEC_PIPE", "1")
    for level, cap in ((1, 32), (2, 32), (3, 16)):
        m = zpq.Model(level=level)
        assert L.zpq_dpipe_applies(m.h, cap, 11) == 0
        assert L.zpq_dpipe_applies(m.h, cap, 12) == 1
        assert L.zpq_dpipe_applies(m.h, cap, 8192) == 1
        assert L.zpq_dpipe_applies(m.h, 33, 8192) == 0
    for level in (4, 5):                                         # a MIX2 follows the chain: the lane-per-component decoder
        assert L.zpq_dpipe_applies(zpq.Model(level=level).h, 12, 8192) == 0


def test_which_general_models_the_wave_pipeline_encodes(zpq, monkeypatch):
    """Host logic (zpq_gpipe_applies, internal): the wave-per-component encoder takes a general model when its program is the
    shipped hash chain, it has at most 15 components and every input names an EARLIER component (a MIX: at most eight)."""
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from inputs import C4B
    from test_gpu_gpipe import LEFT, TAKEN
    from test_gpu_lanes import MODELS, hdr
    L = zpq.lib()
    L.zpq_gpipe_applies.argtypes = [C.c_void_p]
    L.zpq_gpipe_applies.restype = C.c_int
    monkeypatch.delenv("ZPQ_ENC_GPIPE", raising=False)
    assert sorted(TAKEN + LEFT) == sorted(MODELS)
    for name in TAKEN:
        assert L.zpq_gpipe_applies(zpq.Model(header=hdr(MODELS[name])).h) == 1, name
    for name in LEFT:
        assert L.zpq_gpipe_applies(zpq.Model(header=hdr(MODELS[name])).h) == 0, name
    assert L.zpq_gpipe_applies(zpq.Model(header=C4B).h) == 1
    for level in range(1, 6):                                    # (level 1's program is not the hash chain)
        assert L.zpq_gpipe_applies(zpq.Model(level=level).h) == (1 if level >= 2 else 0)
    # a MIX over more than eight inputs, more than 15 components, a program that is not the hash chain
    assert L.zpq_gpipe_applies(zpq.Model(header=hdr([[3, 12]] + [[8, 12, i] for i in range(9)] + [[7, 4, 0, 10, 16, 255]])).h) == 0
    assert L.zpq_gpipe_applies(zpq.Model(header=hdr([[3, 12]] + [[8, 12, i] for i in range(15)])).h) == 0
    other = bytes([3, 6, 0, 0, 3, 2, 12, 40, 3, 12, 8, 12, 1, 0]) + bytes([112, 25, 59, 112, 56, 0])
    assert L.zpq_gpipe_applies(zpq.Model(header=other).h) == 0
    monkeypatch.setenv("ZPQ_ENC_GPIPE", "0")
    assert L.zpq_gpipe_applies(zpq.Model(header=C4B).h) == 0
    # the decoder of the same waves (k_gdec) takes the same models
    L.zpq_gdec_applies.argtypes = [C.c_void_p]
    L.zpq_gdec_applies.restype = C.c_int
    monkeypatch.delenv("ZPQ_DEC_GPIPE", raising=False)
    for name in TAKEN:
        assert L.zpq_gdec_applies(zpq.Model(header=hdr(MODELS[name])).h) == 1, name
    for name in LEFT:
        assert L.zpq_gdec_applies(zpq.Model(header=hdr(MODELS[name])).h) == 0, name
    assert L.zpq_gdec_applies(zpq.Model(header=C4B).h) == 1
    monkeypatch.setenv("ZPQ_DEC_GPIPE", "0")
    assert L.zpq_gdec_applies(zpq.Model(header=C4B).h) == 0


def test_encoder_wave_orders_host_logic(zpq):
    """ZPQ_ENC_SPLIT's wave order is checked on the host before anything is launched (zpq_pipe_split_order_valid, internal): every
    component exactly once -- whole (8 + c), as its history / weights pair (2c, 2c + 1), or, for a three-component chain, both ISSEs
    PAIRED on the halves of one wave (c = H1+H2, d = P1+P2) -- and exactly one coder (6)."""
    L = zpq.lib()
    L.zpq_pipe_split_order_valid.argtypes = [C.c_char_p, C.c_int]
    L.zpq_pipe_split_order_valid.restype = C.c_int
    ok = lambda o, n: L.zpq_pipe_split_order_valid(o.encode(), n)
    for o in ("60231", "6823", "6019", "02316", "689"):
        assert ok(o, 2) == 1, o
    for o in ("6089", "64523", "0123", "60011", "602316", "60c31", "6d02", "60cd1", "", "6", "7089", "x"):
        assert ok(o, 2) == 0, o
    for o in ("6024135", "689a", "682345", "60cd1", "c6d8", "60c351", "6024d1", "86cd", "6cd01"):
        assert ok(o, 3) == 1, o
    for o in ("60231", "6024137", "6802345", "60cd", "60cd12", "6ccd1", "60cd13", "60cd15", "6c2d01", "60cdd1", "cd01"):
        assert ok(o, 3) == 0, o
    assert ok("60231", 0) == 0 and ok("60231", 4) == 0

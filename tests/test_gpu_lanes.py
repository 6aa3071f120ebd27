"""GPU parity: the lane-per-component kernel (k_lanes: any model with <= 64 components, all
nine component types) against the CPU oracle, the golden vectors and the lane-0 interpreter."""
import hashlib
import json
import os
import random
import sys

import pytest

import oracle_lib as O
import workload as W

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from inputs import C4B, INPUTS  # noqa: E402

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))

HC = [74, 18, 104, 95, 0] + [59, 112, 25] * 7 + [59, 112, 56, 0]


def hdr(comps, hh=4, hm=16):
    b = [hh, hm, 0, 0, len(comps)]
    for c in comps:
        b += c
    return bytes(b + [0] + HC)


MODELS = {
    "const": [[1, 160]],
    "cm": [[2, 16, 255]],
    "cm_small_limit": [[2, 8, 3]],
    "match": [[3, 16], [4, 16, 16]],
    "matchonly": [[4, 12, 10]],
    "avg": [[3, 16], [2, 16, 255], [5, 0, 1, 128]],
    "avg_bad_index": [[3, 16], [5, 0, 9, 100]],
    "mix2": [[3, 16], [2, 16, 255], [6, 8, 0, 1, 24, 255]],
    "mix2_mask0": [[3, 16], [2, 12, 20], [6, 0, 0, 1, 30, 0]],
    "isse": [[3, 16], [8, 16, 0]],
    "isse_stale_input": [[8, 12, 1], [3, 12]],           # j >= i: uses last bit's p (quirk Q13)
    "isse_no_input": [[8, 12, 7]],                       # j >= n: clamp2k(w1 >> 10)
    "mix": [[3, 16], [2, 16, 255], [7, 8, 0, 2, 24, 255]],
    "mix_over_range": [[3, 12], [7, 4, 0, 5, 16, 15]],   # m reaches past n: loop stops at n
    "sse": [[3, 16], [9, 8, 0, 32, 255]],
    "sse_out_of_table": [[3, 16], [9, 2, 0, 1, 2]],      # hashes far beyond the table: p = 0 (quirk Q10)
    "unknown_type": [[3, 12], [0], [8, 12, 0]],          # type 0 advances one byte (predictor.v:465-467)
    # forward and self references feed the coded prediction: an input index >= the consumer's own index
    # reads what that component predicted for the PREVIOUS bit (predictor.v:536-668 walks one p[] in order)
    "fwd_avg": [[5, 1, 2, 100], [2, 12, 255], [3, 12], [6, 4, 0, 1, 24, 255]],
    "fwd_isse": [[8, 12, 1], [3, 12], [5, 0, 1, 128]],
    "fwd_mix_self": [[7, 4, 0, 3, 16, 255], [2, 12, 255], [3, 12], [5, 0, 2, 90]],
    "fwd_sse_self": [[9, 4, 0, 32, 255], [3, 12], [5, 0, 1, 60]],
    "fwd_mix2": [[6, 4, 1, 2, 20, 255], [1, 200], [4, 12, 12], [5, 0, 2, 128]],
    "fwd_chain": [[8, 10, 3], [5, 0, 2, 77], [2, 10, 30], [9, 3, 1, 32, 100], [7, 2, 0, 4, 12, 3]],
}


def big_model():
    """20 components, MIX over 19 inputs, final SSE."""
    comps = [[3, 12]] + [[8, 12, i] for i in range(0, 8)] + [[2, 12, 60], [4, 12, 12], [1, 100]] + \
            [[2, 10, 255], [3, 10], [5, 0, 1, 77], [6, 6, 2, 3, 20, 255], [8, 10, 4], [2, 14, 8], [4, 10, 10]]
    comps += [[7, 6, 0, 19, 14, 255], [9, 6, 19, 20, 200]]
    return hdr(comps, hh=5)


def run_parity(zpq, ctx, header, blocks, cap=None):
    offs = O.scan_header(header)
    model = zpq.Model(header=header, offsets=offs)
    F = zpq.FLAG_PP | zpq.FLAG_LANES
    coded, status, _ = ctx.encode_blocks(model, blocks, cap=cap, flags=F)
    # k_rows: n <= 16; k_gpipe (wave per component): hash-chain program and every input an earlier component
    assert ctx.last_kernel_name in ("k_lanes<encode>", "k_rows<encode>", "k_gpipe<encode>")
    assert (status == 0).all(), status
    want = [O.Codec(header, offs).encode(b) for b in blocks]
    assert coded == want
    if ctx.last_kernel_name == "k_gpipe<encode>":        # the lane-per-component encoder stays covered
        os.environ["ZPQ_ENC_GPIPE"] = "0"
        try:
            other, status, _ = ctx.encode_blocks(model, blocks, cap=cap, flags=F)
        finally:
            del os.environ["ZPQ_ENC_GPIPE"]
        assert ctx.last_kernel_name in ("k_lanes<encode>", "k_rows<encode>") and (status == 0).all() and other == want
    dec, status, consumed, _, first = ctx.decode_blocks(model, coded, cap=max(len(b) for b in blocks) + 16, flags=F)
    assert ctx.last_kernel_name in ("k_lanes<decode>", "k_rows<decode>", "k_gdec<decode>")
    assert (status == 0).all() and dec == blocks and (first == 0).all()
    assert [int(c) for c in consumed] == [len(c) for c in coded]
    if ctx.last_kernel_name == "k_gdec<decode>":         # the lane-per-component decoder stays covered
        os.environ["ZPQ_DEC_GPIPE"] = "0"
        try:
            dec, status, consumed, _, first = ctx.decode_blocks(model, coded, cap=max(len(b) for b in blocks) + 16, flags=F)
        finally:
            del os.environ["ZPQ_DEC_GPIPE"]
        assert ctx.last_kernel_name in ("k_lanes<decode>", "k_rows<decode>")
        assert (status == 0).all() and dec == blocks and [int(c) for c in consumed] == [len(c) for c in coded]
    return coded


@pytest.mark.parametrize("name", sorted(MODELS))
def test_single_type_models(zpq, gpu_ctx, name):
    run_parity(zpq, gpu_ctx, hdr(MODELS[name]), [INPUTS["lcg4k"], INPUTS["text2k"], INPUTS["zeros256"], b"", b"a"])


def test_c4b_golden_streams(zpq, gpu_ctx):
    model = zpq.Model(header=C4B)
    for mode in ("pp", "raw"):
        ks = [k for k in sorted(G["streams"]) if k.startswith("c4b/") and k.endswith(mode)]
        blocks = [INPUTS[k.split("/")[1]] for k in ks]
        coded, status, _ = gpu_ctx.encode_blocks(model, blocks, flags=zpq.FLAG_PP if mode == "pp" else 0)
        assert gpu_ctx.last_kernel_name == "k_gpipe<encode>" and (status == 0).all()
        for k, c in zip(ks, coded):
            assert hashlib.sha256(c).hexdigest() == G["streams"][k]["sha256"], k
        dec, status, *_ = gpu_ctx.decode_blocks(model, coded, cap=8192, flags=zpq.FLAG_PP if mode == "pp" else 0)
        assert (status == 0).all() and dec == blocks


def test_c4b_ragged_batch_and_lane0_agreement(zpq, gpu_ctx):
    rnd = random.Random(606)
    blocks = []
    for i in range(70):                                   # more blocks than one workgroup carries
        n = rnd.choice([0, 1, 2, 33, 255, 256, 257, 1000, 2600])
        blocks.append(bytes(rnd.getrandbits(8) for _ in range(n)) if i % 3 else
                      bytes(rnd.choice(b"the quick brown fox ") for _ in range(n)))
    coded = run_parity(zpq, gpu_ctx, C4B, blocks)
    model = zpq.Model(header=C4B)
    coded0, status, _ = gpu_ctx.encode_blocks(model, blocks, flags=zpq.FLAG_PP | zpq.FLAG_GENERIC)
    assert gpu_ctx.last_kernel_name == "k_generic<encode>"
    assert coded0 == coded


def test_twenty_components_mix_over_many_lanes(zpq, gpu_ctx):
    run_parity(zpq, gpu_ctx, big_model(), [INPUTS["lcg4k"][:1500], INPUTS["text2k"], bytes(300)])


def test_random_zpaql_programs_in_a_mixed_model(zpq, gpu_ctx):
    rnd = random.Random(99)
    valid = [op for op in range(256) if op not in (56, 255, 57, 58, 61, 62) and not ((op & 7) in (5, 6) and op < 56)
             and not (120 <= op < 128) and not (240 <= op < 255)]
    for trial in range(6):
        prog = []
        for _ in range(rnd.randint(8, 30)):
            op = rnd.choice(valid)
            prog.append(op)
            if op & 7 == 7:
                prog.append(rnd.choice([0, 1, 2, 3]) if op in (39, 47, 63) else
                            rnd.choice([v for v in range(1, 255) if v not in (39, 47, 63)]))
        prog += [112, 25, 59, 112, 56, 0]
        header = bytes([3, 6, 0, 0, 3, 2, 12, 40, 3, 12, 8, 12, 1, 0]) + bytes(prog)
        blocks = [bytes(rnd.getrandbits(8) for _ in range(400)), INPUTS["text2k"][:500]]
        run_parity(zpq, gpu_ctx, header, blocks)


@pytest.mark.parametrize("level", [1, 2, 3])
def test_shipped_levels_through_lanes_kernel(zpq, gpu_ctx, level):
    """Three independent device implementations (chain, lanes, lane-0) must agree with the oracle."""
    rnd = random.Random(level)
    blocks = [bytes(rnd.getrandbits(8) for _ in range(900)), INPUTS["text2k"], bytes(500), b""]
    coded = run_parity(zpq, gpu_ctx, O.level_header(level), blocks)
    model = zpq.Model(level=level)
    chain, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert gpu_ctx.last_kernel_name == "k_chain<encode>" and chain == coded   # (four blocks: below the pipelined encoder's minimum)


def test_slot_reuse_and_overflow(zpq, gpu_ctx):
    model = zpq.Model(header=C4B)
    zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 5 * model.state_bytes + 100)
    try:
        rnd = random.Random(3)
        blocks = [bytes(rnd.getrandbits(8) for _ in range(rnd.choice([100, 700]))) for _ in range(23)]
        coded, status, _ = gpu_ctx.encode_blocks(model, blocks)
        assert gpu_ctx.last_slots == 5 and (status == 0).all()
        assert coded == [O.Codec(C4B).encode(b) for b in blocks]
    finally:
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)
    _, status, out_len = gpu_ctx.encode_blocks(model, [INPUTS["lcg4k"]], cap=64)
    assert status[0] == -7 and int(out_len[0]) == len(O.Codec(C4B).encode(INPUTS["lcg4k"]))


@pytest.mark.parametrize("nb", [256, 16384])
def test_c4b_at_baseline_block_size(zpq, gpu_ctx, nb):
    """C4b (all nine component types) on 64 KiB blocks of all four classes through the lanes kernel:
    round-trip properties on every block, byte parity with the oracle on a sample.  16 384 blocks = the shape
    bench.py's `secondary` ships (k_rows at its resident capacity, four workgroups per CU)."""
    import torch
    import workload as W
    torch.cuda.empty_cache()
    size = 65536
    arr = W.make_blocks_fast(nb, size)
    model = zpq.Model(header=C4B)
    dev = torch.device("cuda:0")
    cap = size * 6 + 1024
    d_in = torch.from_numpy(arr.reshape(-1)).to(dev)
    i64 = dict(dtype=torch.int64, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    in_off = torch.arange(nb + 1, **i64) * size
    out_off = torch.arange(nb + 1, **i64) * cap
    d_out = torch.zeros(nb * cap, dtype=torch.uint8, device=dev)
    d_dec = torch.zeros(nb * size, dtype=torch.uint8, device=dev)
    d_len, d_st, d_dlen, d_cons, d_code, d_first, d_dst = (torch.zeros(nb, **i32) for _ in range(7))
    torch.cuda.synchronize()                             # order torch's fills before the ctx stream's kernels
    gpu_ctx.encode_blocks_dev(model, nb, d_in.data_ptr(), in_off.data_ptr(), zpq.FLAG_PP, d_out.data_ptr(),
                              out_off.data_ptr(), d_len.data_ptr(), d_st.data_ptr())
    assert gpu_ctx.last_kernel_name == "k_gpipe<encode>"
    if nb == 16384:
        assert gpu_ctx.last_slots == min(nb, gpu_ctx.resident_capacity(model)) >= 8192
    gpu_ctx.decode_blocks_dev(model, nb, d_out.data_ptr(), out_off.data_ptr(), zpq.FLAG_PP, d_dec.data_ptr(),
                              in_off.data_ptr(), d_dlen.data_ptr(), d_cons.data_ptr(), d_code.data_ptr(),
                              d_first.data_ptr(), d_dst.data_ptr())
    gpu_ctx.sync()
    assert gpu_ctx.last_kernel_name == "k_gdec<decode>"
    assert bool((d_st == 0).all()) and bool((d_dst == 0).all()) and bool((d_dlen == size).all())
    assert bool(torch.equal(d_dec, d_in)) and bool(torch.equal(d_cons, d_len)) and bool((d_first == 0).all())
    out, lens = d_out.cpu().numpy(), d_len.cpu().numpy()
    sample = [0, 1, 2, 3, 130, 255] + ([4097, 8190, 12291, 16380, 16383] if nb > 256 else [])
    want = O.encode_blocks(C4B, [arr[i].tobytes() for i in sample], nthreads=6, slack=cap)
    for i, w in zip(sample, want):
        assert out[i * cap:i * cap + int(lens[i])].tobytes() == w, i


def test_four_blocks_per_wave_equals_one_block_per_wave(zpq, gpu_ctx, monkeypatch):
    """Models with at most 16 components and the shipped hash-chain program ride four to a wave (k_rows: a block is a
    16-lane row); the one-block-per-wave kernel stays for everything else.  Same model, same blocks, both kernels:
    identical streams, equal to the oracle; a ragged batch so that rows of one wave end at different times, more blocks
    than one workgroup holds, and a budget that makes rows reuse their slots."""
    monkeypatch.setenv("ZPQ_ENC_GPIPE", "0")             # (the kernels under test here are k_rows / k_lanes, not the wave pipelines)
    monkeypatch.setenv("ZPQ_DEC_GPIPE", "0")
    model = zpq.Model(header=C4B)
    rnd = random.Random(16)
    blocks = [bytes(W.make_block(7 * b + 1, rnd.choice([0, 1, 5, 64, 700, 2500, 6000]))) for b in range(41)]
    want = [O.Codec(C4B).encode(b) for b in blocks]
    rows, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert (status == 0).all() and rows == want
    dec, status, consumed, _, first = gpu_ctx.decode_blocks(model, rows, cap=8192)
    assert (status == 0).all() and dec == blocks and (first == 0).all()
    zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 6 * model.state_bytes + 100)
    try:
        reuse, status, _ = gpu_ctx.encode_blocks(model, blocks)
        assert gpu_ctx.last_slots == 6 and (status == 0).all() and reuse == want
    finally:
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)
    assert gpu_ctx.last_kernel_name == "k_rows<encode>"
    monkeypatch.setenv("ZPQ_LANES_ROWS", "0")
    one, status, _ = gpu_ctx.encode_blocks(model, blocks)
    assert gpu_ctx.last_kernel_name == "k_lanes<encode>"
    assert (status == 0).all() and one == want
    dec, status, *_ = gpu_ctx.decode_blocks(model, one, cap=8192)
    assert (status == 0).all() and dec == blocks


def test_general_programs_ride_four_blocks_per_wave(zpq, gpu_ctx, monkeypatch):
    """Round 3: a model with at most 16 components whose HCOMP program is NOT the shipped hash chain also rides four blocks
    to a wave (k_rows); its program runs through the interpreter on the first lane of every row.  Random programs (every
    opcode group incl. jumps, M and H traffic, R registers), a ragged batch whose rows end at different times, more blocks
    than slots: identical to the one-block-per-wave kernel and to the oracle, decoded back."""
    rnd = random.Random(2026)
    valid = [op for op in range(256) if op not in (56, 255, 57, 58, 61, 62) and not ((op & 7) in (5, 6) and op < 56)
             and not (120 <= op < 128) and not (240 <= op < 255)]
    for trial in range(4):
        prog = []
        for _ in range(rnd.randint(8, 30)):
            op = rnd.choice(valid)
            prog.append(op)
            if op & 7 == 7:
                prog.append(rnd.choice([0, 1, 2, 3]) if op in (39, 47, 63) else
                            rnd.choice([v for v in range(1, 255) if v not in (39, 47, 63)]))
        prog += [112, 25, 59, 112, 56, 0]
        header = bytes([3, 6, 0, 0, 3, 2, 12, 40, 3, 12, 8, 12, 1, 0]) + bytes(prog)
        model = zpq.Model(header=header)
        blocks = [bytes(W.make_block(11 * b + trial, rnd.choice([0, 1, 7, 64, 500, 1500, 3000]))) for b in range(23)]
        want = [O.Codec(header).encode(b) for b in blocks]
        monkeypatch.delenv("ZPQ_LANES_ROWS", raising=False)
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 7 * model.state_bytes + 100)
        try:
            rows, status, _ = gpu_ctx.encode_blocks(model, blocks)
            assert gpu_ctx.last_kernel_name == "k_rows<encode>" and gpu_ctx.last_slots == 7
            assert (status == 0).all() and rows == want
            dec, status, *_ = gpu_ctx.decode_blocks(model, rows, cap=4096)
            assert gpu_ctx.last_kernel_name == "k_rows<decode>" and (status == 0).all() and dec == blocks
        finally:
            zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)
        monkeypatch.setenv("ZPQ_LANES_ROWS", "h")               # round 2's rule: only hash-chain programs ride rows
        one, status, _ = gpu_ctx.encode_blocks(model, blocks)
        assert gpu_ctx.last_kernel_name == "k_lanes<encode>" and (status == 0).all() and one == want
    # a program that never halts: the step cap is a per-block status on either kernel, not a hang
    monkeypatch.delenv("ZPQ_LANES_ROWS", raising=False)
    loop = bytes([3, 6, 0, 0, 3, 2, 12, 40, 3, 12, 8, 12, 1, 0, 63, 0xFD, 0])   # jmp to itself: rel = ((0xFD + 128) & 255) - 127 = -2
    model = zpq.Model(header=loop)
    _, status, _ = gpu_ctx.encode_blocks(model, [b"abc", b"", b"x" * 40])
    assert gpu_ctx.last_kernel_name == "k_rows<encode>"
    assert status[0] == -8 and status[2] == -8                     # ZPQ_E_VMSTEPS (the empty block still codes the PP byte)

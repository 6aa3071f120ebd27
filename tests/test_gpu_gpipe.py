"""GPU parity: the wave-per-component ENCODER of general models (k_gpipe, zpq_gpipe.hip: wave = component, lane = block,
predictions handed from wave to wave through LDS rings) against the CPU oracle and against the lane-per-component encoder
(k_rows / k_lanes) on the same batches; and their bit-synchronous DECODER (k_gdec: the same waves, a barrier per level of the
prediction chain) against k_rows / k_lanes on the same streams."""
import os
import random
import sys

import pytest

import oracle_lib as O
import workload as W

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from inputs import C4B, INPUTS  # noqa: E402
from test_gpu_lanes import MODELS, hdr  # noqa: E402

pytestmark = pytest.mark.gpu

# which of test_gpu_lanes' models the pipeline takes: all nine types, inputs that are EARLIER components only
TAKEN = ["const", "cm", "cm_small_limit", "match", "matchonly", "avg", "mix2", "mix2_mask0", "isse", "mix", "sse", "sse_out_of_table"]
LEFT = ["avg_bad_index", "isse_stale_input", "isse_no_input", "mix_over_range", "unknown_type",
        "fwd_avg", "fwd_isse", "fwd_mix_self", "fwd_sse_self", "fwd_mix2", "fwd_chain"]


def both_encoders(zpq, ctx, model, blocks, flags, cap=None):
    pipe, st1, len1 = ctx.encode_blocks(model, blocks, flags=flags, cap=cap)
    name = ctx.last_kernel_name
    os.environ["ZPQ_GPIPE_BATCH"] = "0"                  # the pipeline's bit-serial stages (one table access per bit, in bit order)
    try:
        serial, st0, len0 = ctx.encode_blocks(model, blocks, flags=flags, cap=cap)
        assert ctx.last_kernel_name == name
    finally:
        del os.environ["ZPQ_GPIPE_BATCH"]
    assert serial == pipe and list(st0) == list(st1) and list(len0) == list(len1)
    os.environ["ZPQ_ENC_GPIPE"] = "0"
    try:
        rows, st2, len2 = ctx.encode_blocks(model, blocks, flags=flags, cap=cap)
        other = ctx.last_kernel_name
    finally:
        del os.environ["ZPQ_ENC_GPIPE"]
    return name, pipe, st1, len1, other, rows, st2, len2


def both_decoders(zpq, ctx, model, coded, blocks, flags, cap, intact=True):
    """The wave-per-component decoder (k_gdec) and the lane-per-component one on the same streams: blocks, status, bytes
    consumed, PP byte and the coder's last window must agree, and equal what was encoded."""
    got = ctx.decode_blocks(model, coded, cap=cap, flags=flags)
    assert ctx.last_kernel_name == "k_gdec<decode>"
    os.environ["ZPQ_DEC_GPIPE"] = "0"
    try:
        ref = ctx.decode_blocks(model, coded, cap=cap, flags=flags)
        assert ctx.last_kernel_name in ("k_rows<decode>", "k_lanes<decode>")
    finally:
        del os.environ["ZPQ_DEC_GPIPE"]
    dec, status, consumed, code, first = got
    assert dec == ref[0] and all(list(a) == list(b) for a, b in zip(got[1:], ref[1:]))
    ok = [i for i in range(len(blocks)) if status[i] == 0 and intact]
    assert all(dec[i] == blocks[i] and int(consumed[i]) == len(coded[i]) for i in ok)
    if flags & zpq.FLAG_PP:
        assert all(int(first[i]) == 0 for i in ok)
    return status


@pytest.mark.parametrize("name", TAKEN)
def test_every_component_type_alone(zpq, gpu_ctx, name):
    header = hdr(MODELS[name])
    model = zpq.Model(header=header)
    blocks = [INPUTS["lcg4k"], INPUTS["text2k"], INPUTS["zeros256"], b"", b"a", bytes(W.make_block(3, 3000)), bytes(W.make_block(2, 1500))]
    for flags in (zpq.FLAG_PP | zpq.FLAG_LANES, zpq.FLAG_LANES):
        pp = bool(flags & zpq.FLAG_PP)
        name_, pipe, st1, _, other, rows, st2, _ = both_encoders(zpq, gpu_ctx, model, blocks, flags)
        assert name_ == "k_gpipe<encode>" and other in ("k_rows<encode>", "k_lanes<encode>")
        assert (st1 == 0).all() and (st2 == 0).all()
        want = [O.Codec(header).encode(b, pp=pp) for b in blocks]
        assert rows == want
        assert pipe == want
        assert (both_decoders(zpq, gpu_ctx, model, pipe, blocks, flags, 4200) == 0).all()


@pytest.mark.parametrize("level", [2, 3, 4, 5])
def test_shipped_levels(zpq, gpu_ctx, level):
    model = zpq.Model(level=level)
    rnd = random.Random(level)
    blocks = [bytes(W.make_block(5 * b + level, rnd.choice([0, 1, 9, 300, 2000, 5000]))) for b in range(9)]
    F = zpq.FLAG_PP | zpq.FLAG_LANES
    name_, pipe, st1, _, other, rows, st2, _ = both_encoders(zpq, gpu_ctx, model, blocks, F)
    assert name_ == "k_gpipe<encode>" and (st1 == 0).all()
    assert pipe == [O.Codec(O.level_header(level)).encode(b) for b in blocks] == rows
    assert (both_decoders(zpq, gpu_ctx, model, pipe, blocks, F, 5016) == 0).all()


def test_all_nine_types_ragged_batch_slot_reuse_and_overflow(zpq, gpu_ctx):
    """More blocks than a workgroup's 64 lanes, lanes that end at different bytes, fewer slots than blocks (a lane codes
    several blocks in turn, the workgroup re-initialises its slots between rounds), an output buffer that is too small."""
    model = zpq.Model(header=C4B)
    rnd = random.Random(77)
    blocks = [bytes(W.make_block(13 * b + 5, rnd.choice([0, 1, 2, 33, 255, 256, 257, 1000, 2600, 7000]))) for b in range(150)]
    want = [O.Codec(C4B).encode(b) for b in blocks]
    name_, pipe, st1, _, other, rows, st2, _ = both_encoders(zpq, gpu_ctx, model, blocks, zpq.FLAG_PP)
    assert name_ == "k_gpipe<encode>" and other == "k_rows<encode>" and (st1 == 0).all() and (st2 == 0).all()
    assert pipe == want and rows == want
    assert (both_decoders(zpq, gpu_ctx, model, want, blocks, zpq.FLAG_PP, 7016) == 0).all()
    zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 70 * model.state_bytes + 100)
    try:
        reuse, status, _ = gpu_ctx.encode_blocks(model, blocks)
        assert gpu_ctx.last_kernel_name == "k_gpipe<encode>" and gpu_ctx.last_slots == 70
        assert (status == 0).all() and reuse == want
        assert (both_decoders(zpq, gpu_ctx, model, want, blocks, zpq.FLAG_PP, 7016) == 0).all() and gpu_ctx.last_slots == 70
    finally:
        zpq.lib().zpq_ctx_set_state_budget(gpu_ctx.h, 150 << 30)
    # a batch of nothing but empty blocks (no input byte to read; with the PP flag one coded byte, without it none)
    for flags in (zpq.FLAG_PP, 0):
        empty = [b""] * 5
        coded, status, _ = gpu_ctx.encode_blocks(model, empty, flags=flags)
        assert gpu_ctx.last_kernel_name == "k_gpipe<encode>" and (status == 0).all()
        assert coded == [O.Codec(C4B).encode(b"", pp=bool(flags)) for _ in empty]
        assert (both_decoders(zpq, gpu_ctx, model, coded, empty, flags, 16) == 0).all()
    small = [INPUTS["lcg4k"], b"abc", INPUTS["text2k"]]
    _, status, out_len = gpu_ctx.encode_blocks(model, small, cap=64)
    assert gpu_ctx.last_kernel_name == "k_gpipe<encode>"
    assert list(status) == [-7, 0, -7] and [int(x) for x in out_len] == [len(O.Codec(C4B).encode(b)) for b in small]
    # a decoder whose output buffer is too small stops that block with the same status; the others are untouched
    coded = [O.Codec(C4B).encode(b) for b in small]
    status = both_decoders(zpq, gpu_ctx, model, coded, small, zpq.FLAG_PP, 100)
    assert list(status) == [-7, 0, -7]
    # a stream cut short decodes to SOMETHING on both decoders alike (zeros are read past the end): no hang, no fault
    both_decoders(zpq, gpu_ctx, model, [c[:len(c) // 2] for c in coded], small, zpq.FLAG_PP, 5000, intact=False)


def test_long_input_distances_need_a_deeper_ring(zpq, gpu_ctx):
    """A final MIX2 / SSE that names component 0 from position 13: predictions live 13 bytes in the ring."""
    comps = [[3, 12], [2, 12, 60]] + [[8, 12, i] for i in range(1, 11)] + [[6, 6, 0, 11, 20, 255], [9, 6, 0, 20, 200]]
    header = hdr(comps)
    model = zpq.Model(header=header)
    blocks = [INPUTS["lcg4k"][:1800], INPUTS["text2k"], bytes(300), b""]
    name_, pipe, st1, _, other, rows, st2, _ = both_encoders(zpq, gpu_ctx, model, blocks, zpq.FLAG_PP)
    assert name_ == "k_gpipe<encode>" and (st1 == 0).all()
    assert pipe == [O.Codec(header).encode(b) for b in blocks] == rows
    assert (both_decoders(zpq, gpu_ctx, model, pipe, blocks, zpq.FLAG_PP, 2100) == 0).all()

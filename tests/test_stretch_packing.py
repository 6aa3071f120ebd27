"""The facts the kernels' eight-instruction stretch rests on (zpq_chain.hip `stretch_of`, zpq_model.cpp's packing), checked
against the CPU oracle's tables: the reference's stretch table (predictor.v:73-96,169-214) rises by at most 1 per entry
everywhere but at its last one, index 0 reads as index 1, and no ICM counter (predictor.v:357-370,701-709;
statetable.v:90-100) ever has index 32767."""
import numpy as np

import oracle_lib as O


def stretch_table():
    L = O.lib()
    return np.array([L.zo_stretch(i) for i in range(32768)], dtype=np.int64)


def test_the_table_has_one_step_above_one_and_it_is_the_last():
    st = stretch_table()
    d = np.diff(st)
    assert d.min() >= 0                                   # monotone
    assert list(np.nonzero(d > 1)[0]) == [32766]          # 375 -> 2047 at index 32767 (predictor.v:81-82)
    assert st[0] == st[1] == -375 and st[32766] == 375 and st[32767] == 2047   # stretch(0) is clamped to stretch(1)


def test_packed_words_decode_every_reachable_index():
    """base (i16) + 15 step bits per 16 entries, word 0 starting at stretch(1): value = base + popcount(steps below q & 15)."""
    st = stretch_table()
    raw = st.copy()
    words = []
    for b in range(2048):
        lo = b * 16
        base = int(raw[max(lo, 1)]) if b == 0 else int(raw[lo])
        bits = 0
        for k in range(1, 16):
            prev = raw[lo + k - 1] if lo + k - 1 >= 1 else raw[1]
            if raw[lo + k] != prev:
                bits |= 1 << k
        words.append((base, bits))
    for q in range(32767):
        base, bits = words[q >> 4]
        k = q & 15
        v = base + bin((bits >> 1) & ((1 << k) - 1)).count("1")
        assert v == st[max(q, 1)], q


def test_no_icm_counter_reaches_the_last_index():
    L = O.lib()
    top = max(L.zo_cminit(s) for s in range(256))
    assert (top >> 8) == 31987 and top < (1 << 23)
    # one update from any counter below index 32767 (predictor.v:701-709: cm += (y*32767 - (cm >> 8)) >> 2) stays below it
    cm = np.arange(0, 32767 * 256, dtype=np.int64)
    up = cm + ((32767 - (cm >> 8)) >> 2)
    down = cm + ((0 - (cm >> 8)) >> 2)
    assert (up >> 8).max() <= 32766 and down.min() >= 0

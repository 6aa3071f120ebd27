"""Synthetic block generator of SURVEY.md 8(d) / BASELINE.md 3: block b has class
b mod 4: 0 zeros, 1 uniform bytes (splitmix64), 2 order-1 Markov "text" over 64
ASCII symbols, 3 periodic repeat of random data with period 16 + (b mod 4080)."""
import numpy as np

MASK = (1 << 64) - 1
SEED0 = 0x5A50415100000000


def _splitmix_stream(seed, n):
    """n successive splitmix64 outputs (vectorised)."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _markov_text_loop(use_fav, fav, uni):
    """The recurrence as SURVEY.md 8(d) states it, one symbol at a time (kept as the checker of the
    vectorised form below: tests/test_sharding_gloo.py::test_markov_generator_forms_agree)."""
    cur = 0
    uf = use_fav.tolist(); fv = fav.tolist(); un = uni.tolist()
    o = bytearray(len(uf))
    for i in range(len(uf)):
        cur = (cur * 7 + fv[i]) & 63 if uf[i] else un[i]
        o[i] = 32 + cur
    return np.frombuffer(bytes(o), dtype=np.uint8).copy()


def _markov_text(use_fav, fav, uni):
    """cur_i = (7 cur_{i-1} + fav_i) mod 64, or uni_i where use_fav_i is false.  7 is a unit mod 64
    (7 * 55 = 1), so with a_i = 55^i cur_i the recurrence is a prefix sum a_i = a_{i-1} + 55^i fav_i
    that restarts at every reset position: one cumsum for the whole block."""
    n = len(fav)
    idx = np.arange(n, dtype=np.int64)
    pw = np.ones(16, dtype=np.int64); ipw = np.ones(16, dtype=np.int64)
    for k in range(1, 16):
        pw[k] = pw[k - 1] * 7 % 64
        ipw[k] = ipw[k - 1] * 55 % 64
    p7 = pw[idx & 15]; i7 = ipw[idx & 15]            # 7^16 = 1 mod 64
    S = np.cumsum(np.where(use_fav, fav * i7, 0))
    r = np.maximum.accumulate(np.where(use_fav, -1, idx))     # last reset position <= i (-1: none yet)
    rr = np.maximum(r, 0)
    base = np.where(r >= 0, uni[rr] * i7[rr] - S[rr], 0)
    cur = (p7 * (base + S)) & 63
    return (32 + cur).astype(np.uint8)


def make_block(b, size=65536):
    cls = b % 4
    seed = (SEED0 + b) & MASK
    if cls == 0:
        return np.zeros(size, dtype=np.uint8)
    r = _splitmix_stream(seed, size)
    if cls == 1:
        return (r & np.uint64(255)).astype(np.uint8)
    if cls == 2:
        # order-1 Markov over 64 symbols: with prob 3/4 pick one of 4 favoured successors
        # (row*7+k) mod 64, else uniform; mapped to ASCII 32..95.
        fav = ((r >> np.uint64(8)) & np.uint64(3)).astype(np.int64)
        uni = ((r >> np.uint64(16)) & np.uint64(63)).astype(np.int64)
        use_fav = ((r & np.uint64(3)) != 0)
        return _markov_text(use_fav, fav, uni)
    per = 16 + (b % 4080)
    base = (r[:per] & np.uint64(255)).astype(np.uint8)
    return np.tile(base, size // per + 1)[:size].copy()


def make_blocks(nblocks, size=65536, start=0):
    """Returns a (nblocks, size) uint8 array: block start + i of SURVEY.md 8(d)'s generator."""
    out = np.empty((nblocks, size), dtype=np.uint8)
    for i in range(nblocks):
        out[i] = make_block(start + i, size)
    return out


_TEXT_CACHE = {}


def make_blocks_fast(nblocks, size=65536, start=0, text_pool=64):
    """Like make_blocks, but class-2 ("text") blocks are drawn from a pool of
    `text_pool` distinct generated blocks (block b uses pool entry (b//4) % pool) so
    that multi-GiB workloads build in seconds.  Stated in bench.py's config."""
    out = np.empty((nblocks, size), dtype=np.uint8)
    for i in range(nblocks):
        b = start + i
        if b % 4 == 2:
            key = (2 + 4 * ((b // 4) % text_pool), size)
            if key not in _TEXT_CACHE:
                _TEXT_CACHE[key] = make_block(key[0], size)
            out[i] = _TEXT_CACHE[key]
        else:
            out[i] = make_block(b, size)
    return out

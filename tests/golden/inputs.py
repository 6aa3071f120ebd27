"""Deterministic small inputs shared by make_golden.py and the tests."""


def lcg_bytes(n, seed=12345):
    x = seed
    out = bytearray()
    for _ in range(n):
        x = (x * 1103515245 + 12345) & 0x7FFFFFFF
        out.append((x >> 16) & 255)
    return bytes(out)


def text_bytes(n):
    s = b"It was the best of times, it was the worst of times, it was the age of wisdom. "
    return (s * (n // len(s) + 1))[:n]


# a header that exercises all nine component types.
C4B = bytes([4, 16, 0, 0, 9,
             1, 160, 2, 16, 255, 3, 16, 4, 16, 16, 5, 1, 2, 128, 6, 8, 3, 4, 24, 255,
             8, 16, 5, 7, 8, 0, 7, 24, 255, 9, 8, 7, 32, 255, 0,
             74, 18, 104, 95, 0] + [59, 112, 25] * 7 + [59, 112, 56, 0])

INPUTS = {
    "empty": b"",
    "a": b"a",
    "hello": b"Hello World!",
    "zeros256": bytes(256),
    "lcg4k": lcg_bytes(4096),
    "text2k": text_bytes(2048),
}

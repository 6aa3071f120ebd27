#!/usr/bin/env python3
"""tools/abort_ab.py LIB...: the round-3 teardown abort (rc 134, std::bad_variant_access) as an A/B over library builds.

For each library a child python does, through raw ctypes only: zpq_ctx_create, zpq_model_create_level(2),
zpq_block_create, one zpq_block_encode_segment, then zpq_ctx_destroy FIRST, zpq_block_destroy AFTER it, and exits 7.
Prints the child's return code per library (7 = clean; 134 / -6 = abort)."""
import subprocess
import sys

CHILD = r"""
import ctypes as C, sys
try:
    import torch  # its HIP runtime first, as the harness does
except Exception:
    pass
L = C.CDLL(sys.argv[1])
vp = C.c_void_p
L.zpq_ctx_create.argtypes = [C.c_int, vp]; L.zpq_ctx_destroy.argtypes = [vp]
L.zpq_model_create_level.argtypes = [C.c_int, vp]; L.zpq_model_destroy.argtypes = [vp]
L.zpq_block_create.argtypes = [vp, vp, vp]; L.zpq_block_destroy.argtypes = [vp]
L.zpq_block_encode_segment.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_uint32, vp, C.c_size_t, vp]
ctx, m, b = vp(), vp(), vp()
assert L.zpq_ctx_create(0, C.byref(ctx)) == 0
assert L.zpq_model_create_level(2, C.byref(m)) == 0
assert L.zpq_block_create(ctx, m, C.byref(b)) == 0
out = C.create_string_buffer(4096); n = C.c_size_t()
assert L.zpq_block_encode_segment(b, b"abc" * 100, 300, 1, out, 4096, C.byref(n)) == 0
L.zpq_ctx_destroy(ctx)
L.zpq_block_destroy(b)          # after its ctx
L.zpq_model_destroy(m)
sys.exit(7)
"""

for lib in sys.argv[1:]:
    r = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True, timeout=600)
    tail = [ln for ln in r.stderr.splitlines() if "amdgpu.ids" not in ln][-3:]
    print("%s: rc=%d %s" % (lib, r.returncode, " | ".join(tail)))

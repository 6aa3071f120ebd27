"""ctypes wrapper around oracle/libzpaq_oracle.so (the CPU checker).

Test infrastructure: imported only by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  Builds the library with oracle/Makefile if missing.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
_LIB = None


class TraceBit(C.Structure):
    _fields_ = [("p", C.c_int32), ("y", C.c_int32), ("low", C.c_uint32), ("high", C.c_uint32)]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # ZPQ_ORACLE_SO: another build of the same source (the sanitizer build, tests/test_oracle_sanitizers.py)
    so = os.path.join(ODIR, os.environ.get("ZPQ_ORACLE_SO", "libzpaq_oracle.so"))
    src = os.path.join(ODIR, "zpaq_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ODIR, "-s"] + (["asan"] if "asan" in os.path.basename(so) else []))
    L = C.CDLL(so)
    vp, sz, i64, u8p = C.c_void_p, C.c_size_t, C.c_int64, C.c_char_p
    L.zo_tables.argtypes = [vp] * 5
    L.zo_level_header.argtypes = [C.c_int, vp, C.c_int]
    L.zo_level_name.restype = C.c_char_p
    L.zo_scan_header.argtypes = [u8p, C.c_int, vp, vp, vp]
    L.zo_vm_new.restype = vp
    L.zo_vm_new.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.zo_vm_free.argtypes = [vp]
    L.zo_vm_run.argtypes = [vp, C.c_uint32]
    for f in ("zo_vm_reg", "zo_vm_h", "zo_vm_m", "zo_vm_r"):
        getattr(L, f).restype = C.c_uint32
    L.zo_vm_reg.argtypes = [vp, C.c_int]
    L.zo_vm_set_reg.argtypes = [vp, C.c_int, C.c_uint32]
    L.zo_vm_h.argtypes = [vp, C.c_uint32]
    L.zo_vm_m.argtypes = [vp, C.c_uint32]
    L.zo_vm_r.argtypes = [vp, C.c_int]
    L.zo_vm_hlen.argtypes = [vp]
    L.zo_vm_mlen.argtypes = [vp]
    L.zo_codec_new.restype = vp
    L.zo_codec_new.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    L.zo_codec_new_level.restype = vp
    L.zo_codec_new_level.argtypes = [C.c_int, vp]
    L.zo_codec_free.argtypes = [vp]
    L.zo_codec_ncomp.argtypes = [vp]
    L.zo_codec_state_bytes.restype = sz
    L.zo_codec_state_bytes.argtypes = [vp]
    L.zo_pred_reset.argtypes = [vp]
    L.zo_pred_predict.argtypes = [vp]
    L.zo_pred_update.argtypes = [vp, C.c_int]
    L.zo_pred_p.argtypes = [vp, C.c_int]
    L.zo_pred_h.restype = C.c_uint32
    L.zo_pred_h.argtypes = [vp, C.c_int]
    L.zo_pred_c8.restype = C.c_uint32
    L.zo_pred_c8.argtypes = [vp]
    L.zo_pred_hmap4.restype = C.c_uint32
    L.zo_pred_hmap4.argtypes = [vp]
    L.zo_encode_segment.restype = i64
    L.zo_encode_segment.argtypes = [vp, u8p, sz, C.c_uint, vp, sz, vp, sz]
    L.zo_decode_segment.restype = i64
    L.zo_decode_segment.argtypes = [vp, u8p, sz, vp, sz, vp, vp, sz]
    L.zo_encode_blocks.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_uint, vp, vp, vp, C.c_int]
    L.zo_decode_blocks.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    L.zo_sha1.argtypes = [u8p, sz, vp]
    L.zo_compress_archive.restype = i64
    L.zo_compress_archive.argtypes = [C.c_int, u8p, u8p, u8p, sz, C.c_int, vp, sz]
    L.zo_decompress_archive.restype = i64
    L.zo_decompress_archive.argtypes = [u8p, sz, vp, vp, sz, vp, sz, vp, sz, vp]
    _LIB = L
    return L


def level_header(level):
    buf = C.create_string_buffer(128)
    n = lib().zo_level_header(level, buf, 128)
    return buf.raw[:n]


def scan_header(hdr):
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().zo_scan_header(hdr, len(hdr), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


class Codec:
    """One ZPAQ block's model state (ZPAQL + Predictor) in the oracle."""

    def __init__(self, header, offsets=None):
        cend, hbegin, hend = offsets if offsets else scan_header(header)
        err = C.c_int()
        self.h = lib().zo_codec_new(header, len(header), cend, hbegin, hend, C.byref(err))
        if not self.h:
            raise ValueError("oracle refused header: %d" % err.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().zo_codec_free(self.h)
            self.h = None

    def encode(self, data, pp=True, ntrace=0, cap=None):
        cap = cap if cap is not None else len(data) * 17 + 4096
        out = C.create_string_buffer(cap)
        tr = (TraceBit * ntrace)() if ntrace else None
        n = lib().zo_encode_segment(self.h, data, len(data), 1 if pp else 0, out, cap, tr, ntrace)
        if n < 0:
            raise OverflowError("oracle encode overflow")
        coded = out.raw[:n]
        if ntrace:
            return coded, [(t.p, t.y, t.low, t.high) for t in tr]
        return coded

    def decode(self, coded, cap=None):
        cap = cap if cap is not None else 1 << 20
        out = C.create_string_buffer(cap)
        cons = C.c_size_t()
        n = lib().zo_decode_segment(self.h, coded, len(coded), out, cap, C.byref(cons), None, 0)
        if n < 0:
            raise OverflowError("oracle decode overflow")
        return out.raw[:n], cons.value


def encode_blocks(header, blocks, pp=True, nthreads=1, slack=None):
    """Fresh model per block, one segment per block. Returns list of coded bytes."""
    import numpy as np
    cend, hbegin, hend = scan_header(header)
    nb = len(blocks)
    in_off = np.zeros(nb + 1, dtype=np.uint64)
    in_off[1:] = np.cumsum([len(b) for b in blocks])
    src = np.frombuffer(b"".join(blocks) or b"\0", dtype=np.uint8)
    caps = [(len(b) * 17 + 4096) if slack is None else slack for b in blocks]
    out_off = np.zeros(nb + 1, dtype=np.uint64)
    out_off[1:] = np.cumsum(caps)
    out = np.zeros(int(out_off[-1]), dtype=np.uint8)
    out_len = np.zeros(nb, dtype=np.int64)
    rc = lib().zo_encode_blocks(header, len(header), cend, hbegin, hend, nb, src.ctypes.data,
                                in_off.ctypes.data, 1 if pp else 0, out.ctypes.data,
                                out_off.ctypes.data, out_len.ctypes.data, nthreads)
    assert rc == 0, rc
    return [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() for i in range(nb)]


def decode_blocks(header, coded, cap, nthreads=1):
    import numpy as np
    cend, hbegin, hend = scan_header(header)
    nb = len(coded)
    in_off = np.zeros(nb + 1, dtype=np.uint64)
    in_off[1:] = np.cumsum([len(b) for b in coded])
    src = np.frombuffer(b"".join(coded) or b"\0", dtype=np.uint8)
    out_off = np.arange(nb + 1, dtype=np.uint64) * np.uint64(cap)
    out = np.zeros(int(out_off[-1]), dtype=np.uint8)
    out_len = np.zeros(nb, dtype=np.int64)
    rc = lib().zo_decode_blocks(header, len(header), cend, hbegin, hend, nb, src.ctypes.data,
                                in_off.ctypes.data, out.ctypes.data, out_off.ctypes.data,
                                out_len.ctypes.data, nthreads)
    assert rc == 0, rc
    return [out[int(out_off[i]):int(out_off[i]) + int(out_len[i])].tobytes() for i in range(nb)]

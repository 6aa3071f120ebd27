"""ctypes binding over the C ABI of libzpaq_hip.so (include/zpaq_hip.h).

This is plumbing only: every compute call goes to the HIP library.  There is no
Python or CPU fallback -- if the shared library is missing, or no GPU is
present, the calls raise ZpqError.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FLAG_PP = 1
FLAG_GENERIC = 2
FLAG_LANES = 8
_LIB = None
# Handle lifetime on the Python side (the C side tolerates any destroy order as well, include/zpaq_hip.h): every live
# Context, Block and PinnedArray is tracked; Context.close() closes its blocks first; at interpreter exit everything
# still alive is closed in that order by ONE atexit hook (registered when the library is loaded, i.e. after torch's own
# hooks, so it runs before them and before any runtime is torn down) and from then on __del__ does nothing: a
# destructor that runs during interpreter finalisation must not call into HIP any more.
_LIVE_CTX = weakref.WeakSet()
_LIVE_PINNED = weakref.WeakSet()
_FINALIZING = False


def _shutdown():
    global _FINALIZING
    for ctx in list(_LIVE_CTX):
        try:
            ctx.close()
        except Exception:  # noqa: BLE001
            pass
    for arr in list(_LIVE_PINNED):
        try:
            arr.free()
        except Exception:  # noqa: BLE001
            pass
    _FINALIZING = True


class ZpqError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("%s: %s (%d)" % (what, status_string(code), code))


def lib_path():
    # ZPQ_LIB_PATH: another build of the same library (kernel experiments, tools/variants.sh)
    return os.environ.get("ZPQ_LIB_PATH") or os.path.join(HERE, "lib", "libzpaq_hip.so")


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise ZpqError(-1, "libzpaq_hip.so is not built (run __graft_entry__.build() or make -C zpaq-v_amd/csrc)")
    # PyTorch-ROCm ships its own HIP runtime; a process that maps this library (and with it the system's libamdhip64) BEFORE
    # torch ends up with two runtimes and the second one to initialise sees no device (measured: build() then smoke() in one
    # process).  Where torch is there -- it is on every box this harness runs on -- its runtime is mapped first, always.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001
        pass
    L = C.CDLL(p)
    vp, u8p, u32, i32, u64 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_int, C.c_uint64
    L.zpq_level_header.argtypes = [i32, vp, i32, vp, vp, vp, vp]
    L.zpq_scan_header.argtypes = [u8p, i32, vp, vp, vp]
    L.zpq_model_create.argtypes = [u8p, i32, i32, i32, i32, vp]
    L.zpq_model_create_level.argtypes = [i32, vp]
    L.zpq_model_destroy.argtypes = [vp]
    L.zpq_model_ncomp.argtypes = [vp]
    L.zpq_model_state_bytes.argtypes = [vp]
    L.zpq_model_state_bytes.restype = u64
    L.zpq_model_has_fast_path.argtypes = [vp]
    L.zpq_ctx_create.argtypes = [i32, vp]
    L.zpq_ctx_destroy.argtypes = [vp]
    L.zpq_ctx_sync.argtypes = [vp]
    L.zpq_ctx_stream.argtypes = [vp]
    L.zpq_ctx_stream.restype = vp
    L.zpq_ctx_set_state_budget.argtypes = [vp, u64]
    L.zpq_ctx_set_max_block_bytes.argtypes = [vp, u64]
    L.zpq_ctx_last_slots.argtypes = [vp]
    L.zpq_ctx_last_line_store.argtypes = [vp]
    L.zpq_ctx_last_line_store.restype = C.c_uint
    L.zpq_ctx_resident_capacity.argtypes = [vp, vp, u32]
    L.zpq_ctx_last_kernel_ms.argtypes = [vp]
    L.zpq_ctx_last_kernel_ms.restype = C.c_float
    L.zpq_ctx_last_kernel_name.argtypes = [vp]
    L.zpq_ctx_last_kernel_name.restype = C.c_char_p
    enc = [vp, vp, i32, vp, vp, u32, vp, vp, vp, vp]
    dec = [vp, vp, i32, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    L.zpq_encode_blocks.argtypes = enc
    L.zpq_encode_blocks_dev.argtypes = enc
    L.zpq_decode_blocks.argtypes = dec
    L.zpq_decode_blocks_dev.argtypes = dec
    L.zpq_host_alloc.argtypes = [C.c_size_t]
    L.zpq_host_alloc.restype = vp
    L.zpq_host_free.argtypes = [vp]
    L.zpq_encode_blocks_multi.argtypes = [vp, i32, vp, i32, vp, vp, u32, vp, vp, vp, vp]
    L.zpq_decode_blocks_multi.argtypes = [vp, i32, vp, i32, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    L.zpq_sha1_blocks.argtypes = [vp, i32, vp, vp, vp]
    L.zpq_sha1_blocks_dev.argtypes = [vp, i32, vp, vp, vp]
    L.zpq_sha1_ranges_dev.argtypes = [vp, i32, vp, vp, vp, vp]
    L.zpq_ctx_device.argtypes = [vp]
    L.zpq_gather_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.zpq_block_create.argtypes = [vp, vp, vp]
    L.zpq_block_destroy.argtypes = [vp]
    L.zpq_block_encode_segment.argtypes = [vp, u8p, C.c_size_t, u32, vp, C.c_size_t, vp]
    L.zpq_block_decode_segment.argtypes = [vp, u8p, C.c_size_t, u32, vp, C.c_size_t, vp, vp, vp, vp]
    L.zpq_tables.argtypes = [vp, vp]
    L.zpq_tables_ex.argtypes = [vp, vp, vp]
    L.zpq_debug_contexts.argtypes = [vp, vp, u8p, C.c_size_t, vp]
    L.zpq_debug_encode_trace.argtypes = [vp, vp, u8p, C.c_size_t, u32, vp, C.c_size_t, vp, vp, C.c_size_t]
    L.zpq_status_string.argtypes = [i32]
    L.zpq_status_string.restype = C.c_char_p
    L.zpq_version.restype = C.c_char_p
    _LIB = L
    atexit.register(_shutdown)
    return L


def status_string(code):
    try:
        return lib().zpq_status_string(code).decode()
    except Exception:
        return "?"


def _ck(rc, what):
    if rc != 0:
        raise ZpqError(rc, what)


def tables_ex():
    """dt[1024], dt2k[256], ns[1024] as the library builds them (zpq_tables_ex)."""
    dt = np.zeros(1024, dtype=np.int32)
    dt2k = np.zeros(256, dtype=np.int32)
    ns = np.zeros(1024, dtype=np.uint8)
    _ck(lib().zpq_tables_ex(dt.ctypes.data, dt2k.ctypes.data, ns.ctypes.data), "zpq_tables_ex")
    return dt, dt2k, ns


def level_header(level):
    """get_compression_level(level).hcomp (reference zpaq/levels.v:26-36)."""
    buf = C.create_string_buffer(256)
    n = C.c_int()
    _ck(lib().zpq_level_header(level, buf, 256, C.byref(n), None, None, None), "zpq_level_header")
    return buf.raw[:n.value]


def scan_header(hdr):
    """cend, hbegin, hend as Compressor.start_block derives them (compressor.v:96-145)."""
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    _ck(lib().zpq_scan_header(hdr, len(hdr), C.byref(a), C.byref(b), C.byref(c)), "zpq_scan_header")
    return a.value, b.value, c.value


class Model:
    def __init__(self, header=None, level=None, offsets=None):
        self.h = C.c_void_p()
        if header is None:
            header = level_header(level)
        self.header = bytes(header)
        self.offsets = tuple(offsets) if offsets else scan_header(self.header)
        _ck(lib().zpq_model_create(self.header, len(self.header), *self.offsets, C.byref(self.h)),
            "zpq_model_create")

    def close(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.zpq_model_destroy(self.h)                 # (a Block built on the model keeps it alive: it is refcounted)
            self.h = None

    def __del__(self):
        if not _FINALIZING:
            self.close()

    @property
    def ncomp(self):
        return lib().zpq_model_ncomp(self.h)

    @property
    def state_bytes(self):
        return lib().zpq_model_state_bytes(self.h)

    @property
    def has_fast_path(self):
        return bool(lib().zpq_model_has_fast_path(self.h))


class PinnedArray:
    """A numpy view over page-locked, device-visible host memory (zpq_host_alloc): what a front end should hand
    to the host-pointer batch calls so that transfers are plain DMA and outputs are packed by the GPU."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().zpq_host_alloc(max(self.nbytes, 1))
        if not self.ptr:
            raise ZpqError(-6, "zpq_host_alloc")
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr))[:self.nbytes]
        _LIVE_PINNED.add(self)

    def free(self):
        if getattr(self, "ptr", None) and _LIB is not None:
            self.array = None
            _LIB.zpq_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        if not _FINALIZING:
            self.free()


def _offsets(lengths):
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    if len(lengths):
        off[1:] = np.cumsum(np.asarray(lengths, dtype=np.uint64))
    return off


class Context:
    """One GPU (zpq_ctx).  Batch helpers take/return Python bytes for tests; the
    *_dev forms take raw device pointers (e.g. torch tensors' data_ptr())."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        self._children = weakref.WeakSet()                 # Blocks (and front-end handles) living on this ctx
        _ck(lib().zpq_ctx_create(device, C.byref(self.h)), "zpq_ctx_create")
        _LIVE_CTX.add(self)

    def close(self):
        """Destroy the ctx; whatever still lives on it is closed first."""
        for child in list(getattr(self, "_children", ())):
            try:
                child.close()
            except Exception:  # noqa: BLE001
                pass
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.zpq_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        if not _FINALIZING:
            self.close()

    def sync(self):
        _ck(lib().zpq_ctx_sync(self.h), "zpq_ctx_sync")

    @property
    def stream(self):
        return lib().zpq_ctx_stream(self.h)

    @property
    def last_kernel_ms(self):
        return lib().zpq_ctx_last_kernel_ms(self.h)

    @property
    def last_kernel_name(self):
        return lib().zpq_ctx_last_kernel_name(self.h).decode()

    @property
    def last_slots(self):
        return lib().zpq_ctx_last_slots(self.h)

    @property
    def last_line_store(self):
        """Lines per hash table of the compact line store the last launch ran with (0 = dense tables)."""
        return lib().zpq_ctx_last_line_store(self.h)

    def resident_capacity(self, model, flags=FLAG_PP):
        """Blocks of `model` one launch keeps resident on this GPU (zpq_ctx_resident_capacity)."""
        n = lib().zpq_ctx_resident_capacity(self.h, model.h, flags)
        if n < 0:
            raise ZpqError(n, "zpq_ctx_resident_capacity")
        return n

    def encode_blocks(self, model, blocks, flags=FLAG_PP, cap=None):
        nb = len(blocks)
        in_off = _offsets([len(b) for b in blocks])
        src = np.frombuffer(b"".join(blocks) + b"\0", dtype=np.uint8)
        caps = [cap if cap is not None else len(b) * 17 + 4096 for b in blocks]
        out_off = _offsets(caps)
        out = np.zeros(int(out_off[-1]) + 1, dtype=np.uint8)
        out_len = np.zeros(nb, dtype=np.uint32)
        status = np.zeros(nb, dtype=np.int32)
        _ck(lib().zpq_encode_blocks(self.h, model.h, nb, src.ctypes.data, in_off.ctypes.data, flags,
                                    out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
                                    status.ctypes.data), "zpq_encode_blocks")
        res = []
        for i in range(nb):
            o = int(out_off[i])
            res.append(out[o:o + min(int(out_len[i]), caps[i])].tobytes())
        return res, status, out_len

    def sha1_blocks(self, blocks):
        """SHA-1 of each byte string, one message per GPU lane (zpq_sha1_blocks; sha1.v:6-146)."""
        nb = len(blocks)
        in_off = _offsets([len(b) for b in blocks])
        src = np.frombuffer(b"".join(blocks) + b"\0", dtype=np.uint8)
        out = np.zeros(20 * nb + 1, dtype=np.uint8)
        _ck(lib().zpq_sha1_blocks(self.h, nb, src.ctypes.data, in_off.ctypes.data, out.ctypes.data), "zpq_sha1_blocks")
        return [out[20 * i:20 * i + 20].tobytes() for i in range(nb)]

    def sha1_blocks_dev(self, nblocks, d_in, d_in_off, d_out20):
        _ck(lib().zpq_sha1_blocks_dev(self.h, nblocks, d_in, d_in_off, d_out20), "zpq_sha1_blocks_dev")

    def decode_blocks(self, model, coded, cap, flags=FLAG_PP):
        nb = len(coded)
        in_off = _offsets([len(b) for b in coded])
        src = np.frombuffer(b"".join(coded) + b"\0", dtype=np.uint8)
        out_off = _offsets([cap] * nb)
        out = np.zeros(int(out_off[-1]) + 1, dtype=np.uint8)
        out_len = np.zeros(nb, dtype=np.uint32)
        consumed = np.zeros(nb, dtype=np.uint32)
        code = np.zeros(nb, dtype=np.uint32)
        first = np.zeros(nb, dtype=np.uint32)
        status = np.zeros(nb, dtype=np.int32)
        _ck(lib().zpq_decode_blocks(self.h, model.h, nb, src.ctypes.data, in_off.ctypes.data, flags,
                                    out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
                                    consumed.ctypes.data, code.ctypes.data, first.ctypes.data,
                                    status.ctypes.data), "zpq_decode_blocks")
        res = []
        for i in range(nb):
            o = int(out_off[i])
            res.append(out[o:o + min(int(out_len[i]), cap)].tobytes())
        return res, status, consumed, code, first

    def encode_blocks_dev(self, model, nblocks, d_in, d_in_off, flags, d_out, d_out_off, d_out_len, d_status):
        _ck(lib().zpq_encode_blocks_dev(self.h, model.h, nblocks, d_in, d_in_off, flags, d_out, d_out_off,
                                        d_out_len, d_status), "zpq_encode_blocks_dev")

    def decode_blocks_dev(self, model, nblocks, d_in, d_in_off, flags, d_out, d_out_off, d_out_len,
                          d_consumed, d_code, d_first, d_status):
        _ck(lib().zpq_decode_blocks_dev(self.h, model.h, nblocks, d_in, d_in_off, flags, d_out, d_out_off,
                                        d_out_len, d_consumed, d_code, d_first, d_status),
            "zpq_decode_blocks_dev")

    def debug_contexts(self, model, data):
        n = model.ncomp
        out = np.zeros(max(1, len(data) * n), dtype=np.uint32)
        _ck(lib().zpq_debug_contexts(self.h, model.h, data, len(data), out.ctypes.data), "zpq_debug_contexts")
        return out[:len(data) * n].reshape(len(data), n)

    def debug_encode_trace(self, model, data, ntrace, flags=FLAG_PP):
        cap = len(data) * 17 + 4096
        out = C.create_string_buffer(cap)
        olen = C.c_size_t()
        tr = np.zeros(ntrace, dtype=np.int32)
        _ck(lib().zpq_debug_encode_trace(self.h, model.h, data, len(data), flags, out, cap, C.byref(olen),
                                         tr.ctypes.data, ntrace), "zpq_debug_encode_trace")
        return out.raw[:olen.value], tr


class Block:
    """One ZPAQ block whose model state persists across segments (zpq_block)."""

    def __init__(self, ctx, model):
        self.ctx, self.model = ctx, model
        self.h = C.c_void_p()
        _ck(lib().zpq_block_create(ctx.h, model.h, C.byref(self.h)), "zpq_block_create")
        ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.zpq_block_destroy(self.h)
            self.h = None

    def __del__(self):
        if not _FINALIZING:
            self.close()

    def encode_segment(self, data, flags=FLAG_PP, cap=None):
        cap = cap if cap is not None else len(data) * 17 + 4096
        out = C.create_string_buffer(cap)
        olen = C.c_size_t()
        _ck(lib().zpq_block_encode_segment(self.h, data, len(data), flags, out, cap, C.byref(olen)),
            "zpq_block_encode_segment")
        return out.raw[:olen.value]

    def decode_segment(self, coded, cap, flags=FLAG_PP):
        out = C.create_string_buffer(max(cap, 1))
        olen, cons = C.c_size_t(), C.c_size_t()
        code, first = C.c_uint32(), C.c_uint32()
        _ck(lib().zpq_block_decode_segment(self.h, coded, len(coded), flags, out, cap, C.byref(olen),
                                           C.byref(cons), C.byref(code), C.byref(first)),
            "zpq_block_decode_segment")
        return out.raw[:olen.value], cons.value, code.value, first.value

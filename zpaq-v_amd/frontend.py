"""Python handles on the C++ front end (include/zpaq_frontend.hpp): Compressor and
Decompresser with the reference's method names and call order
(zpaq/compressor.v:33-418, zpaq/decompressor.v:187-640).  All work happens in
libzpaq_hip.so; these classes only marshal arguments."""
import ctypes as C

from . import binding as B


def _lib():
    L = B.lib()
    if getattr(L, "_zpqf_ready", False):
        return L
    vp, u8p = C.c_void_p, C.c_char_p
    L.zpqf_compressor_new.restype = vp
    L.zpqf_compressor_new.argtypes = [vp]
    L.zpqf_compressor_free.argtypes = [vp]
    L.zpqf_compressor_set_input.argtypes = [vp, u8p, C.c_size_t]
    L.zpqf_compressor_start_block.argtypes = [vp, C.c_int]
    L.zpqf_compressor_start_block_hcomp.argtypes = [vp, u8p, C.c_size_t]
    L.zpqf_compressor_start_segment.argtypes = [vp, u8p, u8p]
    L.zpqf_compressor_compress.argtypes = [vp, C.c_int]
    L.zpqf_compressor_end_segment.argtypes = [vp]
    L.zpqf_compressor_end_block.argtypes = [vp]
    L.zpqf_compressor_last_error.argtypes = [vp]
    L.zpqf_compressor_output.restype = C.c_size_t
    L.zpqf_compressor_output.argtypes = [vp, vp]
    L.zpqf_compressor_sha1.argtypes = [vp, vp]
    L.zpqf_decompresser_new.restype = vp
    L.zpqf_decompresser_new.argtypes = [vp]
    L.zpqf_decompresser_free.argtypes = [vp]
    L.zpqf_decompresser_set_input.argtypes = [vp, u8p, C.c_size_t]
    L.zpqf_decompresser_find_block.argtypes = [vp]
    L.zpqf_decompresser_find_filename.argtypes = [vp]
    L.zpqf_decompresser_filename.restype = C.c_size_t
    L.zpqf_decompresser_filename.argtypes = [vp, vp, C.c_size_t]
    L.zpqf_decompresser_comment.restype = C.c_size_t
    L.zpqf_decompresser_comment.argtypes = [vp, vp, C.c_size_t]
    L.zpqf_decompresser_decompress.argtypes = [vp, C.c_int]
    L.zpqf_decompresser_read_segment_end.argtypes = [vp]
    L.zpqf_decompresser_last_error.argtypes = [vp]
    L.zpqf_decompresser_output.restype = C.c_size_t
    L.zpqf_decompresser_output.argtypes = [vp, vp]
    L.zpqf_decompresser_sha1.argtypes = [vp, vp]
    L.zpqf_archive_add.restype = vp
    L.zpqf_archive_add.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.zpqf_archive_add_fragmented.restype = vp
    L.zpqf_archive_add_fragmented.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, C.c_uint64, vp]
    L.zpqf_archive_add_multi.restype = vp
    L.zpqf_archive_add_multi.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_uint64, vp]
    L.zpqf_archive_extract_multi.restype = vp
    L.zpqf_archive_extract_multi.argtypes = [vp, C.c_int, u8p, C.c_size_t, C.c_int, vp]
    L.zpqf_archive_bytes.restype = C.c_size_t
    L.zpqf_archive_bytes.argtypes = [vp, vp]
    L.zpqf_archive_extract.restype = vp
    L.zpqf_archive_extract.argtypes = [vp, u8p, C.c_size_t, C.c_int, vp]
    L.zpqf_archive_nfiles.argtypes = [vp]
    for fn in ("name", "comment"):
        getattr(L, "zpqf_archive_" + fn).restype = C.c_char_p
        getattr(L, "zpqf_archive_" + fn).argtypes = [vp, C.c_int]
    L.zpqf_archive_size.restype = C.c_uint64
    L.zpqf_archive_size.argtypes = [vp, C.c_int]
    L.zpqf_archive_sha1_ok.argtypes = [vp, C.c_int]
    L.zpqf_archive_status.argtypes = [vp, C.c_int]
    L.zpqf_archive_data.restype = vp
    L.zpqf_archive_data.argtypes = [vp, C.c_int]
    L.zpqf_archive_free.argtypes = [vp]
    L._zpqf_ready = True
    return L


class Compressor:
    """Compressor.new() (compressor.v:33-46).  `ctx` may be None for store mode (level 0)."""

    def __init__(self, ctx=None):
        self._L = _lib()
        self.h = self._L.zpqf_compressor_new(ctx.h if ctx is not None else None)
        if not self.h:
            raise B.ZpqError(-9, "zpqf_compressor_new")
        self._keep = None
        if ctx is not None:
            ctx._children.add(self)                        # closed before the ctx (binding.Context.close)

    def close(self):
        if getattr(self, "h", None) and B._LIB is not None:
            self._L.zpqf_compressor_free(self.h)
            self.h = None

    def __del__(self):
        if not B._FINALIZING:
            self.close()

    def set_input(self, data):
        self._keep = bytes(data)
        self._L.zpqf_compressor_set_input(self.h, self._keep, len(self._keep))

    def start_block(self, level):
        self._L.zpqf_compressor_start_block(self.h, level)

    def start_block_hcomp(self, hcomp):
        self._L.zpqf_compressor_start_block_hcomp(self.h, bytes(hcomp), len(hcomp))

    def start_segment(self, filename, comment):
        self._L.zpqf_compressor_start_segment(self.h, filename.encode(), comment.encode())

    def compress(self, n):
        return bool(self._L.zpqf_compressor_compress(self.h, n))

    def end_segment(self):
        self._L.zpqf_compressor_end_segment(self.h)

    def end_block(self):
        self._L.zpqf_compressor_end_block(self.h)

    @property
    def last_error(self):
        return self._L.zpqf_compressor_last_error(self.h)

    def output_bytes(self):
        p = C.c_void_p()
        n = self._L.zpqf_compressor_output(self.h, C.byref(p))
        return C.string_at(p, n) if n else b""

    def get_sha1(self):
        out = C.create_string_buffer(20)
        self._L.zpqf_compressor_sha1(self.h, out)
        return out.raw


class Decompresser:
    """Decompresser.new() (decompressor.v:187-200)."""

    def __init__(self, ctx=None):
        self._L = _lib()
        self.h = self._L.zpqf_decompresser_new(ctx.h if ctx is not None else None)
        if not self.h:
            raise B.ZpqError(-9, "zpqf_decompresser_new")
        self._keep = None
        if ctx is not None:
            ctx._children.add(self)

    def close(self):
        if getattr(self, "h", None) and B._LIB is not None:
            self._L.zpqf_decompresser_free(self.h)
            self.h = None

    def __del__(self):
        if not B._FINALIZING:
            self.close()

    def set_input(self, data):
        self._keep = bytes(data)
        self._L.zpqf_decompresser_set_input(self.h, self._keep, len(self._keep))

    def find_block(self):
        return bool(self._L.zpqf_decompresser_find_block(self.h))

    def find_filename(self):
        return bool(self._L.zpqf_decompresser_find_filename(self.h))

    def get_filename(self):
        buf = C.create_string_buffer(4096)
        self._L.zpqf_decompresser_filename(self.h, buf, 4096)
        return buf.value.decode(errors="replace")

    def get_comment(self):
        buf = C.create_string_buffer(4096)
        self._L.zpqf_decompresser_comment(self.h, buf, 4096)
        return buf.value.decode(errors="replace")

    def decompress(self, n):
        return bool(self._L.zpqf_decompresser_decompress(self.h, n))

    def read_segment_end(self):
        self._L.zpqf_decompresser_read_segment_end(self.h)

    @property
    def last_error(self):
        return self._L.zpqf_decompresser_last_error(self.h)

    def output_bytes(self):
        p = C.c_void_p()
        n = self._L.zpqf_decompresser_output(self.h, C.byref(p))
        return C.string_at(p, n) if n else b""

    def get_sha1(self):
        out = C.create_string_buffer(20)
        self._L.zpqf_decompresser_sha1(self.h, out)
        return out.raw


def archive_add(ctx, level, files, fragment_bytes=0):
    """zpaq::archive_add: the reference CLI's add loop (cmd/main.v:283-311) for a list of
    (name, comment, data) as ONE GPU batch.  Returns the archive bytes.  fragment_bytes > 0 cuts
    longer files into blocks of that size (continuation blocks carry an empty name)."""
    L = _lib()
    n = len(files)
    names = (C.c_char_p * n)(*[f[0].encode() for f in files])
    comments = (C.c_char_p * n)(*[f[1].encode() for f in files])
    keep = [bytes(f[2]) for f in files]
    data = (C.c_char_p * n)(*keep)
    lens = (C.c_uint64 * n)(*[len(k) for k in keep])
    rc = C.c_int(0)
    if isinstance(ctx, (list, tuple)):                       # several GPUs: block b -> ctx[b mod G]
        arr = (C.c_void_p * len(ctx))(*[c.h for c in ctx])
        h = L.zpqf_archive_add_multi(arr, len(ctx), level, n, names, comments, data, lens, fragment_bytes, C.byref(rc))
    else:
        h = L.zpqf_archive_add_fragmented(ctx.h if ctx is not None else None, level, n, names, comments, data, lens,
                                          fragment_bytes, C.byref(rc))
    try:
        if rc.value != 0 or not h:
            raise B.ZpqError(rc.value or -9, "archive_add")
        p = C.c_void_p()
        k = L.zpqf_archive_bytes(h, C.byref(p))
        return C.string_at(p, k) if k else b""
    finally:
        L.zpqf_archive_free(h)


def archive_extract(ctx, archive, want_data=True, join_unnamed=False):
    """zpaq::archive_extract: every segment of every block in archive order
    (cmd/main.v:342-380,440-465).  Returns dicts name/comment/size/sha1_ok/status/data."""
    L = _lib()
    archive = bytes(archive)
    rc = C.c_int(0)
    mode = (1 if want_data else 0) | (2 if join_unnamed else 0)
    if isinstance(ctx, (list, tuple)):
        arr = (C.c_void_p * len(ctx))(*[c.h for c in ctx])
        h = L.zpqf_archive_extract_multi(arr, len(ctx), archive, len(archive), mode, C.byref(rc))
    else:
        h = L.zpqf_archive_extract(ctx.h if ctx is not None else None, archive, len(archive), mode, C.byref(rc))
    try:
        if rc.value != 0 or not h:
            raise B.ZpqError(rc.value or -9, "archive_extract")
        out = []
        for i in range(L.zpqf_archive_nfiles(h)):
            size = L.zpqf_archive_size(h, i)
            d = C.string_at(L.zpqf_archive_data(h, i), size) if (want_data and size) else b""
            out.append(dict(name=L.zpqf_archive_name(h, i).decode(errors="replace"),
                            comment=L.zpqf_archive_comment(h, i).decode(errors="replace"), size=size,
                            sha1_ok=bool(L.zpqf_archive_sha1_ok(h, i)), status=L.zpqf_archive_status(h, i), data=d))
        return out
    finally:
        L.zpqf_archive_free(h)

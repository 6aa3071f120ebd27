"""zpaq-v_amd: MI355X-native ZPAQ block codec (the context-mixing hot path of
dy-tea/zpaq-v behind its Compressor/Decompresser API).

The package directory name contains '-' (it is fixed by the project layout), so
import it through `load()` in `__graft_entry__.py` / tests/conftest.py, which
registers it as module ``zpaq_v_amd``.
"""
from .binding import (  # noqa: F401
    FLAG_GENERIC, FLAG_LANES, FLAG_PP, Block, Context, Model, PinnedArray, ZpqError, level_header, lib, lib_path,
    scan_header, status_string,
)
from .frontend import Compressor, Decompresser, archive_add, archive_extract  # noqa: F401,E402

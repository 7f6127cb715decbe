"""libflate_amd — MI355X-native DEFLATE hot path behind libflate's own API shape.

Modules mirror the reference crate layout: `deflate`, `zlib`, `gzip`, `lz77`.
The work runs in hand-written HIP kernels reached through the C ABI in include/lfx.h; there is no
CPU fallback.
"""
from . import _ffi  # noqa: F401
from .context import Context, default_context  # noqa: F401
from . import deflate, gzip, lz77, non_blocking, zlib  # noqa: F401

__all__ = ["Context", "default_context", "deflate", "zlib", "gzip", "lz77", "non_blocking"]

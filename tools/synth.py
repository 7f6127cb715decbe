"""ctypes binding of tools/synth.c — deterministic TEXT / LOWENT generators."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblfx_synth.so")
_lib = None

SEED_BASE = 0x5EED0000


def build(force=False):
    src = os.path.join(_HERE, "synth.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(src) > os.path.getmtime(_SO):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src, "-lm"])
    return _SO


def _get():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        for f in (_lib.lfx_synth_text, _lib.lfx_synth_lowent):
            f.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    return _lib


def text(n, seed=SEED_BASE + 2):
    out = np.empty(n, dtype=np.uint8)
    _get().lfx_synth_text(out.ctypes.data, n, seed)
    return out


def lowent(n, seed=SEED_BASE + 5):
    out = np.empty(n, dtype=np.uint8)
    _get().lfx_synth_lowent(out.ctypes.data, n, seed)
    return out

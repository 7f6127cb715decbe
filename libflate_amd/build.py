"""Builds the in-tree native library libflate_amd/liblfx.so (C ABI, include/lfx.h) for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "liblfx.so")
SOURCES = ["lfx_encode_kernels.hip", "lfx_match7.hip", "lfx_match5.hip", "lfx_parse2.hip", "lfx_decode_kernels.hip", "lfx_inflate_fast.hip", "lfx_api.cpp", "lfx_decode.cpp", "lfx_sharded.cpp", "lfx_hostio.cpp"]
HEADERS = ["lfx_common.h", "lfx_container.h", "lfx_device.h", "lfx_decode.h", "lfx_huff.h", "lfx_plan.h", "lfx_ctx.h", "lfx_abi_guard.h", "lfx_hostio.h",
           os.path.join("..", "..", "include", "lfx.h"), os.path.join("..", "..", "include", "lfx_testhooks.h")]


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, defs=(), so=None, tag=""):
    """defs / so / tag: development builds with extra -D macros into a library of another name (tools/exp: kernel variants
    side by side in one GPU call, selected by LFX_SO); the product build takes none of them."""
    so = so or SO
    if not force and not defs and not _stale() and os.path.exists(so):
        return so
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.rsplit(".", 1)[0] + tag + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall",
               "-Wno-unused-result", "-c", os.path.join(CSRC, src), "-o", obj] + ["-D" + d for d in defs]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

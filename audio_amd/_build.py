"""Build libaudio_amd.so in-tree with hipcc for gfx950:  python -m audio_amd._build [--force] [-v] [--lab]

Staleness is decided by CONTENT, not by modification times (a fresh clone's checkout order must not
decide whether the library is rebuilt): the SHA-256 of every source under csrc/, of include/audio_amd.h and of the compile
command is stored beside each library (``<lib>.buildhash``); the library is current when the stored value equals the one
computed now.  ``--lab`` builds ``libaudio_amd_lab.so`` = the same sources with ``-DAAMD_LAB``: the tools-only kernel
instantiations and their environment switches (AAMD_LFW_LAB, AAMD_RSM_LAB, AAMD_MFCC_LAB ...) exist in that library only;
``AAMD_USE_LAB_LIB=1`` makes ``audio_amd._lib`` load it (tools/ scripts; never set by the product).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libaudio_amd.so")
LAB_OUT = os.path.join(HERE, "lib", "libaudio_amd_lab.so")
SOURCES = ["c_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
         # packed fp32 VALU ops run at half rate on gfx950 (no gain) and cost v_mov shuffles
         "-fno-slp-vectorize"]
HEADER = os.path.join(os.path.dirname(HERE), "include", "audio_amd.h")


def _tmp_suffix() -> str:
    """Per-process temporary name: concurrent builders (pytest-xdist workers) each write their own file and rename it into
    place atomically instead of writing into one shared .tmp."""
    return ".tmp.%d" % os.getpid()


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _digest(files, extra) -> str:
    h = hashlib.sha256()
    for path in sorted(files):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for e in extra:
        h.update(str(e).encode() + b"\0")
    return h.hexdigest()


def _stored(out: str):
    try:
        with open(out + ".buildhash", encoding="ascii") as f:
            return f.read().strip()
    except OSError:
        return None


def _store(out: str, digest: str) -> None:
    with open(out + ".buildhash" + _tmp_suffix(), "w", encoding="ascii") as f:
        f.write(digest + "\n")
    os.replace(out + ".buildhash" + _tmp_suffix(), out + ".buildhash")


def _kernel_sources():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != "torch_shim.cpp" and not f.startswith(".")] + [HEADER]


def _extra_flags():
    # AAMD_EXTRA_HIPCC_FLAGS: experiment builds only (e.g. -DAAMD_M400_POOLS=1 for tools/bench_pool_ab.py); never set by the product
    return os.environ.get("AAMD_EXTRA_HIPCC_FLAGS", "").split()


def source_digest(lab: bool = False) -> str:
    return _digest(_kernel_sources(), FLAGS + (["-DAAMD_LAB"] if lab else []) + _extra_flags())


SHIM_OUT = os.path.join(HERE, "lib", "libaudio_amd_torch.so")
SHIM_SRC = os.path.join(CSRC, "torch_shim.cpp")


def _shim_digest() -> str:
    import torch
    # the shim links against libaudio_amd.so by name only (the C ABI is the header's): its inputs are its own source, the header
    # and the torch it was compiled against
    return _digest([SHIM_SRC, HEADER], ["shim-v1", torch.__version__])


def shim_stale() -> bool:
    return not os.path.exists(SHIM_OUT) or _stored(SHIM_OUT) != _shim_digest()


def build_shim(force: bool = False, verbose: bool = False) -> str:
    """libaudio_amd_torch.so: the LibTorch-stable-ABI shim (host code only: g++ against the torch headers), linked
    against libaudio_amd.so through an $ORIGIN rpath so the pair travels together."""
    if not force and not shim_stale():
        return SHIM_OUT
    import torch
    tp = os.path.dirname(torch.__file__)
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-DUSE_ROCM",
           "-I", os.path.join(os.path.dirname(HERE), "include"), "-I", os.path.join(tp, "include"),
           SHIM_SRC, "-o", SHIM_OUT + _tmp_suffix(), "-L", os.path.dirname(OUT), "-laudio_amd",
           "-L", os.path.join(tp, "lib"), "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(tp, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(SHIM_OUT + _tmp_suffix(), SHIM_OUT)
    _store(SHIM_OUT, _shim_digest())
    return SHIM_OUT


def stale(lab: bool = False) -> bool:
    out = LAB_OUT if lab else OUT
    return not os.path.exists(out) or _stored(out) != source_digest(lab)


def build(force: bool = False, verbose: bool = False, lab: bool = False) -> str:
    out = LAB_OUT if lab else OUT
    if not force and not stale(lab):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    digest = source_digest(lab)                     # of what is compiled now (a source edited during the build shows up next time)
    cmd = [hipcc()] + FLAGS + (["-DAAMD_LAB"] if lab else []) + _extra_flags() + \
        [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out + _tmp_suffix()]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out + _tmp_suffix(), out)
    _store(out, digest)
    return out


if __name__ == "__main__":
    lab_ = "--lab" in sys.argv
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, lab=lab_))
    if not lab_:
        print(build_shim(force="--force" in sys.argv, verbose="-v" in sys.argv))

"""Build libaudio_amd.so in-tree with hipcc for gfx950:  python -m audio_amd._build"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libaudio_amd.so")
SOURCES = ["c_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
         # packed fp32 VALU ops run at half rate on gfx950 (no gain) and cost v_mov shuffles
         "-fno-slp-vectorize"]


def _tmp_suffix() -> str:
    """Per-process temporary name: concurrent builders (pytest-xdist workers) each write their own file and rename it into
    place atomically instead of writing into one shared .tmp."""
    return ".tmp.%d" % os.getpid()


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


SHIM_OUT = os.path.join(HERE, "lib", "libaudio_amd_torch.so")
SHIM_SRC = os.path.join(CSRC, "torch_shim.cpp")


def shim_stale() -> bool:
    if not os.path.exists(SHIM_OUT):
        return True
    t = os.path.getmtime(SHIM_OUT)
    deps = [SHIM_SRC, os.path.join(os.path.dirname(HERE), "include", "audio_amd.h"), os.path.abspath(__file__), OUT]
    return any(os.path.getmtime(d) > t for d in deps)


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
    return SHIM_OUT


def stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f != "torch_shim.cpp"]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "audio_amd.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # AAMD_EXTRA_HIPCC_FLAGS: experiment builds only (e.g. -DAAMD_M400_POOLS=1 for tools/bench_pool_ab.py); never set by the product
    extra = os.environ.get("AAMD_EXTRA_HIPCC_FLAGS", "").split()
    cmd = [hipcc()] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT + _tmp_suffix()]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(OUT + _tmp_suffix(), OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
    print(build_shim(force="--force" in sys.argv, verbose="-v" in sys.argv))

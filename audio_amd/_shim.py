"""Loader of libaudio_amd_torch.so -- the dispatcher-level boundary (csrc/torch_shim.cpp).

Same mechanism as the reference's native extension: a LibTorch-stable-ABI library whose static initialisers register
boxed kernels on the CUDA dispatch key, loaded with ``torch.ops.load_library``
(reference: src/torchaudio/_extension/utils.py:50-56, src/libtorchaudio/lfilter.cpp:118-138).

  torch.ops.aamd.{spectrogram, mel_spectrogram, mel_spectrogram_db, mfcc_dct, resample, lfilter, fftconvolve}
  torch.ops.torchaudio._lfilter_core_loop      (CUDA key; the reference's own schema -- see ensure_torchaudio_op)

There is no CPU-key kernel: CPU tensors raise NotImplementedError from the dispatcher.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SHIM_PATH = os.path.join(_HERE, "lib", "libaudio_amd_torch.so")

OPS = ("spectrogram", "mel_spectrogram", "mel_spectrogram_db", "mfcc_dct", "resample", "lfilter", "fftconvolve")

_lock = threading.Lock()
_handle = None
_torchaudio_done = False


def load():
    """Load the shim once (registers torch.ops.aamd.*); raises loudly if it has not been built."""
    global _handle
    if _handle is not None:
        return _handle
    with _lock:
        if _handle is None:
            if not os.path.exists(SHIM_PATH):
                raise RuntimeError(f"audio_amd: {SHIM_PATH} not found. Build it first (python -m audio_amd._build).")
            from . import _lib
            _lib.lib()                                   # libaudio_amd.so first: the shim links against it
            torch.ops.load_library(SHIM_PATH)
            h = C.CDLL(SHIM_PATH)
            h.aamd_torch_shim_abi.restype = C.c_int
            if h.aamd_torch_shim_abi() != 3:
                raise RuntimeError("audio_amd: ABI version mismatch between the torch shim and include/audio_amd.h")
            _handle = h
    return _handle


def available() -> bool:
    return os.path.exists(SHIM_PATH)


def ensure_torchaudio_op() -> None:
    """Make ``torch.ops.torchaudio._lfilter_core_loop`` dispatch to the MI355X kernel for ROCm tensors.

    This is the single native entry point of the unmodified reference (filtering.py:994).  If libtorchaudio is loaded
    it has defined the schema already and only the CUDA-key kernel is added; otherwise the schema fragment is defined
    here with the reference's exact signature (lfilter.cpp:119-123)."""
    global _torchaudio_done
    h = load()
    with _lock:
        if _torchaudio_done:
            return
        try:
            torch._C._get_schema("torchaudio::_lfilter_core_loop", "")
            have_schema = True
        except RuntimeError:
            have_schema = False
        if not have_schema and h.aamd_torch_define_torchaudio_schema() != 0:
            raise RuntimeError("audio_amd: could not define torchaudio::_lfilter_core_loop")
        if torch._C._dispatch_has_kernel_for_dispatch_key("torchaudio::_lfilter_core_loop", "CUDA"):
            raise RuntimeError("audio_amd: torchaudio::_lfilter_core_loop already has a CUDA kernel (libtorchaudio built "
                               "with USE_CUDA); refusing to shadow it")
        if h.aamd_torch_register_torchaudio_cuda() != 0:
            raise RuntimeError("audio_amd: could not register the CUDA kernel of torchaudio::_lfilter_core_loop")
        _torchaudio_done = True

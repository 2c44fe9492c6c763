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
            if h.aamd_torch_shim_abi() != 4:
                raise RuntimeError("audio_amd: ABI version mismatch between the torch shim and include/audio_amd.h")
            _register_fakes()
            _handle = h
    return _handle


def _register_fakes() -> None:
    """Fake (Meta) kernels of the compiled ``aamd::*`` ops -- shapes exactly as csrc/torch_shim.cpp allocates them -- so
    that FakeTensor tracing (torch.compile, torch.export, make_fx) can see through the default route, not only through
    the module-level ``audio_amd::*`` ops (one surface with fake kernels)."""
    reg = torch.library.register_fake

    @reg("aamd::spectrogram")
    def _(wav, window, twiddle, n_fft, hop, pad, center, pad_mode, onesided, n_frames, scale, power):
        n_freq = n_fft // 2 + 1 if onesided else n_fft
        return wav.new_empty((wav.shape[0], n_frames, n_freq * (1 if power > 0.0 else 2)), dtype=torch.float32)

    @reg("aamd::mel_spectrogram")
    def _(wav, window, twiddle, band_lo, band_width, band_weights, lane_order, table400, n_fft, hop, pad, center, pad_mode,
          n_frames, scale, power, table_sig):
        return wav.new_empty((wav.shape[0], n_frames, band_lo.shape[0]), dtype=torch.float32)

    @reg("aamd::mel_spectrogram_db")
    def _(wav, window, twiddle, band_lo, band_width, band_weights, lane_order, table400, n_fft, hop, pad, center, pad_mode,
          n_frames, scale, power, multiplier, amin, db_multiplier, group_max, rows_per_group, table_sig):
        return wav.new_empty((wav.shape[0], n_frames, band_lo.shape[0]), dtype=torch.float32)

    @reg("aamd::mfcc_dct")
    def _(mel, dct_mat, log_mode, group_max, vec_per_group, top_db):
        return mel.new_empty((mel.shape[0], dct_mat.shape[1]))

    @reg("aamd::resample")
    def _(wav, kernel, orig, new, width, out_len, band_tap_lo, tap_span):
        return wav.new_empty((wav.shape[0], out_len))

    @reg("aamd::lfilter")
    def _(waveform, a_coeffs, b_coeffs, n_stages, clamp):
        return torch.empty_like(waveform)

    @reg("aamd::fftconvolve")
    def _(x, y, x_row_of, y_row_of, rows, start, out_len):
        return x.new_empty((rows, out_len))


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

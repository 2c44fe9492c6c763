"""Loader of libaudio_amd_torch.so -- the dispatcher-level boundary (csrc/torch_shim.cpp).

Same mechanism as the reference's native extension: a LibTorch-stable-ABI library whose static initialisers register
boxed kernels on the CUDA dispatch key, loaded with ``torch.ops.load_library``
(reference: src/torchaudio/_extension/utils.py:50-56, src/libtorchaudio/lfilter.cpp:118-138).

  torch.ops.aamd.{spectrogram, mel_spectrogram, mel_spectrogram_db, mfcc_dct, resample, lfilter, fftconvolve} and, since
  round 4, one op for every other compute entry of include/audio_amd.h (`OPS` below)
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

OPS = ("spectrogram", "mel_spectrogram", "mel_spectrogram_db", "mfcc_dct", "resample", "lfilter", "fftconvolve",
       # round 4: the rest of the compute entries of include/audio_amd.h (one op per native entry, as the reference does)
       "mel_spectrogram_lognorm", "mfcc_frag_build", "mfcc_fused", "istft", "istft_f64", "spectrogram_f64", "phase_vocoder",
       "griffinlim_update", "mel_scale", "amplitude_to_db", "amplitude_to_db_clamped", "db_clamp", "spectrogram_grad",
       "mel_spectrogram_grad", "resample_sparse", "kaldi_features", "lfilter_f64", "resample_f64", "fftconvolve_f64",
       # round 5: the staged form of fftconvolve (prepared tap spectra for a repeated impulse response), the prepared tap
       # fragments of the matrix-core resampler
       "fftconvolve_staged", "resample_frag_build")

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
            if h.aamd_torch_shim_abi() != 7:
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
    def _(wav, kernel, orig, new, width, out_len, band_tap_lo, tap_span, frag):
        return wav.new_empty((wav.shape[0], out_len))

    @reg("aamd::resample_frag_build")
    def _(kernel, orig, new, width, band_tap_lo, tap_span):
        from . import _host
        ks = _host._rs_pick_ks(int(tap_span), int(orig))  # rsm::frag_bytes: tiles x (KS / 8) steps x (hi, lo) x 64 lanes x 16 B
        return kernel.new_empty((len(band_tap_lo) * (ks // 8) * 2 * 64 * 4,))

    @reg("aamd::lfilter")
    def _(waveform, a_coeffs, b_coeffs, n_stages, clamp):
        return torch.empty_like(waveform)

    @reg("aamd::fftconvolve")
    def _(x, y, x_row_of, y_row_of, rows, start, out_len):
        return x.new_empty((rows, out_len))

    @reg("aamd::fftconvolve_staged")
    def _(x, y, x_row_of, y_row_of, rows, start, out_len, workspace, stages):
        return x.new_empty((rows if stages & 2 else 0, out_len))

    # ---- round 4 ops ----
    @reg("aamd::mel_spectrogram_lognorm")
    def _(wav, window, twiddle, band_lo, band_width, band_weights, lane_order, table400, n_fft, hop, n_frames, scale, gain,
          mean, invstddev, out_frames, table_sig):
        rows = wav.shape[0] * (wav.shape[2] if wav.dim() == 3 else 1)
        return window.new_empty((rows, out_frames if mean is not None else n_frames, band_lo.shape[0]))

    @reg("aamd::mfcc_frag_build")
    def _(dct_mat, n_mels, n_mfcc):
        from . import _lib
        return dct_mat.new_empty((int(_lib.lib().aamd_mfcc_frag_floats()),))

    @reg("aamd::mfcc_fused")
    def _(wav, window, twiddle, band_lo, band_width, band_weights, lane_order, table400, dct_frag, group_max, n_fft, hop, pad,
          center, pad_mode, n_frames, scale, n_mfcc, multiplier, amin, db_multiplier, top_db, rows_per_group, table_sig):
        return wav.new_empty((wav.shape[0], n_frames, n_mfcc))

    def _istft_fake(spec, window, twiddle, inv_envelope, n_fft, hop, pad, center, pad_mode, length, scale, adjoint):
        return window.new_empty((spec.shape[0], length))
    reg("aamd::istft")(_istft_fake)
    reg("aamd::istft_f64")(_istft_fake)

    @reg("aamd::spectrogram_f64")
    def _(wav, window, twiddle, n_fft, hop, pad, center, pad_mode, n_frames):
        return wav.new_empty((wav.shape[0], n_frames, n_fft // 2 + 1, 2))

    @reg("aamd::phase_vocoder")
    def _(spec, phase_advance, rate, frame_major_out):
        import math
        rows, n_freq, n_in = spec.shape[0], spec.shape[1], spec.shape[2]
        n_out = int(math.ceil(n_in / rate))
        return phase_advance.new_empty((rows, n_out, n_freq, 2) if frame_major_out else (rows, n_freq, n_out, 2))

    @reg("aamd::griffinlim_update")
    def _(rebuilt, tprev, magnitude, momentum):
        return torch.empty_like(rebuilt)

    @reg("aamd::mel_scale")
    def _(spec, band_lo, band_width, band_weights):
        return spec.new_empty((spec.shape[0], spec.shape[1], band_lo.shape[0]))

    @reg("aamd::amplitude_to_db")
    def _(x, multiplier, amin, db_multiplier, group_max, group_size):
        return torch.empty_like(x)

    @reg("aamd::amplitude_to_db_clamped")
    def _(x, multiplier, amin, db_multiplier, group_max, group_size, top_db):
        return torch.empty_like(x)

    @reg("aamd::db_clamp")
    def _(x, group_max, group_size, top_db):
        return torch.empty_like(x)

    @reg("aamd::spectrogram_grad")
    def _(spec, dpower, power):
        return torch.empty_like(spec)

    @reg("aamd::mel_spectrogram_grad")
    def _(spec, dmel, band_lo, band_width, band_weights, power):
        return spec

    @reg("aamd::resample_sparse")
    def _(wav, taps_compact, tap_lo, orig, new, width, out_len):
        return wav.new_empty((wav.shape[0], out_len))

    @reg("aamd::kaldi_features")
    def _(wav, window, twiddle, band_lo, band_width, band_weights, noise, n_frames, opts, fopts):
        return wav.new_empty((wav.shape[0], n_frames, opts[10] if band_lo is not None else opts[0] // 2 + 1))

    @reg("aamd::lfilter_f64")
    def _(waveform, a_coeffs, b_coeffs, n_stages, clamp):
        return torch.empty_like(waveform)

    @reg("aamd::resample_f64")
    def _(wav, kernel, orig, new, width, out_len):
        return wav.new_empty((wav.shape[0], out_len))

    @reg("aamd::fftconvolve_f64")
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

"""``torch.ops.audio_amd.*`` -- the hot path registered as PyTorch custom ops.

The reference reaches native code through operator registration: ``torch.ops.load_library`` runs
``STABLE_TORCH_LIBRARY_FRAGMENT(torchaudio, m){ m.def(schema) }`` / ``..._IMPL(torchaudio, CUDA, m)``
(src/libtorchaudio/lfilter.cpp:118-138, src/torchaudio/_extension/utils.py:50-56).  This module is the
same mechanism from the Python side (``torch.library``): schemas in a new namespace, implementations
under the CUDA dispatch key (ROCm tensors dispatch there) that call libaudio_amd.so's C ABI, and
Meta ("fake") implementations so the ops trace / export.  There is deliberately NO CPU-key
implementation: a CPU tensor fails in the dispatcher ("no kernel for CPU"), loudly.

Schemas follow SURVEY.md 8(b).  ``norm_mode``: 0 none, 1 "frame_length", 2 "window".
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from . import _host
from . import functional as F

_NORM = {0: False, 1: "frame_length", 2: "window"}

_DEF = torch.library.Library("audio_amd", "DEF")
_DEF.define("spectrogram(Tensor waveform, Tensor window, int pad, int n_fft, int hop_length, int win_length, "
            "float? power, int norm_mode, bool center, str pad_mode, bool onesided) -> Tensor")
_DEF.define("mel_spectrogram(Tensor waveform, Tensor window, Tensor fb, int pad, int n_fft, int hop_length, "
            "int win_length, float power, int norm_mode, bool center, str pad_mode) -> Tensor")
_DEF.define("mfcc(Tensor waveform, Tensor window, Tensor fb, Tensor dct_mat, int pad, int n_fft, int hop_length, "
            "int win_length, float power, int norm_mode, bool center, str pad_mode, bool log_mels, float top_db) -> Tensor")
_DEF.define("inverse_spectrogram(Tensor spectrogram, int? length, Tensor window, int pad, int n_fft, int hop_length, "
            "int win_length, int norm_mode, bool center, str pad_mode, bool onesided) -> Tensor")
_DEF.define("amplitude_to_DB(Tensor x, float multiplier, float amin, float db_multiplier, float? top_db) -> Tensor")
_DEF.define("resample_apply(Tensor waveform, Tensor kernel, int orig_freq, int new_freq, int gcd, int width) -> Tensor")
_DEF.define("lfilter(Tensor waveform, Tensor a_coeffs, Tensor b_coeffs, bool clamp, bool batching) -> Tensor")
_DEF.define("lfilter_cascade(Tensor waveform, Tensor a_coeffs, Tensor b_coeffs, bool clamp) -> Tensor")
_DEF.define("fftconvolve(Tensor x, Tensor y, str mode) -> Tensor")
_DEF.define("phase_vocoder(Tensor complex_specgrams, float rate, Tensor phase_advance) -> Tensor")
_DEF.define("griffinlim(Tensor specgram, Tensor window, int n_fft, int hop_length, int win_length, float power, int n_iter, "
            "float momentum, int? length, bool rand_init) -> Tensor")
# round 6: the rest of the public surface, so that every scripted front of functional.py / transforms.py is one operator
_DEF.define("mel_scale(Tensor specgram, Tensor fb) -> Tensor")
_DEF.define("resample(Tensor waveform, int orig_freq, int new_freq, int lowpass_filter_width, float rolloff, "
            "str resampling_method, float? beta) -> Tensor")
_DEF.define("pitch_shift(Tensor waveform, int sample_rate, int n_steps, int bins_per_octave, int n_fft, int? win_length, "
            "int? hop_length, Tensor? window) -> Tensor")
_DEF.define("biquad(Tensor waveform, float b0, float b1, float b2, float a0, float a1, float a2) -> Tensor")
_DEF.define("filtfilt(Tensor waveform, Tensor a_coeffs, Tensor b_coeffs, bool clamp) -> Tensor")
_DEF.define("designed_biquad(Tensor waveform, str kind, int sample_rate, float[] params, bool flag) -> Tensor")
_DEF.define("mfcc_module(Tensor waveform, Tensor window, Tensor fb, Tensor dct_mat, int pad, int n_fft, int hop_length, "
            "int win_length, float power, int norm_mode, bool center, str pad_mode, bool log_mels, float top_db, "
            "float multiplier, float amin, float db_multiplier, int fused, int state) -> Tensor")
_DEF.define("rnnt_features(Tensor waveform, Tensor window, Tensor fb, int n_fft, int hop_length, float gain, Tensor mean, "
            "Tensor invstddev, int right_padding) -> Tensor")

_CUDA = torch.library.Library("audio_amd", "IMPL", "CUDA")
_META = torch.library.Library("audio_amd", "IMPL", "Meta")
_AUTOGRAD = torch.library.Library("audio_amd", "IMPL", "AutogradCUDA")


def _plain_tensor_wants_grad(a) -> bool:
    # (`type(a) is Tensor`: fake / functional tensors of a tracer never take the Python implementation -- they go on to the
    # Meta kernel exactly as before the autograd route existed)
    return type(a) is Tensor and a.requires_grad


def _register(name: str, fn) -> None:
    """CUDA-key kernel = the eager implementation.  The AutogradCUDA kernel makes a SCRIPTED call differentiable the way the
    eager call is (the reference's scripted modules are: they are aten compositions): when autograd is recording and an
    argument asks for a gradient the implementation runs right there, at the autograd level, and its own autograd Functions
    (F._SpectrogramFunction, _LFilterFunction, ...) record the graph; every other call goes below autograd to the CUDA /
    Meta kernel."""
    _CUDA.impl(name, fn)
    op = getattr(torch.ops.audio_amd, name).default

    def at_autograd_level(*args):
        if torch.is_grad_enabled() and any(_plain_tensor_wants_grad(a) for a in args):
            return fn(*args)
        with torch._C._AutoDispatchBelowAutograd():
            return op(*args)

    _AUTOGRAD.impl(name, at_autograd_level)


def _spectrogram(waveform, window, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode, onesided):
    return F.spectrogram(waveform, pad, window, n_fft, hop_length, win_length, power, _NORM[norm_mode], center,
                         pad_mode, onesided)


def _mel_spectrogram(waveform, window, fb, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode):
    out = F._melspectrogram(waveform, pad, window, fb, n_fft, hop_length, win_length, power, _NORM[norm_mode], center,
                            pad_mode)
    return out.view(tuple(waveform.shape[:-1]) + out.shape[-2:]).transpose(-1, -2)


def _mfcc(waveform, window, fb, dct_mat, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode,
          log_mels, top_db):
    return F._mfcc(waveform, pad, window, fb, dct_mat, n_fft, hop_length, win_length, power, _NORM[norm_mode], center,
                   pad_mode, log_mels, top_db)


def _inverse_spectrogram(spectrogram, length, window, pad, n_fft, hop_length, win_length, norm_mode, center, pad_mode,
                         onesided):
    return F.inverse_spectrogram(spectrogram, length, pad, window, n_fft, hop_length, win_length, _NORM[norm_mode],
                                 center, pad_mode, onesided)


def _mel_spectrogram_module(waveform, window, fb, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode):
    return F._melspectrogram_module(waveform, window, fb, pad, n_fft, hop_length, win_length, power, _NORM[norm_mode], center,
                                    pad_mode)


def _mfcc_module(waveform, window, fb, dct_mat, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode,
                 log_mels, top_db, multiplier, amin, db_multiplier, fused, state):
    return F._mfcc_module(waveform, window, fb, dct_mat, pad, n_fft, hop_length, win_length, power, _NORM[norm_mode], center,
                          pad_mode, log_mels, top_db, (multiplier, amin, db_multiplier), fused,
                          F._mfcc_state_of(state) if fused else None)


_register("spectrogram", _spectrogram)
_register("inverse_spectrogram", _inverse_spectrogram)
_register("mel_spectrogram", _mel_spectrogram_module)
_register("mfcc", _mfcc)
_register("amplitude_to_DB", F.amplitude_to_DB)
def _resample_apply(waveform, kernel, orig_freq, new_freq, gcd, width):
    # (the schema puts the tap table second; until round 6 the function itself was registered and received the arguments in the
    # wrong order -- found by the first scripted Resample that ran on a device, tests/test_torchscript.py)
    return F._apply_sinc_resample_kernel(waveform, orig_freq, new_freq, gcd, kernel, width)


_register("resample_apply", _resample_apply)
_register("lfilter", F.lfilter)
_register("lfilter_cascade", F.biquad_cascade)
_register("fftconvolve", F.fftconvolve)
_register("phase_vocoder", F.phase_vocoder)
_register("griffinlim", F.griffinlim)


_register("mfcc_module", _mfcc_module)
_register("mel_scale", F.mel_scale)
_register("resample", F.resample)
_register("pitch_shift", F.pitch_shift)
_register("biquad", F.biquad)
_register("filtfilt", F.filtfilt)
_register("designed_biquad", F._designed_biquad)


def _rnnt_features(waveform, window, fb, n_fft, hop_length, gain, mean, invstddev, right_padding):
    out = F._mel_lognorm(waveform, window, fb, n_fft, hop_length, gain, mean, invstddev, right_padding)
    return out.view(tuple(waveform.shape[:-1]) + out.shape[-2:])


_register("rnnt_features", _rnnt_features)


# ---- Meta implementations: shapes / strides only ---------------------------------------------

def _stft_frames(length, pad, n_fft, hop, center):
    return _host.frame_count(length, n_fft, hop, center, pad)


def _frame_major_view(waveform, n_out, T, dtype=None):
    lead = tuple(waveform.shape[:-1])
    rows = 1
    for d in lead:
        rows *= d
    out = waveform.new_empty((rows, T, n_out), dtype=dtype or waveform.dtype)
    return out.view(lead + (T, n_out)).transpose(-1, -2)


def _spectrogram_meta(waveform, window, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode, onesided):
    T = _stft_frames(waveform.shape[-1], pad, n_fft, hop_length, center)
    n_freq = n_fft // 2 + 1 if onesided else n_fft
    return _frame_major_view(waveform, n_freq, T, torch.complex64 if power is None else None)


def _mel_meta(waveform, window, fb, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode):
    return _frame_major_view(waveform, fb.shape[1], _stft_frames(waveform.shape[-1], pad, n_fft, hop_length, center))


def _mfcc_meta(waveform, window, fb, dct_mat, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode,
               log_mels, top_db):
    return _frame_major_view(waveform, dct_mat.shape[1], _stft_frames(waveform.shape[-1], pad, n_fft, hop_length, center))


def _resample_meta(waveform, kernel, orig_freq, new_freq, gcd, width):
    orig, new = int(orig_freq) // gcd, int(new_freq) // gcd
    return waveform.new_empty(tuple(waveform.shape[:-1]) + (int(math.ceil(new * waveform.shape[-1] / orig)),))


def _lfilter_meta(waveform, a_coeffs, b_coeffs, clamp, batching):
    if a_coeffs.ndim > 1 and not batching:
        return waveform.new_empty(tuple(waveform.shape[:-1]) + (a_coeffs.shape[0], waveform.shape[-1]))
    return torch.empty_like(waveform)


def _fftconvolve_meta(x, y, mode):
    F._check_shape_compatible(x, y)
    F._check_convolve_mode(mode)
    nx, ny = x.shape[-1], y.shape[-1]
    n = {"full": nx + ny - 1, "same": nx, "valid": max(nx, ny) - min(nx, ny) + 1}[mode]
    lead = tuple(max(a, b) for a, b in zip(x.shape[:-1], y.shape[:-1]))
    return x.new_empty(lead + (n,))


_META.impl("spectrogram", _spectrogram_meta)
_META.impl("mel_spectrogram", _mel_meta)
_META.impl("mfcc", _mfcc_meta)
_META.impl("amplitude_to_DB", lambda x, multiplier, amin, db_multiplier, top_db: torch.empty_like(x))
_META.impl("resample_apply", _resample_meta)
_META.impl("lfilter", _lfilter_meta)
_META.impl("lfilter_cascade", lambda waveform, a_coeffs, b_coeffs, clamp: torch.empty_like(waveform))
_META.impl("fftconvolve", _fftconvolve_meta)


def _inverse_spectrogram_meta(spectrogram, length, window, pad, n_fft, hop_length, win_length, norm_mode, center, pad_mode,
                              onesided):
    T = spectrogram.shape[-1]
    n = length if length is not None else n_fft + hop_length * (T - 1) - (2 * (n_fft // 2) if center else 0)
    return spectrogram.new_empty(tuple(spectrogram.shape[:-2]) + (n,), dtype=torch.float32)


def _phase_vocoder_meta(complex_specgrams, rate, phase_advance):
    if rate == 1.0:
        return complex_specgrams
    T = int(math.ceil(complex_specgrams.shape[-1] / rate))
    return complex_specgrams.new_empty(tuple(complex_specgrams.shape[:-1]) + (T,), dtype=torch.complex64)


def _griffinlim_meta(specgram, window, n_fft, hop_length, win_length, power, n_iter, momentum, length, rand_init):
    n = length if length is not None else hop_length * (specgram.shape[-1] - 1)
    return specgram.new_empty(tuple(specgram.shape[:-2]) + (n,), dtype=torch.float32)


def _rnnt_features_meta(waveform, window, fb, n_fft, hop_length, gain, mean, invstddev, right_padding):
    T = _stft_frames(waveform.shape[-1], 0, n_fft, hop_length, True)
    return waveform.new_empty(tuple(waveform.shape[:-1]) + (T + right_padding, fb.shape[1]), dtype=torch.float32)


def _mfcc_module_meta(waveform, window, fb, dct_mat, pad, n_fft, hop_length, win_length, power, norm_mode, center, pad_mode,
                      log_mels, top_db, multiplier, amin, db_multiplier, fused, state):
    return _frame_major_view(waveform, dct_mat.shape[1], _stft_frames(waveform.shape[-1], pad, n_fft, hop_length, center))


def _mel_scale_meta(specgram, fb):
    lead, T = tuple(specgram.shape[:-2]), specgram.shape[-1]
    return specgram.new_empty(lead + (T, fb.shape[1])).transpose(-1, -2)


def _resample_full_meta(waveform, orig_freq, new_freq, lowpass_filter_width, rolloff, resampling_method, beta):
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    return _resample_meta(waveform, None, orig_freq, new_freq, g, 0)


_same_as_waveform = lambda waveform, *rest: torch.empty_like(waveform)        # noqa: E731
_META.impl("mfcc_module", _mfcc_module_meta)
_META.impl("mel_scale", _mel_scale_meta)
_META.impl("resample", _resample_full_meta)
_META.impl("pitch_shift", _same_as_waveform)
_META.impl("biquad", _same_as_waveform)
_META.impl("filtfilt", _same_as_waveform)
_META.impl("designed_biquad", _same_as_waveform)
_META.impl("inverse_spectrogram", _inverse_spectrogram_meta)
_META.impl("phase_vocoder", _phase_vocoder_meta)
_META.impl("griffinlim", _griffinlim_meta)
_META.impl("rnnt_features", _rnnt_features_meta)

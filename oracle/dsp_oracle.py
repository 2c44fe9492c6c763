"""CPU oracle for the audio DSP hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A float64 numpy restatement of the algorithms torchaudio composes for
Spectrogram / MelSpectrogram / MFCC / Resample / lfilter / fftconvolve.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; nothing under ``audio_amd/`` does.

Parity pinning: ``tests/test_oracle_golden.py`` checks every function here
against (i) the reference's own librosa / SoX golden vectors and (ii) outputs of
the reference itself run in the build container, both committed as fixtures
under ``tests/golden/`` by ``tests/golden/make_golden.py``.

Each function cites the reference file:line it restates (paths relative to
/root/reference/src/torchaudio unless prefixed ``torch/``).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np


# --------------------------------------------------------------------------- #
# framing / STFT                                                              #
# --------------------------------------------------------------------------- #

def frame_count(length: int, n_fft: int, hop: int, center: bool) -> int:
    """torch/functional.py:675-681 + aten stft: n_frames = 1 + (L' - n_fft)//hop
    with L' = L + 2*(n_fft//2) when centred."""
    lp = length + 2 * (n_fft // 2) if center else length
    return 1 + (lp - n_fft) // hop


def source_index(i: np.ndarray, length: int, pad_mode: str) -> np.ndarray:
    """Map an index into the centred (padded) signal to a source sample index, or
    -1 where the padded value is a constant zero.

    Restates aten padding used by torch.stft(center=True) (torch/functional.py:675-680):
    reflect: i<0 -> -i ; i>=L -> 2(L-1)-i  (edge sample not repeated)
    replicate: clamp ; circular: modulo ; constant: zeros.
    """
    i = np.asarray(i, dtype=np.int64)
    if pad_mode == "reflect":
        j = np.where(i < 0, -i, i)
        j = np.where(j >= length, 2 * (length - 1) - j, j)
        return j
    if pad_mode == "replicate":
        return np.clip(i, 0, length - 1)
    if pad_mode == "circular":
        return np.mod(i, length)
    if pad_mode == "constant":
        return np.where((i < 0) | (i >= length), -1, i)
    raise ValueError(pad_mode)


def padded_window(window: np.ndarray, n_fft: int) -> np.ndarray:
    """aten stft: a window shorter than n_fft is zero-padded centred,
    left = (n_fft - win_length)//2."""
    window = np.asarray(window, dtype=np.float64)
    wl = window.shape[0]
    if wl == n_fft:
        return window
    left = (n_fft - wl) // 2
    out = np.zeros(n_fft, dtype=np.float64)
    out[left:left + wl] = window
    return out


def hann_window(n: int, periodic: bool = True) -> np.ndarray:
    """torch.hann_window (transforms/_transforms.py:70,86): periodic by default."""
    d = n if periodic else n - 1
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / d)


def frames(x: np.ndarray, n_fft: int, hop: int, center: bool = True,
           pad_mode: str = "reflect") -> np.ndarray:
    """(B, L) -> (B, T, n_fft) framed signal (no window)."""
    x = np.asarray(x, dtype=np.float64)
    B, L = x.shape
    T = frame_count(L, n_fft, hop, center)
    p = n_fft // 2 if center else 0
    idx = (np.arange(T)[:, None] * hop + np.arange(n_fft)[None, :]) - p
    src = source_index(idx, L, pad_mode if center else "constant")
    fr = x[:, np.clip(src, 0, L - 1)]
    fr = np.where(src[None] < 0, 0.0, fr)
    return fr


def stft(x, window, n_fft, hop, center=True, pad_mode="reflect",
         onesided=True, scale=1.0) -> np.ndarray:
    """functional/functional.py:119-137 (torch.stft call): returns complex (B, F, T)."""
    fr = frames(x, n_fft, hop, center, pad_mode) * padded_window(window, n_fft)[None, None, :]
    spec = np.fft.rfft(fr, axis=-1) if onesided else np.fft.fft(fr, axis=-1)
    return np.swapaxes(spec, -1, -2) * scale


def spectrogram(x, pad, window, n_fft, hop, win_length, power, normalized,
                center=True, pad_mode="reflect", onesided=True) -> np.ndarray:
    """functional/functional.py:54-145.  x: (..., L) -> (..., F, T)."""
    x = np.asarray(x, dtype=np.float64)
    if pad > 0:  # :112-114
        x = np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)])
    frame_norm = normalized == "frame_length"
    window_norm = normalized is True or normalized == "window"
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    window = np.asarray(window, dtype=np.float64)
    assert window.shape[0] == win_length
    s = stft(x2, window, n_fft, hop, center, pad_mode, onesided,
             scale=(1.0 / math.sqrt(n_fft)) if frame_norm else 1.0)
    if window_norm:  # :139-140
        s = s / math.sqrt(float((window ** 2).sum()))
    s = s.reshape(shape[:-1] + s.shape[-2:])
    if power is None:
        return s
    if power == 1.0:
        return np.abs(s)
    return np.abs(s) ** power


# --------------------------------------------------------------------------- #
# mel filterbank / dB / DCT                                                   #
# --------------------------------------------------------------------------- #

def hz_to_mel(f: float, mel_scale: str = "htk") -> float:
    """functional/functional.py:425-456."""
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + f / 700.0)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    if f >= min_log_hz:
        mels = min_log_hz / f_sp + math.log(f / min_log_hz) / (math.log(6.4) / 27.0)
    return mels


def mel_to_hz(m: np.ndarray, mel_scale: str = "htk") -> np.ndarray:
    """functional/functional.py:459-489."""
    m = np.asarray(m, dtype=np.float64)
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    f = f_sp * m
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, 1000.0 * np.exp(logstep * (m - min_log_mel)), f)


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None,
                    mel_scale="htk") -> np.ndarray:
    """functional/functional.py:518-587 -> (n_freqs, n_mels), float64."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(hz_to_mel(f_min, mel_scale), hz_to_mel(f_max, mel_scale), n_mels + 2)
    f_pts = mel_to_hz(m_pts, mel_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    if norm == "slaney":
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels]))[None, :]
    return fb


def mel_scale(spec: np.ndarray, fb: np.ndarray) -> np.ndarray:
    """transforms/_transforms.py:413: (..., F, T) x (F, M) -> (..., M, T)."""
    return np.swapaxes(np.swapaxes(spec, -1, -2) @ np.asarray(fb, dtype=np.float64), -1, -2)


def amplitude_to_db(x, multiplier, amin, db_multiplier, top_db: Optional[float] = None):
    """functional/functional.py:356-404 (cut-off per leading item of the
    (-1, C, F, T) view, C = shape[-3] if ndim > 2 else 1)."""
    x = np.asarray(x, dtype=np.float64)
    x_db = multiplier * np.log10(np.maximum(x, amin)) - multiplier * db_multiplier
    if top_db is not None:
        shape = x_db.shape
        c = shape[-3] if x_db.ndim > 2 else 1
        v = x_db.reshape(-1, c, shape[-2], shape[-1])
        cut = v.max(axis=(-3, -2, -1)) - top_db
        v = np.maximum(v, cut[:, None, None, None])
        x_db = v.reshape(shape)
    return x_db


def create_dct(n_mfcc: int, n_mels: int, norm: Optional[str]) -> np.ndarray:
    """functional/functional.py:636-667 -> (n_mels, n_mfcc)."""
    n = np.arange(float(n_mels))
    k = np.arange(float(n_mfcc))[:, None]
    dct = np.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    if norm is None:
        dct = dct * 2.0
    else:
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct = dct * math.sqrt(2.0 / float(n_mels))
    return dct.T


def mel_spectrogram(x, window, fb, n_fft, hop, win_length=None, pad=0, power=2.0,
                    normalized=False, center=True, pad_mode="reflect"):
    """transforms/_transforms.py:612-622."""
    win_length = win_length or n_fft
    s = spectrogram(x, pad, window, n_fft, hop, win_length, power, normalized,
                    center, pad_mode, True)
    return mel_scale(s, fb)


def mfcc(x, window, fb, dct_mat, n_fft, hop, win_length=None, log_mels=False,
         top_db=80.0, **kw):
    """transforms/_transforms.py:692-709."""
    mel = mel_spectrogram(x, window, fb, n_fft, hop, win_length, **kw)
    if log_mels:
        mel = np.log(mel + 1e-6)
    else:
        mel = amplitude_to_db(mel, 10.0, 1e-10, 0.0, top_db)
    return np.swapaxes(np.swapaxes(mel, -1, -2) @ np.asarray(dct_mat, np.float64), -1, -2)


# --------------------------------------------------------------------------- #
# resample                                                                    #
# --------------------------------------------------------------------------- #

def _i0(x: np.ndarray) -> np.ndarray:
    return np.i0(x)


def sinc_resample_kernel(orig_freq, new_freq, gcd, lowpass_filter_width=6, rolloff=0.99,
                         resampling_method="sinc_interp_hann", beta=None):
    """functional/functional.py:1305-1402, evaluated in float64.
    Returns (kernel[new, 2*width+orig] float64, width)."""
    orig = int(orig_freq) // gcd
    new = int(new_freq) // gcd
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = t * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    if resampling_method == "sinc_interp_hann":
        window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    else:
        if beta is None:
            beta = 14.769656459379492
        window = _i0(beta * np.sqrt(1 - (t / lowpass_filter_width) ** 2)) / _i0(np.float64(beta))
    t = t * math.pi
    scale = base / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(t == 0, 1.0, np.sin(t) / t)
    return k * window * scale, width


def apply_sinc_resample_kernel(x, orig_freq, new_freq, gcd, kernel, width):
    """functional/functional.py:1405-1432.  x: (..., L)."""
    x = np.asarray(x, dtype=np.float64)
    kernel = np.asarray(kernel, dtype=np.float64)
    orig = int(orig_freq) // gcd
    new = int(new_freq) // gcd
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    B, L = x2.shape
    xp = np.pad(x2, [(0, 0), (width, width + orig)])
    K = kernel.shape[-1]
    nq = (xp.shape[1] - K) // orig + 1
    idx = np.arange(nq)[:, None] * orig + np.arange(K)[None, :]
    win = xp[:, idx]                       # (B, nq, K)
    y = np.einsum("bqk,pk->bqp", win, kernel.reshape(new, K))
    y = y.reshape(B, nq * new)
    target = int(math.ceil(new * L / orig))
    y = y[:, :target]
    return y.reshape(shape[:-1] + (y.shape[-1],))


def resample(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99,
             resampling_method="sinc_interp_hann", beta=None):
    """functional/functional.py:1435-1490."""
    if orig_freq == new_freq:
        return np.asarray(x, dtype=np.float64)
    g = math.gcd(int(orig_freq), int(new_freq))
    k, w = sinc_resample_kernel(orig_freq, new_freq, g, lowpass_filter_width, rolloff,
                                resampling_method, beta)
    return apply_sinc_resample_kernel(x, orig_freq, new_freq, g, k, w)


# --------------------------------------------------------------------------- #
# lfilter / biquad                                                            #
# --------------------------------------------------------------------------- #

def lfilter(x, a, b, clamp=True):
    """functional/filtering.py:1032-1099 + libtorchaudio/lfilter.cpp:17-48, float64,
    direct-form I with zero initial state.  x: (..., C?, L); a, b 1-D (shared) or
    2-D (C, order+1) applied per channel (batching=True)."""
    x = np.asarray(x, dtype=np.float64)
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    b = np.atleast_2d(np.asarray(b, dtype=np.float64))
    C = a.shape[0]
    shape = x.shape
    xr = x.reshape(-1, C, shape[-1])
    bn = b / a[:, :1]
    an = a / a[:, :1]
    order = a.shape[1]
    L = shape[-1]
    y = np.zeros_like(xr)
    xp = np.pad(xr, [(0, 0), (0, 0), (order - 1, 0)])
    yp = np.zeros_like(xp)
    for n in range(L):
        acc = np.zeros(xr.shape[:2])
        for k in range(order):
            acc += bn[None, :, k] * xp[:, :, n + order - 1 - k]
        for k in range(1, order):
            acc -= an[None, :, k] * yp[:, :, n + order - 1 - k]
        yp[:, :, n + order - 1] = acc
    y = yp[:, :, order - 1:]
    if clamp:
        y = np.clip(y, -1.0, 1.0)
    return y.reshape(shape)


def biquad(x, b0, b1, b2, a0, a1, a2):
    """functional/filtering.py:295-333."""
    return lfilter(x, [a0, a1, a2], [b0, b1, b2])


# --------------------------------------------------------------------------- #
# fftconvolve                                                                 #
# --------------------------------------------------------------------------- #

def fftconvolve(x, y, mode="full"):
    """functional/functional.py:2222-2258 -- true linear convolution along the last
    dim with size-1 broadcasting of leading dims, then the mode crop (:2207-2219)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.ndim != y.ndim:
        raise ValueError("ndim")
    nx, ny = x.shape[-1], y.shape[-1]
    n = nx + ny - 1
    z = np.fft.irfft(np.fft.rfft(x, n=n) * np.fft.rfft(y, n=n), n=n)
    if mode == "full":
        return z
    if mode == "valid":
        tl = max(nx, ny) - min(nx, ny) + 1
        s = (n - tl) // 2
        return z[..., s:s + tl]
    if mode == "same":
        s = (n - nx) // 2
        return z[..., s:s + nx]
    raise ValueError(mode)

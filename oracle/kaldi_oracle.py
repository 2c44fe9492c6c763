"""TEST INFRASTRUCTURE -- float64 numpy restatement of the reference's Kaldi-compatible front-end.

Follows /root/reference/src/torchaudio/compliance/kaldi.py statement by statement, in float64 (the reference computes in the
waveform's dtype, float32): `frames` (:44-83), `window` (:86-113), `log_energy` (:116-122), `conditioned_frames` (:154-217),
`spectrogram` (:229-315), `warp` (:334-433), `mel_banks` (:436-511), `fbank` (:514-645), `mfcc` (:648-814).  Only tests/ may import
this module; the product (audio_amd/compliance/kaldi.py) never does.  Pinned on the 21 reference-run fixtures of
tests/golden/kaldi_goldens.npz (tests/test_oracle_golden.py::test_kaldi_oracle_*), the three dither cases with the recorded draw.

The machine epsilon that floors the logs is float32's (the reference's `_get_epsilon` for a float32 waveform): the device computes
in float32 and the fixtures were produced in float32.
"""
import math

import numpy as np

EPS = float(np.finfo(np.float32).eps)


def sizes(sample_frequency, frame_shift, frame_length, round_to_power_of_two):
    shift = int(sample_frequency * frame_shift * 0.001)
    win = int(sample_frequency * frame_length * 0.001)
    padded = (1 if win == 0 else 2 ** (win - 1).bit_length()) if round_to_power_of_two else win
    return shift, win, padded


def frames(x, win, shift, snip_edges):
    """kaldi.py:44-83.  snip_edges: only whole frames; otherwise the signal is mirrored at both ends ([2, 1, 0 | 0, 1, 2 ...])."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if snip_edges:
        if n < win:
            return np.zeros((0, 0))
        m = 1 + (n - win) // shift
        src = x
    else:
        m = (n + shift // 2) // shift
        pad = win // 2 - shift // 2
        rev = x[::-1]
        src = np.concatenate([rev[n - pad:], x, rev]) if pad > 0 else np.concatenate([x[-pad:], rev])
    idx = shift * np.arange(m)[:, None] + np.arange(win)[None, :]
    return src[idx]


def window(kind, n, blackman_coeff=0.42):
    """kaldi.py:86-113 (symmetric windows: periodic=False)."""
    k = np.arange(n, dtype=np.float64)
    if kind == "hanning":
        return 0.5 - 0.5 * np.cos(2 * np.pi * k / (n - 1))
    if kind == "hamming":
        return 0.54 - 0.46 * np.cos(2 * np.pi * k / (n - 1))
    if kind == "povey":
        return (0.5 - 0.5 * np.cos(2 * np.pi * k / (n - 1))) ** 0.85
    if kind == "rectangular":
        return np.ones(n)
    if kind == "blackman":
        a = 2 * np.pi / (n - 1)
        return blackman_coeff - 0.5 * np.cos(a * k) + (0.5 - blackman_coeff) * np.cos(2 * a * k)
    raise ValueError(kind)


def log_energy(fr, energy_floor):
    e = np.log(np.maximum((fr ** 2).sum(axis=1), EPS))
    return e if energy_floor == 0.0 else np.maximum(e, math.log(energy_floor))


def conditioned_frames(x, padded, win, shift, window_type="povey", blackman_coeff=0.42, snip_edges=True, raw_energy=True,
                       energy_floor=1.0, dither=0.0, remove_dc_offset=True, preemphasis_coefficient=0.97, noise=None):
    """kaldi.py:154-217: dither -> DC removal -> (raw) log-energy -> pre-emphasis -> window -> zero padding -> (windowed) energy."""
    fr = frames(x, win, shift, snip_edges)
    if dither != 0.0:
        fr = fr + np.asarray(noise, dtype=np.float64) * dither
    if remove_dc_offset:
        fr = fr - fr.mean(axis=1, keepdims=True)
    e = log_energy(fr, energy_floor) if raw_energy else None
    if preemphasis_coefficient != 0.0:
        prev = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)          # replicate the first sample
        fr = fr - preemphasis_coefficient * prev
    fr = fr * window(window_type, win, blackman_coeff)[None, :]
    if padded != win:
        fr = np.concatenate([fr, np.zeros((fr.shape[0], padded - win))], axis=1)
    if not raw_energy:
        e = log_energy(fr, energy_floor)
    return fr, e


_WINDOW_KEYS = ("window_type", "blackman_coeff", "snip_edges", "raw_energy", "energy_floor", "dither", "remove_dc_offset",
                "preemphasis_coefficient", "noise")


def spectrogram(x, sample_frequency=16000.0, frame_shift=10.0, frame_length=25.0, round_to_power_of_two=True,
                subtract_mean=False, **kw):
    """kaldi.py:229-315: log power spectrum floored at epsilon, bin 0 replaced by the frame's log-energy."""
    shift, win, padded = sizes(sample_frequency, frame_shift, frame_length, round_to_power_of_two)
    fr, e = conditioned_frames(x, padded, win, shift, **{k: v for k, v in kw.items() if k in _WINDOW_KEYS})
    p = np.log(np.maximum(np.abs(np.fft.rfft(fr, axis=1)) ** 2, EPS))
    p[:, 0] = e
    return p - p.mean(axis=0, keepdims=True) if subtract_mean else p


def mel(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def inv_mel(m):
    return 700.0 * (np.exp(np.asarray(m, dtype=np.float64) / 1127.0) - 1.0)


def warp(vtln_low, vtln_high, low_freq, high_freq, factor, freq):
    """kaldi.py:334-405: piecewise-linear VTLN warp with inflection points l and h; identity outside [low_freq, high_freq]."""
    lo = vtln_low * max(1.0, factor)
    hi = vtln_high * min(1.0, factor)
    scale = 1.0 / factor
    s_left = (scale * lo - low_freq) / (lo - low_freq)
    s_right = (high_freq - scale * hi) / (high_freq - hi)
    f = np.asarray(freq, dtype=np.float64)
    res = np.where(f >= hi, high_freq + s_right * (f - high_freq), scale * f)
    res = np.where(f < lo, low_freq + s_left * (f - low_freq), res)
    return np.where((f < low_freq) | (f > high_freq), f, res)


def mel_banks(num_bins, padded, sample_freq, low_freq=20.0, high_freq=0.0, vtln_low=100.0, vtln_high=-500.0, vtln_warp=1.0):
    """kaldi.py:436-511: (num_bins, padded / 2) triangles on the mel axis."""
    nyq = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyq
    width = sample_freq / padded
    m_lo, m_hi = float(mel(low_freq)), float(mel(high_freq))
    delta = (m_hi - m_lo) / (num_bins + 1)
    if vtln_high < 0.0:
        vtln_high += nyq
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = m_lo + b * delta, m_lo + (b + 1.0) * delta, m_lo + (b + 2.0) * delta
    if vtln_warp != 1.0:
        left, center, right = (mel(warp(vtln_low, vtln_high, low_freq, high_freq, vtln_warp, inv_mel(t))) for t in (left, center, right))
    m = mel(width * np.arange(padded // 2))[None, :]
    up, down = (m - left) / (center - left), (right - m) / (right - center)
    if vtln_warp == 1.0:
        return np.maximum(0.0, np.minimum(up, down))
    bins = np.zeros_like(up)
    ui, di = (m > left) & (m <= center), (m > center) & (m < right)
    bins[ui] = up[ui]
    bins[di] = down[di]
    return bins


def fbank(x, sample_frequency=16000.0, frame_shift=10.0, frame_length=25.0, round_to_power_of_two=True, num_mel_bins=23,
          low_freq=20.0, high_freq=0.0, vtln_low=100.0, vtln_high=-500.0, vtln_warp=1.0, use_power=True, use_log_fbank=True,
          use_energy=False, htk_compat=False, subtract_mean=False, **kw):
    """kaldi.py:514-645."""
    shift, win, padded = sizes(sample_frequency, frame_shift, frame_length, round_to_power_of_two)
    fr, e = conditioned_frames(x, padded, win, shift, **{k: v for k, v in kw.items() if k in _WINDOW_KEYS})
    s = np.abs(np.fft.rfft(fr, axis=1))
    if use_power:
        s = s ** 2
    banks = mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq, vtln_low, vtln_high, vtln_warp)
    banks = np.concatenate([banks, np.zeros((num_mel_bins, 1))], axis=1)        # the Nyquist bin carries no weight
    f = s @ banks.T
    if use_log_fbank:
        f = np.log(np.maximum(f, EPS))
    if use_energy:
        f = np.concatenate([f, e[:, None]], axis=1) if htk_compat else np.concatenate([e[:, None], f], axis=1)
    return f - f.mean(axis=0, keepdims=True) if subtract_mean else f


def mfcc(x, num_ceps=13, num_mel_bins=23, cepstral_lifter=22.0, use_energy=False, htk_compat=False, subtract_mean=False, **kw):
    """kaldi.py:648-814: log mel energies -> orthonormal DCT-II whose first column is sqrt(1 / N) -> lifter -> energy / HTK order."""
    f = fbank(x, num_mel_bins=num_mel_bins, use_energy=use_energy, htk_compat=htk_compat, subtract_mean=False, use_power=True,
              use_log_fbank=True, **kw)
    e = None
    if use_energy:
        e = f[:, num_mel_bins if htk_compat else 0].copy()
        off = int(not htk_compat)
        f = f[:, off:off + num_mel_bins]
    n = np.arange(num_mel_bins, dtype=np.float64)[:, None]
    k = np.arange(num_ceps, dtype=np.float64)[None, :]
    dct = np.cos(math.pi / num_mel_bins * (n + 0.5) * k) * math.sqrt(2.0 / num_mel_bins)     # create_dct(N, N, "ortho")[:, :num_ceps]
    dct[:, 0] = math.sqrt(1.0 / num_mel_bins)
    c = f @ dct
    if cepstral_lifter != 0.0:
        c = c * (1.0 + 0.5 * cepstral_lifter * np.sin(math.pi * np.arange(num_ceps) / cepstral_lifter))[None, :]
    if use_energy:
        c[:, 0] = e
    if htk_compat:
        c0 = c[:, :1] * (1.0 if use_energy else math.sqrt(2.0))
        c = np.concatenate([c[:, 1:], c0], axis=1)
    return c - c.mean(axis=0, keepdims=True) if subtract_mean else c

"""Evaluate a golden-manifest case with the float64 CPU oracle (test-only helper)."""
import math

import numpy as np

from oracle import dsp_oracle as O


def _window(kind, n):
    if kind == "hamming":
        return 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(n) / n)
    return O.hann_window(n)


def spectrogram_case(kw, x):
    kw = dict(kw)
    n_fft = kw.get("n_fft", 400)
    wl = kw.get("win_length") or n_fft
    hop = kw.get("hop_length") or wl // 2
    win = _window(kw.get("window", "hann"), wl)
    return O.spectrogram(x, kw.get("pad", 0), win, n_fft, hop, wl, kw.get("power", 2.0),
                         kw.get("normalized", False), kw.get("center", True),
                         kw.get("pad_mode", "reflect"), kw.get("onesided", True))


def mel_case(kw, x):
    kw = dict(kw)
    sr = kw.get("sample_rate", 16000)
    n_fft = kw.get("n_fft", 400)
    wl = kw.get("win_length") or n_fft
    hop = kw.get("hop_length") or wl // 2
    n_mels = kw.get("n_mels", 128)
    f_max = kw.get("f_max")
    f_max = float(sr // 2) if f_max is None else f_max
    fb = O.melscale_fbanks(n_fft // 2 + 1, kw.get("f_min", 0.0), f_max, n_mels, sr,
                           kw.get("norm"), kw.get("mel_scale", "htk"))
    return O.mel_spectrogram(x, O.hann_window(wl), fb, n_fft, hop, wl, kw.get("pad", 0),
                             kw.get("power", 2.0), kw.get("normalized", False),
                             kw.get("center", True), kw.get("pad_mode", "reflect"))


def mfcc_case(kw, x):
    kw = dict(kw)
    sr = kw.get("sample_rate", 16000)
    mk = dict(kw.get("melkwargs") or {})
    mk["sample_rate"] = sr
    mel = mel_case(mk, x)
    n_mels = mk.get("n_mels", 128)
    dct = O.create_dct(kw.get("n_mfcc", 40), n_mels, kw.get("norm", "ortho"))
    if kw.get("log_mels", False):
        mel = np.log(mel + 1e-6)
    else:
        mel = O.amplitude_to_db(mel, 10.0, 1e-10, 0.0, 80.0)
    return np.swapaxes(np.swapaxes(mel, -1, -2) @ dct, -1, -2)


def db_case(kw, x):
    mult = 10.0 if kw["stype"] == "power" else 20.0
    return O.amplitude_to_db(x, mult, 1e-10, math.log10(max(1e-10, 1.0)), kw["top_db"])


def resample_case(kw, x):
    kw = dict(kw)
    return O.resample(x, kw.pop("orig_freq"), kw.pop("new_freq"), **kw)


def lfilter_case(kw, x, a, b):
    if kw.get("batching") is False:
        x = np.stack([x] * a.shape[0], -2)
    return O.lfilter(x, a, b, kw.get("clamp", True))


def evaluate(case, inputs):
    op, kw = case["op"], case["kwargs"]
    if op == "Spectrogram":
        return spectrogram_case(kw, inputs[0])
    if op == "MelSpectrogram":
        return mel_case(kw, inputs[0])
    if op == "MFCC":
        return mfcc_case(kw, inputs[0])
    if op == "AmplitudeToDB":
        return db_case(kw, inputs[0])
    if op == "MelScale":
        fb = O.melscale_fbanks(kw["n_stft"], 0.0, float(kw["sample_rate"] // 2), kw["n_mels"], kw["sample_rate"])
        return O.mel_scale(inputs[0], fb)
    if op in ("F.resample", "T.Resample"):
        return resample_case(kw, inputs[0])
    if op == "lfilter":
        return lfilter_case(kw, *inputs)
    if op == "biquad":
        return O.biquad(inputs[0], kw["b0"], kw["b1"], kw["b2"], kw["a0"], kw["a1"], kw["a2"])
    if op in ("fftconvolve", "T.FFTConvolve"):
        return O.fftconvolve(inputs[0], inputs[1], kw["mode"])
    if op == "melscale_fbanks":
        return O.melscale_fbanks(**kw)
    if op == "create_dct":
        return O.create_dct(**kw)
    if op in ("sinc_kernel_transform", "sinc_kernel_functional_f32"):
        kw = dict(kw)
        o, n = kw.pop("orig_freq"), kw.pop("new_freq")
        kw.pop("width")
        k, _ = O.sinc_resample_kernel(o, n, math.gcd(o, n), **kw)
        return k[:, None, :]
    return None

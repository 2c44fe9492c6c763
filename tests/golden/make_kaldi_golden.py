#!/usr/bin/env python
"""Fixtures for the Kaldi-compatible front-end from the REFERENCE's compliance/kaldi.py on the CPU.
Run only in the build container:   python tests/golden/make_kaldi_golden.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
import torchaudio.compliance.kaldi as K  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))
g = torch.Generator().manual_seed(7)
t = torch.arange(24000) / 16000.0
wav = (0.3 * torch.sin(2 * torch.pi * 300 * t) + 0.1 * torch.sin(2 * torch.pi * 2500 * t + 1.0)
       + 0.05 * torch.randn(24000, generator=g) + 0.02)                  # DC offset on purpose
wav = torch.stack([wav, 0.5 * torch.randn(24000, generator=g)]) * 32768.0   # Kaldi reads 16-bit scaled samples
out = {"wav": wav.numpy()}
cases = {
    "fbank_default": ("fbank", dict()),
    "fbank_80_energy": ("fbank", dict(num_mel_bins=80, use_energy=True, channel=1)),
    "fbank_htk_nopower": ("fbank", dict(num_mel_bins=40, use_energy=True, htk_compat=True, use_power=False, raw_energy=False)),
    "fbank_nolog_nosnip": ("fbank", dict(use_log_fbank=False, snip_edges=False, remove_dc_offset=False, preemphasis_coefficient=0.0,
                                          window_type="hamming", energy_floor=0.0)),
    "fbank_vtln": ("fbank", dict(num_mel_bins=30, vtln_warp=1.1, low_freq=40.0, high_freq=-200.0, subtract_mean=True)),
    "fbank_22k": ("fbank", dict(sample_frequency=22050.0, num_mel_bins=64, window_type="blackman")),
    "fbank_44k": ("fbank", dict(sample_frequency=44100.0, num_mel_bins=40, window_type="hanning", snip_edges=False)),
    "fbank_8k": ("fbank", dict(sample_frequency=8000.0, num_mel_bins=23, use_energy=True)),
    "spec_default": ("spectrogram", dict()),
    "spec_rect_nosnip": ("spectrogram", dict(window_type="rectangular", snip_edges=False, raw_energy=False, subtract_mean=True)),
    "mfcc_default": ("mfcc", dict()),
    "mfcc_energy_htk": ("mfcc", dict(use_energy=True, htk_compat=True, num_ceps=20, num_mel_bins=40, cepstral_lifter=0.0)),
    "mfcc_htk_noenergy": ("mfcc", dict(htk_compat=True, channel=1, subtract_mean=True)),
    # round 2: padded window = frame length (round_to_power_of_two=False), any even size
    "fbank_np2_16k_400": ("fbank", dict(round_to_power_of_two=False, num_mel_bins=40, use_energy=True)),
    "spec_np2_16k_400": ("spectrogram", dict(round_to_power_of_two=False, snip_edges=False)),
    "mfcc_np2_8k_200": ("mfcc", dict(round_to_power_of_two=False, sample_frequency=8000.0, channel=1)),
    "fbank_np2_44k_1102": ("fbank", dict(round_to_power_of_two=False, sample_frequency=44100.0, num_mel_bins=64,
                                         raw_energy=False, use_energy=True, window_type="hamming")),
    "fbank_np2_22k_551pad": ("fbank", dict(round_to_power_of_two=False, sample_frequency=22050.0, frame_length=25.04,
                                           num_mel_bins=30)),        # 552 samples = 2^3 * 3 * 23: a prime radix stage
}
# dither: the reference draws torch.randn(frames.shape); the recorded draw is stored so that the device path can be fed the
# SAME noise (audio_amd.compliance.kaldi._randn is the substitution point)
DITHER = {"fbank_dither": ("fbank", dict(dither=1.0, num_mel_bins=40)),
          "spec_dither_np2": ("spectrogram", dict(dither=0.5, round_to_power_of_two=False, snip_edges=False)),
          "mfcc_dither": ("mfcc", dict(dither=2.0, use_energy=True))}
meta = {}
for name, (fn, kw) in cases.items():
    y = getattr(K, fn)(wav, **kw)
    out[name] = y.numpy()
    meta[name] = {"fn": fn, "kw": kw}
    print(name, tuple(y.shape), float(y.abs().max()))
_real_randn = torch.randn
for name, (fn, kw) in DITHER.items():
    drawn = []

    def recording_randn(*a, **k):
        r = _real_randn(*a, **k)
        drawn.append(r.clone())
        return r
    torch.manual_seed(1234)
    torch.randn = recording_randn
    try:
        y = getattr(K, fn)(wav, **kw)
    finally:
        torch.randn = _real_randn
    assert len(drawn) == 1
    out[name] = y.numpy()
    out[f"noise/{name}"] = drawn[0].numpy()
    meta[name] = {"fn": fn, "kw": kw, "noise": f"noise/{name}"}
    print(name, tuple(y.shape), tuple(drawn[0].shape))
# host-side constants of the reference, for bit-exact comparison with audio_amd/_host.py
for tag, args in {"banks_default": (23, 512, 16000.0, 20.0, 0.0, 100.0, -500.0, 1.0),
                  "banks_80": (80, 512, 16000.0, 20.0, -400.0, 100.0, -500.0, 1.0),
                  "banks_vtln": (30, 512, 16000.0, 40.0, -200.0, 100.0, -500.0, 1.1),
                  "banks_44k": (40, 2048, 44100.0, 20.0, 0.0, 100.0, -500.0, 0.9)}.items():
    bins, centers = K.get_mel_banks(*args)
    out[f"const/{tag}/bins"] = bins.numpy(); out[f"const/{tag}/centers"] = centers.numpy()
    out[f"const/{tag}/args"] = np.array(args, dtype=np.float64)
for wt in ("povey", "hanning", "hamming", "rectangular", "blackman"):
    out[f"const/win_{wt}"] = K._feature_window_function(wt, 400, 0.42, torch.device("cpu"), torch.float32).numpy()
out["const/dct_13_23"] = K._get_dct_matrix(13, 23).numpy()
out["const/lifter_13_22"] = K._get_lifter_coeffs(13, 22.0).numpy()
out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
np.savez_compressed(os.path.join(HERE, "kaldi_goldens.npz"), **out)
print("kaldi_goldens.npz", os.path.getsize(os.path.join(HERE, "kaldi_goldens.npz")) // 1024, "KiB")

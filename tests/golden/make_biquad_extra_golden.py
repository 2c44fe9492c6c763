#!/usr/bin/env python
"""Fixtures for the two biquad designers with fixed tables (deemph_biquad, riaa_biquad) from the REFERENCE on the CPU: the six
coefficients each design hands to `biquad` (captured by wrapping the reference's own `biquad`) and the filtered noise.
Run only in the build container:   python tests/golden/make_biquad_extra_golden.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
import torchaudio.functional.filtering as RF  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))
g = torch.Generator().manual_seed(2024)
x = (0.3 * torch.randn(2, 6000, generator=g)).clamp_(-1, 1)
out = {"noise": x.numpy()}
captured = []
real_biquad = RF.biquad


def spy(waveform, b0, b1, b2, a0, a1, a2):
    captured.append([float(v) for v in (b0, b1, b2, a0, a1, a2)])
    return real_biquad(waveform, b0, b1, b2, a0, a1, a2)


RF.biquad = spy
for name, fn, rates in (("deemph", RF.deemph_biquad, (44100, 48000)), ("riaa", RF.riaa_biquad, (44100, 48000, 88200, 96000))):
    for sr in rates:
        y = fn(x, sr)
        out[f"{name}_{sr}"] = y.numpy()
        out[f"{name}_{sr}_coeffs"] = np.array(captured[-1], dtype=np.float64)
RF.biquad = real_biquad
np.savez_compressed(os.path.join(HERE, "biquad_extra_goldens.npz"), **out)
print({k: v.shape for k, v in out.items()})

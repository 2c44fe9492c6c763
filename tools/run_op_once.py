#!/usr/bin/env python
"""Launch one op a few times at its BASELINE shape (for rocprofv3 counter passes).
  python tools/run_op_once.py {mel|spec|mfcc|resample|lfilter|fftconv} [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_amd.functional as F
import audio_amd.transforms as T

op = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)


def noise(*shape):
    return (0.5 * torch.randn(*shape, device=dev, generator=g)).clamp_(-1, 1)


if op == "mel":
    x, t = noise(256, 160000), T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
    fn = lambda: t(x)
elif op == "spec":
    x, t = noise(256, 160000), T.Spectrogram(n_fft=400, hop_length=160).to(dev)
    fn = lambda: t(x)
elif op == "spec512":
    x, t = noise(256, 160000), T.Spectrogram(n_fft=512, hop_length=128).to(dev)
    fn = lambda: t(x)
elif op == "mel1024":
    x, t = noise(256, 160000), T.MelSpectrogram(sample_rate=16000, n_fft=1024, hop_length=256, n_mels=128).to(dev)
    fn = lambda: t(x)
elif op == "istft400":
    x = noise(256, 160000)
    X = T.Spectrogram(n_fft=400, hop_length=160, power=None).to(dev)(x)
    t = T.InverseSpectrogram(n_fft=400, hop_length=160).to(dev)
    fn = lambda: t(X, 160000)
elif op == "mfcc":
    x = noise(512, 160000)
    t = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
    fn = lambda: t(x)
elif op == "resample":
    x = noise(128, 2, 1323000)
    t = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                   rolloff=0.9475937167399596, beta=14.769656459379492).to(dev)
    fn = lambda: t(x)
elif op == "lfilter":
    x = torch.rand(32, 8, 480000, device=dev, generator=g) - 0.5
    a = torch.tensor([1.0, -1.2, 0.5], device=dev)
    b = torch.tensor([0.1, 0.2, 0.1], device=dev)
    fn = lambda: F.lfilter(x, a, b)
elif op == "fftconv":
    x = torch.rand(32, 8, 480000, device=dev, generator=g) - 0.5
    rir = torch.randn(1, 1, 24000, device=dev, generator=g) * 0.05
    fn = lambda: F.fftconvolve(x, rir)
elif op == "cascade":
    x = torch.rand(32, 8, 480000, device=dev, generator=g) - 0.5
    a = torch.tensor([[1.0, -1.2, 0.5], [1.0, -0.9, 0.3], [1.0, -0.5, 0.2], [1.0, -0.2, 0.1]], device=dev)
    b = torch.tensor([[0.1, 0.2, 0.1], [0.2, 0.3, 0.2], [0.3, 0.2, 0.1], [0.2, 0.1, 0.05]], device=dev)
    fn = lambda: F.biquad_cascade(x, a, b)
else:
    raise SystemExit("unknown op")
with torch.no_grad():
    for _ in range(n):
        y = fn()
torch.cuda.synchronize()
print("ok", tuple(y.shape))

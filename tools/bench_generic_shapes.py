#!/usr/bin/env python
"""MelSpectrogram / Spectrogram on shapes served by the GENERIC kernel (256 x 10 s @16 kHz), vs the algorithmic bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_amd.transforms as T


def timeit(fn, warm=20, iters=100):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = torch.device("cuda")
x = (0.5 * torch.randn(256, 160000, device=dev)).clamp_(-1, 1)
import warnings
warnings.simplefilter("ignore")
with torch.no_grad():
    for n_fft, hop, n_mels in [(400, 160, 80), (400, 200, 128), (400, 100, 80), (512, 160, 80), (512, 128, 80), (1024, 256, 128),
                               (2048, 512, 128), (320, 160, 40), (480, 160, 64)]:
        m = T.MelSpectrogram(sample_rate=16000, n_fft=n_fft, hop_length=hop, n_mels=n_mels).to(dev)
        y = m(x)
        us = timeit(lambda: m(x))
        by = x.numel() * 4 + y.numel() * 4
        print(f"mel  n_fft={n_fft:5d} hop={hop:4d} n_mels={n_mels:4d}: {us:9.1f} us  {by / us / 1e3:8.1f} GB/s algorithmic", flush=True)
    for n_fft, hop in [(512, 128), (1024, 256)]:
        s = T.Spectrogram(n_fft=n_fft, hop_length=hop).to(dev)
        y = s(x)
        us = timeit(lambda: s(x), 5, 30)
        by = x.numel() * 4 + y.numel() * 4
        print(f"spec n_fft={n_fft:5d} hop={hop:4d}             : {us:9.1f} us  {by / us / 1e3:8.1f} GB/s algorithmic", flush=True)

#!/usr/bin/env python
"""Launch the headline MelSpectrogram a few times (for rocprofv3 counter passes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_amd.transforms as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
x = (0.5 * torch.randn(256, 160000, device=dev, generator=g)).clamp_(-1, 1)
mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
with torch.no_grad():
    for _ in range(n):
        y = mel(x)
torch.cuda.synchronize()
print("ok", tuple(y.shape))

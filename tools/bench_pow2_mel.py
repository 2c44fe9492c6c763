#!/usr/bin/env python
"""MelSpectrogram on the power-of-two wave-FFT kernel (stft_pow2.h): us per call on the cfg2 batch (256 x 10 s @16 kHz), tools only.
    python tools/bench_pow2_mel.py            (AAMD_USE_LAB_LIB=1 AAMD_NO_TORCH_SHIM=1 for a lab build of the library)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T

CASES = [(400, 160, 80), (512, 160, 80), (512, 128, 64), (1024, 256, 80), (1024, 256, 128), (2048, 512, 128), (256, 64, 40)]


def main():
    global CASES
    if os.environ.get("P2_ONLY"):                      # e.g. P2_ONLY=512,160,80 (PMC runs)
        CASES = [tuple(int(v) for v in os.environ["P2_ONLY"].split(","))]
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(3)
    xs = [(0.5 * torch.randn(256, 160000, device=dev, generator=g)).clamp_(-1, 1) for _ in range(3)]
    for n_fft, hop, n_mels in CASES:
        m = T.MelSpectrogram(16000, n_fft=n_fft, hop_length=hop, n_mels=n_mels).to(dev)
        with torch.no_grad():
            for i in range(30):
                y = m(xs[i % 3])
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(60):
                    y = m(xs[i % 3])
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 60 * 1e3)
        byts = xs[0].numel() * 4 + y.numel() * 4
        print(json.dumps({"n_fft": n_fft, "hop": hop, "n_mels": n_mels, "us": round(best, 1), "frac_hbm": round(byts / (best * 1e-6) / 8e12, 3),
                          "checksum": float(y.double().sum())}), flush=True)


if __name__ == "__main__":
    main()

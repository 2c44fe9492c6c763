#!/usr/bin/env python
"""Barrier / phase ablation of the fused 4-biquad cascade (cfg5a shard): AAMD_LFW_LAB selects variants of
lfilter_wave_kernel (wrong results by design).  One variant per process:  AAMD_LFW_LAB=N python tools/lfw_lab.py"""
import math, os, sys
# the tools-only kernel variants live in libaudio_amd_lab.so (python -m audio_amd._build --lab), reached through ctypes
os.environ.setdefault("AAMD_USE_LAB_LIB", "1")
os.environ.setdefault("AAMD_NO_TORCH_SHIM", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.functional as F

dev = torch.device("cuda")
x = torch.rand(32, 8, 480000, device=dev) - 0.5
A, B = [], []
for fc in (8000.0, 6000.0, 4000.0, 3000.0):
    w0 = 2 * math.pi * fc / 48000
    alpha = math.sin(w0) / 2 / 0.707
    A.append([1 + alpha, -2 * math.cos(w0), 1 - alpha])
    B.append([(1 - math.cos(w0)) / 2, 1 - math.cos(w0), (1 - math.cos(w0)) / 2])
a4, b4 = torch.tensor(A, device=dev), torch.tensor(B, device=dev)
with torch.no_grad():
    for _ in range(10):
        F.biquad_cascade(x, a4, b4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        F.biquad_cascade(x, a4, b4)
    e1.record()
    torch.cuda.synchronize()
print("AAMD_LFW_LAB=%s: %.1f us per launch" % (os.environ.get("AAMD_LFW_LAB", "0"), e0.elapsed_time(e1) / 40 * 1e3))

#!/usr/bin/env python
"""One-kernel MFCC on the cfg4 noise batch under the epilogue's lab switches (AAMD_MFCC_LAB: 1 no fragment loads, 2 no MFMA,
4 no tile minimum, 8 no stores; read once per process): us per call.  Run once per value:
    for v in 0 1 2 4 8 15; do AAMD_MFCC_LAB=$v python tools/bench_mfcc_lab.py; done"""
import json, os, sys
# the tools-only kernel variants live in libaudio_amd_lab.so (python -m audio_amd._build --lab), reached through ctypes
os.environ.setdefault("AAMD_USE_LAB_LIB", "1")
os.environ.setdefault("AAMD_NO_TORCH_SHIM", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
with torch.no_grad():
    xs = [(0.5 * torch.randn(512, 160000, device=dev, generator=g)).clamp_(-1, 1) for _ in range(3)]
    out = {"AAMD_MFCC_LAB": os.environ.get("AAMD_MFCC_LAB", "0")}
    for fused in (True, False):
        m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
        m.fused = fused
        for i in range(40):
            m(xs[i % 3])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100):
            m(xs[i % 3])
        e1.record()
        torch.cuda.synchronize()
        out["fused" if fused else "two_kernel"] = round(e0.elapsed_time(e1) * 10, 1)
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
    for i in range(40):
        mel(xs[i % 3])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100):
        mel(xs[i % 3])
    e1.record()
    torch.cuda.synchronize()
    out["mel_only_512"] = round(e0.elapsed_time(e1) * 10, 1)
    print(json.dumps(out), flush=True)

#!/usr/bin/env python
"""Is the MFCC module call (cfg4 batch) bound by the host or by the GPU?  Per call: the time the host needs to ISSUE it (loop
without synchronisation, clock stopped before the final synchronize) against the time until the GPU has finished."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
xs = [(0.5 * torch.randn(512, 160000, device=dev, generator=g)).clamp_(-1, 1) for _ in range(3)]
out = {}
with torch.no_grad():
    for name, mk in (("mfcc_fused", lambda: T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80))),
                     ("mel", lambda: T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80))):
        m = mk().to(dev)
        if name == "mfcc_fused":
            m.fused = True
        for i in range(100):
            m(xs[i % 3])
        torch.cuda.synchronize()
        res = []
        for r in range(3):
            t0 = time.perf_counter()
            for i in range(300):
                m(xs[i % 3])
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res.append({"issue_us": round((t1 - t0) / 300 * 1e6, 1), "done_us": round((t2 - t0) / 300 * 1e6, 1)})
        out[name] = res
print(json.dumps(out))

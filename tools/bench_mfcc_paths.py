#!/usr/bin/env python
"""cfg4 (MFCC n_mfcc = 40, 80 mels, 512 x 10 s @ 16 kHz): the one-kernel path (+ fix-up launch) against the exact two-kernel
path, on batches with no / some / mostly clamped tiles.  One JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
ALGO = 512 * 160000 * 4 + 512 * 1001 * 40 * 4


def timed(fn, warmup=150, steps=150):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


with torch.no_grad():
    x = (0.5 * torch.randn(512, 160000, device=dev, generator=g)).clamp_(-1, 1)
    cases = {"noise (nothing clamped)": x}
    x10 = x.clone(); x10[:51, 80000:] = 0.0
    cases["5 % of the tiles digital silence"] = x10
    x50 = x.clone(); x50[:, 80000:] = 0.0
    cases["50 % of the tiles digital silence"] = x50
    # zero-padded tails of every clip (a length-sorted, padded batch): where does the two-kernel path take over?
    for pct in (10, 15, 20, 25, 30, 40):
        xp = x.clone(); xp[:, int(160000 * (1 - pct / 100.0)):] = 0.0
        cases[f"every clip: last {pct} % zero padding"] = xp
    for name, inp in cases.items():
        rec = {"case": name}
        for label, fused in (("fused", True), ("two_kernel", False)) + ((("auto", "auto"),) if "padding" not in name else ()):
            m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
            m.fused = fused
            ring = [inp, inp.clone(), inp.clone()]          # 3 x 328 MB in + outputs: beyond the 256 MiB Infinity Cache
            it = [0]

            def step():
                it[0] += 1
                return m(ring[it[0] % 3])
            us = timed(step)
            rep = m.fused_report()
            rec[label] = {"us": round(us, 1), "frac_of_hbm_peak": round(ALGO / us / 8e6, 3), "path": rep["path"],
                          "redone_share": rep["redone_share"]}
        print(json.dumps(rec), flush=True)

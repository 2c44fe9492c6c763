#!/usr/bin/env python
"""lfilter / biquad cascades across batch shapes and orders: ms per launch and fraction of the HBM peak (read + write once)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.functional as F

dev = torch.device("cuda")


def timed(fn, warmup=5, steps=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def stable(order, gen):
    # poles at radius < 1: product of first / second order sections
    import numpy as np
    a = np.array([1.0])
    rng = np.random.default_rng(order)
    left = order
    while left > 0:
        if left >= 2:
            r, th = rng.uniform(0.5, 0.95), rng.uniform(0.1, 3.0)
            a = np.convolve(a, [1.0, -2 * r * np.cos(th), r * r]); left -= 2
        else:
            a = np.convolve(a, [1.0, -rng.uniform(-0.9, 0.9)]); left -= 1
    b = rng.standard_normal(order + 1) * 0.2
    return torch.tensor(a, dtype=torch.float32, device=dev), torch.tensor(b, dtype=torch.float32, device=dev)


with torch.no_grad():
    for (rows, n), order in [((256, 480000), 1), ((256, 480000), 2), ((256, 480000), 4), ((256, 480000), 8), ((256, 480000), 16),
                             ((4096, 16000), 2), ((8, 4800000), 2), ((32, 160000), 2), ((65536, 4000), 2)]:
        x = (torch.rand(rows, n, device=dev) - 0.5)
        a, b = stable(order, None)
        ms = timed(lambda: F.lfilter(x, a, b, clamp=True))
        print(json.dumps({"rows": rows, "n": n, "order": order, "ms": round(ms, 4),
                          "frac_of_hbm_peak": round(2 * x.numel() * 4 / (ms * 1e-3) / 8e12, 3)}), flush=True)
    x = (torch.rand(256, 480000, device=dev) - 0.5)
    for name, fn in [("lowpass_biquad", lambda: F.lowpass_biquad(x, 48000, 4000.0)),
                     ("highpass_biquad", lambda: F.highpass_biquad(x, 48000, 100.0)),
                     ("deemph_biquad", lambda: F.deemph_biquad(x, 48000)),
                     ("equalizer_biquad", lambda: F.equalizer_biquad(x, 48000, 1000.0, 3.0))]:
        if not hasattr(F, name):
            continue
        ms = timed(fn)
        print(json.dumps({"op": name, "rows": 256, "n": 480000, "ms": round(ms, 4),
                          "frac_of_hbm_peak": round(2 * x.numel() * 4 / (ms * 1e-3) / 8e12, 3)}), flush=True)

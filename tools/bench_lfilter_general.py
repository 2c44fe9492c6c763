#!/usr/bin/env python
"""The general-order lfilter kernel (float64 state, csrc/lfilter.h) beside the second-order-section route on the cfg5a
batch shape (256 rows x 480 000 samples): ms per launch, fraction of the HBM peak (read + write once), and the forward +
backward of a learnable filter (the path that always takes the general-order kernel)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from scipy import signal
import audio_amd.functional as F

dev = torch.device("cuda")


def timed(fn, warmup=3, steps=10):
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


designs = {4: signal.butter(4, 0.2), 6: signal.cheby1(6, 1, 0.1), 8: signal.ellip(8, 0.5, 60, 0.3),
           12: signal.butter(6, [0.2, 0.5], "bandpass"), 16: signal.butter(8, [0.25, 0.6], "bandpass")}
x = (torch.rand(256, 480000, device=dev) - 0.5)
nbytes = 2 * x.numel() * 4
for order, (b, a) in designs.items():
    at, bt = torch.tensor(a, dtype=torch.float32, device=dev), torch.tensor(b, dtype=torch.float32, device=dev)
    row = {"order": order}
    for route in ("sections", "general"):
        prev = F.set_lfilter_sections(route == "sections")
        try:
            with torch.no_grad():
                if route == "sections" and F._lfilter_sections(at, bt, at.reshape(1, -1), bt.reshape(1, -1)) is None:
                    row[route] = None
                    continue
                ms = timed(lambda: F.lfilter(x, at, bt, clamp=True))
        finally:
            F.set_lfilter_sections(prev)
        row[route + "_ms"] = round(ms, 4)
        row[route + "_frac_hbm"] = round(nbytes / (ms * 1e-3) / 8e12, 3)
    # learnable coefficients: forward + backward (dx, da, db) on 32 rows
    xs = x[:32].clone().requires_grad_(True)
    ag, bg = at.clone().requires_grad_(True), bt.clone().requires_grad_(True)

    def step():
        y = F.lfilter(xs, ag, bg, clamp=False)
        y.sum().backward()
        xs.grad = ag.grad = bg.grad = None
    row["learnable_fwd_bwd_32rows_ms"] = round(timed(step, 2, 5), 4)
    print(json.dumps(row), flush=True)

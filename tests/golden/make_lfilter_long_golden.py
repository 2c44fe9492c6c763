#!/usr/bin/env python
"""Fixtures for biquads whose impulse response outlives the 2048 samples a wave of the biquad kernels owns (|pole| up to
0.9999), from the REFERENCE on the CPU with its own compiled core loop (oracle/build_ref.py binds src/libtorchaudio/lfilter.cpp
as `torchaudio.functional.filtering._lfilter_core_loop`): single resonators without clamp and a 4-stage cascade with the
reference's default clamp between the stages.  Run only in the build container:
    python tests/golden/make_lfilter_long_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

RF = build_ref.load()          # the reference's functional.filtering with its native core loop bound
HERE = os.path.dirname(os.path.abspath(__file__))
g = torch.Generator().manual_seed(77)
x = 0.1 * torch.randn(3, 12000, generator=g)
designs = [(0.9995, 0.3), (0.9999, 2.0), (0.999, 1.2), (0.99, 0.05)]
out = {"noise": x.numpy(), "designs": np.array(designs, dtype=np.float64)}
A, B = [], []
for i, (r, th) in enumerate(designs):
    a = torch.tensor([1.0, -2 * r * np.cos(th), r * r], dtype=torch.float32)
    b = torch.tensor([1 - r, 0.0, 0.0], dtype=torch.float32)
    A.append(a)
    B.append(b)
    out[f"single_{i}"] = RF.lfilter(x, a, b, clamp=False).numpy()
y = x
for a, b in zip(A, B):
    y = RF.lfilter(y, a, b, clamp=True)
out["cascade_clamped"] = y.numpy()
out["a"] = torch.stack(A).numpy()
out["b"] = torch.stack(B).numpy()
np.savez_compressed(os.path.join(HERE, "lfilter_long_goldens.npz"), **out)
print({k: v.shape for k, v in out.items()})

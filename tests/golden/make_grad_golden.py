#!/usr/bin/env python
"""Gradient fixtures for F.lfilter from the REFERENCE's autograd (functional/filtering.py:941-1024) on the CPU.
Run only in the build container:   PYTHONPATH=/root/reference/src python tests/golden/make_grad_golden.py
Loss = sum(y * r) with a fixed random r, so dL/dy = r; stores inputs, r, y and the three gradients."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import build_ref  # noqa: E402
build_ref.load()           # the reference's compiled CPU core (the Python fallback is not differentiable)
import torchaudio.functional as F  # noqa: E402  (the reference)

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
g = torch.Generator().manual_seed(2024)
cases = {
    "shared_clamp": dict(shape=(2, 2, 300), a=[1.0, -1.2, 0.5], b=[0.8, 0.9, 0.4], clamp=True, scale=0.12),
    "shared_noclamp": dict(shape=(3, 257), a=[0.9, -0.7, 0.2], b=[0.3, -0.1, 0.2], clamp=False, scale=0.5),
    "per_channel": dict(shape=(2, 2, 300), a=[[1.0, -1.5, 0.7], [1.1, -0.4, 0.1]], b=[[0.2, 0.1, 0.05], [0.5, 0.3, -0.2]],
                        clamp=True, scale=0.9),
    "order4": dict(shape=(2, 200), a=[1.0, -0.9, 0.5, -0.2, 0.05], b=[0.2, 0.1, 0.3, -0.1, 0.05], clamp=False, scale=0.5),
}
for name, c in cases.items():
    x = (c["scale"] * torch.randn(*c["shape"], generator=g, dtype=torch.float64)).requires_grad_()
    a = torch.tensor(c["a"], dtype=torch.float64, requires_grad=True)
    b = torch.tensor(c["b"], dtype=torch.float64, requires_grad=True)
    r = torch.randn(*c["shape"], generator=g, dtype=torch.float64)
    y = F.lfilter(x, a, b, clamp=c["clamp"])
    (y * r).sum().backward()
    out[f"{name}/x"] = x.detach().numpy(); out[f"{name}/a"] = a.detach().numpy(); out[f"{name}/b"] = b.detach().numpy()
    out[f"{name}/r"] = r.numpy(); out[f"{name}/y"] = y.detach().numpy(); out[f"{name}/clamp"] = np.array(int(c["clamp"]))
    out[f"{name}/dx"] = x.grad.numpy(); out[f"{name}/da"] = a.grad.numpy(); out[f"{name}/db"] = b.grad.numpy()
    print(name, "clamped:", int(((y.detach().abs() >= 1.0)).sum()), "of", y.numel())
np.savez_compressed(os.path.join(HERE, "reference_grads.npz"), **out)
print("reference_grads.npz:", len(out), "arrays")

#!/usr/bin/env python
"""Resample across common rate pairs (the band width KS of the MFMA kernel differs per pair): ms per launch and fraction of the
HBM peak on a 128 x stereo x 30 s shard.  One JSON line per pair."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T

dev = torch.device("cuda")
pairs = [(48000, 16000, {}), (16000, 8000, {}), (44100, 48000, {}), (48000, 44100, {}), (8000, 16000, {}), (44100, 16000, {}),
         (44100, 16000, dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596,
                             beta=14.769656459379492))]
with torch.no_grad():
    for o, n, kw in pairs:
        x = (0.5 * torch.randn(128, 2, 30 * o, device=dev)).clamp_(-1, 1)
        rs = T.Resample(o, n, **kw).to(dev)
        y = rs(x)
        for _ in range(5):
            rs(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rs(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        algo = (x.numel() + y.numel()) * 4
        print(json.dumps({"orig": o, "new": n, "kaiser_best": bool(kw), "ms": round(ms, 4),
                          "frac_of_hbm_peak": round(algo / (ms * 1e-3) / 8e12, 3)}), flush=True)

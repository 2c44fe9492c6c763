#!/usr/bin/env python
"""cfg3 shard (Resample 44.1k -> 16k kaiser_best, 128 x stereo x 30 s): the binary16 hi/lo-split MFMA kernel against the
fp32 MFMA kernel (AAMD_POLICY_RESAMPLE_FP32)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T
from audio_amd import _lib

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
x = (0.5 * torch.randn(128, 2, 1323000, device=dev, generator=g)).clamp_(-1, 1)
rs = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                rolloff=0.9475937167399596, beta=14.769656459379492).to(dev)
ALGO = x.numel() * 4 + 128 * 2 * 480000 * 4
FLOPS = 128 * 2 * 480000 * 373 * 2


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


with torch.no_grad():
    for name, flags in (("binary16 hi/lo split (default)", 0), ("fp32 MFMA", _lib.POLICY_RESAMPLE_FP32)):
        with _lib.kernel_policy(flags):
            ms = timed(lambda: rs(x))
        print(json.dumps({"kernel": name, "ms_per_shard": round(ms, 4), "audio_sec_per_sec": round(128 * 30.0 / (ms * 1e-3)),
                          "frac_of_hbm_peak": round(ALGO / (ms * 1e-3) / 8e12, 3),
                          "effective_TFLOPs": round(FLOPS / (ms * 1e-3) / 1e12, 1)}), flush=True)

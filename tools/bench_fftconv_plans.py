#!/usr/bin/env python
"""Overlap-save fftconvolve: the frequency-domain delay-line plan against the recompute plan (policy switches), per tap count
and batch shape; what the launcher's cost model picks on its own.  One JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.functional as F
from audio_amd import _lib

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


cases = [((256, 480000), 700), ((256, 480000), 2400), ((256, 480000), 8192), ((256, 480000), 9000), ((256, 480000), 12000), ((256, 480000), 16000), ((256, 480000), 24000), ((256, 480000), 30000),
         ((32, 480000), 24000), ((8, 1920000), 24000), ((1024, 120000), 24000), ((256, 100000), 24000)]
with torch.no_grad():
    for (rows, nx), taps in cases:
        x = torch.rand(rows, nx, device=dev) - 0.5
        h = torch.randn(1, taps, device=dev) * 0.01
        rec = {"rows": rows, "nx": nx, "taps": taps}
        if taps > 8192:
            with _lib.kernel_policy(_lib.POLICY_FFTCONV_FDL):
                rec["delay_line_ms"] = round(timed(lambda: F.fftconvolve(x, h)), 4)
        with _lib.kernel_policy(_lib.POLICY_FFTCONV_NO_FDL):
            rec["recompute_ms"] = round(timed(lambda: F.fftconvolve(x, h)), 4)
        rec["default_ms"] = round(timed(lambda: F.fftconvolve(x, h)), 4)
        rec["default_plan"] = {1: "recompute", 2: "complex-block delay line", 3: "real-block delay line (round 4)"}[
            _lib.lib().aamd_fftconvolve_plan(rows, nx, taps, nx + taps - 1)]
        with _lib.kernel_policy(_lib.POLICY_FFTCONV_COMPLEX):
            rec["cost_model_among_complex_plans"] = {1: "recompute", 2: "delay line"}[
                _lib.lib().aamd_fftconvolve_plan(rows, nx, taps, nx + taps - 1)]
        print(json.dumps(rec), flush=True)

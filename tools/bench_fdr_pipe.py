#!/usr/bin/env python
"""A/B of the two walks of the real-block delay line (csrc/fftconv_fdr.h): AAMD_FDR_PIPE=0 (one spectrum buffer, 7 barriers
per block) against 1 (two buffers, inverse of block j beside the forward of block j + 1, 4 barriers).  Each setting runs in its
own process (the launcher reads the variable once); every child checks its result against a float64 FFT convolution first."""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    import audio_amd.functional as F
    from audio_amd import _lib
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(4321)
    out = {"pipe": os.environ.get("AAMD_FDR_PIPE")}
    # correctness: 3 and 2 partitions, segments, odd offsets
    for (rows, nx, taps, mode) in ((5, 70001, 24000, "full"), (3, 50000, 9000, "same"), (1, 300000, 20000, "full")):
        x = torch.randn(rows, nx, device=dev, generator=g)
        h = torch.randn(1, taps, device=dev, generator=g) * torch.exp(-torch.arange(taps, device=dev) / (0.3 * taps))
        got = F.fftconvolve(x, h, mode)
        n = nx + taps - 1
        full = torch.fft.irfft(torch.fft.rfft(x.double(), n=n) * torch.fft.rfft(h.double(), n=n), n=n)
        if mode == "same":
            s0 = (n - nx) // 2
            full = full[..., s0:s0 + nx]
        err = float((got.double() - full).abs().max() / full.abs().max())
        assert _lib.lib().aamd_fftconvolve_plan(rows, nx, taps, got.shape[-1]) == 3
        assert err <= 1e-5, (rows, nx, taps, err)
        out[f"err_{rows}x{nx}_{taps}"] = err

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
        x = torch.rand(256, 480000, device=dev, generator=g) - 0.5
        for taps in (24000, 12000):
            h = torch.randn(1, taps, device=dev, generator=g) * 0.01
            out[f"ms_256x480000_{taps}"] = round(timed(lambda: F.fftconvolve(x, h)), 4)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for rep in range(2):
            for pipe in ("0", "1"):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, AAMD_FDR_PIPE=pipe),
                                   capture_output=True, text=True, timeout=400)
                print(r.stdout.strip() or ("FAILED pipe=%s: " % pipe + r.stderr[-1500:]), flush=True)

#!/usr/bin/env python
"""A/B of the n_fft = 400 kernel's tail pools (csrc/melspec400.h, pool_tile) through the product API: the BASELINE shapes cfg2
(MelSpectrogram 256 x 10 s), the Spectrogram of the same batch and cfg4 (MFCC 512 x 10 s), each with the pools (default) and
under AAMD_POLICY_MEL400_NO_POOL (static tile runs, what every launch did before round 5), interleaved round by round on four
rotating input batches; outputs compared bit for bit, and the pooled launch repeated for run-to-run stability.
AAMD_MEL400_POOL_P=<P> (environment, read once per process) overrides the launcher's share for a sweep.

The product is built WITHOUT the pools (the measured gain is 0.5 us of 70, profiles/r05_i_mel400_pool_sweep.txt): build
the experiment with  AAMD_EXTRA_HIPCC_FLAGS=-DAAMD_M400_POOLS=1 python -m audio_amd._build --force  first.  In the default build both
arms of this A/B run the same static hand-out."""
import json
import os
# the tools-only kernel variants live in libaudio_amd_lab.so (python -m audio_amd._build --lab), reached through ctypes
os.environ.setdefault("AAMD_USE_LAB_LIB", "1")
os.environ.setdefault("AAMD_NO_TORCH_SHIM", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T
from audio_amd import _lib


def main():
    dev = torch.device("cuda")
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    g = torch.Generator(device="cuda").manual_seed(0)
    work = [
        ("cfg2", T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80).to(dev), 256),
        ("spec", T.Spectrogram(n_fft=400, hop_length=160).to(dev), 256),
        ("cfg4", T.MFCC(16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev), 512),
    ]
    for name, mod, rows in work:
        xs = [(0.5 * torch.randn(rows, 160000, device=dev, generator=g)).clamp_(-1, 1) for _ in range(4)]
        with torch.no_grad():
            a = mod(xs[0]).clone()
            with _lib.kernel_policy(_lib.POLICY_MEL400_NO_POOL):
                b = mod(xs[0]).clone()
            same = bool(torch.equal(a, b))
            stable = True
            for i in range(12):
                y = mod(xs[i % 4])
                if i % 4 == 0:
                    stable = stable and bool(torch.equal(y, a))
            torch.cuda.synchronize()
            for i in range(400):              # clock ramp
                mod(xs[i % 4])
            torch.cuda.synchronize()
            res = {"pool": [], "static": []}
            for r in range(rounds):
                for key in ("pool", "static"):
                    pol = _lib.kernel_policy(_lib.POLICY_MEL400_NO_POOL) if key == "static" else None
                    if pol is not None:
                        pol.__enter__()
                    for i in range(20):
                        mod(xs[i % 4])
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for i in range(calls):
                        mod(xs[i % 4])
                    e1.record()
                    torch.cuda.synchronize()
                    if pol is not None:
                        pol.__exit__(None, None, None)
                    res[key].append(e0.elapsed_time(e1) * 1e3 / calls)
        print(json.dumps({"workload": name, "pool_share_env": os.environ.get("AAMD_MEL400_POOL_P"), "bit_equal_static": same,
                          "repeat_stable": stable,
                          "us_per_call_pool": [round(v, 2) for v in res["pool"]], "us_per_call_static": [round(v, 2) for v in res["static"]],
                          "mean_pool": round(sum(res["pool"]) / rounds, 2), "mean_static": round(sum(res["static"]) / rounds, 2)}), flush=True)
        del xs


if __name__ == "__main__":
    main()

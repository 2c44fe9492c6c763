#!/usr/bin/env python
"""Independent batches issued round-robin on S HIP streams (tools only): how much of the headline kernel's per-launch tail
(XCDs finish up to 7 % apart, workgroups 63.9 ... 71.4 us) and of the launch-to-launch gap a second queue fills.

  python tools/bench_streams.py [--op mel|mfcc] [--steps 1000] [--streams 1 2 3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="mel")
    ap.add_argument("--steps", type=int, nargs="+", default=[20, 1000])
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3])
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    if a.op == "mel":
        batch, ring = 256, 5
        mod = T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
    else:
        batch, ring = 512, 3
        mod = T.MFCC(16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
    g = torch.Generator(device=dev).manual_seed(1234)
    xs = [(0.5 * torch.randn(batch, 160000, device=dev, generator=g)).clamp_(-1, 1) for _ in range(ring)]
    ys = [None] * ring
    with torch.no_grad():
        for i in range(600):
            ys[i % ring] = mod(xs[i % ring])
        torch.cuda.synchronize()
        ref = [mod(x).clone() for x in xs]
        for rnd in range(a.rounds):
            for S in a.streams:
                streams = [torch.cuda.Stream(dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
                for K in a.steps:
                    for i in range(50):
                        with torch.cuda.stream(streams[i % S]):
                            ys[i % ring] = mod(xs[i % ring])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(K):
                        with torch.cuda.stream(streams[i % S]):
                            ys[i % ring] = mod(xs[i % ring])
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    ok = all(torch.equal(ys[j], ref[j]) for j in range(ring) if ys[j] is not None)
                    print(json.dumps({"op": a.op, "streams": S, "steps": K, "us_per_step": dt / K * 1e6, "round": rnd,
                                      "bit_equal_to_single_stream": ok}), flush=True)


if __name__ == "__main__":
    main()

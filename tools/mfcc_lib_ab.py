#!/usr/bin/env python
"""A/B of two builds of libaudio_amd.so on the fused MFCC (cfg4 batch), interleaved in ONE process on one box: both libraries are
loaded side by side through ctypes and the binding's handle is swapped between the timed blocks.
    python tools/mfcc_lib_ab.py audio_amd/lib/libaudio_amd.so audio_amd/lib/libaudio_amd_prev.so [--steps 200 --rounds 4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_amd.transforms as T
from audio_amd import _lib


def load(path):
    _lib._lib = None
    _lib.LIB_PATH = os.path.abspath(path)
    return _lib.lib()


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs=2)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    handles = [load(p) for p in args.libs]
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1234)
    base = [(0.5 * torch.randn(512, 160000, device=dev, generator=g)).clamp_(-1, 1) for _ in range(3)]
    print(f"# A = {args.libs[0]}, B = {args.libs[1]}; cfg4 batch, 3 input batches rotate; us per call, {args.rounds} interleaved rounds")
    for label, silent in (("nothing clamped", 0), ("26 clips in a row silent", 26), ("every clip: last 20 % silent", -1)):
        xs = []
        for b in base:
            x = b.clone()
            if silent > 0:
                x[100:100 + silent] = 0.0
            elif silent < 0:
                x[:, 128000:] = 0.0
            xs.append(x)
        for shape_label, view in (("(B, L)", lambda x: x), ("(B, 1, L)", lambda x: x[:, None, :])):
            m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
            m.fused = True
            it = [0]

            def step():
                it[0] += 1
                return m(view(xs[it[0] % 3]))

            res, outs, shares = [[], []], [], []
            with torch.no_grad():
                for k in (0, 1):
                    _lib._lib = handles[k]
                    outs.append(m(view(xs[0])).clone())
                    shares.append(m.fused_report()["redone_share"])
                for _ in range(args.rounds):
                    for k in (0, 1):
                        _lib._lib = handles[k]
                        res[k].append(timed(step, 100, args.steps))
            a, b = sorted(res[0]), sorted(res[1])
            print(f"{label:30s} {shape_label:10s} redone {shares[0]:.4f} / {shares[1]:.4f}  bit-equal {bool(torch.equal(outs[0], outs[1]))}  "
                  f"A {a[0]:7.1f} / {a[len(a) // 2]:7.1f}   B {b[0]:7.1f} / {b[len(b) // 2]:7.1f}   (min / median)  "
                  f"A - B {a[len(a) // 2] - b[len(b) // 2]:+6.1f} us", flush=True)


if __name__ == "__main__":
    main()

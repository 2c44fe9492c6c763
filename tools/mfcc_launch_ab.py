#!/usr/bin/env python
"""A/B of the fused MFCC's launch forms on the cfg4 batch (512 x 10 s @16 kHz, n_mfcc 40), interleaved in one process on one box:
  one   = aamd_mfcc_fused_f32 pass 2 as ONE launch (in-kernel grid barrier, strided fix-up shares)
  three = the same entry under AAMD_POLICY_MFCC_THREE_LAUNCHES (pass 0 + mfcc_fix_list_kernel + fix-up launch: rounds 3-4)
for a batch in which nothing reaches the cut-off, one with 5 % of the clips silent (clustered fix-up work) and one with half of
them silent, for ONE batch-global cut-off ((B, L) input) and per-item cut-offs ((B, 1, L)).
    python tools/mfcc_launch_ab.py [--steps 200] > profiles/rNN_mfcc_launch_ab.txt"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_amd.transforms as T
from audio_amd import _lib


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
    return e0.elapsed_time(e1) / steps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1234)
    base = [(0.5 * torch.randn(512, 160000, device=dev, generator=g)).clamp_(-1, 1) for _ in range(3)]
    print(f"# {torch.cuda.get_device_name(0)}; cfg4 batch 512 x 160000, 3 input batches rotate; us per call "
          f"(steps {args.steps}, {args.rounds} interleaved rounds: min / median)")
    for label, silent in (("nothing clamped", 0), ("5 % of the clips silent (26 clips in a row)", 26),
                          ("50 % of the clips silent", 256)):
        xs = []
        for b in base:
            x = b.clone()
            if silent:
                x[100:100 + silent] = 0.0
            xs.append(x)
        for shape_label, view in (("(B, L): one batch-global cut-off", lambda x: x),
                                  ("(B, 1, L): per-item cut-offs", lambda x: x[:, None, :])):
            m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
            m.fused = True
            it = [0]

            def step():
                it[0] += 1
                return m(view(xs[it[0] % 3]))

            res = {"one": [], "three": []}
            with torch.no_grad():
                y1 = m(view(xs[0])).clone()
                share = m.fused_report()["redone_share"]
                with _lib.kernel_policy(_lib.POLICY_MFCC_THREE_LAUNCHES):
                    y3 = m(view(xs[0])).clone()
                same = bool(torch.equal(y1, y3))
                for _ in range(args.rounds):
                    res["one"].append(timed(step, args.warmup, args.steps))
                    with _lib.kernel_policy(_lib.POLICY_MFCC_THREE_LAUNCHES):
                        res["three"].append(timed(step, args.warmup, args.steps))
                census = ""
                m(view(xs[1]))
                torch.cuda.synchronize()
                sync = m._fused_state.last_sync
                if sync is not None:
                    # lines of 32 dwords: arrival counter, one flag line per workgroup, one census line per workgroup whose first
                    # 8 dwords are four 100 MHz stamps (the full grid: as many workgroups as the chip has CUs)
                    w = sync.view(torch.int32).cpu().numpy().reshape(-1, 32)
                    n_wg = (w.shape[0] - 1) // 2
                    st = w[1 + n_wg:1 + 2 * n_wg, 0:8].copy().view("int64")   # (workgroups, 4): entry, pass done, barrier passed, checked
                    st = st[st[:, 1] > 0]
                    if len(st):
                        t0 = st[:, 0].min()
                        done, passed, checked = (st[:, 1] - t0) / 100.0, (st[:, 2] - t0) / 100.0, (st[:, 3] - t0) / 100.0
                        import numpy as np
                        census = (f"\n      census of one launch ({len(st)} workgroups, us after the first entry): entry spread "
                                  f"{(st[:, 0].max() - t0) / 100.0:.1f}; first pass done min / median / max "
                                  f"{done.min():.1f} / {np.median(done):.1f} / {done.max():.1f}; barrier passed median / max "
                                  f"{np.median(passed):.1f} / {passed.max():.1f}; candidates checked max {checked.max():.1f}")
            o, t = sorted(res["one"]), sorted(res["three"])
            print(f"{label:45s} {shape_label:36s} redone share {share:.4f}  bit-equal {same}  "
                  f"one launch {o[0]:7.1f} / {o[len(o) // 2]:7.1f}   three launches {t[0]:7.1f} / {t[len(t) // 2]:7.1f}   "
                  f"delta {o[len(o) // 2] - t[len(t) // 2]:+6.1f} us" + census, flush=True)
        del xs


if __name__ == "__main__":
    main()

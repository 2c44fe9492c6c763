#!/usr/bin/env python
"""What the fix-up launch of the one-kernel MFCC costs on the cfg4 noise batch (no tile is flagged there): the module with
`fused = True` as it ships (pass 0 + fix-up launch) against the same module with the second C-ABI call suppressed (tools only:
the result is only right because nothing is flagged).  us per call, un-profiled, interleaved rounds."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T
from audio_amd import _lib

dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1234)
L = _lib.lib()
real = L.aamd_mfcc_fused_f32
skip = {"on": False}


def entry(*args):
    if skip["on"] and args[-2]._obj.pass_ == 1:
        return 0
    return real(*args)


L.aamd_mfcc_fused_f32 = entry
with torch.no_grad():
    xs = [(0.5 * torch.randn(512, 160000, device=dev, generator=g)).clamp_(-1, 1) for _ in range(3)]
    m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
    m.fused = True
    for i in range(60):
        m(xs[i % 3])
    torch.cuda.synchronize()
    res = {"with_fixup": [], "pass0_only": []}
    for r in range(5):
        for name, on in (("with_fixup", False), ("pass0_only", True)):
            skip["on"] = on
            for i in range(10):
                m(xs[i % 3])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(200):
                m(xs[i % 3])
            e1.record()
            torch.cuda.synchronize()
            res[name].append(round(e0.elapsed_time(e1) * 5, 2))
    skip["on"] = False
    print(json.dumps({"us_per_call": res, "redone_share": m.fused_report()["redone_share"]}), flush=True)

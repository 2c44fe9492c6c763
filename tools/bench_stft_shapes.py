#!/usr/bin/env python
"""Spectrogram / MelSpectrogram / MFCC across common parameterisations (256 x 10 s @16 kHz unless noted): ms per call and the
fraction of the HBM peak on input + output bytes.  One JSON line per case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.transforms as T

dev = torch.device("cuda")


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


cases = [("MelSpectrogram defaults (n_fft 400, hop 200, 128 mels)", lambda: T.MelSpectrogram(16000)),
         ("MelSpectrogram 400/160/80 (headline)", lambda: T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80)),
         ("MelSpectrogram 400/160/128", lambda: T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=128)),
         ("MelSpectrogram 512/160/80", lambda: T.MelSpectrogram(16000, n_fft=512, hop_length=160, n_mels=80)),
         ("MelSpectrogram 512/128/64 win 400", lambda: T.MelSpectrogram(16000, n_fft=512, win_length=400, hop_length=128, n_mels=64)),
         ("MelSpectrogram 1024/256/128", lambda: T.MelSpectrogram(16000, n_fft=1024, hop_length=256, n_mels=128)),
         ("MelSpectrogram 2048/512/128", lambda: T.MelSpectrogram(16000, n_fft=2048, hop_length=512, n_mels=128)),
         ("MelSpectrogram 800/200/80", lambda: T.MelSpectrogram(16000, n_fft=800, hop_length=200, n_mels=80)),
         ("MelSpectrogram 480/160/80", lambda: T.MelSpectrogram(16000, n_fft=480, hop_length=160, n_mels=80)),
         ("Spectrogram defaults (400, hop 200)", lambda: T.Spectrogram()),
         ("Spectrogram 512/128 power None (complex)", lambda: T.Spectrogram(n_fft=512, hop_length=128, power=None)),
         ("Spectrogram 1024/256", lambda: T.Spectrogram(n_fft=1024, hop_length=256)),
         ("MFCC defaults (n_mfcc 40, mel defaults)", lambda: T.MFCC(16000)),
         ("MFCC 13 / 400/160/40 mels", lambda: T.MFCC(16000, n_mfcc=13, melkwargs=dict(n_fft=400, hop_length=160, n_mels=40)))]
with torch.no_grad():
    x = (0.5 * torch.randn(256, 160000, device=dev)).clamp_(-1, 1)
    for name, make in cases:
        try:
            m = make().to(dev)
            y = m(x)
            ms = timed(lambda: m(x))
            nbytes = x.numel() * 4 + y.numel() * y.element_size()
            print(json.dumps({"case": name, "ms": round(ms, 4), "out_shape": list(y.shape),
                              "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 8e12, 3)}), flush=True)
        except Exception as e:  # a sweep: report and go on
            print(json.dumps({"case": name, "error": repr(e)[:200]}), flush=True)

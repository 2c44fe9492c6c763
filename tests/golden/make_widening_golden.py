#!/usr/bin/env python
"""Fixtures for the SURVEY 8(f) rank-2 callers (InverseSpectrogram, GriffinLim, TimeStretch / phase_vocoder,
PitchShift, Speed) from the REFERENCE on the CPU (float32, as a user would run it).
Run only in the build container:   python tests/golden/make_widening_golden.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
import torchaudio.functional as F  # noqa: E402  (the reference)
import torchaudio.transforms as T  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
g = torch.Generator().manual_seed(99)


def noise(*shape):
    return (0.3 * torch.randn(*shape, generator=g)).clamp_(-1, 1)


def tone(n, sr=16000):
    t = torch.arange(n) / sr
    return 0.4 * torch.sin(2 * torch.pi * 440 * t) + 0.2 * torch.sin(2 * torch.pi * 1230 * t + 0.3)


# phase vocoder / TimeStretch
for name, (n_fft, hop, rate, L) in {"pv_fast": (400, 160, 1.3, 8000), "pv_slow": (512, 128, 0.7, 6000),
                                     "pv_big": (1024, 256, 2.5, 30000)}.items():
    x = torch.stack([noise(L), tone(L) + 0.05 * noise(L)])
    spec = T.Spectrogram(n_fft=n_fft, hop_length=hop, power=None)(x)
    ts = T.TimeStretch(hop_length=hop, n_freq=n_fft // 2 + 1, fixed_rate=rate)
    y = ts(spec)
    # the same float32 spectrogram pushed through the reference in float64: the yardstick for float32 noise
    y64 = T.TimeStretch(hop_length=hop, n_freq=n_fft // 2 + 1, fixed_rate=rate).double()(spec.to(torch.complex128))
    out[f"{name}/spec"] = spec.numpy(); out[f"{name}/out"] = y.numpy(); out[f"{name}/out64"] = y64.to(torch.complex64).numpy()
    out[f"{name}/cfg"] = np.array([n_fft, hop, rate], dtype=np.float64)
    print(name, tuple(spec.shape), "->", tuple(y.shape), "ref32 vs ref64:",
          float((y - y64).abs().max() / y64.abs().max()))

# inverse spectrogram
for name, (n_fft, hop, L, length) in {"inv_400": (400, 160, 8000, 8000), "inv_512": (512, 128, 5000, None)}.items():
    x = noise(2, L)
    spec = T.Spectrogram(n_fft=n_fft, hop_length=hop, power=None)(x)
    y = T.InverseSpectrogram(n_fft=n_fft, hop_length=hop)(spec, length)
    out[f"{name}/spec"] = spec.numpy(); out[f"{name}/out"] = y.numpy()
    out[f"{name}/cfg"] = np.array([n_fft, hop, -1 if length is None else length], dtype=np.float64)

# Griffin-Lim (deterministic start)
for name, (n_fft, hop, power, n_iter, momentum, L) in {"gl_400": (400, 200, 2.0, 8, 0.99, 6000),
                                                        "gl_512": (512, 128, 1.0, 5, 0.0, 5000)}.items():
    x = torch.stack([tone(L) + 0.02 * noise(L), noise(L)])
    spec = T.Spectrogram(n_fft=n_fft, hop_length=hop, power=power)(x)
    y = T.GriffinLim(n_fft=n_fft, hop_length=hop, power=power, n_iter=n_iter, momentum=momentum, length=L,
                     rand_init=False)(spec)
    y64 = T.GriffinLim(n_fft=n_fft, hop_length=hop, power=power, n_iter=n_iter, momentum=momentum, length=L,
                       rand_init=False).double()(spec.double())
    out[f"{name}/spec"] = spec.numpy(); out[f"{name}/out"] = y.numpy(); out[f"{name}/out64"] = y64.float().numpy()
    out[f"{name}/cfg"] = np.array([n_fft, hop, power, n_iter, momentum, L], dtype=np.float64)
    print(name, tuple(y.shape), "ref32 vs ref64:", float((y - y64).abs().max() / y64.abs().max()))

# PitchShift, Speed
for name, (sr, n_steps, L) in {"ps_up": (16000, 4, 8000), "ps_down": (16000, -3, 6000)}.items():
    x = torch.stack([tone(L), 0.5 * tone(L) + 0.1 * noise(L)]).reshape(2, 1, L)
    y = T.PitchShift(sr, n_steps)(x)
    yf = F.pitch_shift(x, sr, n_steps)
    y64 = F.pitch_shift(x.double(), sr, n_steps)
    out[f"{name}/x"] = x.numpy(); out[f"{name}/out"] = y.detach().numpy(); out[f"{name}/out_f"] = yf.numpy()
    out[f"{name}/out64"] = y64.float().numpy()
    out[f"{name}/cfg"] = np.array([sr, n_steps], dtype=np.float64)
    print(name, tuple(y.shape), "ref32 vs ref64:", float((yf - y64).abs().max() / y64.abs().max()))
x = noise(3, 4000)
lengths = torch.tensor([4000.0, 3000.0, 1234.0])
y, yl = T.Speed(16000, 1.1)(x, lengths)
out["speed/x"] = x.numpy(); out["speed/out"] = y.numpy(); out["speed/lengths"] = lengths.numpy(); out["speed/out_lengths"] = yl.numpy()
# RNN-T feature extractor: the reference's own Sequential (rnnt_pipeline.py:319-326) with synthetic global statistics
import json, tempfile  # noqa: E402
from torchaudio.pipelines import rnnt_pipeline as RP  # noqa: E402
stats = {"mean": (10 + 3 * torch.randn(80, generator=g)).tolist(), "invstddev": (0.2 + torch.rand(80, generator=g)).tolist()}
with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
    f.write(json.dumps(stats))
ref_fe = RP._ModuleFeatureExtractor(torch.nn.Sequential(
    T.MelSpectrogram(sample_rate=16000, n_fft=400, n_mels=80, hop_length=160),
    RP._FunctionalModule(lambda x: x.transpose(1, 0)),
    RP._FunctionalModule(lambda x: RP._piecewise_linear_log(x * RP._gain)),
    RP._GlobalStatsNormalization(f.name),
    RP._FunctionalModule(lambda x: torch.nn.functional.pad(x, (0, 0, 0, 4))),
))
xs = [0.1 * noise(16000), torch.cat([0.3 * tone(9000), torch.zeros(3000), 1e-4 * noise(4321)])]
for i, xw in enumerate(xs):
    feats, length = ref_fe(xw.clone())
    out[f"rnnt{i}/x"] = xw.numpy(); out[f"rnnt{i}/out"] = feats.numpy(); out[f"rnnt{i}/length"] = length.numpy()
    print("rnnt", i, tuple(feats.shape), int(length), "linear-branch share", float((feats == feats).float().mean()))
out["rnnt/mean"] = np.array(stats["mean"], dtype=np.float32); out["rnnt/invstddev"] = np.array(stats["invstddev"], dtype=np.float32)
np.savez_compressed(os.path.join(HERE, "widening_goldens.npz"), **out)
print("widening_goldens.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "widening_goldens.npz")) // 1024, "KiB")

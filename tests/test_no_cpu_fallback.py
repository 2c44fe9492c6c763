"""The product path has NO CPU fallback: every public entry point refuses a CPU tensor loudly (RuntimeError /
NotImplementedError from the dispatcher) instead of computing with torch, and nothing under audio_amd/ imports oracle/."""
import os
import re

import pytest
import torch

import audio_amd.functional as F
import audio_amd.transforms as T
import audio_amd.compliance.kaldi as K
from audio_amd.pipelines import RNNTFeatureExtractor

X = torch.randn(2, 4000)


@pytest.mark.parametrize("call", [
    lambda: T.Spectrogram(n_fft=400)(X),
    lambda: T.Spectrogram(n_fft=512, power=None)(X),
    lambda: T.MelSpectrogram(sample_rate=16000, n_fft=400, n_mels=40)(X),
    lambda: T.MFCC(sample_rate=16000, n_mfcc=13, melkwargs=dict(n_fft=400, n_mels=40))(X),
    lambda: T.Resample(16000, 8000)(X),
    lambda: T.InverseSpectrogram(n_fft=400)(torch.randn(2, 201, 21, dtype=torch.complex64)),
    lambda: T.GriffinLim(n_fft=400, n_iter=2)(torch.rand(2, 201, 21)),
    lambda: T.TimeStretch(n_freq=201, fixed_rate=1.2)(torch.randn(2, 201, 21, dtype=torch.complex64)),
    lambda: T.PitchShift(16000, 2)(X),
    lambda: T.Speed(16000, 1.1)(X),
    lambda: T.AmplitudeToDB()(X.abs()),
    lambda: T.FFTConvolve()(X, torch.randn(2, 300)),
    lambda: F.lfilter(X, torch.tensor([1.0, -0.5]), torch.tensor([0.3, 0.2])),
    lambda: F.fftconvolve(X, torch.randn(2, 30)),
    lambda: F.phase_vocoder(torch.randn(2, 201, 21, dtype=torch.complex64), 1.3, torch.zeros(201, 1)),
    lambda: K.fbank(X * 1000),
    lambda: K.spectrogram(X * 1000),
    lambda: K.mfcc(X * 1000),
    lambda: RNNTFeatureExtractor({"mean": [0.0] * 80, "invstddev": [1.0] * 80})(X),
    lambda: RNNTFeatureExtractor({"mean": [0.0] * 80, "invstddev": [1.0] * 80})((X * 1000).to(torch.int16)),
])
def test_cpu_tensors_are_refused(call):
    with pytest.raises((RuntimeError, NotImplementedError), match="(ROCm|MI355X|CPU|device)"):
        call()


def test_product_never_imports_the_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|\boracle\.(dsp_oracle|torch_cpu_ref|build_ref|cpu_baselines)\b")
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(".py"):
                for i, line in enumerate(open(os.path.join(d, f)), 1):
                    assert not pat.search(line), f"{f}:{i}: {line.strip()}"

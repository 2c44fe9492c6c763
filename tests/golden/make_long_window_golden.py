#!/usr/bin/env python
"""Fixtures for LONG analysis windows (round 5: n_fft / padded Kaldi windows of about 5 750 .. 8 192 samples, the layout of the generic
kernels without an LDS twiddle table) from the REFERENCE on the CPU.  Run only in the build container:
    python tests/golden/make_long_window_golden.py
Inputs and outputs go to tests/golden/long_window_goldens.npz (float32; the GPU box has no /root/reference)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference/src")
import torchaudio.compliance.kaldi as K  # noqa: E402  (the reference)
import torchaudio.transforms as T  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
g = torch.Generator().manual_seed(11)
t = torch.arange(50000) / 44100.0
x = torch.stack([0.4 * torch.sin(2 * torch.pi * 441.0 * t) + 0.05 * torch.randn(50000, generator=g),
                 0.3 * torch.randn(50000, generator=g),
                 0.2 * torch.sin(2 * torch.pi * 5000.0 * t + 0.3) * torch.linspace(0, 1, 50000)])
out = {"x": x.numpy()}
with torch.no_grad():
    out["spec_8192_p2"] = T.Spectrogram(n_fft=8192, hop_length=2048)(x).numpy()
    cplx = T.Spectrogram(n_fft=8192, hop_length=2048, power=None)(x)
    out["spec_8192_complex"] = torch.view_as_real(cplx).numpy()
    out["spec_7000_p1_norm"] = T.Spectrogram(n_fft=7000, hop_length=1750, power=1.0, normalized=True)(x).numpy()
    out["spec_8192_win6000_nocenter"] = T.Spectrogram(n_fft=8192, win_length=6000, hop_length=3000, center=False)(x).numpy()
    out["mel_8192_128"] = T.MelSpectrogram(sample_rate=44100, n_fft=8192, hop_length=2048, n_mels=128)(x).numpy()
    out["mfcc_8192"] = T.MFCC(sample_rate=44100, n_mfcc=20, melkwargs=dict(n_fft=8192, hop_length=2048, n_mels=64))(x).numpy()
    out["inverse_8192"] = T.InverseSpectrogram(n_fft=8192, hop_length=2048)(cplx, 50000).numpy()
    # Kaldi: 16-bit scaled samples, 16 kHz; 400 ms frames pad to 8192, 450 ms frames stay at 7200 with round_to_power_of_two=False
    wav = (x[:2, :24000] * 32768.0).contiguous()
    out["kaldi_wav"] = wav.numpy()
    out["kaldi_fbank_400ms"] = K.fbank(wav, frame_length=400.0, frame_shift=50.0, num_mel_bins=40, use_energy=True).numpy()
    out["kaldi_spec_450ms_np2"] = K.spectrogram(wav, frame_length=450.0, frame_shift=100.0, round_to_power_of_two=False,
                                                snip_edges=False, channel=1).numpy()
    out["kaldi_mfcc_400ms"] = K.mfcc(wav, frame_length=400.0, frame_shift=50.0, num_mel_bins=40, num_ceps=13).numpy()
for k, v in out.items():
    print(k, v.shape, float(np.abs(v).max()))
np.savez_compressed(os.path.join(HERE, "long_window_goldens.npz"), **{k: np.asarray(v, dtype=np.float32) for k, v in out.items()})

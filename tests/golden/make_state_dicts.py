#!/usr/bin/env python
"""State dicts of the REFERENCE's transform modules, for the drop-in check "a checkpoint written by torchaudio loads
into audio_amd's module of the same name with strict=True, and vice versa" (reference test:
test/torchaudio_unittest/transforms/transforms_test.py:54-85; SURVEY.md section 5 "Checkpoint/resume").

Run ONLY in the build container:   PYTHONPATH=/root/reference/src python tests/golden/make_state_dicts.py
Writes tests/golden/state_dicts.npz: '<case>/<state_dict key>' -> array, plus '<case>/__keys__'."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
import torchaudio.transforms as T  # noqa: E402  (the reference, pure-python mode)

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "Spectrogram": lambda: T.Spectrogram(n_fft=400, hop_length=160),
    "Spectrogram_win300": lambda: T.Spectrogram(n_fft=512, win_length=300, hop_length=128),
    "MelScale": lambda: T.MelScale(),
    "MelScale_slaney": lambda: T.MelScale(n_mels=64, sample_rate=22050, n_stft=513, norm="slaney", mel_scale="slaney"),
    "MelSpectrogram": lambda: T.MelSpectrogram(),
    "MelSpectrogram_headline": lambda: T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80),
    "MFCC": lambda: T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)),
    "Resample_44100_16000": lambda: T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser",
                                               lowpass_filter_width=64, rolloff=0.9475937167399596,
                                               beta=14.769656459379492),
    "Resample_8000_16000": lambda: T.Resample(8000, 16000),
    "InverseSpectrogram": lambda: T.InverseSpectrogram(n_fft=400, hop_length=160),
    "GriffinLim": lambda: T.GriffinLim(n_fft=400, hop_length=160),
    "AmplitudeToDB": lambda: T.AmplitudeToDB("power", 80.0),
    "TimeStretch": lambda: T.TimeStretch(hop_length=160, n_freq=201, fixed_rate=1.3),
    "PitchShift": lambda: T.PitchShift(16000, 4),
}


def main():
    out = {}
    for name, make in CASES.items():
        m = make()
        if name == "PitchShift":       # lazily initialised kernel buffer (transforms/_transforms.py PitchShift)
            import torch
            m.initialize_parameters(torch.zeros(1, 16000))
        sd = m.state_dict()
        out[f"{name}/__keys__"] = np.array(list(sd.keys()), dtype=object).astype(str)
        for k, v in sd.items():
            if v.numel() > (1 << 22):      # PitchShift's 8000 x 10095 tap table: shape only (values: test_kaldi-style fixtures)
                out[f"{name}/{k}/__shape__"] = np.array(v.shape)
            else:
                out[f"{name}/{k}"] = v.detach().cpu().numpy()
    np.savez_compressed(os.path.join(HERE, "state_dicts.npz"), **out)
    for k in sorted(out):
        print(k, getattr(out[k], "shape", None))


if __name__ == "__main__":
    main()

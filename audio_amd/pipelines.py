"""RNN-T feature extraction front-end (reference: torchaudio/pipelines/rnnt_pipeline.py:16-47, 80-107, 319-326).

The reference builds, per utterance,

    MelSpectrogram(sample_rate, n_fft, n_mels, hop_length) -> transpose(1, 0) -> _piecewise_linear_log(x * _gain)
    -> _GlobalStatsNormalization -> pad(right_padding frames)

i.e. five element-wise passes over the mel features after the STFT.  Here the whole chain is ONE launch of the
headline kernel (`EPI400_MEL_NORM`, csrc/melspec400.h): the features are born frame-major -- the layout the
transpose produces -- normalised, in rows that already contain the right padding.  Batches are first class:
`forward` takes (..., time) and returns (..., frames + right_padding, n_mels); `__call__` on a 1-D waveform returns
`(features, length)` exactly like the reference's `_ModuleFeatureExtractor`.

The upstream step is fused as well: an **int16** waveform (the PCM samples the decoder produces; torchaudio's loaders hand
out `int16 / 32768` as float32) is read directly by the kernel -- no conversion pass, half the input bytes.
"""
from __future__ import annotations

import json
import math
from typing import Optional, Tuple, Union

import torch
from torch import Tensor

from . import functional as F
from . import transforms as T

__all__ = ["RNNTFeatureExtractor", "piecewise_linear_log", "GAIN"]

_decibel = 2 * 20 * math.log10(torch.iinfo(torch.int16).max)
GAIN = pow(10, 0.05 * _decibel)                     # rnnt_pipeline.py:16-17


def piecewise_linear_log(x: Tensor) -> Tensor:
    """rnnt_pipeline.py:20-23, out of place, with the reference's exact semantics: its two in-place masked
    assignments evaluate the second mask (x <= e) AFTER the first one wrote log(x), so values in (e, e^e] end up as
    log(x) / e.  (x <= e: x / e;  e < x <= e^e: log(x) / e;  x > e^e: log(x).)"""
    t = torch.where(x > math.e, torch.log(torch.clamp(x, min=math.e)), x)
    return torch.where(t <= math.e, t / math.e, t)


class RNNTFeatureExtractor(torch.nn.Module):
    r"""Drop-in for the module `RNNTBundle.get_feature_extractor()` returns (rnnt_pipeline.py:303-327).

    Args:
        global_stats (str or dict): path of the bundle's ``global_stats.json`` or a dict with ``mean`` / ``invstddev``.
        sample_rate, n_fft, n_mels, hop_length, right_padding: the bundle's `_sample_rate`, `_n_fft`, `_n_mels`,
            `_hop_length`, `_right_padding` (defaults: EMFORMER_RNNT_BASE_LIBRISPEECH, rnnt_pipeline.py:353-371).
    """

    def __init__(self, global_stats: Union[str, dict], sample_rate: int = 16000, n_fft: int = 400, n_mels: int = 80,
                 hop_length: int = 160, right_padding: int = 4, gain: float = GAIN) -> None:
        super().__init__()
        if isinstance(global_stats, str):
            with open(global_stats) as f:
                global_stats = json.loads(f.read())
        self.register_buffer("mean", torch.as_tensor(global_stats["mean"], dtype=torch.float32))
        self.register_buffer("invstddev", torch.as_tensor(global_stats["invstddev"], dtype=torch.float32))
        self.mel = T.MelSpectrogram(sample_rate=sample_rate, n_fft=n_fft, n_mels=n_mels, hop_length=hop_length)
        self.right_padding = right_padding
        self.gain = gain

    def features(self, waveform: Tensor, channels_first: bool = True) -> Tensor:
        """(..., time) float32 in [-1, 1], or int16 PCM -> (..., frames + right_padding, n_mels).

        ``channels_first=False`` (EXTENSION; the flag name is the reference loader's, torchaudio/_torchcodec.py:150-152):
        ``waveform`` is int16 PCM in the decoder's interleaved order ``(..., time, channels)``; the result is
        ``(..., channels, frames + right_padding, n_mels)`` -- what the reference computes from the transposed, converted
        tensor -- with the de-interleave done inside the kernel's load for mono and stereo."""
        sp = self.mel.spectrogram
        if not channels_first:
            chans = int(waveform.shape[-1])
            out = F._mel_lognorm(waveform, sp.window, self.mel.mel_scale.fb, sp.n_fft, sp.hop_length, self.gain, self.mean,
                                 self.invstddev, self.right_padding, interleaved_channels=chans)
            return out.view(tuple(waveform.shape[:-2]) + (chans,) + out.shape[-2:])
        out = F._mel_lognorm(waveform, sp.window, self.mel.mel_scale.fb, sp.n_fft, sp.hop_length, self.gain, self.mean,
                             self.invstddev, self.right_padding)
        return out.view(tuple(waveform.shape[:-1]) + out.shape[-2:])

    def forward(self, input: Tensor) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        """1-D waveform: `(features (length, n_mels), length (1,))` as the reference; batched input: features only."""
        feats = self.features(input)
        if input.dim() == 1:
            return feats, torch.tensor([feats.shape[0]])
        return feats

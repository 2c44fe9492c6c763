"""Kaldi-compatible features on the MI355X kernels (reference: torchaudio/compliance/kaldi.py).

`spectrogram`, `fbank` and `mfcc` keep the reference's signatures and defaults.  The framing, per-frame conditioning
(DC removal, log-energy, pre-emphasis, window, zero padding), the FFT, the mel banks and the logs run in ONE HIP kernel
(`p2::kaldi_pow2_kernel`, csrc/stft_pow2.h, through `aamd_kaldi_features_f32`); `mfcc` adds the DCT on the matrix-core kernel
of the MFCC path.  Constants (window function, mel banks, DCT matrix, lifter) are built on the host exactly as the reference
builds them.  Padded windows of 256 / 512 / 1024 / 2048 samples run on the register FFT; every other even size
(`round_to_power_of_two=False`: 400 at 16 kHz, 200 at 8 kHz, 1102 at 44.1 kHz ...) on the mixed-radix LDS kernel
(csrc/kaldi_generic.h).  `dither` draws its Gaussian noise exactly as the reference does -- `torch.randn(frames.shape)` on
the waveform's device (kaldi.py:180-183) -- so a seeded run reproduces the reference's run on the same device.

EXTENSION (not in the reference, which takes one channel of one utterance per call): `spectrogram_batch`, `fbank_batch` and
`mfcc_batch` take (B, n) equal-length utterances and return (B, m, cols) from ONE launch; row b is bit-identical to the
single-utterance call on `waveforms[b]` (`subtract_mean` is per utterance).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Tuple

import torch
from torch import Tensor

from .. import _host, _lib
from .. import functional as F

__all__ = ["get_mel_banks", "inverse_mel_scale", "inverse_mel_scale_scalar", "mel_scale", "mel_scale_scalar", "spectrogram",
           "fbank", "mfcc", "vtln_warp_freq", "vtln_warp_mel_freq", "spectrogram_batch", "fbank_batch", "mfcc_batch"]

EPSILON = torch.tensor(torch.finfo(torch.float).eps)
MILLISECONDS_TO_SECONDS = 0.001
HAMMING, HANNING, POVEY, RECTANGULAR, BLACKMAN = "hamming", "hanning", "povey", "rectangular", "blackman"
WINDOWS = [HAMMING, HANNING, POVEY, RECTANGULAR, BLACKMAN]

# host-side constants, verbatim semantics of the reference (kaldi.py:318-511)
inverse_mel_scale_scalar = _host.kaldi_inverse_mel_scale_scalar
inverse_mel_scale = _host.kaldi_inverse_mel_scale
mel_scale_scalar = _host.kaldi_mel_scale_scalar
mel_scale = _host.kaldi_mel_scale
vtln_warp_freq = _host.kaldi_vtln_warp_freq
vtln_warp_mel_freq = _host.kaldi_vtln_warp_mel_freq
get_mel_banks = _host.kaldi_get_mel_banks


_ALL = object()        # `channel` sentinel of the *_batch functions


def _randn(shape, device, dtype) -> Tensor:
    """The reference's dither draw (tests substitute a recorded draw here)."""
    return torch.randn(shape, device=device, dtype=dtype)


def _next_power_of_2(x: int) -> int:
    return 1 if x == 0 else 2 ** (x - 1).bit_length()


def _num_frames(num_samples: int, window_size: int, window_shift: int, snip_edges: bool) -> int:
    """kaldi.py:44-83 (_get_strided)."""
    if snip_edges:
        return 0 if num_samples < window_size else 1 + (num_samples - window_size) // window_shift
    return (num_samples + (window_shift // 2)) // window_shift


def _properties(waveform: Tensor, channel: int, sample_frequency: float, frame_shift: float, frame_length: float,
                round_to_power_of_two: bool, preemphasis_coefficient: float) -> Tuple[Tensor, int, int, int]:
    """kaldi.py:125-151 with the reference's assertions."""
    if channel is _ALL:                       # the *_batch extension: every row is an utterance
        assert waveform.dim() == 2, "waveforms must be (B, n)"
    else:
        channel = max(channel, 0)
        assert channel < waveform.size(0), "Invalid channel {} for size {}".format(channel, waveform.size(0))
        waveform = waveform[channel, :]
    window_shift = int(sample_frequency * frame_shift * MILLISECONDS_TO_SECONDS)
    window_size = int(sample_frequency * frame_length * MILLISECONDS_TO_SECONDS)
    padded_window_size = _next_power_of_2(window_size) if round_to_power_of_two else window_size
    assert 2 <= window_size <= waveform.size(-1), "choose a window size {} that is [2, {}]".format(window_size, waveform.size(-1))
    assert 0 < window_shift, "`window_shift` must be greater than 0"
    assert padded_window_size % 2 == 0, (
        "the padded `window_size` must be divisible by two." " use `round_to_power_of_two` or change `frame_length`")
    assert 0.0 <= preemphasis_coefficient <= 1.0, "`preemphasis_coefficient` must be between [0,1]"
    assert sample_frequency > 0, "`sample_frequency` must be greater than zero"
    return waveform, window_shift, window_size, padded_window_size


def _features(waveform: Tensor, window_shift: int, window_size: int, padded: int, window_type: str, blackman_coeff: float,
              snip_edges: bool, raw_energy: bool, energy_floor: float, dither: float, remove_dc_offset: bool,
              preemphasis_coefficient: float, bands, use_power: bool, use_log: bool, energy_col: int, first_col: int,
              n_cols: int) -> Tensor:
    if padded > 8192:       # (~5 750 .. 8 192 run the generic kernel's layout without an LDS twiddle table; beyond that nothing fits the LDS)
        raise NotImplementedError(f"audio_amd: padded windows above 8192 samples are not supported, got {padded}")
    if not waveform.is_cuda:
        raise RuntimeError(f"audio_amd: waveform must be on an MI355X (ROCm) device, got {waveform.device}. "
                           "The HIP kernels have no CPU fallback.")
    if waveform.dtype != torch.float32:
        raise TypeError(f"audio_amd: kaldi features need float32 waveforms, got {waveform.dtype}")
    dev = waveform.device
    x = waveform if waveform.stride(-1) == 1 else waveform.contiguous()
    batch = x.dim() == 2
    n_utt = x.size(0) if batch else 1
    n = x.size(-1)
    if batch and n_utt > 1 and x.stride(0) < n:
        x = x.contiguous()
    m = _num_frames(n, window_size, window_shift, snip_edges)
    if m == 0 or n_utt == 0:
        return torch.empty((n_utt, 0, 0) if batch else (0, 0), dtype=torch.float32, device=dev)
    key = ("kaldi_win", window_type, window_size, padded, blackman_coeff, str(dev))
    win = F._cached(key, lambda: torch.nn.functional.pad(
        _host.kaldi_window(window_type, window_size, blackman_coeff), (0, padded - window_size)).to(dev).contiguous())
    noise = None
    if dither != 0.0:
        # kaldi.py:180-183: rand_gauss = torch.randn(strided_input.shape, device, dtype); frames += rand_gauss * dither
        noise = _randn((n_utt, m, window_size) if batch else (m, window_size), device=dev, dtype=torch.float32).contiguous()
    ops = F._ops()
    if ops is not None and x.is_contiguous():
        # the dispatcher-level op (csrc/torch_shim.cpp aamd::kaldi_features): options as two flat lists, (n_utt, n) rows
        res = ops.kaldi_features(x.view(n_utt, n), win, F._twiddles(padded, dev),
                                 None if bands is None else bands.lo, None if bands is None else bands.width,
                                 None if bands is None else bands.weights,
                                 None if noise is None else noise.view(n_utt, m, window_size), m,
                                 [padded, window_shift, window_size, int(snip_edges), int(remove_dc_offset), int(raw_energy),
                                  int(use_power), int(use_log), energy_col, first_col, n_cols],
                                 [float(preemphasis_coefficient), float(energy_floor), float(dither)])
        return res if batch else res.view(m, n_cols)
    out = torch.empty((n_utt, m, n_cols) if batch else (m, n_cols), dtype=torch.float32, device=dev)
    d = _lib.KaldiDesc(n, m, padded, window_shift, window_size, int(snip_edges), float(preemphasis_coefficient),
                       int(remove_dc_offset), int(raw_energy), float(energy_floor), int(use_power), int(use_log),
                       energy_col, first_col, n_cols, float(dither), None if noise is None else noise.data_ptr(),
                       n_utt, x.stride(0) if batch and n_utt > 1 else n)
    L = _lib.lib()
    with torch.cuda.device(dev):
        _lib.check(L.aamd_kaldi_features_f32(x.data_ptr(), win.data_ptr(), F._twiddles(padded, dev).data_ptr(),
                                             None if bands is None else C.byref(bands.struct), out.data_ptr(), C.byref(d),
                                             _lib.current_stream(dev)))
    return out


def _subtract_column_mean(tensor: Tensor, subtract_mean: bool) -> Tensor:
    if subtract_mean:
        tensor = tensor - torch.mean(tensor, dim=-2, keepdim=True)
    return tensor


def spectrogram(
    waveform: Tensor, blackman_coeff: float = 0.42, channel: int = -1, dither: float = 0.0,
    energy_floor: float = 1.0, frame_length: float = 25.0, frame_shift: float = 10.0, min_duration: float = 0.0,
    preemphasis_coefficient: float = 0.97, raw_energy: bool = True, remove_dc_offset: bool = True,
    round_to_power_of_two: bool = True, sample_frequency: float = 16000.0, snip_edges: bool = True,
    subtract_mean: bool = False, window_type: str = POVEY,
) -> Tensor:
    r"""Kaldi's compute-spectrogram-feats (reference: compliance/kaldi.py:229-315): (m, padded_window_size // 2 + 1)."""
    waveform, window_shift, window_size, padded = _properties(
        waveform, channel, sample_frequency, frame_shift, frame_length, round_to_power_of_two, preemphasis_coefficient)
    if waveform.size(-1) < min_duration * sample_frequency:
        return torch.empty(0)
    out = _features(waveform, window_shift, window_size, padded, window_type, blackman_coeff, snip_edges, raw_energy,
                    energy_floor, dither, remove_dc_offset, preemphasis_coefficient, None, True, True, -1, 0, padded // 2 + 1)
    return _subtract_column_mean(out, subtract_mean)


def fbank(
    waveform: Tensor, blackman_coeff: float = 0.42, channel: int = -1, dither: float = 0.0,
    energy_floor: float = 1.0, frame_length: float = 25.0, frame_shift: float = 10.0, high_freq: float = 0.0,
    htk_compat: bool = False, low_freq: float = 20.0, min_duration: float = 0.0, num_mel_bins: int = 23,
    preemphasis_coefficient: float = 0.97, raw_energy: bool = True, remove_dc_offset: bool = True,
    round_to_power_of_two: bool = True, sample_frequency: float = 16000.0, snip_edges: bool = True,
    subtract_mean: bool = False, use_energy: bool = False, use_log_fbank: bool = True, use_power: bool = True,
    vtln_high: float = -500.0, vtln_low: float = 100.0, vtln_warp: float = 1.0, window_type: str = POVEY,
) -> Tensor:
    r"""Kaldi's compute-fbank-feats (reference: compliance/kaldi.py:514-645): (m, num_mel_bins + use_energy)."""
    device, dtype = waveform.device, waveform.dtype
    waveform, window_shift, window_size, padded = _properties(
        waveform, channel, sample_frequency, frame_shift, frame_length, round_to_power_of_two, preemphasis_coefficient)
    if waveform.size(-1) < min_duration * sample_frequency:
        return torch.empty(0, device=device, dtype=dtype)
    key = ("kaldi_banks", num_mel_bins, padded, sample_frequency, low_freq, high_freq, vtln_low, vtln_high, vtln_warp,
           str(device))

    def make():
        bins, _ = get_mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq, vtln_low, vtln_high, vtln_warp)
        bins = torch.nn.functional.pad(bins.to(torch.float32), (0, 1), mode="constant", value=0)   # (num_bins, padded/2 + 1)
        return F.MelBandsOnDevice(bins.T.contiguous(), device)
    bands = F._cached(key, make)
    n_cols = num_mel_bins + int(use_energy)
    energy_col = -1 if not use_energy else (num_mel_bins if htk_compat else 0)
    first_col = 1 if (use_energy and not htk_compat) else 0
    out = _features(waveform, window_shift, window_size, padded, window_type, blackman_coeff, snip_edges, raw_energy,
                    energy_floor, dither, remove_dc_offset, preemphasis_coefficient, bands, use_power, use_log_fbank,
                    energy_col, first_col, n_cols)
    return _subtract_column_mean(out, subtract_mean)


def _get_dct_matrix(num_ceps: int, num_mel_bins: int) -> Tensor:
    """kaldi.py:648-658."""
    dct_matrix = _host.create_dct(num_mel_bins, num_mel_bins, "ortho")
    dct_matrix[:, 0] = math.sqrt(1 / float(num_mel_bins))
    return dct_matrix[:, :num_ceps]


def _get_lifter_coeffs(num_ceps: int, cepstral_lifter: float) -> Tensor:
    """kaldi.py:661-666."""
    i = torch.arange(num_ceps)
    return 1.0 + 0.5 * cepstral_lifter * torch.sin(math.pi * i / cepstral_lifter)


def mfcc(
    waveform: Tensor, blackman_coeff: float = 0.42, cepstral_lifter: float = 22.0, channel: int = -1,
    dither: float = 0.0, energy_floor: float = 1.0, frame_length: float = 25.0, frame_shift: float = 10.0,
    high_freq: float = 0.0, htk_compat: bool = False, low_freq: float = 20.0, num_ceps: int = 13,
    min_duration: float = 0.0, num_mel_bins: int = 23, preemphasis_coefficient: float = 0.97,
    raw_energy: bool = True, remove_dc_offset: bool = True, round_to_power_of_two: bool = True,
    sample_frequency: float = 16000.0, snip_edges: bool = True, subtract_mean: bool = False,
    use_energy: bool = False, vtln_high: float = -500.0, vtln_low: float = 100.0, vtln_warp: float = 1.0,
    window_type: str = POVEY,
) -> Tensor:
    r"""Kaldi's compute-mfcc-feats (reference: compliance/kaldi.py:669-813): (m, num_ceps)."""
    assert num_ceps <= num_mel_bins, "num_ceps cannot be larger than num_mel_bins: %d vs %d" % (num_ceps, num_mel_bins)
    device = waveform.device
    feature = fbank(waveform=waveform, blackman_coeff=blackman_coeff, channel=channel, dither=dither,
                    energy_floor=energy_floor, frame_length=frame_length, frame_shift=frame_shift, high_freq=high_freq,
                    htk_compat=htk_compat, low_freq=low_freq, min_duration=min_duration, num_mel_bins=num_mel_bins,
                    preemphasis_coefficient=preemphasis_coefficient, raw_energy=raw_energy,
                    remove_dc_offset=remove_dc_offset, round_to_power_of_two=round_to_power_of_two,
                    sample_frequency=sample_frequency, snip_edges=snip_edges, subtract_mean=False, use_energy=use_energy,
                    use_log_fbank=True, use_power=True, vtln_high=vtln_high, vtln_low=vtln_low, vtln_warp=vtln_warp,
                    window_type=window_type)
    if feature.numel() == 0:
        return feature
    if use_energy:
        signal_log_energy = feature[..., num_mel_bins if htk_compat else 0]
        mel_offset = int(not htk_compat)
        feature = feature[..., mel_offset:(num_mel_bins + mel_offset)]
    dct_matrix = F._cached(("kaldi_dct", num_ceps, num_mel_bins, str(device)),
                           lambda: _get_dct_matrix(num_ceps, num_mel_bins).to(dtype=torch.float32, device=device).contiguous())
    # (m, num_mel_bins) @ (num_mel_bins, num_ceps) on the MFCC path's DCT kernel (log_mode 0: plain product)
    lead = feature.shape[:-1]
    feature = F._dct_rows(feature.reshape(-1, num_mel_bins).contiguous(), dct_matrix).reshape(lead + (num_ceps,))
    if cepstral_lifter != 0.0:
        feature = feature * _get_lifter_coeffs(num_ceps, cepstral_lifter).to(device=device, dtype=torch.float32)
    if use_energy:
        feature[..., 0] = signal_log_energy
    if htk_compat:
        energy = feature[..., 0].unsqueeze(-1)
        feature = feature[..., 1:]
        if not use_energy:
            energy = energy * math.sqrt(2)
        feature = torch.cat((feature, energy), dim=-1)
    return _subtract_column_mean(feature, subtract_mean)


def spectrogram_batch(waveforms: Tensor, **kwargs) -> Tensor:
    """EXTENSION: `spectrogram` of B equal-length utterances (B, n) -> (B, m, padded // 2 + 1) in one launch; keyword
    arguments as `spectrogram` (without `channel`)."""
    assert "channel" not in kwargs, "the batch functions take every row as an utterance"
    return spectrogram(waveforms, channel=_ALL, **kwargs)


def fbank_batch(waveforms: Tensor, **kwargs) -> Tensor:
    """EXTENSION: `fbank` of B equal-length utterances (B, n) -> (B, m, num_mel_bins + use_energy) in one launch."""
    assert "channel" not in kwargs, "the batch functions take every row as an utterance"
    return fbank(waveforms, channel=_ALL, **kwargs)


def mfcc_batch(waveforms: Tensor, **kwargs) -> Tensor:
    """EXTENSION: `mfcc` of B equal-length utterances (B, n) -> (B, m, num_ceps): one feature launch + one DCT launch."""
    assert "channel" not in kwargs, "the batch functions take every row as an utterance"
    return mfcc(waveforms, channel=_ALL, **kwargs)

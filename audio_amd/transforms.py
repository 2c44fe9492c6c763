"""Drop-in replacements for the hot-path modules of ``torchaudio.transforms``:
Spectrogram, MelScale, MelSpectrogram, AmplitudeToDB, MFCC, Resample, FFTConvolve.

Constructor / forward signatures, registered buffer names (``window``, ``fb``, ``dct_mat``,
``kernel``), shapes, strides, warnings and error messages follow
src/torchaudio/transforms/_transforms.py; ``forward`` calls the HIP kernels through
``audio_amd.functional``.
"""
from __future__ import annotations

import ctypes as C
import math
import secrets
import warnings
from typing import Callable, Optional, Tuple, Union

import torch
from torch import Tensor

from . import _host, _lib
from . import functional as F
from . import _ops  # noqa: F401  (registers torch.ops.audio_amd.* -- the opaque module-level ops torch.compile sees)


_norm_mode = F._norm_mode      # `normalized` as the op schemas' integer: 0 none, 1 "frame_length", 2 "window" / True

# TorchScript (reference: test/torchaudio_unittest/transforms/torchscript_consistency_impl.py:13-76 scripts every module of this
# file).  Each ``forward`` below has two bodies: called from Python it runs ``_forward_eager`` (plan caches, autograd Functions,
# ctypes -- not compiled, the call sits behind ``torch.jit.is_scripting()``); inside a scripted program it is ONE operator of
# the ``audio_amd`` namespace (audio_amd/_ops.py) whose kernel runs the same launches, so scripted == eager bit for bit and the
# scripted module keeps ``state_dict`` keys, buffers and attributes.

__all__ = ["Spectrogram", "InverseSpectrogram", "GriffinLim", "TimeStretch", "PitchShift", "Speed", "SpeedPerturbation",
           "MelScale", "MelSpectrogram", "AmplitudeToDB", "MFCC", "Resample", "FFTConvolve"]


class Spectrogram(torch.nn.Module):
    r"""Spectrogram of ``(..., time)`` audio (reference: _transforms.py:25-123)."""
    __constants__ = ["n_fft", "win_length", "hop_length", "pad", "power", "normalized"]

    def __init__(
        self,
        n_fft: int = 400,
        win_length: Optional[int] = None,
        hop_length: Optional[int] = None,
        pad: int = 0,
        window_fn: Callable[..., Tensor] = torch.hann_window,
        power: Optional[float] = 2.0,
        normalized: Union[bool, str] = False,
        wkwargs: Optional[dict] = None,
        center: bool = True,
        pad_mode: str = "reflect",
        onesided: bool = True,
        return_complex: Optional[bool] = None,
    ) -> None:
        super().__init__()
        torch._C._log_api_usage_once("torchaudio.transforms.Spectrogram")
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window)
        self.pad = pad
        self.power = power
        self.normalized = normalized
        self.center = center
        self.pad_mode = pad_mode
        self.onesided = onesided
        if return_complex is not None:
            warnings.warn(
                "`return_complex` argument is now deprecated and is not effective."
                "`torchaudio.transforms.Spectrogram(power=None)` always returns a tensor with "
                "complex dtype. Please remove the argument in the function call."
            )

    def forward(self, waveform: Tensor) -> Tensor:
        if not torch.jit.is_scripting():
            return self._forward_eager(waveform)
        return torch.ops.audio_amd.spectrogram(waveform, self.window, self.pad, self.n_fft, self.hop_length, self.win_length,
                                               self.power, _norm_mode(self.normalized), self.center, self.pad_mode,
                                               self.onesided)

    def _forward_eager(self, waveform: Tensor) -> Tensor:
        if torch.compiler.is_compiling():          # one opaque op with a fake kernel (see MelSpectrogram.forward)
            return torch.ops.audio_amd.spectrogram(waveform, self.window, self.pad, self.n_fft, self.hop_length,
                                                   self.win_length, self.power, _norm_mode(self.normalized), self.center,
                                                   self.pad_mode, self.onesided)
        return F.spectrogram(waveform, self.pad, self.window, self.n_fft, self.hop_length, self.win_length,
                             self.power, self.normalized, self.center, self.pad_mode, self.onesided)


class InverseSpectrogram(torch.nn.Module):
    r"""Least-squares inverse of a complex spectrogram (reference: _transforms.py:126-208)."""
    __constants__ = ["n_fft", "win_length", "hop_length", "pad", "power", "normalized"]

    def __init__(
        self,
        n_fft: int = 400,
        win_length: Optional[int] = None,
        hop_length: Optional[int] = None,
        pad: int = 0,
        window_fn: Callable[..., Tensor] = torch.hann_window,
        normalized: Union[bool, str] = False,
        wkwargs: Optional[dict] = None,
        center: bool = True,
        pad_mode: str = "reflect",
        onesided: bool = True,
    ) -> None:
        super().__init__()
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window)
        self.pad = pad
        self.normalized = normalized
        self.center = center
        self.pad_mode = pad_mode
        self.onesided = onesided

    def forward(self, spectrogram: Tensor, length: Optional[int] = None) -> Tensor:
        return F.inverse_spectrogram(spectrogram, length, self.pad, self.window, self.n_fft, self.hop_length,
                                     self.win_length, self.normalized, self.center, self.pad_mode, self.onesided)


class GriffinLim(torch.nn.Module):
    r"""Waveform from a magnitude spectrogram by Griffin-Lim (reference: _transforms.py:212-297)."""
    __constants__ = ["n_fft", "n_iter", "win_length", "hop_length", "power", "length", "momentum", "rand_init"]

    def __init__(
        self,
        n_fft: int = 400,
        n_iter: int = 32,
        win_length: Optional[int] = None,
        hop_length: Optional[int] = None,
        window_fn: Callable[..., Tensor] = torch.hann_window,
        power: float = 2.0,
        wkwargs: Optional[dict] = None,
        momentum: float = 0.99,
        length: Optional[int] = None,
        rand_init: bool = True,
    ) -> None:
        super().__init__()
        if not (0 <= momentum < 1):
            raise ValueError("momentum must be in the range [0, 1). Found: {}".format(momentum))
        self.n_fft = n_fft
        self.n_iter = n_iter
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window)
        self.length = length
        self.power = power
        self.momentum = momentum
        self.rand_init = rand_init

    def forward(self, specgram: Tensor) -> Tensor:
        return F.griffinlim(specgram, self.window, self.n_fft, self.hop_length, self.win_length, self.power,
                            self.n_iter, self.momentum, self.length, self.rand_init)


class TimeStretch(torch.nn.Module):
    r"""Stretch a complex STFT in time without modifying pitch (reference: _transforms.py:1014-1083)."""
    __constants__ = ["fixed_rate"]

    def __init__(self, hop_length: Optional[int] = None, n_freq: int = 201, fixed_rate: Optional[float] = None) -> None:
        super().__init__()
        self.fixed_rate = fixed_rate
        n_fft = (n_freq - 1) * 2
        hop_length = hop_length if hop_length is not None else n_fft // 2
        self.register_buffer("phase_advance", torch.linspace(0, math.pi * hop_length, n_freq)[..., None])

    def forward(self, complex_specgrams: Tensor, overriding_rate: Optional[float] = None) -> Tensor:
        if not torch.is_complex(complex_specgrams):
            raise ValueError("audio_amd: the input to TimeStretch must be a complex tensor")
        if overriding_rate is None:
            if self.fixed_rate is None:
                raise ValueError("If no fixed_rate is specified, must pass a valid rate to the forward method.")
            rate = self.fixed_rate
        else:
            rate = overriding_rate
        return F.phase_vocoder(complex_specgrams, rate, self.phase_advance)


class PitchShift(torch.nn.Module):
    r"""Shift the pitch of a waveform by ``n_steps`` steps (reference: _transforms.py:1674-1780)."""
    __constants__ = ["sample_rate", "n_steps", "bins_per_octave", "n_fft", "win_length", "hop_length"]

    def __init__(
        self,
        sample_rate: int,
        n_steps: int,
        bins_per_octave: int = 12,
        n_fft: int = 512,
        win_length: Optional[int] = None,
        hop_length: Optional[int] = None,
        window_fn: Callable[..., Tensor] = torch.hann_window,
        wkwargs: Optional[dict] = None,
    ) -> None:
        super().__init__()
        self.n_steps = n_steps
        self.bins_per_octave = bins_per_octave
        self.sample_rate = sample_rate
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 4
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window)
        rate = 2.0 ** (-float(n_steps) / bins_per_octave)
        self.orig_freq = int(sample_rate / rate)
        self.gcd = math.gcd(int(self.orig_freq), int(sample_rate))
        self.width = -1
        self.kernel = None          # built on the first call (the reference uses a lazy UninitializedParameter)

    def forward(self, waveform: Tensor) -> Tensor:
        if not torch.jit.is_scripting():
            return self._forward_eager(waveform)
        # (the scripted program rebuilds nothing either: F.pitch_shift keeps the tap table per parameter set)
        return torch.ops.audio_amd.pitch_shift(waveform, self.sample_rate, self.n_steps, self.bins_per_octave, self.n_fft,
                                               self.win_length, self.hop_length, self.window)

    def _forward_eager(self, waveform: Tensor) -> Tensor:
        if waveform.dtype in F.LOW_PRECISION:         # float16 / bfloat16: float32 arithmetic, the input dtype back (F._reduced_precision_io)
            return self._forward_eager(waveform.float()).to(waveform.dtype)
        F._require_device(waveform, "waveform")
        shape = waveform.size()
        stretch = F._stretch_waveform(waveform, self.n_steps, self.bins_per_octave, self.n_fft, self.win_length,
                                      self.hop_length, self.window)
        if self.orig_freq != self.sample_rate:
            if self.kernel is None or self.kernel.device != waveform.device:
                kernel, self.width = _host.sinc_resample_kernel(self.orig_freq, self.sample_rate, self.gcd,
                                                                dtype=waveform.dtype)
                self.kernel = kernel.to(waveform.device)
            shift = F._apply_sinc_resample_kernel(stretch, self.orig_freq, self.sample_rate, self.gcd, self.kernel,
                                                  self.width)
        else:
            shift = stretch
        return F._fix_waveform_shape(shift, shape)


class Speed(torch.nn.Module):
    r"""Adjust waveform speed (reference: _transforms.py:1958-2001)."""

    def __init__(self, orig_freq, factor) -> None:
        super().__init__()
        self.orig_freq = orig_freq
        self.factor = factor
        source = int(factor * orig_freq)
        target = int(orig_freq)
        gcd = math.gcd(source, target)
        self.source_sample_rate, self.target_sample_rate = source // gcd, target // gcd
        self.resampler = Resample(orig_freq=self.source_sample_rate, new_freq=self.target_sample_rate)

    def forward(self, waveform: Tensor, lengths: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        if lengths is None:
            out_lengths = None
        else:
            out_lengths = torch.ceil(lengths * self.target_sample_rate / self.source_sample_rate).to(lengths.dtype)
        return self.resampler(waveform), out_lengths


class SpeedPerturbation(torch.nn.Module):
    r"""Pick one of ``factors`` uniformly at random and adjust the speed by it (reference: _transforms.py:2004-2055)."""

    def __init__(self, orig_freq: int, factors) -> None:
        super().__init__()
        self.speeders = torch.nn.ModuleList([Speed(orig_freq=orig_freq, factor=factor) for factor in factors])

    def forward(self, waveform: Tensor, lengths: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        pick = int(torch.randint(len(self.speeders), ()))
        # (a walk instead of `self.speeders[pick]`: TorchScript indexes a ModuleList with constants only)
        for i, speeder in enumerate(self.speeders):
            if i == pick:
                return speeder(waveform, lengths)
        raise RuntimeError("SpeedPerturbation: no speeder was picked")


class AmplitudeToDB(torch.nn.Module):
    r"""Power/amplitude -> dB (reference: _transforms.py:300-346)."""
    __constants__ = ["multiplier", "amin", "ref_value", "db_multiplier"]

    def __init__(self, stype: str = "power", top_db: Optional[float] = None) -> None:
        super().__init__()
        self.stype = stype
        if top_db is not None and top_db < 0:
            raise ValueError("top_db must be positive value")
        self.top_db = top_db
        self.multiplier = 10.0 if stype == "power" else 20.0
        self.amin = 1e-10
        self.ref_value = 1.0
        self.db_multiplier = math.log10(max(self.amin, self.ref_value))

    def forward(self, x: Tensor) -> Tensor:
        if not torch.jit.is_scripting():
            if torch.compiler.is_compiling():
                return torch.ops.audio_amd.amplitude_to_DB(x, self.multiplier, self.amin, self.db_multiplier, self.top_db)
        return F.amplitude_to_DB(x, self.multiplier, self.amin, self.db_multiplier, self.top_db)


class MelScale(torch.nn.Module):
    r"""STFT bins -> mel bins with triangular filters (reference: _transforms.py:349-415)."""
    __constants__ = ["n_mels", "sample_rate", "f_min", "f_max"]

    def __init__(
        self,
        n_mels: int = 128,
        sample_rate: int = 16000,
        f_min: float = 0.0,
        f_max: Optional[float] = None,
        n_stft: int = 201,
        norm: Optional[str] = None,
        mel_scale: str = "htk",
    ) -> None:
        super().__init__()
        self.n_mels = n_mels
        self.sample_rate = sample_rate
        self.f_max = f_max if f_max is not None else float(sample_rate // 2)
        self.f_min = f_min
        self.norm = norm
        self.mel_scale = mel_scale
        if f_min > self.f_max:
            raise ValueError("Require f_min: {} <= f_max: {}".format(f_min, self.f_max))
        fb = F.melscale_fbanks(n_stft, self.f_min, self.f_max, self.n_mels, self.sample_rate, self.norm, self.mel_scale)
        self.register_buffer("fb", fb)

    def forward(self, specgram: Tensor) -> Tensor:
        return F.mel_scale(specgram, self.fb)


class MelSpectrogram(torch.nn.Module):
    r"""MelSpectrogram (reference: _transforms.py:506-622).  ``forward`` is ONE fused kernel:
    framing -> window -> FFT -> |X|^power -> banded mel; the (rows, T, freq) spectrogram the
    reference materialises between its two sub-modules never touches HBM."""
    __constants__ = ["sample_rate", "n_fft", "win_length", "hop_length", "pad", "n_mels", "f_min"]

    def __init__(
        self,
        sample_rate: int = 16000,
        n_fft: int = 400,
        win_length: Optional[int] = None,
        hop_length: Optional[int] = None,
        f_min: float = 0.0,
        f_max: Optional[float] = None,
        pad: int = 0,
        n_mels: int = 128,
        window_fn: Callable[..., Tensor] = torch.hann_window,
        power: float = 2.0,
        normalized: bool = False,
        wkwargs: Optional[dict] = None,
        center: bool = True,
        pad_mode: str = "reflect",
        onesided: Optional[bool] = None,
        norm: Optional[str] = None,
        mel_scale: str = "htk",
    ) -> None:
        super().__init__()
        torch._C._log_api_usage_once("torchaudio.transforms.MelSpectrogram")
        if onesided is not None:
            warnings.warn(
                "Argument 'onesided' has been deprecated and has no influence on the behavior of this module."
            )
        self.sample_rate = sample_rate
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.pad = pad
        self.power = power
        self.normalized = normalized
        self.n_mels = n_mels
        self.f_max = f_max
        self.f_min = f_min
        self.spectrogram = Spectrogram(
            n_fft=self.n_fft, win_length=self.win_length, hop_length=self.hop_length, pad=self.pad,
            window_fn=window_fn, power=self.power, normalized=self.normalized, wkwargs=wkwargs,
            center=center, pad_mode=pad_mode, onesided=True,
        )
        self.mel_scale = MelScale(self.n_mels, self.sample_rate, self.f_min, self.f_max, self.n_fft // 2 + 1,
                                  norm, mel_scale)
        self._plans: dict = {}      # per-(shape, device, buffer version) launch plans of the inference path

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_plans"] = {}            # launch plans hold op handles and device tensors: rebuilt on first use
        return state

    def _frame_major(self, waveform: Tensor, db=None) -> Tensor:
        sp = self.spectrogram
        return F._melspectrogram(waveform, sp.pad, sp.window, self.mel_scale.fb, sp.n_fft, sp.hop_length,
                                 sp.win_length, sp.power, sp.normalized, sp.center, sp.pad_mode, db=db)

    def forward(self, waveform: Tensor) -> Tensor:
        if not torch.jit.is_scripting():
            return self._forward_eager(waveform)
        return torch.ops.audio_amd.mel_spectrogram(waveform, self.spectrogram.window, self.mel_scale.fb, self.spectrogram.pad,
                                                   self.spectrogram.n_fft, self.spectrogram.hop_length,
                                                   self.spectrogram.win_length, float(self.power),
                                                   _norm_mode(self.spectrogram.normalized), self.spectrogram.center,
                                                   self.spectrogram.pad_mode)

    def _forward_eager(self, waveform: Tensor) -> Tensor:
        sp = self.spectrogram
        if torch.compiler.is_compiling():
            # torch.compile / torch.export: the module is ONE opaque op with a fake kernel (audio_amd/_ops.py); the host-side
            # plan caches (tensor identities, data pointers, ctypes) are not something a tracer should look into
            return torch.ops.audio_amd.mel_spectrogram(waveform, sp.window, self.mel_scale.fb, sp.pad, sp.n_fft,
                                                       sp.hop_length, sp.win_length, float(sp.power),
                                                       _norm_mode(sp.normalized), sp.center, sp.pad_mode)
        # low-precision / float64 / learnable / training / steady-state serving: F._melspectrogram_module (the kernel of the
        # operator above is the same function), with this module's own launch plans
        return F._melspectrogram_module(waveform, sp.window, self.mel_scale.fb, sp.pad, sp.n_fft, sp.hop_length, sp.win_length,
                                        sp.power, sp.normalized, sp.center, sp.pad_mode, self._plans)


class MFCC(torch.nn.Module):
    r"""MFCC (reference: _transforms.py:625-709): fused mel kernel, then dB (+ top_db = 80 with the
    reference's grouping of the cut-off) or log, then the DCT-II as one kernel."""
    __constants__ = ["sample_rate", "n_mfcc", "dct_type", "top_db", "log_mels"]

    def __init__(
        self,
        sample_rate: int = 16000,
        n_mfcc: int = 40,
        dct_type: int = 2,
        norm: str = "ortho",
        log_mels: bool = False,
        melkwargs: Optional[dict] = None,
    ) -> None:
        super().__init__()
        supported_dct_types = [2]
        if dct_type not in supported_dct_types:
            raise ValueError("DCT type not supported: {}".format(dct_type))
        self.sample_rate = sample_rate
        self.n_mfcc = n_mfcc
        self.dct_type = dct_type
        self.norm = norm
        self.top_db = 80.0
        self.amplitude_to_DB = AmplitudeToDB("power", self.top_db)
        melkwargs = melkwargs or {}
        self.MelSpectrogram = MelSpectrogram(sample_rate=self.sample_rate, **melkwargs)
        if self.n_mfcc > self.MelSpectrogram.n_mels:
            raise ValueError("Cannot select more MFCC coefficients than # mel bins")
        dct_mat = F.create_dct(self.n_mfcc, self.MelSpectrogram.n_mels, self.norm)
        self.register_buffer("dct_mat", dct_mat)
        self.log_mels = log_mels
        #: optional hook ``fn(group_max: Tensor) -> None`` run between the dB pass and the clamp;
        #: audio_amd.distributed installs an all-reduce(MAX) here when a batch is sharded.
        self.group_max_hook: Optional[Callable[[Tensor], None]] = None
        #: EXTENSION: which MFCC path runs.  True = the one-kernel MFCC (DCT on the f16 matrix pipe in the mel kernel's epilogue,
        #: operands split into two binary16 numbers, + a fix-up launch over the compacted list of tiles the top_db cut-off
        #: reaches); False = the exact two-kernel path; both are bit-reproducible call to call.  "auto" (default) decides ONCE
        #: per module, at its first eligible call, from the share of tiles that call had to redo (> 15 %: two kernels are
        #: cheaper) and keeps that arithmetic for every later call (F.MfccFusedState; round 3 re-decided from a polled event,
        #: so equal inputs could return different bits).  A group_max_hook (sharded batches) or a HIP-graph capture in
        #: progress run the one-kernel path without deciding.  Same results within 3e-5 dB.  Measured on the cfg4 batch
        #: (profiles/r03_d_mfcc_paths.txt): 211 us against 229 us for two kernels on noise, 220 against 234 with 5 % clamped
        #: tiles, 297 against 220 with 50 %.
        self.fused = "auto"
        #: the handle of this module's "auto" decision (F._mfcc_state_of): an integer, so that a scripted copy carries it and
        #: shares the decision -- and therefore the bits -- of the module it was scripted from
        self._fused_handle = secrets.randbits(62)

    __jit_unused_properties__ = ["_fused_state"]

    @property
    def _fused_state(self) -> "F.MfccFusedState":
        return F._mfcc_state_of(self._fused_handle)

    def fused_report(self) -> dict:
        """EXTENSION: which MFCC path ran last, the module's "auto" decision, and what the last one-kernel call's fix-up pass
        had to redo (this read synchronises with the device)."""
        st = F._mfcc_state_of(self._fused_handle)
        share = None
        if st.last_count is not None:
            share = float(st.last_count.item()) / max(st.last_tiles, 1)
        return {"path": st.path, "redone_share": share, "decided": st.decided, "decided_share": st.decided_share,
                "calls_fused": st.calls_fused, "calls_two_kernel": st.calls_two_kernel}

    def reset_fused_decision(self) -> None:
        """EXTENSION: forget the "auto" decision (the next eligible call takes it again, e.g. after the kind of batch changed)."""
        F._mfcc_state_of(self._fused_handle).reset()

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_fused_handle"] = secrets.randbits(62)     # a copy / an unpickled module decides for itself
        return d

    def forward(self, waveform: Tensor) -> Tensor:
        if not torch.jit.is_scripting():
            return self._forward_eager(waveform)
        return torch.ops.audio_amd.mfcc_module(
            waveform, self.MelSpectrogram.spectrogram.window, self.MelSpectrogram.mel_scale.fb, self.dct_mat,
            self.MelSpectrogram.spectrogram.pad, self.MelSpectrogram.spectrogram.n_fft,
            self.MelSpectrogram.spectrogram.hop_length, self.MelSpectrogram.spectrogram.win_length,
            float(self.MelSpectrogram.power), _norm_mode(self.MelSpectrogram.spectrogram.normalized),
            self.MelSpectrogram.spectrogram.center, self.MelSpectrogram.spectrogram.pad_mode, self.log_mels, self.top_db,
            self.amplitude_to_DB.multiplier, self.amplitude_to_DB.amin, self.amplitude_to_DB.db_multiplier,
            _fused_mode(self.fused), self._fused_handle)

    def _forward_eager(self, waveform: Tensor) -> Tensor:
        sp = self.MelSpectrogram.spectrogram
        if torch.compiler.is_compiling():          # one opaque op with a fake kernel (see MelSpectrogram.forward)
            return torch.ops.audio_amd.mfcc(waveform, sp.window, self.MelSpectrogram.mel_scale.fb, self.dct_mat, sp.pad,
                                            sp.n_fft, sp.hop_length, sp.win_length, float(sp.power),
                                            _norm_mode(sp.normalized), sp.center, sp.pad_mode, self.log_mels, self.top_db)
        a2db = self.amplitude_to_DB
        fused = _fused_mode(getattr(self, "fused", "auto"))
        return F._mfcc_module(waveform, sp.window, self.MelSpectrogram.mel_scale.fb, self.dct_mat, sp.pad, sp.n_fft,
                              sp.hop_length, sp.win_length, sp.power, sp.normalized, sp.center, sp.pad_mode, self.log_mels,
                              self.top_db, (a2db.multiplier, a2db.amin, a2db.db_multiplier), fused,
                              F._mfcc_state_of(self._fused_handle) if fused else None, self.group_max_hook,
                              self.MelSpectrogram._plans)


def _fused_mode(fused: Union[str, bool]) -> int:
    """``MFCC.fused`` as the op schema's integer: 0 = False (two kernels), 1 = True (one kernel), 2 = "auto"."""
    if isinstance(fused, str):
        return 2
    return 1 if fused else 0


class Resample(torch.nn.Module):
    r"""Resample (reference: _transforms.py:899-980): the tap table is computed once in float64
    and cached as the float32 buffer ``kernel``; forward is the polyphase HIP kernel."""

    def __init__(
        self,
        orig_freq: int = 16000,
        new_freq: int = 16000,
        resampling_method: str = "sinc_interp_hann",
        lowpass_filter_width: int = 6,
        rolloff: float = 0.99,
        beta: Optional[float] = None,
        *,
        dtype: Optional[torch.dtype] = None,
    ) -> None:
        super().__init__()
        self.orig_freq = orig_freq
        self.new_freq = new_freq
        self.gcd = math.gcd(int(self.orig_freq), int(self.new_freq))
        self.resampling_method = resampling_method
        self.lowpass_filter_width = lowpass_filter_width
        self.rolloff = rolloff
        self.beta = beta
        if self.orig_freq != self.new_freq:
            kernel, self.width = _host.sinc_resample_kernel(
                self.orig_freq, self.new_freq, self.gcd, self.lowpass_filter_width, self.rolloff,
                self.resampling_method, beta, dtype=dtype)
            self.register_buffer("kernel", kernel)

    def forward(self, waveform: Tensor) -> Tensor:
        if self.orig_freq == self.new_freq:
            return waveform
        if not torch.jit.is_scripting():
            if torch.compiler.is_compiling():          # one opaque op with a fake kernel (see MelSpectrogram.forward)
                return torch.ops.audio_amd.resample_apply(waveform, self.kernel, self.orig_freq, self.new_freq, self.gcd,
                                                          self.width)
        return F._apply_sinc_resample_kernel(waveform, self.orig_freq, self.new_freq, self.gcd, self.kernel,
                                             self.width)


class FFTConvolve(torch.nn.Module):
    r"""Convolution along the last dim (reference: _transforms.py:1906-1948)."""

    def __init__(self, mode: str = "full") -> None:
        super().__init__()
        F._check_convolve_mode(mode)
        self.mode = mode

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        if not torch.jit.is_scripting():
            if torch.compiler.is_compiling():
                return torch.ops.audio_amd.fftconvolve(x, y, self.mode)
        return F.fftconvolve(x, y, mode=self.mode)

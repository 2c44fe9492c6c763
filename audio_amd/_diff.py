"""Arbitrarily-often differentiable, dtype-generic (float32 / float64) building blocks on the HIP kernels.

The reference guarantees ``gradcheck`` AND ``gradgradcheck`` of its filtering and spectral ops and runs them in float64
(test/torchaudio_unittest/functional/autograd_impl.py:21-35, transforms/autograd_test_impl.py:30-45).  Its own autograd
Functions get there by writing every ``backward`` out of differentiable pieces (functional/filtering.py:941-1024:
``DifferentiableIIR.backward`` calls ``DifferentiableIIR.apply`` on the flipped gradient).  The same construction here:

* each LINEAR operator of the path is a ``torch.autograd.Function`` whose backward is the ``apply`` of its adjoint
  operator, itself such a Function -- the complex STFT and its adjoint (``aamd_spectrogram_* power <= 0`` /
  ``aamd_istft_*(adjoint=1)``) differentiate into each other for ever;
* everything non-linear around them (|X|^p, the filterbank product, dB, clamp) is ordinary torch arithmetic on the
  kernels' outputs, so autograd differentiates it to any order.

The fused float32 Functions of functional.py (one launch forward, three backward) stay the fast first-order path; their
``backward`` switches to these blocks when it runs under ``create_graph=True``.  float64 inputs take these blocks in the
forward pass too: float64 is the precision path (generic kernels, csrc/f64_paths.h), not the throughput path.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

import numpy as np
import torch
from torch import Tensor

from . import _host, _lib

FLOATS = (torch.float32, torch.float64)


def _suffix(dtype) -> str:
    return "f64" if dtype == torch.float64 else "f32"


def twiddles(n_fft: int, device, dtype) -> Tensor:
    from . import functional as F
    if dtype == torch.float32:
        return F._twiddles(n_fft, device)

    def make():
        t = np.arange(n_fft, dtype=np.float64)
        ang = 2.0 * np.pi * t / n_fft
        return torch.from_numpy(np.stack([np.cos(ang), -np.sin(ang)], axis=1)).to(device).contiguous()
    return F._cached(("tw64", n_fft, str(device)), make)


Cfg = Tuple[int, int, int, bool, str]     # n_fft, hop, pad, center, pad_mode


def _desc(rows: int, length: int, cfg: Cfg, n_frames: int, scale: float = 1.0):
    n_fft, hop, pad, center, pad_mode = cfg
    return _lib.StftDesc(rows, length, max(length, 1), n_fft, hop, pad, int(center), _lib.PAD_MODES[pad_mode], 1, n_frames,
                         scale, 0.0)


def n_frames_of(length: int, cfg: Cfg) -> int:
    n_fft, hop, pad, center, _ = cfg
    return _host.frame_count(length, n_fft, hop, center, pad)


class Stft(torch.autograd.Function):
    """x (rows, L) -> X (rows, T, F, 2): the onesided complex STFT with unit scale.  Linear; backward = StftAdjoint."""

    @staticmethod
    def forward(ctx, x2: Tensor, wp: Tensor, cfg: Cfg):
        x2 = x2.contiguous()
        rows, length = x2.shape
        T = n_frames_of(length, cfg)
        n_freq = cfg[0] // 2 + 1
        from . import functional as F
        ops = F._ops()
        if ops is not None and rows * T > 0:
            # the dispatcher-level ops (csrc/torch_shim.cpp): float32 = aamd::spectrogram with power <= 0 (complex frames)
            if x2.dtype == torch.float64:
                out = ops.spectrogram_f64(x2, wp, twiddles(cfg[0], x2.device, x2.dtype), cfg[0], cfg[1], cfg[2], bool(cfg[3]),
                                          _lib.PAD_MODES[cfg[4]], T)
            else:
                out = ops.spectrogram(x2, wp, twiddles(cfg[0], x2.device, x2.dtype), cfg[0], cfg[1], cfg[2], bool(cfg[3]),
                                      _lib.PAD_MODES[cfg[4]], True, T, 1.0, 0.0).view(rows, T, n_freq, 2)
            ctx.save_for_backward(wp)
            ctx.cfg, ctx.length = cfg, length
            return out
        out = torch.empty((rows, T, n_freq, 2), dtype=x2.dtype, device=x2.device)
        if out.numel():
            entry = getattr(_lib.lib(), "aamd_spectrogram_" + _suffix(x2.dtype))
            d = _desc(rows, length, cfg, T)
            with torch.cuda.device(x2.device):
                _lib.check(entry(x2.data_ptr(), wp.data_ptr(), twiddles(cfg[0], x2.device, x2.dtype).data_ptr(),
                                 out.data_ptr(), C.byref(d), _lib.current_stream(x2.device)))
        ctx.save_for_backward(wp)
        ctx.cfg, ctx.length = cfg, length
        return out

    @staticmethod
    def backward(ctx, dX):
        (wp,) = ctx.saved_tensors
        return StftAdjoint.apply(dX, wp, ctx.cfg, ctx.length), None, None


class StftAdjoint(torch.autograd.Function):
    """G (rows, T, F, 2) -> x (rows, L): the exact adjoint of Stft (padding map run backwards).  backward = Stft."""

    @staticmethod
    def forward(ctx, G: Tensor, wp: Tensor, cfg: Cfg, length: int):
        G = G.contiguous()
        rows, T = G.shape[0], G.shape[1]
        from . import functional as F
        ops = F._ops()
        if ops is not None and rows * length > 0 and T:
            op = ops.istft_f64 if G.dtype == torch.float64 else ops.istft
            out = op(G, wp, twiddles(cfg[0], G.device, G.dtype), None, cfg[0], cfg[1], cfg[2], bool(cfg[3]),
                     _lib.PAD_MODES[cfg[4]], length, 1.0, True)
            ctx.save_for_backward(wp)
            ctx.cfg = cfg
            return out
        out = torch.zeros((rows, length), dtype=G.dtype, device=G.device)
        if out.numel() and T:
            entry = getattr(_lib.lib(), "aamd_istft_" + _suffix(G.dtype))
            d = _desc(rows, length, cfg, T)
            with torch.cuda.device(G.device):
                _lib.check(entry(G.data_ptr(), wp.data_ptr(), twiddles(cfg[0], G.device, G.dtype).data_ptr(), None,
                                 out.data_ptr(), C.byref(d), 1, _lib.current_stream(G.device)))
        ctx.save_for_backward(wp)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, dx):
        (wp,) = ctx.saved_tensors
        return Stft.apply(dx, wp, ctx.cfg), None, None, None


def stft_complex(x2: Tensor, wp: Tensor, cfg: Cfg) -> Tensor:
    """(rows, L) -> complex (rows, T, F), differentiable to any order."""
    return torch.view_as_complex(Stft.apply(x2, wp, cfg))


def hermitian_extend(spec_f: Tensor, n_fft: int) -> Tensor:
    """(rows, T, n_fft // 2 + 1) onesided spectrum of a REAL signal -> (rows, T, n_fft): X[N - k] = conj X[k]
    (what aten::stft returns for onesided=False; plain tensor arithmetic, so autograd sees through it)."""
    mirror = spec_f[..., 1:(n_fft + 1) // 2].flip(-1).conj()
    return torch.cat([spec_f, mirror], dim=-1)


def onesided_part(spec: Tensor, n_fft: int) -> Tensor:
    """(..., n_fft, T) two-sided spectrum -> its first n_fft // 2 + 1 bins.  aten::istft with onesided=False does exactly
    this before its Hermitian inverse FFT (the mirror half is ignored, not folded in -- checked against torch.istft:
    tests/test_gpu_autograd_f64.py::test_inverse_spectrogram_gradcheck_other_shapes)."""
    return spec[..., : n_fft // 2 + 1, :]


def stft_complex_learnable_window(x2: Tensor, w_padded: Tensor, cfg: Cfg) -> Tensor:
    """(rows, L) -> complex (rows, T, F) with gradients into the WINDOW as well (round 5; the reference propagates into a
    learnable window through aten::stft, transforms/_transforms.py:101-123).  The STFT is bilinear in (signal, window): the
    frames are cut by tensor arithmetic the way the kernel cuts them (explicit zero padding, then the centre padding of
    `pad_mode`), multiplied by the window -- both differentiable by autograd -- and transformed by the SAME onesided kernel,
    called on the windowed frames as rows with a unit window, hop = n_fft and no padding (`Stft`, whose backward is its adjoint
    kernel).  Differentiable to any order in x and in the window; a training path, not the throughput path (it materialises
    the (rows, T, n_fft) frame tensor)."""
    n_fft, hop, pad, center, pad_mode = cfg
    rows = x2.shape[0]
    xp = x2
    if pad > 0:
        xp = torch.nn.functional.pad(xp, (pad, pad))
    if center:
        mode = {"constant": "constant", "reflect": "reflect", "replicate": "replicate", "circular": "circular"}[pad_mode]
        xp = torch.nn.functional.pad(xp.unsqueeze(0), (n_fft // 2, n_fft // 2), mode=mode).squeeze(0)
    fr = xp.unfold(-1, n_fft, hop)                                      # (rows, T, n_fft) view
    T = fr.shape[1]
    fr = (fr * w_padded).reshape(rows * T, n_fft)
    ones = torch.ones(n_fft, dtype=x2.dtype, device=x2.device)
    X = Stft.apply(fr, ones, (n_fft, n_fft, 0, False, "constant"))     # (rows * T, 1, F, 2)
    return torch.view_as_complex(X.reshape(rows, T, n_fft // 2 + 1, 2))


def resample_learnable_kernel(x2: Tensor, kern: Tensor, orig: int, new: int, width: int) -> Tensor:
    """(rows, L) x (new, 2 width + orig) tap table -> (rows, ceil(new L / orig)) with gradients into the TAP TABLE (and the
    signal): the reference's own formulation -- a strided correlation, functional/functional.py:1419-1431 -- as an unfold and
    a matrix product, which autograd differentiates in both operands to any order.  A training path for a learnable resampling
    filter; the throughput path is the banded MFMA kernel (csrc/resample_mfma.h)."""
    rows, length = x2.shape
    xp = torch.nn.functional.pad(x2, (width, width + orig))
    fr = xp.unfold(-1, 2 * width + orig, orig)                          # (rows, Q, taps)
    y = torch.matmul(fr, kern.t())                                      # (rows, Q, new)
    y = y.reshape(rows, -1)
    target = -(-new * length // orig)
    return y[:, :target]


def spectrogram(waveform: Tensor, pad: int, window: Tensor, n_fft: int, hop_length: int, win_length: int, power,
                normalized, center: bool, pad_mode: str, onesided: bool = True) -> Tensor:
    """The reference composition (functional/functional.py:112-145) with torch.stft replaced by `Stft`:
    (..., L) -> (..., F, T) in the waveform's dtype (float32 or float64); onesided=False extends the spectrum of the real
    signal by its Hermitian symmetry (functional.py:123-134 passes the flag to aten::stft for every dtype)."""
    from . import functional as F
    frame_length_norm, window_norm = F._get_spec_norms(normalized)
    shape = waveform.size()
    x2 = waveform.reshape(-1, shape[-1])
    w = window.to(device=waveform.device, dtype=waveform.dtype)
    if pad_mode not in _lib.PAD_MODES:
        raise NotImplementedError(f"audio_amd: pad_mode {pad_mode!r} is not supported")
    if torch.is_grad_enabled() and w.requires_grad:
        # a learnable window: the bilinear form (gradients into the window and the signal)
        spec_f = stft_complex_learnable_window(x2, _host.center_pad_window(w, n_fft), (n_fft, hop_length, pad, bool(center), pad_mode))
    else:
        wp = _host.center_pad_window(w.detach(), n_fft).contiguous()
        spec_f = stft_complex(x2, wp, (n_fft, hop_length, pad, bool(center), pad_mode))       # (rows, T, F)
    if frame_length_norm:
        spec_f = spec_f * (float(n_fft) ** -0.5)
    if window_norm:
        spec_f = spec_f / w.pow(2.0).sum().sqrt()
    if not onesided:
        spec_f = hermitian_extend(spec_f, n_fft)
    spec_f = spec_f.transpose(-1, -2)
    spec_f = spec_f.reshape(tuple(shape[:-1]) + spec_f.shape[-2:])
    if power is not None:
        if power == 1.0:
            return spec_f.abs()
        return spec_f.abs().pow(power)
    return spec_f


def istft(fm: Tensor, wp: Tensor, n_fft: int, hop_length: int, center: bool, out_len: int, inv_env: Tensor,
          scale: float) -> Tensor:
    """Differentiable inverse STFT: complex (rows, T, F) frame-major onesided spectrum -> (rows, out_len).

    aten::istft is LINEAR in the spectrum: Hermitian inverse FFT of every frame (interior bins count twice, the imaginary
    parts of DC / Nyquist are dropped), window, overlap-add, envelope division, centre trimming.  That is the adjoint of the
    zero-padded onesided STFT applied to the spectrum with its interior bins doubled:
        y = inv_env * (scale / N) * StftAdjoint_{constant padding}(c * S),   c = (1, 2, ..., 2, 1),
    and `StftAdjoint` differentiates into `Stft` and back for ever (the module docstring), so this is differentiable to any
    order in float32 and float64.  The pair is taken at the length the T frames COVER after the centre trim,
    hop (T - 1) + n_fft / 2 (centre) or n_fft (not centre) -- torch.istft returns those samples when asked for a longer
    `length` -- where the forward operator has T' >= T frames: the spectrum is extended by T' - T zero frames, which makes the
    operator pair consistent and changes nothing else.  The caller's length then trims or zero-extends."""
    rows, T, n_freq = fm.shape
    c = torch.full((n_freq,), 2.0, dtype=wp.dtype, device=fm.device)
    c[0] = 1.0
    if n_fft % 2 == 0:
        c[-1] = 1.0
    G = torch.view_as_real(fm * c)
    cfg = (n_fft, hop_length, 0, bool(center), "constant")
    covered = hop_length * (T - 1) + (n_fft // 2 if center else n_fft)
    t_op = n_frames_of(covered, cfg)
    if t_op > T:
        G = torch.nn.functional.pad(G, (0, 0, 0, 0, 0, t_op - T))
    x = StftAdjoint.apply(G, wp, cfg, covered)
    if out_len < covered:
        x = x[:, :out_len]
    elif out_len > covered:
        x = torch.nn.functional.pad(x, (0, out_len - covered))
    return x * (inv_env * (scale / n_fft))


def phase_vocoder(spec: Tensor, rate: float, phase_advance: Tensor) -> Tensor:
    """The phase vocoder as tensor arithmetic in the spectrum's own precision (functional/functional.py:765-803:
    interpolated magnitudes, unwrapped phase increments, running phase sum) -- the differentiable / complex128 route of this
    thin caller; the throughput route is the HIP kernel csrc/vocoder.h.  spec: complex (..., F, T)."""
    import math
    shape = spec.size()
    rdt = torch.float64 if spec.dtype == torch.complex128 else torch.float32
    z = torch.nn.functional.pad(spec.reshape((-1,) + tuple(shape[-2:])), [0, 2])
    t = torch.arange(0, shape[-1], rate, device=spec.device, dtype=rdt)
    frac = t % 1.0
    i0 = t.long()
    z0, z1 = z.index_select(-1, i0), z.index_select(-1, i0 + 1)
    pa = phase_advance.to(device=spec.device, dtype=rdt)
    step = z1.angle() - z0.angle() - pa
    step = step - 2 * math.pi * torch.round(step / (2 * math.pi)) + pa
    phase = torch.cumsum(torch.cat([z[..., :1].angle(), step[..., :-1]], dim=-1), -1)
    out = torch.polar(frac * z1.abs() + (1 - frac) * z0.abs(), phase)
    return out.reshape(tuple(shape[:-2]) + out.shape[1:])

"""gradcheck AND gradgradcheck in float64 on the device -- what the reference guarantees for this path and how it tests it
(test/torchaudio_unittest/functional/autograd_impl.py:21-120, transforms/autograd_test_impl.py:24-135, 183-200, 320-333):
same inputs (white noise of the same length), same transforms moved to float64 with `.to(dtype=torch.float64)`, default
gradcheck tolerances.  The float64 kernels are csrc/f64_paths.h + the generic STFT / inverse-STFT kernels instantiated on
double; every backward is built from differentiable pieces (audio_amd/_diff.py), so the second-order check passes too.
`nondet_tol` as in the reference: its replication-pad backward, and here the atomic overlap-add of the STFT adjoint, sum in
a run-dependent order (differences ~1e-17)."""
import math
from functools import partial

import pytest
import torch
from torch.autograd import gradcheck, gradgradcheck

pytestmark = pytest.mark.gpu


def _noise(n_channels, n, seed=0, scale=0.3):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(n_channels, n, generator=g)).clamp_(-1, 1)


def _assert_grad(fn, inputs, enable_all_grad=True, nondet_tol=0.0):
    ins = []
    for i in inputs:
        if torch.is_tensor(i):
            req = i.requires_grad
            i = i.detach().to(dtype=torch.complex128 if i.is_complex() else torch.float64, device="cuda")
            i.requires_grad = True if enable_all_grad else req
        ins.append(i)
    assert gradcheck(fn, ins, nondet_tol=nondet_tol)
    assert gradgradcheck(fn, ins, nondet_tol=nondet_tol)


# ---------------------------------------------------------------------------------------------------- lfilter family

A = torch.tensor([0.7, 0.2, 0.6])
B = torch.tensor([0.4, 0.2, 0.9])


@pytest.mark.parametrize("which", ["x", "a", "b", "all"])
def test_lfilter(which):
    import audio_amd.functional as F
    x = _noise(2, 220)           # 22050 Hz x 0.01 s, as the reference
    a, b = A.clone(), B.clone()
    if which != "all":
        {"x": x, "a": a, "b": b}[which].requires_grad = True
    _assert_grad(F.lfilter, (x, a, b), enable_all_grad=(which == "all"))


def test_lfilter_filterbanks_and_batching():
    import audio_amd.functional as F
    a = torch.tensor([[0.7, 0.2, 0.6], [0.8, 0.2, 0.9]])
    b = torch.tensor([[0.4, 0.2, 0.9], [0.7, 0.2, 0.6]])
    _assert_grad(partial(F.lfilter, batching=False), (_noise(3, 220), a, b))
    _assert_grad(F.lfilter, (_noise(2, 220), a, b))


def test_lfilter_without_clamp_and_float32_second_order_runs():
    import audio_amd.functional as F
    _assert_grad(partial(F.lfilter, clamp=False), (_noise(2, 120), A.clone(), B.clone()))
    # float32: the same differentiable-backward construction (too coarse for gradgradcheck's tolerances, so: it runs and
    # agrees with the float64 second derivative)
    x32 = _noise(2, 120).cuda().requires_grad_()
    a32, b32 = A.cuda().requires_grad_(), B.cuda().requires_grad_()
    y = F.lfilter(x32, a32, b32)
    (gx,) = torch.autograd.grad(y.pow(2).sum(), x32, create_graph=True)
    (ga,) = torch.autograd.grad(gx.pow(2).sum(), a32)
    x64 = x32.detach().double().requires_grad_()
    a64, b64 = A.cuda().double().requires_grad_(), B.cuda().double().requires_grad_()
    y64 = F.lfilter(x64, a64, b64)
    (gx64,) = torch.autograd.grad(y64.pow(2).sum(), x64, create_graph=True)
    (ga64,) = torch.autograd.grad(gx64.pow(2).sum(), a64)
    assert torch.allclose(ga.double(), ga64, rtol=2e-3, atol=1e-4 * float(ga64.abs().max()))


@pytest.mark.parametrize("which", ["a", "b", "all"])
def test_filtfilt(which):
    import audio_amd.functional as F
    x = _noise(2, 220)
    a, b = A.clone(), B.clone()
    if which != "all":
        {"a": a, "b": b}[which].requires_grad = True
    _assert_grad(F.filtfilt, (x, a, b), enable_all_grad=(which == "all"))


def test_biquad_and_designers():
    import audio_amd.functional as F
    x = _noise(1, 220)
    _assert_grad(F.biquad, (x, B[0], B[1], B[2], A[0], A[1], A[2]))
    sr = 22050
    _assert_grad(lambda w, f, q: F.lowpass_biquad(w, sr, f, q), (x, torch.tensor(4000.0), torch.tensor(0.7)))
    _assert_grad(lambda w, f, q: F.band_biquad(w, sr, f, q, True), (x, torch.tensor(800.0), torch.tensor(0.7)))
    _assert_grad(lambda w, g, f, q: F.equalizer_biquad(w, sr, f, g, q), (x, torch.tensor(-6.0), torch.tensor(800.0), torch.tensor(0.7)))


# --------------------------------------------------------------------------------------------------------- STFT family


@pytest.mark.parametrize("kwargs", [
    {"pad": 0, "normalized": False, "power": None}, {"pad": 3, "normalized": True, "power": None},
    {"pad": 0, "normalized": False, "power": 1.0}, {"pad": 3, "normalized": True, "power": 1.0},
    {"pad": 0, "normalized": False, "power": 2.0}, {"pad": 3, "normalized": False, "power": 2.0},
    {"pad": 0, "normalized": True, "power": 2.0}, {"pad": 3, "normalized": True, "power": 2.0},
], ids=str)
def test_spectrogram(kwargs):
    import audio_amd.transforms as T
    t = T.Spectrogram(**kwargs).to(dtype=torch.float64, device="cuda")
    _assert_grad(t, [_noise(2, 400)], nondet_tol=1e-10)           # 8000 Hz x 0.05 s, default n_fft = 400


def test_spectrogram_small_n_fft_and_other_pad_modes():
    import audio_amd.transforms as T
    for pm in ("reflect", "constant", "replicate", "circular"):
        t = T.Spectrogram(n_fft=64, hop_length=16, win_length=48, pad_mode=pm).to(dtype=torch.float64, device="cuda")
        _assert_grad(t, [_noise(1, 150, seed=3)], nondet_tol=1e-10)


def test_melspectrogram_and_melscale():
    import audio_amd.transforms as T
    t = T.MelSpectrogram(sample_rate=8000).to(dtype=torch.float64, device="cuda")
    _assert_grad(t, [_noise(2, 400)], nondet_tol=1e-10)
    ms = T.MelScale(sample_rate=8000, n_stft=201).to(dtype=torch.float64, device="cuda")
    _assert_grad(ms, [torch.rand(2, 201, 6, generator=torch.Generator().manual_seed(1))])


@pytest.mark.parametrize("log_mels", [False, True])
def test_mfcc(log_mels):
    import audio_amd.transforms as T
    t = T.MFCC(sample_rate=8000, log_mels=log_mels).to(dtype=torch.float64, device="cuda")
    _assert_grad(t, [_noise(2, 400)], nondet_tol=1e-10)


def test_amplitude_to_db():
    import audio_amd.transforms as T
    spec = torch.rand(2, 30, 8, generator=torch.Generator().manual_seed(2)) + 0.1
    _assert_grad(T.AmplitudeToDB().to(dtype=torch.float64, device="cuda"), [spec])
    _assert_grad(T.AmplitudeToDB("magnitude", top_db=20.0).to(dtype=torch.float64, device="cuda"), [spec])


# ---------------------------------------------------------------------------------------------- resample / fftconvolve


@pytest.mark.parametrize("orig_freq,new_freq", [(8000, 8000), (8000, 4000), (4000, 8000), (8000, 6000)])
def test_resample(orig_freq, new_freq):
    import audio_amd.transforms as T
    t = T.Resample(orig_freq=orig_freq, new_freq=new_freq).to(dtype=torch.float64, device="cuda")
    _assert_grad(t, [_noise(2, 400)])


@pytest.mark.parametrize("mode", ["full", "valid", "same"])
def test_fftconvolve(mode):
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 25, generator=g)
    y = torch.rand(1, 3, 11, generator=g)
    _assert_grad(T.FFTConvolve(mode=mode), [x, y])


def test_float64_forward_matches_float32_kernels_and_oracle():
    """float64 entry points against the fp32 fast kernels (1e-6) and the float64 oracle (1e-12)."""
    import numpy as np
    import audio_amd.functional as F
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    x = _noise(2, 3000, seed=9).cuda()
    mel32 = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=40).cuda()
    mel64 = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=40).to(dtype=torch.float64, device="cuda")
    with torch.no_grad():
        y32, y64 = mel32(x), mel64(x.double())
    assert y64.dtype == torch.float64 and float((y32.double() - y64).abs().max() / y64.abs().max()) <= 2e-6
    want = O.mel_spectrogram(x.cpu().numpy().astype(np.float64), mel64.spectrogram.window.cpu().numpy(),
                             mel64.mel_scale.fb.cpu().numpy(), 400, 160)
    assert float(np.abs(y64.cpu().numpy() - want).max() / np.abs(want).max()) <= 1e-12
    a = torch.tensor([1.0, -1.5, 0.7], dtype=torch.float64).cuda()
    b = torch.tensor([0.3, 0.1, 0.2], dtype=torch.float64).cuda()
    with torch.no_grad():
        z = F.lfilter(x.double(), a, b)
    wz = O.lfilter(x.cpu().numpy().astype(np.float64), a.cpu().numpy(), b.cpu().numpy())
    assert float(np.abs(z.cpu().numpy() - wz).max()) <= 1e-12
    with torch.no_grad():
        r = F.resample(x.double(), 16000, 11025)
    wr = O.resample(x.cpu().numpy().astype(np.float64), 16000, 11025)
    assert float(np.abs(r.cpu().numpy() - wr).max()) <= 1e-12


# ---- float64 (complex128) through the two widening ops that used to be float32-only ------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(n_fft=400, hop=160, L=8000), dict(n_fft=512, hop=128, L=6000, length=6000),
                                 dict(n_fft=200, hop=50, win_length=160, L=2222)])
def test_inverse_spectrogram_complex128(cfg):
    """complex128 spectrograms run on the float64 inverse-STFT kernel and return float64 (the reference accepts
    complex128, and its tests run in float64): against torch.istft in float64 on the CPU, 1e-11 of the peak."""
    import audio_amd.functional as F
    n_fft, hop, L = cfg["n_fft"], cfg["hop"], cfg["L"]
    wl = cfg.get("win_length", n_fft)
    g = torch.Generator().manual_seed(n_fft)
    x = 0.5 * torch.randn(3, L, generator=g, dtype=torch.float64)
    w = torch.hann_window(wl, dtype=torch.float64)
    X = torch.stft(x, n_fft, hop, wl, w, center=True, pad_mode="reflect", return_complex=True)
    ref = torch.istft(X, n_fft, hop, wl, w, center=True, length=cfg.get("length"))
    with torch.no_grad():
        got = F.inverse_spectrogram(X.cuda(), cfg.get("length"), 0, w.cuda(), n_fft, hop, wl, False)
    assert got.dtype == torch.float64 and got.shape == ref.shape
    assert float((got.cpu() - ref).abs().max()) <= 1e-11 * float(ref.abs().max())


@pytest.mark.gpu
def test_phase_vocoder_complex128():
    """complex128 input: the float64 precision path (reference formula on device tensors) -- equal to the same formula
    evaluated on the CPU, and consistent with the float32 kernel on the same data at the float32 phase-sum noise level."""
    import audio_amd.functional as F
    g = torch.Generator().manual_seed(3)
    spec = torch.randn(2, 201, 120, generator=g, dtype=torch.float64) + 1j * torch.randn(2, 201, 120, generator=g,
                                                                                         dtype=torch.float64)
    pa = torch.linspace(0, math.pi * 160, 201, dtype=torch.float64)[..., None]
    with torch.no_grad():
        got = F.phase_vocoder(spec.cuda(), 1.3, pa.cuda())
        from audio_amd import _diff
        cpu = _diff.phase_vocoder(spec, 1.3, pa)
        f32 = F.phase_vocoder(spec.to(torch.complex64).cuda(), 1.3, pa.float().cuda())
    assert got.dtype == torch.complex128 and got.shape == (2, 201, 93)
    assert float((got.cpu() - cpu).abs().max()) <= 1e-9 * float(cpu.abs().max())
    assert float((got.abs().float() - f32.abs()).abs().max()) <= 1e-5 * float(cpu.abs().max())


# ---- SURVEY 8(f) rank 2 in training mode (VERDICT r3 missing 3): the reference's own cases --------------------------------
def _get_spectrogram(waveform, n_fft, hop_length=None, power=None):
    """test/torchaudio_unittest/common_utils/data_utils.py:121-159 (hop n_fft // 4, Hann window, centre / reflect)."""
    hop_length = hop_length or n_fft // 4
    spec = torch.stft(waveform, n_fft=n_fft, hop_length=hop_length, win_length=n_fft, window=torch.hann_window(n_fft),
                      center=True, pad_mode="reflect", return_complex=True)
    return spec if power is None else spec.abs() ** power


class _Deterministic(torch.nn.Module):
    def __init__(self, transform, seed=0):
        super().__init__()
        self.seed, self.transform = seed, transform

    def forward(self, x):
        torch.random.manual_seed(self.seed)
        return self.transform(x)


def test_inverse_spectrogram_gradcheck():
    """transforms/autograd_test_impl.py:76-83: a realistic complex spectrogram, the waveform's length as `length`."""
    import audio_amd.transforms as T
    waveform = _noise(2, 400)
    spec = _get_spectrogram(waveform, n_fft=400)
    inv = T.InverseSpectrogram(n_fft=400).to(dtype=torch.float64, device="cuda")
    _assert_grad(lambda s: inv(s, 400), [spec], nondet_tol=1e-10)


@pytest.mark.parametrize("kw", [dict(n_fft=64, hop_length=16), dict(n_fft=64, hop_length=16, length=150),
                                dict(n_fft=64, hop_length=16, length=300), dict(n_fft=48, hop_length=12, center=False),
                                dict(n_fft=64, hop_length=16, normalized=True, onesided=False)], ids=str)
def test_inverse_spectrogram_gradcheck_other_shapes(kw):
    """natural length, trimmed, zero-extended, centre off, two-sided + normalised -- and the VALUES of the training path against
    torch.istft on the CPU (the forward of the differentiable operator is a different kernel call than inference)."""
    import audio_amd.transforms as T
    kw = dict(kw)
    length = kw.pop("length", None)
    n_fft, hop = kw["n_fft"], kw["hop_length"]
    onesided, center = kw.get("onesided", True), kw.get("center", True)
    x = _noise(2, 200, seed=4).double()
    wfn = torch.hann_window if center else torch.hamming_window          # (centre off: the Hann window's zero end fails NOLA)
    inv = T.InverseSpectrogram(window_fn=wfn, **kw).to(dtype=torch.float64, device="cuda")
    w = inv.window.detach().cpu()            # the module's buffer: float32 values widened (as the reference's .to(float64) does)
    spec = torch.stft(x, n_fft, hop, n_fft, w, center=center, pad_mode="reflect", onesided=onesided, return_complex=True)
    if not onesided:
        spec = spec + 0.05 * torch.randn(spec.shape, dtype=torch.complex128, generator=torch.Generator().manual_seed(1))
    ref = torch.istft(spec, n_fft, hop, n_fft, w, center=center, onesided=onesided, length=length)
    if kw.get("normalized"):          # torchaudio's normalized=True is the WINDOW norm (functional.py:205-207), not aten's 1/sqrt(n_fft)
        ref = ref * w.pow(2).sum().sqrt()
    s_dev = spec.cuda().requires_grad_()
    got = inv(s_dev, length)
    assert got.shape == ref.shape and float((got.detach().cpu() - ref).abs().max()) <= 1e-11 * float(ref.abs().max())
    with torch.no_grad():
        plain = inv(spec.cuda(), length)                                  # inference kernel: same values
    assert float((plain - got.detach()).abs().max()) <= 1e-12 * float(ref.abs().max())
    _assert_grad(lambda s: inv(s, length), [spec], nondet_tol=1e-10)


def test_inverse_spectrogram_float32_training_path_matches_inference():
    import audio_amd.transforms as T
    x = _noise(3, 8000, seed=8)
    spec = _get_spectrogram(x, n_fft=400, hop_length=160).cuda()
    inv = T.InverseSpectrogram(n_fft=400, hop_length=160).cuda()
    with torch.no_grad():
        plain = inv(spec, 8000)
    s = spec.clone().requires_grad_()
    got = inv(s, 8000)
    assert float((got.detach() - plain).abs().max()) <= 2e-6 * float(plain.abs().max())
    (g,) = torch.autograd.grad(got.square().sum(), s)
    # d/dS sum(istft(S)^2) = 2 * adjoint(istft)(y): compare with torch's own autograd through torch.istft on the CPU
    sc = spec.cpu().clone().requires_grad_()
    yc = torch.istft(sc, 400, 160, 400, torch.hann_window(400), length=8000)
    (gc,) = torch.autograd.grad(yc.square().sum(), sc)
    assert float((g.cpu() - gc).abs().max()) <= 1e-5 * float(gc.abs().max())


@pytest.mark.parametrize("momentum", [0, 0.99])
@pytest.mark.parametrize("rand_init", [False, True])
def test_griffinlim(momentum, rand_init):
    """transforms/autograd_test_impl.py:99-109."""
    import audio_amd.transforms as T
    n_fft, power, n_iter = 80, 1, 2
    spec = _get_spectrogram(_noise(2, 80), n_fft=n_fft, power=power)
    t = _Deterministic(T.GriffinLim(n_fft=n_fft, n_iter=n_iter, momentum=momentum, rand_init=rand_init, power=power))
    _assert_grad(t.to(dtype=torch.float64, device="cuda"), [spec], nondet_tol=1e-10)


def test_griffinlim_training_path_matches_reference_composition_and_inference():
    """The differentiable iteration against (a) the same algorithm on torch.stft / torch.istft in float64 on the CPU (values AND
    gradient) and (b) the float32 inference kernels (the fused phase update) on the same spectrogram."""
    import audio_amd.transforms as T
    n_fft, hop = 400, 100
    x = _noise(2, 3200, seed=6).double()
    gl = T.GriffinLim(n_fft=n_fft, hop_length=hop, n_iter=3, momentum=0.9, rand_init=False, power=2.0, length=3200)
    gl = gl.to(dtype=torch.float64, device="cuda")
    w = gl.window.detach().cpu()             # the module's buffer (float32 values widened), as in the reference's own float64 tests
    spec = (torch.stft(x, n_fft, hop, n_fft, w, return_complex=True).abs() ** 2)

    def cpu_gl(sp):
        mag = sp.pow(0.5)
        ang = torch.full(mag.size(), 1, dtype=torch.complex128)
        tprev = None
        for _ in range(3):
            inv = torch.istft(mag * ang, n_fft, hop, n_fft, w, length=3200)
            reb = torch.stft(inv, n_fft, hop, n_fft, w, return_complex=True)
            ang = reb if tprev is None else reb - tprev * (0.9 / 1.9)
            ang = ang / (ang.abs() + 1e-16)
            tprev = reb
        return torch.istft(mag * ang, n_fft, hop, n_fft, w, length=3200)

    sc = spec.clone().requires_grad_()
    yc = cpu_gl(sc)
    (gc,) = torch.autograd.grad(yc.square().sum(), sc)
    sd = spec.cuda().requires_grad_()
    yd = gl(sd)
    (gd,) = torch.autograd.grad(yd.square().sum(), sd)
    assert float((yd.detach().cpu() - yc.detach()).abs().max()) <= 1e-9 * float(yc.abs().max())
    assert float((gd.cpu() - gc).abs().max()) <= 1e-7 * float(gc.abs().max())
    gl32 = T.GriffinLim(n_fft=n_fft, hop_length=hop, n_iter=3, momentum=0.9, rand_init=False, power=2.0, length=3200).cuda()
    with torch.no_grad():
        y32 = gl32(spec.float().cuda())
    assert float((y32.double().cpu() - yc.detach()).abs().max()) <= 2e-4 * float(yc.abs().max())


@pytest.mark.parametrize("rate", [0.7, 0.8, 0.9, 1.0, 1.3])
def test_timestretch_non_zero(rate):
    """transforms/autograd_test_impl.py:238-262: spectrogram points near zero are pushed away from the origin (atan2)."""
    import audio_amd.transforms as T
    n_fft = 16
    t = T.TimeStretch(n_freq=n_fft // 2 + 1, fixed_rate=rate).to(device="cuda")
    spec = _get_spectrogram(_noise(2, 40, scale=1.0), n_fft=n_fft)
    eps = 2e-2
    close = spec.abs() < eps
    spec[close] = eps * spec[close] / spec[close].abs()
    _assert_grad(t, [spec])


def test_timestretch_float32_training_path_matches_the_kernel():
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(2)
    spec = torch.complex(torch.randn(2, 201, 60, generator=g), torch.randn(2, 201, 60, generator=g)).cuda()
    t = T.TimeStretch(n_freq=201, hop_length=160, fixed_rate=1.2).cuda()
    with torch.no_grad():
        plain = t(spec)
    s = spec.clone().requires_grad_()
    got = t(s)
    assert got.shape == plain.shape
    assert float((got.detach().abs() - plain.abs()).abs().max()) <= 1e-5 * float(plain.abs().max())
    (gr,) = torch.autograd.grad(got.abs().square().sum(), s)
    assert torch.isfinite(torch.view_as_real(gr)).all()


@pytest.mark.parametrize("kwargs", [{"power": None}, {"power": 2.0, "normalized": True}, {"power": 1.0, "pad": 3}], ids=str)
def test_spectrogram_two_sided(kwargs):
    """VERDICT r3 missing 5: onesided=False in float64 (and in training mode) -- values against torch.stft, then gradcheck."""
    import audio_amd.transforms as T
    t = T.Spectrogram(n_fft=64, hop_length=16, onesided=False, **kwargs).to(dtype=torch.float64, device="cuda")
    x = _noise(2, 200, seed=5).double()
    pad = kwargs.get("pad", 0)
    xp = torch.nn.functional.pad(x, (pad, pad))
    w = t.window.detach().cpu()              # the module's buffer (float32 values widened by .to(float64), as in the reference)
    ref = torch.stft(xp, 64, 16, 64, w, center=True, pad_mode="reflect", normalized=False, onesided=False, return_complex=True)
    if kwargs.get("normalized"):
        ref = ref / w.pow(2).sum().sqrt()
    if kwargs["power"] is not None:
        ref = ref.abs().pow(kwargs["power"])
    got = t(x.cuda())
    assert got.shape == ref.shape == (2, 64, ref.shape[-1])
    assert float((got.cpu() - ref).abs().max()) <= 1e-11 * float(ref.abs().max())
    _assert_grad(t, [_noise(2, 200, seed=5)], nondet_tol=1e-10)
    # float32 in training mode takes the same route; inference runs the two-sided kernel
    t32 = T.Spectrogram(n_fft=64, hop_length=16, onesided=False, **kwargs).cuda()
    xa = x.float().cuda().requires_grad_()
    ya = t32(xa)
    with torch.no_grad():
        yb = t32(x.float().cuda())
    assert float((ya.detach() - yb).abs().max()) <= 5e-6 * float(yb.abs().max())
    torch.autograd.grad((ya.abs() if ya.is_complex() else ya).sum(), xa)

"""TorchScript drop-in (VERDICT r5 missing #3): every module / function of the path compiles with ``torch.jit.script``, survives a
``save`` / ``load`` round trip with its ``state_dict`` keys, and -- on the device -- returns the SAME BITS as the eager call.

Mirrors the reference's own consistency suites (parameter values from
/root/reference/test/torchaudio_unittest/transforms/torchscript_consistency_impl.py:13-76, 195-241 and
functional/torchscript_consistency_impl.py:57-105, 249-300, 588-625, 750-786; the harness there is
`common_utils.torch_script`: script -> save -> load).  CPU part: compilation, round trip, keys, shapes / strides on the meta device, the
host-side constant builders bit for bit.  GPU part (-m gpu): scripted == eager with ``torch.equal``, gradients through a scripted
call.
"""
import io

import pytest
import torch

import audio_amd  # noqa: F401
import audio_amd.functional as F
import audio_amd.transforms as T


def torch_script(obj):
    """script -> save -> load (what the reference's `common_utils.torch_script` does)."""
    scripted = torch.jit.script(obj)
    buf = io.BytesIO()
    torch.jit.save(scripted, buf)
    buf.seek(0)
    return torch.jit.load(buf)


def _modules():
    return {
        "Spectrogram": lambda: T.Spectrogram(),
        "Spectrogram_complex": lambda: T.Spectrogram(power=None),
        "Spectrogram_frame_length": lambda: T.Spectrogram(normalized="frame_length"),
        "Spectrogram_window": lambda: T.Spectrogram(normalized=True, power=1.0),
        "InverseSpectrogram": lambda: T.InverseSpectrogram(n_fft=400, hop_length=100),
        "GriffinLim": lambda: T.GriffinLim(length=1000, rand_init=False),
        "AmplitudeToDB": lambda: T.AmplitudeToDB(),
        "AmplitudeToDB_top_db": lambda: T.AmplitudeToDB("magnitude", top_db=60.0),
        "MelScale": lambda: T.MelScale(n_stft=201),
        "MelSpectrogram": lambda: T.MelSpectrogram(),
        "MelSpectrogram_headline": lambda: T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80),
        "MFCC": lambda: T.MFCC(),
        "MFCC_headline": lambda: T.MFCC(16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)),
        "MFCC_log_mels": lambda: T.MFCC(log_mels=True),
        "Resample": lambda: T.Resample(16000, 8000),
        "Resample_kaiser": lambda: T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                                              rolloff=0.9475937167399596, beta=14.769656459379492),
        "TimeStretch": lambda: T.TimeStretch(n_freq=201, hop_length=100, fixed_rate=1.3),
        "PitchShift": lambda: T.PitchShift(sample_rate=8000, n_steps=-3),
        "FFTConvolve_full": lambda: T.FFTConvolve("full"),
        "FFTConvolve_valid": lambda: T.FFTConvolve("valid"),
        "FFTConvolve_same": lambda: T.FFTConvolve("same"),
        "Speed": lambda: T.Speed(1000, 0.9),
        "SpeedPerturbation": lambda: T.SpeedPerturbation(1000, [0.9]),
    }


FUNCTIONS = ["spectrogram", "inverse_spectrogram", "phase_vocoder", "griffinlim", "pitch_shift", "speed", "mel_scale",
             "amplitude_to_DB", "resample", "lfilter", "biquad", "filtfilt", "fftconvolve", "lowpass_biquad", "highpass_biquad",
             "allpass_biquad", "bandpass_biquad", "bandreject_biquad", "equalizer_biquad", "band_biquad", "treble_biquad",
             "bass_biquad", "deemph_biquad", "riaa_biquad", "biquad_cascade", "melscale_fbanks", "create_dct"]


# --------------------------------------------------------------------------- CPU: compile, round trip, keys, shapes


@pytest.mark.parametrize("name", sorted(_modules()))
def test_module_scripts_and_round_trips_with_its_state_dict_keys(name):
    eager = _modules()[name]()
    loaded = torch_script(eager)
    assert sorted(loaded.state_dict()) == sorted(eager.state_dict())
    for k, v in eager.state_dict().items():
        assert torch.equal(loaded.state_dict()[k], v), k
    # the scripted program is ONE audio_amd operator per module (plus, for the thin callers, torch glue)
    graph = str(loaded.inlined_graph)
    assert "audio_amd::" in graph, graph


@pytest.mark.parametrize("name", FUNCTIONS)
def test_function_scripts(name):
    torch_script(getattr(F, name))


def test_scripted_modules_give_the_reference_shapes_and_strides_on_the_meta_device():
    x = torch.empty(3, 2, 16000, device="meta")

    def run(name, *inputs):
        return torch_script(_modules()[name]().to("meta"))(*inputs)

    s = run("Spectrogram", x)
    assert s.shape == (3, 2, 201, 81) and s.stride()[-2:] == (1, 201)
    assert run("Spectrogram_complex", x).dtype == torch.complex64
    m = run("MelSpectrogram_headline", x)
    assert m.shape == (3, 2, 80, 101) and m.stride()[-2:] == (1, 80)
    k = run("MFCC_headline", x)
    assert k.shape == (3, 2, 40, 101) and k.stride()[-2:] == (1, 40)
    assert run("Resample", x).shape == (3, 2, 8000)
    assert run("MelScale", s).shape == (3, 2, 128, 81)
    assert run("AmplitudeToDB", s).shape == s.shape
    c = torch.empty(2, 201, 50, dtype=torch.complex64, device="meta")
    assert run("TimeStretch", c).shape == (2, 201, 39)
    assert run("InverseSpectrogram", c).shape == (2, 4900)
    assert run("FFTConvolve_full", torch.empty(2, 3, 32, device="meta"), torch.empty(2, 3, 55, device="meta")).shape == (2, 3, 86)
    y, n = run("Speed", torch.empty(3, 2, 200, device="meta"), None)
    assert y.shape == (3, 2, 223) and n is None


def test_scripted_constant_builders_equal_eager_bit_for_bit():
    """reference: functional/torchscript_consistency_impl.py:131-141, 168-175 (melscale_fbanks, create_dct) -- host-side, so the
    whole comparison runs without a device."""
    fb = torch_script(F.melscale_fbanks)
    for args in [(100, 0.0, 4000.0, 20, 8000, "slaney", "htk"), (201, 0.0, 8000.0, 80, 16000, None, "htk"),
                 (257, 40.0, 7600.0, 64, 16000, "slaney", "slaney")]:
        assert torch.equal(fb(*args), F.melscale_fbanks(*args))
    dct = torch_script(F.create_dct)
    for args in [(40, 128, "ortho"), (13, 80, None)]:
        assert torch.equal(dct(*args), F.create_dct(*args))


def test_a_cpu_tensor_fails_loudly_in_a_scripted_module_too():
    scripted = torch_script(T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80))
    with pytest.raises((NotImplementedError, RuntimeError), match="CPU"):
        scripted(torch.zeros(2, 4000))


def test_an_invalid_normalized_string_raises_in_the_scripted_program():
    f = torch_script(F.spectrogram)
    with pytest.raises(Exception, match="Invalid normalized parameter"):
        f(torch.zeros(1, 1000, device="meta"), 0, torch.ones(400, device="meta"), 400, 200, 400, 2.0, "bogus", True, "reflect", True,
          None)


# --------------------------------------------------------------------------- GPU: scripted == eager, bit for bit


def _noise(*shape, seed=0, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).clamp_(-1, 1)


def _same(a, b):
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b)
        for u, v in zip(a, b):
            _same(u, v)
        return
    if a is None or b is None:
        assert a is None and b is None
        return
    assert a.dtype == b.dtype and a.shape == b.shape and a.stride() == b.stride(), (a.dtype, b.dtype, a.shape, b.shape)
    assert torch.equal(a, b), float((a - b).abs().max())


def _module_inputs(name, dev):
    rand = lambda *s, seed=0: torch.rand(*s, generator=torch.Generator().manual_seed(seed)).to(dev)  # noqa: E731
    if name.startswith("Spectrogram") or name in ("MelSpectrogram", "MFCC", "MFCC_log_mels"):
        return (rand(1, 1000),)
    if name in ("MelSpectrogram_headline", "MFCC_headline"):
        return (_noise(8, 160000, seed=3).to(dev),)
    if name == "InverseSpectrogram":
        w = _noise(1, 8000, seed=1).to(dev)
        return (F.spectrogram(w, 0, torch.hann_window(400, device=dev), 400, 100, 400, None, False),)
    if name == "GriffinLim":
        return (rand(1, 201, 6),)
    if name.startswith("AmplitudeToDB"):
        return (rand(6, 201),)
    if name == "MelScale":
        return (rand(1, 201, 6),)
    if name == "Resample":
        return (_noise(1, 16000, seed=2).to(dev),)
    if name == "Resample_kaiser":
        return (_noise(4, 2, 44100, seed=2).to(dev),)
    if name == "TimeStretch":
        g = torch.Generator().manual_seed(5)
        return (torch.view_as_complex(torch.randn(2, 201, 50, 2, generator=g)).to(dev),)
    if name == "PitchShift":
        return (_noise(2, 8000, seed=4).to(dev),)
    if name.startswith("FFTConvolve"):
        return (rand(2, 3, 2, 32, seed=1), rand(2, 3, 2, 55, seed=2))
    if name in ("Speed", "SpeedPerturbation"):
        return (rand(3, 2, 200), torch.randint(1, 200, (3, 2), generator=torch.Generator().manual_seed(0)).float().to(dev))
    raise KeyError(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(_modules()))
def test_scripted_module_equals_eager_bit_for_bit(name):
    dev = torch.device("cuda:0")
    eager = _modules()[name]().to(dev)
    scripted = torch_script(eager)
    inputs = _module_inputs(name, dev)
    with torch.no_grad():
        want = eager(*inputs)
        got = scripted(*inputs)
        again = scripted(*inputs)
    _same(got, want)
    _same(again, want)
    assert sorted(scripted.state_dict()) == sorted(eager.state_dict())


@pytest.mark.gpu
def test_scripted_speed_without_lengths():
    dev = torch.device("cuda:0")
    for mod in (T.Speed(1000, 0.9), T.SpeedPerturbation(1000, [0.9])):
        eager = mod.to(dev)
        x = torch.rand(3, 2, 200, device=dev)
        _same(torch_script(eager)(x, None), eager(x, None))


@pytest.mark.gpu
def test_scripted_mfcc_shares_the_auto_decision_of_the_module_it_was_scripted_from():
    """`MFCC.fused = "auto"` decides its arithmetic once per module; the scripted copy carries the module's handle, so whichever of the
    two runs first decides for both (a zero-padded batch here: the decision is "two-kernel")."""
    dev = torch.device("cuda:0")
    x = _noise(16, 48000, seed=7).to(dev)
    x[:, 12000:] = 0.0                                    # padded clips: the cut-off reaches most tiles
    eager = T.MFCC(16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
    scripted = torch_script(eager)
    with torch.no_grad():
        got = scripted(x)                                 # the scripted copy decides ...
        assert eager.fused_report()["decided"] == "two-kernel"
        want = eager(x)                                   # ... and the eager module follows
    _same(got, want)
    for fused in (True, False):
        eager.fused = fused
        scripted = torch_script(eager)
        with torch.no_grad():
            _same(scripted(x), eager(x))


def _function_cases(dev):
    wn = _noise(2, 16000, seed=11).to(dev)
    win = torch.hann_window(400, device=dev)
    cases = []
    for normalize in (True, False, "window", "frame_length"):
        cases.append(("spectrogram", (wn, 0, win, 400, 200, 400, None, normalize, True, "reflect", True, True)))
    cases.append(("spectrogram", (wn, 0, win, 400, 160, 400, 2.0, False, True, "reflect", True, None)))
    spec = F.spectrogram(_noise(1, 400, seed=12).to(dev), 0, win, 400, 200, 400, None, False)
    for normalize in (True, False, "window", "frame_length"):
        cases.append(("inverse_spectrogram", (spec, 400, 0, win, 400, 200, 400, normalize, True, "reflect", True)))
    g = torch.Generator().manual_seed(13)
    cases.append(("griffinlim", (torch.rand(1, 201, 6, generator=g).to(dev), win, 400, 200, 400, 2.0, 32, 0.99, 1000, False)))
    cases.append(("amplitude_to_DB", (torch.rand(2, 2, 10, generator=g).to(dev), 20.0, 1e-10, 0.0, 80.0)))
    cases.append(("amplitude_to_DB", (torch.rand(2, 2, 10, generator=g).to(dev), 10.0, 1e-10, 0.0, None)))
    cases.append(("mel_scale", (torch.rand(2, 201, 30, generator=g).to(dev), F.melscale_fbanks(201, 0.0, 8000.0, 80, 16000).to(dev))))
    b = torch.tensor([0.00299893, -0.0051152, 0.00841964, -0.00747802, 0.00841964, -0.0051152, 0.00299893], device=dev)
    a = torch.tensor([1.0, -4.8155751, 10.2217618, -12.14481273, 8.49018171, -3.3066882, 0.56088705], device=dev)
    cases.append(("lfilter", (wn, a, b, True, True)))
    cases.append(("filtfilt", (wn[:, :8000], torch.rand(4, generator=g).to(dev), torch.rand(4, generator=g).to(dev), True)))
    cases.append(("biquad", (wn, 0.4, 0.2, 0.9, 1.0, 0.2, 0.6)))
    w44 = _noise(2, 44100, seed=14).to(dev)
    cases += [("lowpass_biquad", (w44, 44100, 1000.0, 0.707)), ("highpass_biquad", (w44, 44100, 2000.0, 0.707)),
              ("allpass_biquad", (w44, 44100, 1000.0, 0.707)), ("bandpass_biquad", (w44, 44100, 1000.0, 0.707, True)),
              ("bandpass_biquad", (w44, 44100, 1000.0, 0.707, False)), ("bandreject_biquad", (w44, 44100, 1000.0, 0.707)),
              ("band_biquad", (w44, 44100, 1000.0, 0.707, True)), ("band_biquad", (w44, 44100, 1000.0, 0.707, False)),
              ("treble_biquad", (w44, 44100, 40.0, 3000.0, 0.707)), ("bass_biquad", (w44, 44100, 40.0, 100.0, 0.707)),
              ("deemph_biquad", (w44, 44100)), ("riaa_biquad", (w44, 44100)),
              ("equalizer_biquad", (w44, 44100, 300.0, 13.0, 0.707))]
    cases.append(("resample", (wn, 16000, 8000, 6, 0.99, "sinc_interp_hann", None)))
    for beta in (None, 6.0):
        cases.append(("resample", (wn, 16000, 8000, 6, 0.99, "sinc_interp_kaiser", beta)))
    pv = torch.view_as_complex(torch.randn(2, 1025, 400, 2, generator=g)).to(dev)
    cases.append(("phase_vocoder", (pv, 0.5, torch.linspace(0, 3.14 * 256, 1025, device=dev)[..., None])))
    x, y = torch.rand(2, 3, 2, 32, generator=g).to(dev), torch.rand(2, 3, 2, 55, generator=g).to(dev)
    for mode in ("full", "valid", "same"):
        cases.append(("fftconvolve", (x, y, mode)))
    wv = torch.rand(3, 2, 200, generator=g).to(dev)
    cases.append(("speed", (wv, 1000, 1.1, torch.randint(1, 200, (3, 2), generator=g).float().to(dev))))
    cases.append(("speed", (wv, 1000, 1.1, None)))
    cases.append(("pitch_shift", (_noise(2, 8000, seed=15).to(dev), 8000, 4, 12, 512, None, None, None)))
    cases.append(("biquad_cascade", (wn, torch.tensor([[1.0, -1.2, 0.5], [1.0, -0.3, 0.2]], device=dev),
                                     torch.tensor([[0.3, 0.2, 0.1], [0.5, 0.1, 0.0]], device=dev), True)))
    return cases


@pytest.mark.gpu
def test_scripted_functions_equal_eager_bit_for_bit():
    dev = torch.device("cuda:0")
    scripted = {}
    for name, args in _function_cases(dev):
        fn = getattr(F, name)
        if name not in scripted:
            scripted[name] = torch_script(fn)
        with torch.no_grad():
            torch.random.manual_seed(40)
            want = fn(*args)
            torch.random.manual_seed(40)
            got = scripted[name](*args)
        _same(got, want)


@pytest.mark.gpu
def test_a_scripted_closure_over_the_functions_equals_eager():
    """The reference scripts small closures around its functions (functional/torchscript_consistency_impl.py:294-305)."""
    dev = torch.device("cuda:0")

    def chain(tensor: torch.Tensor) -> torch.Tensor:
        y = F.lowpass_biquad(tensor, 44100, 1000.0)
        y = F.resample(y, 44100, 16000, resampling_method="sinc_interp_kaiser")
        return F.amplitude_to_DB(F.spectrogram(y, 0, torch.hann_window(400, device=tensor.device), 400, 160, 400, 2.0, False),
                                 10.0, 1e-10, 0.0, 80.0)

    x = _noise(2, 44100, seed=21).to(dev)
    with torch.no_grad():
        _same(torch_script(chain)(x), chain(x))


@pytest.mark.gpu
def test_gradients_flow_through_scripted_calls_as_through_eager_ones():
    """The reference's scripted modules are aten compositions and therefore differentiable; the operators behind the scripted fronts
    run the eager implementation at the autograd level when an argument asks for a gradient (audio_amd/_ops.py `_register`)."""
    dev = torch.device("cuda:0")
    cases = [
        (T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80), (_noise(2, 8000, seed=31),)),
        (T.Spectrogram(n_fft=400, hop_length=160), (_noise(2, 8000, seed=32),)),
        (T.MFCC(16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)), (_noise(2, 8000, seed=33),)),
        (T.Resample(16000, 8000), (_noise(2, 8000, seed=34),)),
        (T.FFTConvolve("full"), (_noise(2, 500, seed=35), _noise(2, 64, seed=36))),
    ]
    for mod, inputs in cases:
        eager = mod.to(dev)
        scripted = torch_script(eager)
        leaves_e = [t.to(dev).requires_grad_(True) for t in inputs]
        leaves_s = [t.to(dev).requires_grad_(True) for t in inputs]
        ye, ys = eager(*leaves_e), scripted(*leaves_s)
        assert ys.requires_grad
        _same(ys.detach(), ye.detach())
        w = torch.randn(ye.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        (ye * w).sum().backward()
        (ys * w).sum().backward()
        for a, b in zip(leaves_s, leaves_e):
            assert a.grad is not None and torch.equal(a.grad, b.grad)
    # a scripted function: lfilter into its coefficients
    f = torch_script(F.lfilter)
    x = _noise(2, 4000, seed=37).to(dev)
    a0 = torch.tensor([1.0, -0.4, 0.1], device=dev)
    b0 = torch.tensor([0.3, 0.2, 0.1], device=dev)
    grads = []
    for fn in (F.lfilter, f):
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        fn(x, a, b, True, True).pow(2).sum().backward()
        grads.append((a.grad, b.grad))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])

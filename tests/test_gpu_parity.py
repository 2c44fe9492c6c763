"""Parity tests proper: the HIP path (through the C ABI) on a real MI355X against
 (a) the outputs of the reference implementation itself (committed fixtures), and
 (b) the float64 CPU oracle on the same inputs.
Tolerances: integer work (shapes, strides, frame indexing) exact; floating point
<= 1e-4 peak-relative (north-star), tighter where the reference's own tests are tighter."""
import math
import os

import warnings
import numpy as np
import pytest
import torch

from conftest import (check_click_and_quiet_tone, click_and_quiet_tone, floor_rel_err, peak_rel_err, ref_runs,
                      windowed_rel_err)
import oracle_dispatch as OD
import product_dispatch as PD

pytestmark = pytest.mark.gpu

TOL = {
    "Spectrogram": 1e-4, "MelSpectrogram": 1e-4, "MFCC": 1e-4, "AmplitudeToDB": 1e-4, "MelScale": 1e-4,
    "F.resample": 1e-4, "T.Resample": 1e-4, "lfilter": 1e-4, "biquad": 1e-4, "filtfilt": 1e-4,
    "lowpass_cascade": 1e-4, "fftconvolve": 1e-4, "T.FFTConvolve": 1e-4,
    "lowpass_biquad": 1e-4, "highpass_biquad": 1e-4, "allpass_biquad": 1e-4, "bandpass_biquad": 1e-4,
    "bandreject_biquad": 1e-4, "equalizer_biquad": 1e-4, "bass_biquad": 1e-4, "treble_biquad": 1e-4,
    "band_biquad": 1e-4,
}
CASES = [c for c in ref_runs().cases if c["op"] in TOL]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X (run through gpurun)"
    from audio_amd import _lib
    _lib.lib()


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['id']}-{c['op']}-{c.get('tag')}")
def test_against_reference_and_oracle(case):
    rr = ref_runs()
    inputs = rr.inputs(case)
    got_t = PD.run(case, inputs)
    assert got_t is not None
    exp = rr.output(case)
    assert tuple(got_t.shape) == tuple(exp.shape)          # integer work: exact
    got = got_t.cpu().numpy()
    tol = TOL[case["op"]]
    if case.get("tag") in ("order4", "order8"):
        tol = 5e-4    # reference's own fp32 recursion drifts at high order (see test_oracle_golden)
    e_ref = peak_rel_err(got, exp)
    assert e_ref <= tol, f"vs reference: {e_ref}"
    orc = OD.evaluate(case, inputs)
    if orc is not None:
        e_orc = peak_rel_err(got, orc)
        assert e_orc <= tol, f"vs oracle: {e_orc}"
        # SURVEY A9 asks for BOTH metrics: element-wise error relative to max(|ref|, floor x peak).  The float64
        # oracle is the yardstick; the floor is where fp32 rounding of the LARGEST terms of a sum sits (a bin 1e-5 below
        # the peak of its own frame carries ~1e-7 x peak of FFT rounding noise whatever computes it -- the reference's
        # fp32 CPU path shows the same), so magnitudes use 1e-4 at a 1e-3 floor; dB / cepstral features are compared
        # on an absolute scale already by the peak metric.
        if case["op"] in ("Spectrogram", "MelSpectrogram", "MelScale", "F.resample", "T.Resample", "fftconvolve",
                          "T.FFTConvolve") and not np.iscomplexobj(orc):
            e_floor = floor_rel_err(got, orc, floor=1e-3)
            e_floor_ref = floor_rel_err(exp, orc, floor=1e-3)
            assert e_floor <= max(1e-4, 2.0 * e_floor_ref), f"element-wise vs oracle: {e_floor} (reference itself: {e_floor_ref})"
            # ... and at a 1e-6 floor (at 1e-3 the bins 60 dB under the peak were judged on the absolute
            # scale).  Down there the error of ANY fp32 evaluation is the rounding noise of the frame's largest terms, so the
            # yardstick is the reference's own fp32 CPU result against the same float64 oracle: never more than 3 x its error.
            # (magnitudes element by element; WAVEFORMS -- resample, fftconvolve -- against the local amplitude of every 32-sample
            # passage instead: at a zero crossing |b| is arbitrarily small while the rounding noise is not, conftest.windowed_rel_err)
            metric = floor_rel_err if case["op"] in ("Spectrogram", "MelSpectrogram", "MelScale") else windowed_rel_err
            e6 = metric(got, orc, floor=1e-6)
            e6_ref = metric(exp, orc, floor=1e-6)
            assert e6 <= max(1e-4, 3.0 * e6_ref), f"element-wise vs oracle at the 1e-6 floor: {e6} (reference itself: {e6_ref})"


def test_output_strides_match_reference():
    """The reference returns (..., F, T) views of frame-major memory."""
    import audio_amd.transforms as T
    x = torch.randn(4, 16000, device="cuda")
    s = T.Spectrogram(n_fft=400, hop_length=160).cuda()(x)
    assert s.shape == (4, 201, 101) and s.stride() == (20301, 1, 201)
    m = T.MelSpectrogram(n_fft=400, hop_length=160, n_mels=80).cuda()(x)
    assert m.shape == (4, 80, 101) and m.stride() == (8080, 1, 80)
    c = T.MFCC(n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()(x)
    assert c.shape == (4, 40, 101) and c.stride() == (4040, 1, 40)
    z = T.Spectrogram(n_fft=400, hop_length=160, power=None).cuda()(x)
    assert z.dtype == torch.complex64 and z.shape == (4, 201, 101)


def test_librosa_goldens_on_gpu(librosa_goldens):
    """The reference's own librosa golden vectors, through the HIP path (fp32 vs f64 goldens)."""
    import audio_amd.transforms as T
    xw = torch.from_numpy(librosa_goldens["whitenoise_16k"]).cuda()
    xs = torch.from_numpy(librosa_goldens["sinusoid_16k"]).cuda()
    for i, (n_fft, hop, power) in enumerate([(400, 200, 2.0), (600, 100, 2.0), (400, 200, 3.0), (200, 50, 2.0)]):
        got = T.Spectrogram(n_fft=n_fft, hop_length=hop, power=power).cuda()(xw)[0].cpu().numpy()
        assert peak_rel_err(got, librosa_goldens[f"spectrogram_{i}"]) <= 1e-4
    sizes = [(400, 200, 64), (600, 100, 128), (200, 50, 32)]
    for i in range(12):
        n_fft, hop, n_mels = sizes[i // 4]
        norm = [None, "slaney"][(i // 2) % 2]
        scale = ["htk", "slaney"][i % 2]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t = T.MelSpectrogram(sample_rate=16000, n_fft=n_fft, hop_length=hop, n_mels=n_mels, norm=norm,
                                 mel_scale=scale).cuda()
        got = t(xs)[0].cpu().numpy()
        assert peak_rel_err(got, librosa_goldens[f"melspectrogram_{i:02d}"]) <= 1e-4
    for i, (n_fft, hop, n_mels, n_mfcc) in enumerate([(400, 200, 64, 40), (600, 100, 128, 20), (200, 50, 32, 25)]):
        t = T.MFCC(sample_rate=16000, n_mfcc=n_mfcc, norm="ortho",
                   melkwargs={"hop_length": hop, "n_fft": n_fft, "n_mels": n_mels}).cuda()
        got = t(xw)[0].cpu().numpy()
        # the reference's own tolerance for this comparison (librosa_compatibility_test_impl.py:114-134)
        np.testing.assert_allclose(got, librosa_goldens[f"mfcc_{i}"], atol=5e-4, rtol=1e-5)


def test_sox_golden_biquad_on_gpu(sox_goldens):
    import audio_amd.functional as F
    x = torch.from_numpy(sox_goldens["noise_8k"]).cuda()
    got = F.lfilter(x, torch.tensor([0.7, 0.2, 0.6]).cuda(), torch.tensor([0.4, 0.2, 0.9]).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, sox_goldens["perf_biquad_filtering"], atol=1e-4, rtol=1e-5)
    for name, fn, kw, atol in [
        ("lowpass", F.lowpass_biquad, dict(sample_rate=8000, cutoff_freq=3000), 1.5e-4),
        ("highpass", F.highpass_biquad, dict(sample_rate=8000, cutoff_freq=2000), 1.5e-4),
        ("allpass", F.allpass_biquad, dict(sample_rate=8000, central_freq=1000, Q=0.707), 1e-4),
        ("bandreject", F.bandreject_biquad, dict(sample_rate=8000, central_freq=1000, Q=0.707), 1e-4),
        ("equalizer", F.equalizer_biquad, dict(sample_rate=8000, center_freq=300, gain=1, Q=0.707), 1e-4),
    ]:
        got = fn(x, **kw).cpu().numpy()
        np.testing.assert_allclose(got, sox_goldens[name], atol=atol, rtol=1e-5)


def test_table_driven_biquad_designers_against_reference_runs():
    """deemph_biquad / riaa_biquad on the device vs the reference's CPU output on the same noise (fixture from
    tests/golden/make_biquad_extra_golden.py), the tolerance of the other designer fixtures."""
    import os
    import audio_amd.functional as F
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "biquad_extra_goldens.npz"))
    x = torch.from_numpy(gold["noise"]).cuda()
    for name, fn, rates in (("deemph", F.deemph_biquad, (44100, 48000)), ("riaa", F.riaa_biquad, (44100, 48000, 88200, 96000))):
        for sr in rates:
            np.testing.assert_allclose(fn(x, sr).cpu().numpy(), gold[f"{name}_{sr}"], atol=1e-4, rtol=1e-5)


def test_mel400_fast_path_equals_generic(monkeypatch):
    """Headline register-FFT kernel vs the generic LDS Stockham kernel on ragged / edge inputs."""
    import os
    import audio_amd.transforms as T
    t = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    for L in (401, 560, 961, 1600, 16000, 16001, 48017):
        x = torch.randn(3, L, device="cuda").clamp_(-1, 1)
        fast = t(x)
        gen = _force_generic(lambda: t(x))
        assert fast.shape == gen.shape
        e = (fast - gen).abs().max() / gen.abs().max()
        assert float(e) <= 2e-6, (L, float(e))


def test_headline_shape_properties():
    """BASELINE config 2 (256 x 10 s): size-independent checks at full size --
    batch consistency (item-wise == batched, batch_consistency_test.py:102-107),
    linearity of the power spectrum under scaling, and Parseval on the |X|^2 frames."""
    import audio_amd.transforms as T
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = (0.5 * torch.randn(256, 160000, device="cuda", generator=g)).clamp_(-1, 1)
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    y = mel(x)
    assert y.shape == (256, 80, 1001) and y.stride() == (80080, 1, 80)
    assert torch.isfinite(y).all()
    for b in (0, 17, 255):
        yb = mel(x[b:b + 1])
        assert torch.equal(yb[0], y[b])                      # bit-identical item-wise
    y2 = mel(2.0 * x[:8])
    assert float(((y2 - 4.0 * y[:8]).abs().max() / y2.abs().max())) < 1e-6
    # Parseval per frame: sum_k' |X_k|^2 (two-sided) = N * sum_n (w x)^2 -- use a Spectrogram
    spec = T.Spectrogram(n_fft=400, hop_length=160).cuda()(x[:4])          # (4, 201, 1001)
    two_sided = spec[:, 0] + spec[:, 200] + 2 * spec[:, 1:200].sum(1)
    w = torch.hann_window(400, device="cuda")
    xp = torch.nn.functional.pad(x[:4, None], (200, 200), mode="reflect")[:, 0]
    fr = xp.unfold(-1, 400, 160) * w
    assert float(((two_sided - 400 * (fr ** 2).sum(-1)).abs().max() / two_sided.abs().max())) < 1e-5


def test_empty_and_error_behaviour():
    import audio_amd.functional as F
    import audio_amd.transforms as T
    with pytest.raises(RuntimeError):
        T.Spectrogram(n_fft=400)(torch.randn(2, 100, device="cuda"))      # reflect pad >= length
    with pytest.raises(ValueError, match="same dimension"):
        F.fftconvolve(torch.randn(2, 5, device="cuda"), torch.randn(5, device="cuda"))
    with pytest.raises(ValueError, match="Unrecognized mode"):
        F.fftconvolve(torch.randn(5, device="cuda"), torch.randn(5, device="cuda"), "bogus")
    with pytest.raises(ValueError, match="same size"):
        F.lfilter(torch.randn(5, device="cuda"), torch.ones(3).cuda(), torch.ones(2).cuda())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        T.MelSpectrogram()(torch.randn(1, 16000))
    out = T.MelSpectrogram(n_fft=400, hop_length=160, n_mels=80).cuda()(torch.zeros(0, 16000, device="cuda"))
    assert out.shape == (0, 80, 101)
    y = F.resample(torch.randn(2, 1000, device="cuda"), 16000, 16000)
    assert y.shape == (2, 1000)
    assert F.resample(torch.randn(2, 1001, device="cuda"), 16000, 8000).shape == (2, 501)


def _force_generic(fn):
    from audio_amd import _lib
    with _lib.kernel_policy(_lib.POLICY_FORCE_GENERIC):
        return fn()


@pytest.mark.parametrize("power", [2.0, 1.0, 0.5])
def test_spec400_fast_path_equals_generic_and_oracle(power):
    """Spectrogram epilogue of the radix-20x20 kernel vs the generic Stockham kernel and the oracle."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    t = T.Spectrogram(n_fft=400, hop_length=160, power=power).cuda()
    g = torch.Generator().manual_seed(17)
    # powers < 1 amplify the rounding of near-zero bins without bound (d|X|^p ~ |X|^(p-1) d|X|), in the
    # reference too: compare in the power-spectrum domain, where the 1e-4 bar of the north star is defined
    back = 2.0 / power
    for L in (401, 560, 961, 1283, 1600, 16000, 16001, 48017):
        x = torch.randn(3, L, generator=g).clamp_(-1, 1).cuda()
        fast = t(x)
        gen = _force_generic(lambda: t(x))
        assert fast.shape == gen.shape and fast.stride() == gen.stride()
        e = (fast.pow(back) - gen.pow(back)).abs().max() / gen.pow(back).abs().max()
        assert float(e) <= 3e-6, (L, float(e))
        if L <= 16001:
            exp = O.spectrogram(x.cpu().numpy().astype(np.float64), 0, O.hann_window(400), 400, 160, 400, power, False)
            assert peak_rel_err(fast.double().pow(back).cpu().numpy(), exp ** back) <= 1e-4, L


@pytest.mark.parametrize("shape", [(6, 4000), (3, 2, 4000), (2, 2, 2, 2400), (4000,)])
def test_mfcc_fused_db_path_equals_generic_and_oracle(shape):
    """MFCC = mel kernel with fused dB + per-group max, then the matrix-core DCT: compared with
    the unfused generic kernels and the float64 oracle, with silence so that top_db clamps."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(7)
    x = (0.5 * torch.randn(*shape, generator=g)).clamp_(-1, 1)
    x[..., 1000:1700] = 0.0                          # digital silence -> -100 dB -> clamped
    if x.dim() > 1:
        x[0] *= 1e-3
    m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()
    fast = m(x.cuda())
    gen = _force_generic(lambda: m(x.cuda()))
    assert fast.shape == gen.shape and fast.stride() == gen.stride()
    assert float((fast - gen).abs().max() / gen.abs().max()) <= 2e-5
    fb = O.melscale_fbanks(201, 0.0, 8000.0, 80, 16000)
    exp = O.mfcc(x.numpy().astype(np.float64), O.hann_window(400), fb, O.create_dct(40, 80, "ortho"), 400, 160)
    assert peak_rel_err(fast.cpu().numpy(), exp) <= 1e-4


def test_mfcc_headline_shape_properties():
    """BASELINE config 4 (512 x 10 s) at full size: per-item cut-offs ((B,1,L) input) make the
    batched result equal the item-wise one; the (B,L) input shares ONE cut-off."""
    import audio_amd.transforms as T
    g = torch.Generator(device="cuda").manual_seed(99)
    x = (0.5 * torch.randn(512, 160000, device="cuda", generator=g)).clamp_(-1, 1)
    x[5] *= 1e-4
    m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()
    y3 = m(x[:, None, :])
    assert y3.shape == (512, 1, 40, 1001) and torch.isfinite(y3).all()
    for b in (0, 5, 511):
        yb = m(x[b:b + 1, None, :])
        assert float((yb[0] - y3[b]).abs().max()) <= 1e-4 * float(y3[b].abs().max())
    y2 = m(x)
    assert y2.shape == (512, 40, 1001) and y2.stride() == (40040, 1, 40)
    # one global cut-off: the quiet item is clamped in the 2-D call but not in the 3-D call
    assert float((y2[5] - y3[5, 0]).abs().max()) > 1.0
    assert float((y2[0] - y3[0, 0]).abs().max()) <= 1e-4 * float(y2[0].abs().max())


@pytest.mark.parametrize("rates", [(44100, 16000, "kaiser_best"), (44100, 16000, "default"), (48000, 44100, "kaiser_best"),
                                   (16000, 44100, "default"), (8000, 16000, "default"), (48000, 16000, "kaiser_best")])
def test_resample_matrix_core_path_equals_scalar_and_oracle(rates):
    """Banded MFMA resampler vs the scalar polyphase kernel and the float64 oracle, ragged lengths."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    o, n, kind = rates
    kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596,
              beta=14.769656459379492) if kind == "kaiser_best" else {}
    r = T.Resample(o, n, **kw).cuda()
    g = torch.Generator().manual_seed(11)
    for shape in [(3, 20011), (2, 2, 7001), (1, 501)]:
        x = (0.5 * torch.randn(*shape, generator=g)).clamp_(-1, 1)
        fast = r(x.cuda())
        gen = _force_generic(lambda: r(x.cuda()))
        assert fast.shape == gen.shape
        assert float((fast - gen).abs().max()) <= 2e-6 * max(float(gen.abs().max()), 1e-3)
        exp = O.resample(x.numpy().astype(np.float64), o, n, **kw)
        assert fast.shape == exp.shape
        assert peak_rel_err(fast.cpu().numpy(), exp) <= 1e-5


def test_resample_headline_shape_properties():
    """BASELINE config 3 shape at 1/64 of the batch (16 x stereo x 30 s, 44.1k -> 16k kaiser_best):
    output length, batch consistency, linearity and DC gain 1 (each phase of the table sums to 1)."""
    import audio_amd.transforms as T
    r = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                   rolloff=0.9475937167399596, beta=14.769656459379492).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (0.5 * torch.randn(16, 2, 1323000, device="cuda", generator=g)).clamp_(-1, 1)
    y = r(x)
    assert y.shape == (16, 2, 480000) and torch.isfinite(y).all()
    assert torch.equal(r(x[3:4])[0], y[3])
    y2 = r(0.5 * x[:2])
    assert float((y2 - 0.5 * y[:2]).abs().max()) <= 1e-6
    dc = r(torch.ones(1, 100000, device="cuda"))
    assert float((dc[0, 200:-200] - 1.0).abs().max()) <= 1e-5


def test_registered_ops_equal_modules():
    """torch.ops.audio_amd.* (the dispatcher-registered surface) == the drop-in modules."""
    import audio_amd.functional as F
    import audio_amd.transforms as T
    x = torch.randn(2, 2, 8000, device="cuda").clamp_(-1, 1)
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    got = torch.ops.audio_amd.mel_spectrogram(x, mel.spectrogram.window, mel.mel_scale.fb, 0, 400, 160, 400, 2.0, 0,
                                              True, "reflect")
    assert torch.equal(got, mel(x)) and got.stride() == mel(x).stride()
    mf = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()
    mf.fused = False        # the stateless op is the exact two-kernel path; the module's default ("auto") may take the one-kernel one
    got = torch.ops.audio_amd.mfcc(x, mel.spectrogram.window, mel.mel_scale.fb, mf.dct_mat, 0, 400, 160, 400, 2.0, 0,
                                   True, "reflect", False, 80.0)
    assert torch.equal(got, mf(x))
    a = torch.tensor([1.0, -1.2, 0.5], device="cuda")
    b = torch.tensor([0.1, 0.2, 0.1], device="cuda")
    assert torch.equal(torch.ops.audio_amd.lfilter(x, a, b, True, True), F.lfilter(x, a, b))
    y = torch.randn(1, 1, 300, device="cuda")
    assert torch.equal(torch.ops.audio_amd.fftconvolve(x, y, "same"), F.fftconvolve(x, y, "same"))
    # the widened surface: inverse STFT, phase vocoder, Griffin-Lim, RNN-T features (float and int16 PCM)
    from audio_amd.pipelines import GAIN, RNNTFeatureExtractor
    sp = T.Spectrogram(n_fft=400, hop_length=160, power=None).cuda()
    X = sp(x)
    inv = T.InverseSpectrogram(n_fft=400, hop_length=160).cuda()
    got = torch.ops.audio_amd.inverse_spectrogram(X, 8000, inv.window, 0, 400, 160, 400, 0, True, "reflect", True)
    assert peak_rel_err(got.cpu().numpy(), inv(X, 8000).cpu().numpy()) <= 1e-6      # atomics: order-dependent rounding
    ts = T.TimeStretch(hop_length=160, n_freq=201, fixed_rate=1.2).cuda()
    assert torch.equal(torch.ops.audio_amd.phase_vocoder(X, 1.2, ts.phase_advance), ts(X))
    gl = T.GriffinLim(n_fft=400, hop_length=160, n_iter=3, rand_init=False, length=8000).cuda()
    got = torch.ops.audio_amd.griffinlim(X.abs().pow(2.0), gl.window, 400, 160, 400, 2.0, 3, 0.99, 8000, False)
    assert peak_rel_err(got.cpu().numpy(), gl(X.abs().pow(2.0)).cpu().numpy()) <= 1e-4
    stats = {"mean": torch.randn(80).tolist(), "invstddev": (0.5 + torch.rand(80)).tolist()}
    fe = RNNTFeatureExtractor(stats).cuda()
    got = torch.ops.audio_amd.rnnt_features(x, mel.spectrogram.window, mel.mel_scale.fb, 400, 160, GAIN, fe.mean, fe.invstddev, 4)
    assert torch.equal(got, fe(x))
    pcm = (x * 20000).to(torch.int16)
    got = torch.ops.audio_amd.rnnt_features(pcm, mel.spectrogram.window, mel.mel_scale.fb, 400, 160, GAIN, fe.mean, fe.invstddev, 4)
    assert torch.equal(got, fe(pcm))


@pytest.mark.parametrize("case", [((3, 40000), (3, 9000), "full"), ((2, 2, 20000), (1, 1, 700), "same"),
                                  ((4, 30000), (1, 17000), "valid"), ((1, 500), (2, 20000), "full"),
                                  ((2, 48000), (2, 24000), "full")])
def test_fftconvolve_overlap_save_equals_direct_and_oracle(case):
    """Overlap-save LDS-FFT path vs the time-domain kernel and scipy-checked float64 oracle."""
    import audio_amd.functional as F
    from oracle import dsp_oracle as O
    xs, ys, mode = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(*xs, generator=g)
    y = torch.randn(*ys, generator=g) * torch.exp(-torch.arange(ys[-1]) / (0.3 * ys[-1]))
    fast = F.fftconvolve(x.cuda(), y.cuda(), mode)
    direct = _force_generic(lambda: F.fftconvolve(x.cuda(), y.cuda(), mode))
    assert fast.shape == direct.shape
    assert float((fast - direct).abs().max() / direct.abs().max()) <= 1e-5
    lead = np.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    xe = np.broadcast_to(x.numpy().astype(np.float64), lead + x.shape[-1:])
    ye = np.broadcast_to(y.numpy().astype(np.float64), lead + y.shape[-1:])
    exp = O.fftconvolve(xe, ye, mode)
    assert fast.shape == exp.shape
    assert peak_rel_err(fast.cpu().numpy(), exp) <= 1e-5


@pytest.mark.parametrize("case", [((6, 700000), (1, 24000), "full"), ((3, 2, 300000), (3, 2, 16000), "same"),
                                  ((2, 900000), (2, 30000), "valid"), ((300, 70000), (1, 20000), "full"),
                                  ((4, 60000), (1, 20000), "full"), ((5, 40000), (5, 9000), "same")])
def test_fftconvolve_delay_line_plan_against_oracle(case):
    """The frequency-domain delay-line plan of the overlap-save path (2 / 3 / 4 uniform partitions, row segments when there
    are fewer rows than CUs, per-row taps, slices that start inside the convolution, rows of only 5 blocks) vs the float64
    oracle -- forced by policy, so every shape runs it whatever the cost model says -- and against the plan the cost model
    picks itself; the plan query agrees with the policy."""
    import audio_amd.functional as F
    from audio_amd import _lib
    from oracle import dsp_oracle as O
    xs, ys, mode = case
    g = torch.Generator().manual_seed(xs[-1] + ys[-1])
    x = torch.randn(*xs, generator=g)
    y = torch.randn(*ys, generator=g) * torch.exp(-torch.arange(ys[-1]) / (0.3 * ys[-1]))
    lead = np.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    xe = np.broadcast_to(x.numpy().astype(np.float64), lead + x.shape[-1:])
    ye = np.broadcast_to(y.numpy().astype(np.float64), lead + y.shape[-1:])
    exp = O.fftconvolve(xe, ye, mode)
    rows = int(np.prod(lead))
    xd, yd = x.cuda(), y.cuda()
    with _lib.kernel_policy(_lib.POLICY_FFTCONV_FDL):
        assert _lib.lib().aamd_fftconvolve_plan(rows, xs[-1], ys[-1], exp.shape[-1]) == 2
        fdl = F.fftconvolve(xd, yd, mode)
    with _lib.kernel_policy(_lib.POLICY_FFTCONV_NO_FDL):
        assert _lib.lib().aamd_fftconvolve_plan(rows, xs[-1], ys[-1], exp.shape[-1]) == 1
        rec = F.fftconvolve(xd, yd, mode)
    with _lib.kernel_policy(_lib.POLICY_FFTCONV_COMPLEX):       # the cost model among the complex-block plans (rounds 1-3)
        cplan = _lib.lib().aamd_fftconvolve_plan(rows, xs[-1], ys[-1], exp.shape[-1])
        cown = F.fftconvolve(xd, yd, mode)
    assert cplan in (1, 2) and torch.equal(cown, fdl if cplan == 2 else rec)
    own = F.fftconvolve(xd, yd, mode)
    plan = _lib.lib().aamd_fftconvolve_plan(rows, xs[-1], ys[-1], exp.shape[-1])
    # default: the real-block kernel (plan 3) serves 193 .. 32768 taps (round 5: a third delayed spectrum in registers)
    assert plan == (3 if 192 < ys[-1] <= 32768 else cplan)
    assert fdl.shape == exp.shape == own.shape
    assert peak_rel_err(fdl.cpu().numpy(), exp) <= 1e-5
    assert peak_rel_err(rec.cpu().numpy(), exp) <= 1e-5
    assert peak_rel_err(own.cpu().numpy(), exp) <= 1e-5
    if plan != 3:
        assert torch.equal(own, cown)


@pytest.mark.parametrize("case", [((4, 60001), (1, 193), "full"), ((2, 2, 50000), (2, 2, 2400), "same"), ((3, 40000), (3, 8192), "valid"),
                                  ((6, 20001), (1, 4001), "full"),
                                  ((3, 50001), (1, 8193), "full"), ((2, 3, 33333), (2, 3, 12345), "same"),
                                  ((5, 70001), (5, 24576), "valid"), ((1, 200000), (1, 17000), "full"),
                                  ((7, 16385), (1, 16384), "full"), ((2, 9000), (2, 8500), "full"),
                                  ((3, 90001), (1, 24577), "full"), ((2, 2, 70000), (2, 1, 32768), "same"),
                                  ((260, 40000), (1, 30001), "valid")])
def test_fftconvolve_real_block_delay_line_edges(case):
    """Plan 3 (csrc/fftconv_fdr.h) at its edges: the smallest / largest tap counts it serves, odd row lengths (rows on odd
    float offsets take the scalar load / store paths), outputs that end inside a block, a single row cut into segments, taps
    as long as the signal, per-row taps, all three modes -- against the float64 oracle, and bit-equal on a second call."""
    import audio_amd.functional as F
    from audio_amd import _lib
    from oracle import dsp_oracle as O
    xs, ys, mode = case
    g = torch.Generator().manual_seed(xs[-1] * 3 + ys[-1])
    x = torch.randn(*xs, generator=g)
    y = torch.randn(*ys, generator=g) * torch.exp(-torch.arange(ys[-1]) / (0.3 * ys[-1]))
    lead = np.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    exp = O.fftconvolve(np.broadcast_to(x.numpy().astype(np.float64), lead + x.shape[-1:]),
                        np.broadcast_to(y.numpy().astype(np.float64), lead + y.shape[-1:]), mode)
    rows = int(np.prod(lead))
    assert _lib.lib().aamd_fftconvolve_plan(rows, xs[-1], ys[-1], exp.shape[-1]) == 3
    got = F.fftconvolve(x.cuda(), y.cuda(), mode)
    assert got.shape == exp.shape
    assert peak_rel_err(got.cpu().numpy(), exp) <= 1e-5
    assert torch.equal(got, F.fftconvolve(x.cuda(), y.cuda(), mode))


def test_fftconvolve_headline_shape_properties():
    """BASELINE config 5b shape at 1/8 of the batch (32 x 8 ch x 10 s @48 kHz, 0.5 s RIR = 24000 taps,
    shared RIR): length, linearity, and an impulse RIR reproduces the (delayed) input exactly."""
    import audio_amd.functional as F
    g = torch.Generator(device="cuda").manual_seed(4321)
    x = torch.rand(32, 8, 480000, device="cuda", generator=g) - 0.5
    t = torch.arange(24000, device="cuda") / 48000.0
    rir = (torch.randn(1, 1, 24000, device="cuda", generator=g) * torch.exp(-t / 0.1) * 0.05)
    from audio_amd import _lib
    assert _lib.lib().aamd_fftconvolve_plan(256, 480000, 24000, 503999) == 3      # the real-block delay line serves this shape
    y = F.fftconvolve(x, rir)
    assert y.shape == (32, 8, 503999) and torch.isfinite(y).all()
    y2 = F.fftconvolve(0.5 * x[:2], rir)
    assert float((y2 - 0.5 * y[:2]).abs().max()) <= 2e-6 * float(y.abs().max())
    imp = torch.zeros(1, 1, 24000, device="cuda")
    imp[..., 1234] = 1.0
    z = F.fftconvolve(x[:1], imp)
    assert float((z[..., 1234:1234 + 480000] - x[:1]).abs().max()) <= 2e-6
    # spot-check a few outputs against a float64 dot product
    xr = x[3, 5].double().cpu().numpy()
    h = rir[0, 0].double().cpu().numpy()
    for n in (0, 23999, 100000, 503998):
        lo, hi = max(0, n - 23999), min(n, 479999)
        ref = float(np.dot(xr[lo:hi + 1], h[n - np.arange(lo, hi + 1)]))
        assert abs(float(y[3, 5, n]) - ref) <= 2e-5 * float(y.abs().max())


def test_lfilter_wave_kernel_equals_workgroup_kernel_and_oracle():
    """Big-batch biquad path (one wave per sequence, shuffle scan) vs the workgroup-scan kernel and the
    sequential float64 oracle: shared and per-channel coefficients, ragged length, no clamp, 1st order."""
    import audio_amd.functional as F
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(3)
    x = (0.3 * torch.randn(40, 2, 9001, generator=g))
    cases = [
        (torch.tensor([1.0, -1.8, 0.85]), torch.tensor([0.02, 0.04, 0.02]), True),
        (torch.tensor([[1.0, -1.8, 0.85], [0.9, -0.7, 0.2]]), torch.tensor([[0.02, 0.04, 0.02], [0.3, -0.1, 0.2]]), True),
        (torch.tensor([1.0, -0.6]), torch.tensor([0.5, 0.4]), False),
    ]
    for a, b, clamp in cases:
        fast = F.lfilter(x.cuda(), a.cuda(), b.cuda(), clamp=clamp)
        gen = _force_generic(lambda: F.lfilter(x.cuda(), a.cuda(), b.cuda(), clamp=clamp))
        assert float((fast - gen).abs().max()) <= 2e-5 * max(float(gen.abs().max()), 1e-3)
        if a.ndim == 1:
            exp = O.lfilter(x.numpy().astype(np.float64), a.numpy(), b.numpy(), clamp)
        else:
            exp = np.stack([O.lfilter(x[:, c].numpy().astype(np.float64), a[c].numpy(), b[c].numpy(), clamp) for c in range(2)], 1)
        assert peak_rel_err(fast.cpu().numpy(), exp) <= 1e-4


@pytest.mark.parametrize("design", ["butter4", "butter6_high", "band4", "cheby6", "butter3", "per_channel", "delay"])
def test_lfilter_orders_3_to_8_as_second_order_sections(design):
    """F.lfilter of order 3 .. 8 on big batches runs as second-order sections on the biquad-class kernels (clamp after the
    last section only): against the sequential float64 oracle of the DIRECT form and against the general-order kernel, with
    an input loud enough that the clamp matters, clamp on and off, shared and per-channel coefficients."""
    from scipy import signal
    import audio_amd.functional as F
    from audio_amd import _lib
    from oracle import dsp_oracle as O
    C = 4
    if design == "per_channel":
        ba = [signal.butter(4, w) for w in (0.1, 0.2, 0.3, 0.45)]
        b = torch.tensor(np.stack([x[0] for x in ba]), dtype=torch.float32)
        a = torch.tensor(np.stack([x[1] for x in ba]), dtype=torch.float32)
    else:
        bb, aa = {"butter4": signal.butter(4, 0.2), "butter6_high": signal.butter(6, 0.3, "high"),
                  "band4": signal.butter(4, [0.1, 0.3], "band"), "cheby6": signal.cheby1(6, 1, 0.2),
                  "butter3": signal.butter(3, 0.25),
                  "delay": (np.array([0.0, 0.0, 0.3, 0.1]), np.array([1.0, -0.5, 0.2, -0.1]))}[design]
        b, a = torch.tensor(bb, dtype=torch.float32), torch.tensor(aa, dtype=torch.float32)
    g = torch.Generator().manual_seed(len(design))
    x = (torch.rand(5, C, 220000, generator=g) - 0.5) * 6.0
    xd, ad, bd = x.cuda(), a.cuda(), b.cuda()
    assert F._lfilter_sections(ad, bd, ad.reshape(-1, a.shape[-1]), bd.reshape(-1, b.shape[-1])) is not None
    for clamp in (True, False):
        got = F.lfilter(xd, ad, bd, clamp=clamp)
        with _lib.kernel_policy(_lib.POLICY_FORCE_GENERIC):
            gen = F._lfilter_launch(xd[:1].contiguous(), ad.reshape(1, -1, a.shape[-1]), bd.reshape(1, -1, b.shape[-1]), clamp)
        # float64 direct form: scipy over the whole length, the sequential oracle on a prefix (causal: the prefix of the
        # output depends on the prefix of the input only)
        a64, b64 = np.atleast_2d(a.numpy().astype(np.float64)), np.atleast_2d(b.numpy().astype(np.float64))
        exp = np.stack([np.stack([signal.lfilter(b64[c % len(b64)], a64[c % len(a64)], x[n, c].numpy().astype(np.float64))
                                  for c in range(C)]) for n in range(2)])
        loud = np.abs(exp).max() > 1.05
        if clamp:
            exp = np.clip(exp, -1.0, 1.0)
        pre = O.lfilter(x[:1, :, :3000].numpy().astype(np.float64), a.numpy().astype(np.float64),
                        b.numpy().astype(np.float64), clamp)
        assert np.abs(pre - exp[:1, :, :3000]).max() <= 1e-9
        # (of the peak AFTER the clamp, i.e. of 1.0, while the recursion runs at the input's +-3: hence not 1e-5)
        assert peak_rel_err(got[:2].cpu().numpy(), exp) <= (3e-5 if a.shape[-1] <= 5 else 5e-5)
        # the general-order kernel carries its state in float64 (round 3; the float32 scan of round 2 lost 2e-3 .. 8e-2 of
        # the peak on the 6th-order designs): every design, same bar
        assert peak_rel_err(gen.cpu().numpy(), exp[:1]) <= 1e-4
        if clamp:
            assert float(got.abs().max()) <= 1.0
            assert not loud or float((got[:2].abs() == 1.0).float().mean()) > 0.0        # the clamp was hit where it must be


def test_lfilter_cascade_headline_shape_properties():
    """BASELINE config 5a at 1/8 of the batch (32 x 8 ch x 10 s @48 kHz): the fused 4-biquad cascade
    equals four sequential F.lfilter calls (each clamped), and filtering is causal + time invariant."""
    import audio_amd.functional as F
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.rand(32, 8, 480000, device="cuda", generator=g) - 0.5
    sr = 48000
    A, B = [], []
    for fc in (8000.0, 6000.0, 4000.0, 3000.0):
        w0 = 2 * math.pi * fc / sr
        alpha = math.sin(w0) / 2 / 0.707
        A.append([1 + alpha, -2 * math.cos(w0), 1 - alpha])
        B.append([(1 - math.cos(w0)) / 2, 1 - math.cos(w0), (1 - math.cos(w0)) / 2])
    a = torch.tensor(A, device="cuda")
    b = torch.tensor(B, device="cuda")
    y = F.biquad_cascade(x, a, b)
    assert y.shape == x.shape and torch.isfinite(y).all()
    ref = x[:2]
    for s in range(4):
        ref = F.lfilter(ref, a[s], b[s])
    assert float((y[:2] - ref).abs().max()) <= 1e-5
    # causality / time invariance: delaying the input by d samples delays the output by d
    d = 777
    xd = torch.zeros_like(x[:1])
    xd[..., d:] = x[:1, :, :-d]
    yd = F.biquad_cascade(xd, a, b)
    assert float((yd[..., d:] - y[:1, :, :-d]).abs().max()) <= 1e-5
    assert float(yd[..., :d].abs().max()) == 0.0


@pytest.mark.parametrize("n_mels,mel_scale,norm,f_min,f_max", [
    (40, "htk", None, 0.0, None), (64, "slaney", "slaney", 20.0, 7600.0), (128, "htk", None, 0.0, None),
    (160, "slaney", None, 50.0, None), (23, "htk", "slaney", 0.0, 4000.0), (80, "htk", None, 0.0, None)])
def test_mel400_fast_path_filterbank_matrix(n_mels, mel_scale, norm, f_min, f_max):
    """The radix-20x20 kernel with other filterbanks than the headline one (band tables, lane order,
    rounds with unused rows, n_mels not a multiple of 4 / 20) vs the float64 oracle, ragged length."""
    import warnings
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(n_mels)
    x = (0.5 * torch.randn(3, 5003, generator=g)).clamp_(-1, 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=n_mels, mel_scale=mel_scale,
                             norm=norm, f_min=f_min, f_max=f_max).cuda()
        fb = O.melscale_fbanks(201, f_min, f_max if f_max is not None else 8000.0, n_mels, 16000, norm, mel_scale)
    got = t(x.cuda())
    exp = O.mel_spectrogram(x.numpy().astype(np.float64), O.hann_window(400), fb, 400, 160)
    assert got.shape == exp.shape
    assert peak_rel_err(got.cpu().numpy(), exp) <= 1e-4
    gen = _force_generic(lambda: t(x.cuda()))
    assert float((got - gen).abs().max() / gen.abs().max()) <= 3e-6


def test_mel400_fast_path_layout_edge_cases():
    """Inputs the LDS-DMA staging cannot take as-is: unaligned row starts (sliced views), padded row
    strides, a window shorter than n_fft, normalized spectra, hamming window; and a long clip."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(9)
    base = (0.5 * torch.randn(4, 9000, generator=g)).clamp_(-1, 1).cuda()
    t = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    fb = O.melscale_fbanks(201, 0.0, 8000.0, 80, 16000)
    for view in (base[:, 1:], base[:, 3:8004], base[:, ::1][:, 2:-1], base[1:3, 5:]):
        got = t(view)
        exp = O.mel_spectrogram(view.cpu().numpy().astype(np.float64), O.hann_window(400), fb, 400, 160)
        assert got.shape == exp.shape and peak_rel_err(got.cpu().numpy(), exp) <= 1e-4
    # win_length < n_fft (centre-padded window), hamming, normalized="window"
    t2 = T.MelSpectrogram(sample_rate=16000, n_fft=400, win_length=320, hop_length=160, n_mels=80,
                          window_fn=torch.hamming_window, normalized=True).cuda()
    got = t2(base)
    w = torch.hamming_window(320).numpy().astype(np.float64)
    exp = O.mel_spectrogram(base.cpu().numpy().astype(np.float64), w, fb, 400, 160, win_length=320, normalized=True)
    assert peak_rel_err(got.cpu().numpy(), exp) <= 1e-4
    fastgen = _force_generic(lambda: t2(base))
    assert float((got - fastgen).abs().max() / fastgen.abs().max()) <= 3e-6
    # one long clip (20 minutes at 16 kHz): many tiles per row, fast == generic on the tail frames
    long = (0.5 * torch.randn(1, 19_200_000, generator=g)).clamp_(-1, 1).cuda()
    y = t(long)
    assert y.shape == (1, 80, 120001) and torch.isfinite(y).all()
    tail = t(long[:, -48000:])
    assert float((y[..., -200:] - tail[..., -200:]).abs().max() / tail.abs().max()) <= 3e-6


def test_lfilter_wave_kernel_shape_edges():
    """Workgroup-width selection of the biquad kernel: one long sequence (16 waves), thousands of short
    ones (1 wave each, shorter than a block), gain-only 'filter', unaligned views."""
    import audio_amd.functional as F
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(13)
    a = torch.tensor([1.0, -1.5, 0.7])
    b = torch.tensor([0.2, 0.1, 0.05])
    for shape in [(1, 70001), (3000, 100), (7, 2048), (2, 3, 2049)]:
        x = 0.3 * torch.randn(*shape, generator=g)
        got = F.lfilter(x.cuda(), a.cuda(), b.cuda())
        exp = O.lfilter(x.numpy().astype(np.float64), a.numpy(), b.numpy(), True)
        assert got.shape == exp.shape and peak_rel_err(got.cpu().numpy(), exp) <= 1e-4, shape
    x = 0.3 * torch.randn(4, 5000, generator=g)
    got = F.lfilter(x.cuda(), torch.tensor([2.0]).cuda(), torch.tensor([0.5]).cuda())          # order 0
    assert float((got.cpu() - (0.25 * x).clamp(-1, 1)).abs().max()) <= 1e-6
    xv = (0.3 * torch.randn(4, 9001, generator=g)).cuda()[:, 1:]                                 # unaligned rows
    got = F.lfilter(xv, a.cuda(), b.cuda(), clamp=False)
    exp = O.lfilter(xv.cpu().numpy().astype(np.float64), a.numpy(), b.numpy(), False)
    assert peak_rel_err(got.cpu().numpy(), exp) <= 1e-4


@pytest.mark.parametrize("n_stages,length", [(1, 70004), (2, 40964), (3, 16384 + 8), (4, 65536), (8, 20000),
                                              (5, 2 * 16384 + 6148), (6, 49156), (7, 3 * 16384 - 4), (3, 5 * 16384)])
def test_lfilter_pipelined_kernel_against_oracle(n_stages, length):
    """Long, 16-byte aligned rows take the two-tile kernel (copies and stores issued a few pieces per stage): whole blocks,
    a ragged last block (waves partly / wholly past the end), a length that is an exact multiple of the block, every
    piece-per-stage split (1 .. 8 stages; from 3 stages on the mover kernel, whose stores run in n and copies in n - 1 slots per
    block since round 6) -- against the sequential float64 cascade, clamped per stage."""
    import audio_amd.functional as F
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(100 + n_stages)
    x = (0.6 * torch.randn(3, 2, length, generator=g)).clamp_(-1, 1)
    A, B = [], []
    for s in range(n_stages):
        w0 = 2 * math.pi * (300.0 + 700.0 * s) / 16000
        alpha = math.sin(w0) / 2 / (0.6 + 0.1 * s)
        A.append([1 + alpha, -2 * math.cos(w0), 1 - alpha])
        B.append([1.3 * (1 - math.cos(w0)) / 2, 1.3 * (1 - math.cos(w0)), 1.3 * (1 - math.cos(w0)) / 2])   # gain > 1: clamps
    a, b = torch.tensor(A), torch.tensor(B)
    got = F.biquad_cascade(x.cuda(), a.cuda(), b.cuda()) if n_stages > 1 else F.lfilter(x.cuda(), a[0].cuda(), b[0].cuda())
    exp = x.numpy().astype(np.float64)
    for s in range(n_stages):
        exp = O.lfilter(exp, a[s].numpy().astype(np.float64), b[s].numpy().astype(np.float64), True)
    assert got.shape == x.shape
    assert peak_rel_err(got.cpu().numpy(), exp) <= 2e-4, (n_stages, length)
    assert float(np.abs(got.cpu().numpy()[..., -5:] - exp[..., -5:]).max()) <= 2e-4          # the ragged tail is written


@pytest.mark.parametrize("hop,n_mfcc,shape", [(160, 40, (6, 48000)), (200, 40, (2, 3, 30011)), (100, 16, (5, 12345)),
                                               (160, 48, (3, 16000))])
def test_mfcc_one_kernel_path_equals_two_kernel_path_and_oracle(hop, n_mfcc, shape):
    """The fused MFCC (mel -> dB -> DCT in the radix-20x20 kernel's epilogue + the fix-up launch for the tiles the top_db
    cut-off reaches) against the exact two-kernel path and the float64 oracle: loud noise (nothing clamped), a quiet clip and
    digital silence (clamped tiles), 2-D input (ONE batch-global cut-off) and 3-D input (one per batch item), ragged tails."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(hop + n_mfcc)
    x = (0.4 * torch.randn(*shape, generator=g)).clamp_(-1, 1)
    xs = x.clone()
    xs.view(-1, shape[-1])[1] *= 1e-4
    xs.view(-1, shape[-1])[0, shape[-1] // 2:] = 0.0
    kw = dict(sample_rate=16000, n_mfcc=n_mfcc, melkwargs=dict(n_fft=400, hop_length=hop, n_mels=80))
    fused, exact = T.MFCC(**kw).cuda(), T.MFCC(**kw).cuda()
    fused.fused, exact.fused = True, False
    for inp, clamps in ((x, False), (xs, True)):
        with torch.no_grad():
            a, b = fused(inp.cuda()), exact(inp.cuda())
        assert fused.fused_report()["path"] == "fused" and exact._fused_state.path is None
        assert a.shape == b.shape == tuple(shape[:-1]) + (n_mfcc, shape[-1] // hop + 1)
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * scale + 2e-4, (hop, n_mfcc, clamps)
        torch.cuda.synchronize()
        share = fused.fused_report()["redone_share"]
        assert share is not None and ((share > 0) if clamps else (share == 0.0)), share
        exp = O.mfcc(inp.numpy().astype(np.float64), fused.MelSpectrogram.spectrogram.window.cpu().numpy().astype(np.float64),
                     fused.MelSpectrogram.mel_scale.fb.cpu().numpy().astype(np.float64),
                     fused.dct_mat.cpu().numpy().astype(np.float64), 400, hop)
        assert float(np.abs(a.cpu().numpy() - exp).max()) <= 2e-6 * float(np.abs(exp).max()) + 5e-4      # MFCC absolute (dB scale)


def test_group_max_scratch_pool_hands_out_fresh_slices():
    """The -inf scratch of the top_db group maxima comes from a per-(device, stream) pool filled once per ~32 calls (one
    launch less per MFCC / amplitude_to_DB call): every slice is -inf when handed out, no two calls share one, a second stream
    gets its own pool, a capture in progress bypasses the pool, and results do not change across a refill."""
    import audio_amd.functional as F
    import audio_amd.transforms as T
    dev = torch.device("cuda", 0)
    pool = F._NegInfPool(calls_per_fill=4)
    seen, live = [], []
    for i in range(11):
        t = pool.take(300, dev)
        assert t.shape == (300,) and bool(torch.isinf(t).all()) and bool((t < 0).all())
        seen.append((t.untyped_storage().data_ptr(), t.storage_offset()))
        live.append(t)                               # (a dropped pool's memory may come back from the allocator: compare LIVE slices)
        t.fill_(float(i))                            # a call max-reduces into its slice: the next one must not see that
    assert len(set(seen)) == len(seen)
    assert len({p for p, _ in seen}) >= 2            # refilled at least once
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        u = pool.take(300, dev)
    assert u.untyped_storage().data_ptr() not in {p for p, _ in seen} and bool(torch.isinf(u).all())
    big = pool.take(100000, dev)
    assert big.storage_offset() == 0 and big.numel() == 100000
    g = torch.Generator().manual_seed(3)
    x = (0.4 * torch.randn(6, 20000, generator=g)).clamp_(-1, 1).cuda()
    x[2, 9000:] = 0.0
    db = T.AmplitudeToDB(top_db=60.0).cuda()
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    with torch.no_grad():
        p = mel(x)
        first = db(p).clone()
        for i in range(70):                          # across two refills of the module-level pool
            assert torch.equal(db(p), first), i


def test_mfcc_path_choice_is_taken_once_per_module():
    """`fused="auto"`: the module decides at its FIRST eligible call (synchronised) from the share of tiles that call had to
    redo, and keeps that arithmetic; a zero-padded first batch puts it on the two-kernel path -- including for that first
    call, so every call of a module is the same arithmetic; shapes the fused kernel does not serve (n_mels != 80,
    n_mfcc % 4) take the two-kernel path directly."""
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(5)
    kw0 = dict(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80))
    m = T.MFCC(**kw0).cuda()
    assert m.fused == "auto"
    loud = (0.4 * torch.randn(8, 32000, generator=g)).clamp_(-1, 1).cuda()
    padded = loud.clone()
    padded[:, 4000:] = 0.0
    exact = T.MFCC(**kw0).cuda()
    exact.fused = False
    one = T.MFCC(**kw0).cuda()
    one.fused = True
    with torch.no_grad():
        y0 = m(loud)
        rep = m.fused_report()
        assert rep["decided"] == "fused" and rep["decided_share"] == 0.0 and rep["path"] == "fused"
        assert torch.equal(y0, one(loud))
        frag = m._fused_state.frag              # the DCT operand fragments are built ONCE per module: the buffer is a
        y1 = m(padded)                          # transposed view, and its kernel-ready copy must not be remade per call
        assert m._fused_state.frag is frag
        assert m.fused_report()["path"] == "fused" and m.fused_report()["redone_share"] > 0.5
        assert torch.equal(y1, one(padded))
        m2 = T.MFCC(**kw0).cuda()               # a module whose first batch is zero padded decides the other way ...
        z1 = m2(padded)
        rep = m2.fused_report()
        assert rep["decided"] == "two-kernel" and rep["decided_share"] > 0.5 and rep["path"] == "two-kernel"
        assert torch.equal(z1, exact(padded))   # ... and already that first call is the two-kernel arithmetic
        assert torch.equal(m2(loud), exact(loud)) and m2.fused_report()["calls_fused"] == 0
        m2.reset_fused_decision()
        assert torch.equal(m2(loud), one(loud)) and m2.fused_report()["decided"] == "fused"
        for kw in (dict(n_mfcc=13, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)),
                   dict(n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=64))):
            mm = T.MFCC(sample_rate=16000, **kw).cuda()
            mm(loud)
            assert mm.fused_report()["path"] == "two-kernel"


@pytest.mark.parametrize("fused", [True, False, "auto"])
def test_mfcc_repeated_calls_are_bit_identical(fused):
    """50 calls on a half-silent batch return the same bits, whatever the path setting."""
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(12)
    x = (0.4 * torch.randn(16, 24000, generator=g)).clamp_(-1, 1)
    x[8:] = 0.0
    x[3, 9000:] = 0.0
    xd = x.cuda()
    m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()
    m.fused = fused
    with torch.no_grad():
        first = m(xd).clone()
        other = (0.2 * torch.randn(5, 16000, generator=g)).cuda()
        for i in range(50):
            if i % 7 == 3:
                m(other)                         # another batch in between does not move the path
            assert torch.equal(m(xd), first), i
    rep = m.fused_report()
    assert (rep["calls_fused"] == 0) or (rep["calls_two_kernel"] == 0)


def test_mfcc_default_module_under_hip_graph_capture():
    """the default module is capture-safe -- while a capture is in progress "auto" takes no decision (no host
    read, no pinned copy, no event) and runs the one-kernel path; the replayed graph returns the eager result."""
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(21)
    x = (0.4 * torch.randn(4, 16000, generator=g)).clamp_(-1, 1).cuda()
    x[2, 5000:] = 0.0
    m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()
    with torch.no_grad():
        m.fused = True                           # warm the module's side tables (band image, DCT fragments) without deciding
        want = m(x).clone()
        m.fused = "auto"
        assert m.fused_report()["decided"] is None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m.fused = True
            m(x)
            m.fused = "auto"
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y = m(x)
        assert m._fused_state.decided is None and m._fused_state.path == "fused"
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, want)


def test_resample_matrix_core_layout_edges():
    """Unaligned / strided inputs (scalar loader path) and a length shorter than one chunk."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    r = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                   rolloff=0.9475937167399596, beta=14.769656459379492).cuda()
    kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596,
              beta=14.769656459379492)
    g = torch.Generator().manual_seed(19)
    base = (0.5 * torch.randn(3, 30011, generator=g)).clamp_(-1, 1).cuda()
    for view in (base[:, 1:], base[:, 2:20001], base[:, :300], base[1:, 7:]):
        got = r(view)
        exp = O.resample(view.cpu().numpy().astype(np.float64), 44100, 16000, **kw)
        assert got.shape == exp.shape and peak_rel_err(got.cpu().numpy(), exp) <= 1e-5


@pytest.mark.parametrize("gain", [1.0, 1e-4, 3e4])
def test_resample_binary16_split_kernel_against_fp32_kernel_and_oracle(gain):
    """cfg3's filter (44.1k -> 16k kaiser_best) on the f16 matrix pipe (operands split into two binary16 numbers, chunk-wise
    power-of-two scaling) against the fp32-MFMA kernel (policy switch) and the float64 oracle: normal, very quiet and
    16-bit-range amplitudes, a clip that falls silent, ragged length."""
    import audio_amd.transforms as T
    from audio_amd import _lib
    from oracle import dsp_oracle as O
    kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596,
              beta=14.769656459379492)
    r = T.Resample(44100, 16000, **kw).cuda()
    g = torch.Generator().manual_seed(23)
    x = (0.5 * torch.randn(3, 2, 70013, generator=g)).clamp_(-1, 1) * gain
    x[1, :, 30000:] *= 1e-3
    x[2, 1, 20000:] = 0.0
    with torch.no_grad():
        got = r(x.cuda())
        with _lib.kernel_policy(_lib.POLICY_RESAMPLE_FP32):
            ref32 = r(x.cuda())
    exp = O.resample(x.numpy().astype(np.float64), 44100, 16000, **kw)
    assert got.shape == exp.shape
    assert peak_rel_err(got.cpu().numpy(), exp) <= 1e-5 and peak_rel_err(ref32.cpu().numpy(), exp) <= 1e-5     # (fp32 taps vs the oracle's float64 taps)
    assert float((got - ref32).abs().max()) <= 2e-6 * float(ref32.abs().max())                  # measured: see profiles/r02_v_resample_f16.txt
    q = got[1, :, 12000:].cpu().numpy(), exp[1, :, 12000:]                       # the quiet part, against ITS peak
    assert float(np.abs(q[0] - q[1]).max()) <= 2e-5 * float(np.abs(q[1]).max())
    assert float(got[2, 1, 7400:].abs().max()) == 0.0                             # silence stays silence


def test_resample_8_byte_operand_layout_against_the_4_byte_one():
    """Round 5: for odd reduced `orig` and bands of 257 .. 448 taps (cfg3: 441 : 160, kaiser_best) the binary16-split resampler
    deals the output groups to its two MFMA tiles by parity and walks the contraction steps in a permuted order in the odd lane groups, so
    that its LDS operand reads are 8-byte aligned ds_read_b64 with one two-way conflicted step of 13 (csrc/resample_mfma.h, b64_sigma).  Same products, another summation
    order: against the 4-byte layout (AAMD_POLICY_RESAMPLE_B32) <= 2e-6 of the peak, both <= 1e-5 of the float64 oracle; ragged
    lengths, unaligned views (the 16-byte phase of a chunk changes which tile is the aligned one), a row that falls silent; and
    a rate pair outside the layout (even orig) is bit-identical under the switch."""
    import audio_amd.transforms as T
    from audio_amd import _lib
    from oracle import dsp_oracle as O
    kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596,
              beta=14.769656459379492)
    r = T.Resample(44100, 16000, **kw).cuda()
    g = torch.Generator().manual_seed(29)
    base = (0.5 * torch.randn(5, 90017, generator=g)).clamp_(-1, 1)
    base[3, 40000:] = 0.0
    base = base.cuda()
    for view in (base, base[:, 1:], base[:, 2:70001], base[1:, 3:], base[:, :14113], base[:, :500]):
        with torch.no_grad():
            got = r(view)
            with _lib.kernel_policy(_lib.POLICY_RESAMPLE_B32):
                ref = r(view)
        exp = O.resample(view.cpu().numpy().astype(np.float64), 44100, 16000, **kw)
        assert got.shape == exp.shape == ref.shape
        peak = float(np.abs(exp).max())
        assert float((got - ref).abs().max()) <= 2e-6 * peak
        assert peak_rel_err(got.cpu().numpy(), exp) <= 1e-5 and peak_rel_err(ref.cpu().numpy(), exp) <= 1e-5
    assert float(r(base)[3, 14600:].abs().max()) == 0.0                  # silence stays silence
    r2 = T.Resample(48000, 44100, **kw).cuda()                             # 160 : 147, even orig: the 4-byte layout either way
    x = base[:, :50000]
    with torch.no_grad():
        a = r2(x)
        with _lib.kernel_policy(_lib.POLICY_RESAMPLE_B32):
            b = r2(x)
    assert torch.equal(a, b)


def test_resample_prepared_tap_fragments_are_bit_identical_to_the_in_kernel_split():
    """Round 5 (C ABI 7): the packed binary16 (hi, lo) tap fragments of the matrix-core resampler are prepared once per kernel
    tensor (aamd_resample_frag_build_f32; 13 % of a BASELINE config-3 launch when every workgroup forms them itself) and read by
    both operand layouts.  Same values, same arithmetic: bit-identical to the call without them (C entry with frag = NULL), on
    cfg3's filter (8-byte layout), under AAMD_POLICY_RESAMPLE_B32 (4-byte layout), on a default-quality pair with filled phase
    tiles (48 k -> 16 k), on a pair with more than 14 phase tiles (several build launches), and after the kernel tensor was
    changed in place (the cache key holds its version)."""
    import ctypes as C
    import audio_amd.transforms as T
    from audio_amd import _host, _lib
    g = torch.Generator().manual_seed(31)
    x = (0.5 * torch.randn(3, 60011, generator=g)).clamp_(-1, 1).cuda()

    def unprepared(r, orig, new, xin):           # the C entry without fragments, on the tensors the module holds
        kern = r.kernel.reshape(r.kernel.shape[0], -1).contiguous()
        lo, span = _host.resample_band_table(kern.cpu().numpy())
        lo = np.ascontiguousarray(lo, dtype=np.int32)
        bands = _lib.ResampleBands(lo.shape[0], span, lo.ctypes.data_as(C.POINTER(C.c_int32)))
        out_len = -(-new * xin.shape[1] // orig)
        out = torch.empty(xin.shape[0], out_len, device="cuda")
        _lib.check(_lib.lib().aamd_resample_prepared_f32(xin.data_ptr(), kern.data_ptr(), out.data_ptr(), xin.shape[0], xin.shape[1],
                                                         xin.shape[1], orig, new, r.width, out_len, C.byref(bands), None,
                                                         _lib.current_stream(xin.device)))
        return out

    kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596, beta=14.769656459379492)
    r = T.Resample(44100, 16000, **kw).cuda()
    with torch.no_grad():
        got = r(x)
        assert torch.equal(got, unprepared(r, 441, 160, x))
        with _lib.kernel_policy(_lib.POLICY_RESAMPLE_B32):
            assert torch.equal(r(x), unprepared(r, 441, 160, x))
        r.kernel.mul_(0.5)                                        # in place: a new version, new fragments
        assert torch.equal(r(x), unprepared(r, 441, 160, x))
        assert float((r(x) - 0.5 * got).abs().max()) <= 1e-6 * float(got.abs().max())
        r3 = T.Resample(44100, 48000, **kw).cuda()                # 147 : 160 -> 160 phases = 10 tiles; and a long table:
        assert torch.equal(r3(x), unprepared(r3, 147, 160, x))
        r4 = T.Resample(16000, 22050).cuda()                      # 320 : 441 -> 28 phase tiles: two build launches
        assert torch.equal(r4(x), unprepared(r4, 320, 441, x))
        with _lib.kernel_policy(_lib.POLICY_RESAMPLE_FP32):
            y4f = r4(x)
        assert float((r4(x) - y4f).abs().max()) <= 2e-6 * float(y4f.abs().max())
        r2 = T.Resample(48000, 16000).cuda()                      # one phase: the host fills the 16-phase tile (3 : 1 -> 48 : 16)
        y2 = r2(x)
        with _lib.kernel_policy(_lib.POLICY_RESAMPLE_FP32):       # the exact-fp32 MFMA kernel ignores the fragments
            y2f = r2(x)
        assert float((y2 - y2f).abs().max()) <= 2e-6 * float(y2f.abs().max())


def test_resample_click_and_minus_100_db_tone_in_one_chunk():
    """the block-floating scaling of the binary16-split resampler is weakest where ONE chunk holds a
    full-scale click AND a passage 100 dB below it: the chunk's power-of-two scale is set by the click, the quiet samples land
    near the bottom of binary16's normal range and their low parts in its subnormals.  Against the float64 oracle and against
    the fp32-MFMA kernel on the same input (`check_click_and_quiet_tone`); the same case runs through the CPU replay of both
    kernels in tests/test_cpu_sim.py."""
    import audio_amd.transforms as T
    from audio_amd import _lib
    from oracle import dsp_oracle as O
    kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596,
              beta=14.769656459379492)
    r = T.Resample(44100, 16000, **kw).cuda()
    x = click_and_quiet_tone()
    xt = torch.from_numpy(x)
    with torch.no_grad():
        got = r(xt.cuda()).cpu().numpy()
        with _lib.kernel_policy(_lib.POLICY_RESAMPLE_FP32):
            ref32 = r(xt.cuda()).cpu().numpy()
    exp = O.resample(x.astype(np.float64), 44100, 16000, **kw)
    check_click_and_quiet_tone(got, ref32, exp)


@pytest.mark.parametrize("hop", [100, 200])
def test_fft400_other_hops_fast_path(hop):
    """hop = 100 and 200 (torchaudio's default n_fft // 2) also take the radix-20x20 kernel: mel, MFCC and
    spectrogram vs the generic kernel and the float64 oracle, ragged lengths."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(hop)
    fb = O.melscale_fbanks(201, 0.0, 8000.0, 80, 16000)
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=hop, n_mels=80).cuda()
    spec = T.Spectrogram(n_fft=400, hop_length=hop, power=1.0).cuda()
    mfcc = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=hop, n_mels=80)).cuda()
    for L in (401, 1009, 16000, 48017):
        x = (0.5 * torch.randn(3, L, generator=g)).clamp_(-1, 1)
        for t in (mel, spec, mfcc):
            fast = t(x.cuda())
            gen = _force_generic(lambda: t(x.cuda()))
            assert fast.shape == gen.shape and fast.stride() == gen.stride()
            assert float((fast - gen).abs().max() / gen.abs().max()) <= 2e-5, (hop, L, type(t).__name__)
        exp = O.mel_spectrogram(x.numpy().astype(np.float64), O.hann_window(400), fb, 400, hop)
        assert peak_rel_err(mel(x.cuda()).cpu().numpy(), exp) <= 1e-4


@pytest.mark.parametrize("hop", [100, 160, 200])
def test_spec400_complex_fast_path(hop):
    """power=None (complex STFT) on the radix-20x20 kernel vs the generic kernel and the oracle."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator().manual_seed(hop + 1)
    t = T.Spectrogram(n_fft=400, hop_length=hop, power=None).cuda()
    for L in (401, 1283, 16000, 48017):
        x = (0.5 * torch.randn(3, L, generator=g)).clamp_(-1, 1)
        fast = t(x.cuda())
        gen = _force_generic(lambda: t(x.cuda()))
        assert fast.dtype == torch.complex64 and fast.shape == gen.shape and fast.stride() == gen.stride()
        assert float((fast - gen).abs().max() / gen.abs().max()) <= 3e-6
        if L <= 16000:
            exp = O.spectrogram(x.numpy().astype(np.float64), 0, O.hann_window(400), 400, hop, 400, None, False)
            assert np.abs(fast.cpu().numpy() - exp).max() / np.abs(exp).max() <= 1e-4


@pytest.mark.parametrize("name", ["shared_clamp", "shared_noclamp", "per_channel", "order4"])
def test_lfilter_autograd_vs_reference_gradients(name):
    """dL/dx, dL/da, dL/db of F.lfilter (adjoint filters run by the HIP kernels) against gradients recorded
    from the REFERENCE's autograd with its compiled CPU core (tests/golden/make_grad_golden.py)."""
    import audio_amd.functional as F
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_grads.npz"))
    x = torch.tensor(G[f"{name}/x"], dtype=torch.float32, device="cuda", requires_grad=True)
    a = torch.tensor(G[f"{name}/a"], dtype=torch.float32, device="cuda", requires_grad=True)
    b = torch.tensor(G[f"{name}/b"], dtype=torch.float32, device="cuda", requires_grad=True)
    r = torch.tensor(G[f"{name}/r"], dtype=torch.float32, device="cuda")
    y = F.lfilter(x, a, b, clamp=bool(G[f"{name}/clamp"]))
    assert peak_rel_err(y.detach().cpu().numpy(), G[f"{name}/y"]) <= 1e-5
    (y * r).sum().backward()
    for got, key in ((x.grad, "dx"), (a.grad, "da"), (b.grad, "db")):
        exp = G[f"{name}/{key}"]
        assert got.shape == exp.shape
        assert peak_rel_err(got.cpu().numpy(), exp) <= 2e-4, key
    # the thin dB caller is differentiable since round 2 (reference formula under autograd), the inverse STFT since round 4
    # (the adjoint-operator pair of _diff.py; gradcheck in tests/test_gpu_autograd_f64.py)
    import audio_amd.transforms as T
    db = T.AmplitudeToDB()(x.reshape(-1, x.shape[-1]).abs() + 1e-3)
    assert db.requires_grad
    z = torch.randn(1, 201, 5, dtype=torch.complex64, device="cuda", requires_grad=True)
    w = T.InverseSpectrogram(n_fft=400)(z)
    assert w.requires_grad and torch.autograd.grad(w.square().sum(), z)[0].shape == z.shape


@pytest.mark.parametrize("shapes,mode", [(((3, 700), (3, 90)), "full"), (((2, 2, 5000), (1, 1, 400)), "same"),
                                         (((4, 3000), (1, 1200)), "valid"), (((1, 300), (2, 9000)), "full")])
def test_fftconvolve_autograd(shapes, mode):
    """Gradients of F.fftconvolve in both operands (adjoint = convolution with the time-reversed operand on the
    same kernels, broadcast dims summed) vs autograd through the float64 FFT composition on the CPU."""
    import audio_amd.functional as F
    g = torch.Generator().manual_seed(31)
    x = torch.randn(*shapes[0], generator=g, dtype=torch.float64)
    y = torch.randn(*shapes[1], generator=g, dtype=torch.float64) * 0.3
    xr, yr = x.clone().requires_grad_(), y.clone().requires_grad_()
    n = x.shape[-1] + y.shape[-1] - 1
    full = torch.fft.irfft(torch.fft.rfft(xr, n=n) * torch.fft.rfft(yr, n=n), n=n)
    if mode == "full":
        ref = full
    else:
        m = x.shape[-1] if mode == "same" else max(x.shape[-1], y.shape[-1]) - min(x.shape[-1], y.shape[-1]) + 1
        s0 = (n - m) // 2
        ref = full[..., s0:s0 + m]
    r = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * r).sum().backward()
    xg = x.float().cuda().requires_grad_()
    yg = y.float().cuda().requires_grad_()
    z = F.fftconvolve(xg, yg, mode)
    assert peak_rel_err(z.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    (z * r.float().cuda()).sum().backward()
    assert peak_rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) <= 2e-5
    assert peak_rel_err(yg.grad.cpu().numpy(), yr.grad.numpy()) <= 2e-5


@pytest.mark.parametrize("rates", [(44100, 16000, True), (16000, 44100, False), (48000, 16000, False), (8000, 12000, False)])
def test_resample_autograd(rates):
    """dL/dx of Resample: one more launch of the polyphase kernel with the adjoint tap table, vs autograd
    through the reference's pad + conv1d composition (float64, CPU)."""
    import audio_amd.transforms as T
    from oracle import torch_cpu_ref as R
    o, n, best = rates
    kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596,
              beta=14.769656459379492) if best else {}
    t = T.Resample(o, n, **kw).cuda()
    g = torch.Generator().manual_seed(5)
    x = (0.5 * torch.randn(2, 3, 2011, generator=g, dtype=torch.float64))
    xr = x.clone().requires_grad_()
    gcd = math.gcd(o, n)
    ref = R.resample(xr, t.kernel.cpu().double(), o // gcd, n // gcd, t.width)
    r = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * r).sum().backward()
    xg = x.float().cuda().requires_grad_()
    y = t(xg)
    assert y.shape == ref.shape and peak_rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
    (y * r.float().cuda()).sum().backward()
    assert xg.grad.shape == x.shape
    assert peak_rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) <= 2e-5


def _ref_spectrogram(x, window, n_fft, hop, pad, power, normalized, center, pad_mode):
    """The reference composition (functional/functional.py:123-145) in float64 on the CPU, differentiable."""
    if pad > 0:
        x = torch.nn.functional.pad(x, (pad, pad), "constant")
    shape = x.shape
    X = torch.stft(x.reshape(-1, shape[-1]), n_fft, hop, window.shape[0], window, center, pad_mode, False, True,
                   return_complex=True)
    X = X.reshape(shape[:-1] + X.shape[-2:])
    if normalized == "window" or normalized is True:
        X = X / window.pow(2.0).sum().sqrt()
    elif normalized == "frame_length":
        X = X / math.sqrt(n_fft)
    if power is None:
        return X
    return X.abs().pow(2.0) if power == 2.0 else X.abs().pow(power)


@pytest.mark.parametrize("cfg", [
    dict(n_fft=400, hop=160, L=4000, power=2.0),                                    # radix-20x20 fast path
    dict(n_fft=400, hop=160, L=4000, power=1.0),
    dict(n_fft=400, hop=200, L=3001, power=None),
    dict(n_fft=400, hop=200, L=12000, power=2.0),                 # last interior tile ends exactly at the row end
    dict(n_fft=400, hop=200, L=12000, power=None, pad_mode="circular"),
    dict(n_fft=400, hop=100, L=6700, power=1.0, pad_mode="replicate"),
    dict(n_fft=512, hop=128, L=3000, power=2.0, pad_mode="constant"),
    dict(n_fft=256, hop=64, L=2000, power=2.0, center=False, normalized="window"),
    dict(n_fft=200, hop=50, L=1777, power=1.0, pad=37, normalized="frame_length"),
    dict(n_fft=97, hop=31, L=1500, power=2.0, pad_mode="replicate"),
    dict(n_fft=128, hop=32, win_length=100, L=1500, power=3.0, pad_mode="circular"),
    dict(n_fft=2048, hop=512, L=9000, power=None),
    dict(n_fft=64, hop=16, L=700, power=2.0, lead=(2, 3)),
])
def test_spectrogram_autograd(cfg):
    """dL/dwaveform of Spectrogram: backward = complex STFT recompute + the adjoint overlap-add kernel
    (csrc/istft.h, adjoint = 1), vs torch autograd through torch.stft in float64 on the CPU."""
    import audio_amd.transforms as T
    n_fft, hop, L, power = cfg["n_fft"], cfg["hop"], cfg["L"], cfg["power"]
    win_length = cfg.get("win_length", n_fft)
    kw = dict(pad=cfg.get("pad", 0), normalized=cfg.get("normalized", False), center=cfg.get("center", True),
              pad_mode=cfg.get("pad_mode", "reflect"))
    g = torch.Generator().manual_seed(n_fft * 7 + hop)
    lead = cfg.get("lead", (3,))
    x = 0.5 * torch.randn(*lead, L, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_()
    w = torch.hann_window(win_length, dtype=torch.float64)
    ref = _ref_spectrogram(xr, w, n_fft, hop, kw["pad"], power, kw["normalized"], kw["center"], kw["pad_mode"])
    r = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    if power is None:
        r = torch.complex(r, torch.randn(ref.shape, generator=g, dtype=torch.float64))
        (ref * r.conj()).real.sum().backward()
    else:
        (ref * r).sum().backward()
    t = T.Spectrogram(n_fft=n_fft, hop_length=hop, win_length=win_length, power=power, **kw).cuda()
    xg = x.float().cuda().requires_grad_()
    y = t(xg)
    assert y.shape == ref.shape and y.stride() == t(xg.detach()).stride()
    if power is None:
        assert peak_rel_err(torch.view_as_real(y.detach()).cpu().numpy(), torch.view_as_real(ref.detach()).numpy()) <= 1e-5
        (y * r.to(torch.complex64).cuda().conj()).real.sum().backward()
    else:
        assert peak_rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-5
        (y * r.float().cuda()).sum().backward()
    assert xg.grad.shape == x.shape
    assert peak_rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) <= 2e-5


@pytest.mark.parametrize("which", ["mel", "mfcc", "mfcc_log", "mel_generic"])
def test_mel_and_mfcc_autograd(which):
    """Gradients of MelSpectrogram / MFCC w.r.t. the waveform (HIP spectrogram + adjoint, torch tail) vs torch
    autograd through the reference composition in float64 on the CPU."""
    import audio_amd.transforms as T
    from oracle import torch_cpu_ref as R
    g = torch.Generator().manual_seed(77)
    x = 0.5 * torch.randn(2, 2, 4800, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_()
    n_fft, hop = (400, 160) if which != "mel_generic" else (512, 200)
    if which.startswith("mel"):
        t = T.MelSpectrogram(sample_rate=16000, n_fft=n_fft, hop_length=hop, n_mels=80).cuda()
        ref = R.mel_spectrogram(xr, torch.hann_window(n_fft, dtype=torch.float64), t.mel_scale.fb.cpu().double(), n_fft, hop)
    else:
        t = T.MFCC(sample_rate=16000, n_mfcc=40, log_mels=(which == "mfcc_log"),
                   melkwargs=dict(n_fft=n_fft, hop_length=hop, n_mels=80)).cuda()
        fb, dct = t.MelSpectrogram.mel_scale.fb.cpu().double(), t.dct_mat.cpu().double()
        if which == "mfcc_log":
            mel = R.mel_spectrogram(xr, torch.hann_window(n_fft, dtype=torch.float64), fb, n_fft, hop)
            ref = torch.matmul(torch.log(mel + 1e-6).transpose(-1, -2), dct).transpose(-1, -2)
        else:
            ref = R.mfcc(xr, torch.hann_window(n_fft, dtype=torch.float64), fb, dct, n_fft, hop)
    r = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * r).sum().backward()
    xg = x.float().cuda().requires_grad_()
    y = t(xg)
    assert y.shape == ref.shape
    assert peak_rel_err(y.detach().cpu().numpy(), ref.detach().numpy()) <= 1e-4
    with torch.no_grad():
        assert peak_rel_err(y.detach().cpu().numpy(), t(xg.detach()).cpu().numpy()) <= 1e-5     # fused path agrees
    (y * r.float().cuda()).sum().backward()
    assert peak_rel_err(xg.grad.cpu().numpy(), xr.grad.numpy()) <= 1e-4


@pytest.mark.parametrize("cfg", [
    dict(n_fft=400, hop=160, L=4000),
    dict(n_fft=400, hop=160, L=4000, length=3900),
    dict(n_fft=512, hop=128, L=5000, normalized="window"),
    dict(n_fft=256, hop=64, L=3000, normalized="frame_length", length=3000, pad=20),
    dict(n_fft=200, hop=50, win_length=160, L=2222, lead=(2, 2)),
    dict(n_fft=97, hop=24, L=1500),
    dict(n_fft=2048, hop=512, L=20000, length=20000),
    dict(n_fft=128, hop=32, L=1000, center=False),
])
def test_inverse_spectrogram_vs_torch_istft(cfg):
    """InverseSpectrogram (inverse FFT + window + overlap-add + envelope division in one HIP kernel) against
    torch.istft in float64 on the CPU (what functional/functional.py:148-225 calls), and the round trip."""
    import audio_amd.transforms as T
    n_fft, hop, L = cfg["n_fft"], cfg["hop"], cfg["L"]
    win_length = cfg.get("win_length", n_fft)
    normalized, center = cfg.get("normalized", False), cfg.get("center", True)
    length, pad = cfg.get("length"), cfg.get("pad", 0)
    g = torch.Generator().manual_seed(n_fft + hop)
    x = 0.5 * torch.randn(*cfg.get("lead", (3,)), L, generator=g, dtype=torch.float64)
    w = torch.hann_window(win_length, dtype=torch.float64)
    if not center:
        w = w + 0.1                                   # torch.istft needs a nonzero envelope at the edges
    X = _ref_spectrogram(x, w, n_fft, hop, pad, None, normalized, center, "reflect")
    Xs = X
    if normalized == "window":
        Xs = X * w.pow(2.0).sum().sqrt()
    elif normalized == "frame_length":
        Xs = X * math.sqrt(n_fft)
    shape = Xs.shape
    ref = torch.istft(Xs.reshape(-1, shape[-2], shape[-1]), n_fft, hop, win_length, w, center, False, True,
                      length + 2 * pad if length is not None else None, False)
    if length is not None and pad > 0:
        ref = ref[:, pad:-pad]
    ref = ref.reshape(shape[:-2] + ref.shape[-1:])
    wfn = (lambda n: torch.hann_window(n)) if center else (lambda n: torch.hann_window(n) + 0.1)
    t = T.InverseSpectrogram(n_fft=n_fft, hop_length=hop, win_length=win_length, normalized=normalized, center=center,
                             pad=pad, window_fn=wfn).cuda()
    got = t(X.to(torch.complex64).cuda(), length)
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert peak_rel_err(got.cpu().numpy(), ref.numpy()) <= 1e-5
    # round trip through this repo's own Spectrogram (power=None)
    s = T.Spectrogram(n_fft=n_fft, hop_length=hop, win_length=win_length, power=None, normalized=normalized,
                      center=center, pad=pad, window_fn=wfn).cuda()
    xb = x.float().cuda()
    back = t(s(xb), L)
    frames = X.shape[-1]
    covered = n_fft + hop * (frames - 1) - (2 * (n_fft // 2) if center else 0) - 2 * pad
    n = min(back.shape[-1], L, covered)
    assert peak_rel_err(back[..., :n].cpu().numpy(), x[..., :n].numpy()) <= 1e-5


def _widening():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "widening_goldens.npz"))


@pytest.mark.parametrize("name", ["pv_fast", "pv_slow", "pv_big"])
def test_phase_vocoder_vs_reference(name):
    """T.TimeStretch / F.phase_vocoder (one HIP kernel) against the reference's float32 CPU output.  Magnitudes are
    interpolated (tight); the phase is a running sum of up to ~1e5 rad held in float32, so the reference's own
    float32 result differs from its float64 result by 2-3e-4 of the peak (stored as out64; pv_slow's float64 run
    even picks different frames) -- the complex tolerance is that noise level, the 99.9 % quantile is 1e-4."""
    import audio_amd.functional as F
    import audio_amd.transforms as T
    G = _widening()
    n_fft, hop, rate = G[f"{name}/cfg"]
    n_fft, hop, rate = int(n_fft), int(hop), float(rate)
    spec = torch.tensor(G[f"{name}/spec"]).cuda()
    ref = G[f"{name}/out"]
    with torch.no_grad():
        t = T.TimeStretch(hop_length=hop, n_freq=n_fft // 2 + 1, fixed_rate=rate).cuda()
        y = t(spec)
        assert y.shape == ref.shape and y.dtype == torch.complex64 and y.is_contiguous()
        got = y.cpu().numpy()
        assert peak_rel_err(np.abs(got), np.abs(ref)) <= 1e-5
        d = np.abs(got - ref) / np.abs(ref).max()
        assert d.max() <= 5e-4 and np.quantile(d, 0.999) <= 1e-4
        if name != "pv_slow":
            noise = np.abs(ref - G[f"{name}/out64"]).max() / np.abs(ref).max()
            assert d.max() <= 1.5 * noise           # no worse than float32 is for the reference itself
        # frame-major (transposed-view) input, the layout Spectrogram returns: same bits
        fm = spec.transpose(-1, -2).contiguous().transpose(-1, -2)
        assert torch.equal(F.phase_vocoder(fm, rate, t.phase_advance), y)
        assert F.phase_vocoder(spec, 1.0, t.phase_advance) is spec
        with pytest.raises(ValueError):
            T.TimeStretch(hop_length=hop, n_freq=n_fft // 2 + 1).cuda()(spec)


@pytest.mark.parametrize("name", ["inv_400", "inv_512"])
def test_inverse_spectrogram_vs_reference_fixture(name):
    import audio_amd.transforms as T
    G = _widening()
    n_fft, hop, length = (int(v) for v in G[f"{name}/cfg"])
    spec = torch.tensor(G[f"{name}/spec"]).cuda()
    with torch.no_grad():
        y = T.InverseSpectrogram(n_fft=n_fft, hop_length=hop).cuda()(spec, None if length < 0 else length)
    assert y.shape == G[f"{name}/out"].shape
    assert peak_rel_err(y.cpu().numpy(), G[f"{name}/out"]) <= 1e-5


@pytest.mark.parametrize("name", ["gl_400", "gl_512"])
def test_griffinlim_vs_reference(name):
    """T.GriffinLim (rand_init=False) against the reference's float32 CPU output.  The iteration divides by |angles|,
    so float32 rounding is amplified at weak bins: the reference's float32 and float64 runs differ by up to 5e-4
    of the peak (out64); tolerance 2e-3, and the result must be as consistent with the magnitudes as the reference's."""
    import audio_amd.transforms as T
    G = _widening()
    n_fft, hop, power, n_iter, momentum, L = G[f"{name}/cfg"]
    spec = torch.tensor(G[f"{name}/spec"]).cuda()
    with torch.no_grad():
        t = T.GriffinLim(n_fft=int(n_fft), hop_length=int(hop), power=float(power), n_iter=int(n_iter),
                         momentum=float(momentum), length=int(L), rand_init=False).cuda()
        y = t(spec)
        ref = G[f"{name}/out"]
        assert y.shape == ref.shape
        assert peak_rel_err(y.cpu().numpy(), ref) <= 2e-3
        assert peak_rel_err(y.cpu().numpy(), G[f"{name}/out64"]) <= 2e-3
        # spectral convergence of the reconstruction (how well |STFT(y)| matches the target magnitudes)
        s = T.Spectrogram(n_fft=int(n_fft), hop_length=int(hop), power=float(power)).cuda()
        sc = lambda w: float(((s(w) - spec).norm() / spec.norm()))   # noqa: E731
        assert sc(y) <= sc(torch.tensor(ref).cuda()) * 1.01 + 1e-6
        # random start: different phases, same fixed-point quality class
        t2 = T.GriffinLim(n_fft=int(n_fft), hop_length=int(hop), power=float(power), n_iter=int(n_iter),
                          momentum=float(momentum), length=int(L), rand_init=True).cuda()
        assert t2(spec).shape == ref.shape
    with pytest.raises(ValueError):
        T.GriffinLim(momentum=1.0)


@pytest.mark.parametrize("name", ["ps_up", "ps_down"])
def test_pitch_shift_vs_reference(name):
    """T.PitchShift / F.pitch_shift = STFT -> phase vocoder -> inverse STFT -> polyphase resampler, all HIP.  The phase
    vocoder's float32 phase noise (see test_phase_vocoder_vs_reference) passes through: the reference's float32 and
    float64 runs differ by 1.1e-3 / 4.5e-4 of the peak (out64); tolerance 3e-3 on the peak, 3e-4 on the 99 % quantile."""
    import audio_amd.functional as F
    import audio_amd.transforms as T
    G = _widening()
    sr, n_steps = (int(v) for v in G[f"{name}/cfg"])
    x = torch.tensor(G[f"{name}/x"]).cuda()
    with torch.no_grad():
        y = T.PitchShift(sr, n_steps).cuda()(x)
        yf = F.pitch_shift(x, sr, n_steps)
    for got, key in ((y, "out"), (yf, "out_f")):
        ref = G[f"{name}/{key}"]
        assert got.shape == ref.shape
        d = np.abs(got.cpu().numpy() - ref) / np.abs(ref).max()
        assert d.max() <= 3e-3 and np.quantile(d, 0.99) <= 3e-4


def test_speed_and_speed_perturbation_vs_reference():
    import audio_amd.functional as F
    import audio_amd.transforms as T
    G = _widening()
    x = torch.tensor(G["speed/x"]).cuda()
    lengths = torch.tensor(G["speed/lengths"]).cuda()
    with torch.no_grad():
        y, yl = T.Speed(16000, 1.1).cuda()(x, lengths)
        y2, none = F.speed(x, 16000, 1.1)
        assert none is None
        assert peak_rel_err(y.cpu().numpy(), G["speed/out"]) <= 1e-5
        assert peak_rel_err(y2.cpu().numpy(), G["speed/out"]) <= 1e-5
        assert np.array_equal(yl.cpu().numpy(), G["speed/out_lengths"])
        torch.manual_seed(0)
        sp = T.SpeedPerturbation(16000, [0.9, 1.0, 1.1]).cuda()
        shapes = {tuple(sp(x)[0].shape) for _ in range(12)}
        assert (3, 4000) in shapes and len(shapes) >= 2


@pytest.mark.parametrize("n_fft,hop", [(256, 64), (256, 100), (512, 128), (512, 160), (1024, 256), (1024, 411), (2048, 512)])
def test_pow2_wave_fft_equals_generic_and_torch_stft(n_fft, hop):
    """The register-resident wave FFT (csrc/stft_pow2.h; n_fft = 512 / 1024 / 2048) against the generic Stockham
    kernel and torch.stft in float64 on the CPU: power, magnitude, complex and mel outputs, every padding mode,
    ragged lengths (odd frame counts, rows shorter than a frame)."""
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(n_fft + hop)
    w64 = torch.hann_window(n_fft, dtype=torch.float64)
    for L, pad_mode, center in ((n_fft // 2 + 3, "reflect", True), (3 * n_fft + 17, "reflect", True),
                                (5 * n_fft, "constant", True), (4 * n_fft + 1, "replicate", True),
                                (2 * n_fft + 100, "circular", True), (6 * n_fft + 5, "reflect", False)):
        x = torch.randn(3, L, generator=g).clamp_(-1, 1)
        xc = x.cuda()
        ref = torch.stft(x.double(), n_fft, hop, n_fft, w64, center, pad_mode, False, True, return_complex=True)
        for power in (2.0, 1.0, None):
            t = T.Spectrogram(n_fft=n_fft, hop_length=hop, power=power, center=center, pad_mode=pad_mode).cuda()
            fast = t(xc)
            gen = _force_generic(lambda: t(xc))
            assert fast.shape == gen.shape == ref.shape and fast.stride() == gen.stride()
            if power is None:
                exp = torch.view_as_real(ref).numpy()
                got, gg = torch.view_as_real(fast).cpu().numpy(), torch.view_as_real(gen).cpu().numpy()
            else:
                exp = ref.abs().pow(power).numpy()
                got, gg = fast.cpu().numpy(), gen.cpu().numpy()
            assert peak_rel_err(got, gg) <= 3e-6, (L, pad_mode, power)
            assert peak_rel_err(got, exp) <= 1e-5, (L, pad_mode, power)
        if center and pad_mode == "reflect":
            m = T.MelSpectrogram(sample_rate=16000, n_fft=n_fft, hop_length=hop, n_mels=96, normalized=True).cuda()
            fast = m(xc)
            gen = _force_generic(lambda: m(xc))
            assert fast.shape == gen.shape and fast.stride() == gen.stride()
            assert peak_rel_err(fast.cpu().numpy(), gen.cpu().numpy()) <= 3e-6
            exp = torch.matmul((ref / w64.pow(2).sum().sqrt()).abs().pow(2).transpose(-1, -2), m.mel_scale.fb.cpu().double())
            assert peak_rel_err(fast.cpu().numpy(), exp.transpose(-1, -2).numpy()) <= 1e-5
    # wide filterbank (does not fit the LDS table): weights are read from global memory
    m = T.MelSpectrogram(sample_rate=48000, n_fft=n_fft, hop_length=hop, n_mels=24, f_min=20.0).cuda()
    x = torch.randn(2, 4 * n_fft, generator=g).clamp_(-1, 1).cuda()
    assert peak_rel_err(m(x).cpu().numpy(), _force_generic(lambda: m(x)).cpu().numpy()) <= 3e-6


@pytest.mark.parametrize("n_fft,hop", [(256, 64), (512, 160), (1024, 256), (2048, 512)])
def test_pow2_mel_band_walk_lanes_per_mel(n_fft, hop):
    """Round 6: the last round of the power-of-two kernel's band walk runs on G = 1 / 2 / 4 / 8 lanes per mel, G from
    n_mels alone (mel_tail_lanes): every G, the last round being the only one (n_mels <= 64), full rounds (64, 128), a ragged
    last frame pair -- against the generic kernel and the float64 composition of torch.stft with the module's own filterbank."""
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(n_fft)
    x = torch.randn(3, 7 * n_fft + 33, generator=g).clamp_(-1, 1)
    w64 = torch.hann_window(n_fft, dtype=torch.float64)
    ref = torch.stft(x.double(), n_fft, hop, n_fft, w64, True, "reflect", False, True, return_complex=True).abs().pow(2)
    for n_mels in (1, 8, 20, 33, 40, 64, 65, 72, 80, 96, 100, 128, 136):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # (narrow filters at the low end have no bin at small n_fft: the reference warns too)
            m = T.MelSpectrogram(sample_rate=16000, n_fft=n_fft, hop_length=hop, n_mels=n_mels).cuda()
        fast = m(x.cuda())
        gen = _force_generic(lambda: m(x.cuda()))
        exp = torch.matmul(ref.transpose(-1, -2), m.mel_scale.fb.cpu().double()).transpose(-1, -2)
        assert fast.shape == gen.shape == exp.shape and fast.stride() == gen.stride()
        assert peak_rel_err(fast.cpu().numpy(), gen.cpu().numpy()) <= 3e-6, n_mels
        assert peak_rel_err(fast.cpu().numpy(), exp.numpy()) <= 1e-5, n_mels


def _assert_rnnt_features_close(got, ref, x_cpu, fe):
    """Feature-domain comparison with the conditioning of the transform written out.  The mel spectrogram itself is
    held to 2e-6 of each frame's peak (fp32 FFT noise floor, ours and the reference's); the post-processing maps a mel
    error dm to dm * gain * plog'(y) * invstddev, which for bins 1e-12 below the frame peak (a tone's far leakage
    bins, amplified by gain = 1.07e9) is not small, and plog itself JUMPS from 1 to e at y = e^e (rnnt_pipeline.py:20-23
    as evaluated).  So: |got - ref| <= 2e-4 + invstddev * gain * plog'(y) * 2e-6 * peak_mel(frame), elements within
    0.5 % of the two break points excluded."""
    import math
    from audio_amd.pipelines import GAIN
    from oracle import torch_cpu_ref as R
    fb = fe.mel.mel_scale.fb.cpu().double()
    mel = R.mel_spectrogram(x_cpu.double()[None], torch.hann_window(400, dtype=torch.float64), fb, 400, 160)[0].transpose(0, 1)
    y = (mel * GAIN).numpy()
    inv = fe.invstddev.cpu().numpy().astype(np.float64)
    e, ee = math.e, math.e ** math.e
    dplog = np.where(y <= e, 1.0 / e, np.where(y <= ee, 1.0 / (np.maximum(y, e) * e), 1.0 / np.maximum(y, e)))
    peak = mel.numpy().max(axis=1, keepdims=True)
    tol = 2e-4 + inv[None, :] * GAIN * dplog * 2e-6 * peak
    near = (np.abs(y / e - 1) < 5e-3) | (np.abs(y / ee - 1) < 5e-3)
    bad = (np.abs(got - ref) > tol) & ~near
    assert not bad.any(), (int(bad.sum()), float(np.abs(got - ref)[bad].max()))
    # and the well-conditioned elements (bins within 1e-5 of the frame peak) are tight
    strong = (mel.numpy() > 1e-5 * peak) & ~near
    assert np.abs(got - ref)[strong].max() <= 2e-4


def test_rnnt_feature_extractor_vs_reference():
    """audio_amd.pipelines.RNNTFeatureExtractor (MelSpectrogram + transpose + gain / piecewise log + global-stats
    normalisation + right padding in ONE launch of the headline kernel) against the reference's own Sequential
    (pipelines/rnnt_pipeline.py:319-326) run on the CPU, and against the unfused composition on the GPU."""
    from audio_amd.pipelines import GAIN, RNNTFeatureExtractor, piecewise_linear_log
    import audio_amd.transforms as T
    G = _widening()
    stats = {"mean": G["rnnt/mean"].tolist(), "invstddev": G["rnnt/invstddev"].tolist()}
    fe = RNNTFeatureExtractor(stats).cuda()
    with torch.no_grad():
        for i in range(2):
            x = torch.tensor(G[f"rnnt{i}/x"]).cuda()
            feats, length = fe(x)
            ref = G[f"rnnt{i}/out"]
            assert feats.shape == ref.shape and int(length) == int(G[f"rnnt{i}/length"].item())
            assert torch.all(feats[-4:] == 0)
            _assert_rnnt_features_close(feats.cpu().numpy()[:-4], ref[:-4], x.cpu(), fe)
        # batched, vs the same steps unfused on the device
        xb = torch.stack([torch.tensor(G["rnnt0/x"])[:15000], torch.tensor(G["rnnt1/x"])[:15000]]).cuda().reshape(2, 1, 15000)
        fb = fe(xb)
        mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, n_mels=80, hop_length=160).cuda()(xb).transpose(-1, -2)
        ref = (piecewise_linear_log(mel * GAIN) - fe.mean) * fe.invstddev
        assert fb.shape == (2, 1, ref.shape[-2] + 4, 80)
        assert float((fb[..., :-4, :] - ref).abs().max()) <= 2e-4 and torch.all(fb[..., -4:, :] == 0)
        # a shape outside the n_fft = 400 fast path: generic mel + the element-wise kernel + torch padding
        fe2 = RNNTFeatureExtractor(stats, n_fft=512, hop_length=128).cuda()
        f2 = fe2(xb)
        mel2 = T.MelSpectrogram(sample_rate=16000, n_fft=512, n_mels=80, hop_length=128).cuda()(xb).transpose(-1, -2)
        ref2 = (piecewise_linear_log(mel2 * GAIN) - fe.mean) * fe.invstddev
        assert f2.shape[-2] == ref2.shape[-2] + 4
        assert float((f2[..., :-4, :] - ref2).abs().max()) <= 2e-4


@pytest.mark.parametrize("hop", [160, 200])
def test_rnnt_features_from_int16_pcm(hop):
    """int16 PCM read directly by the headline kernel (rank 4, upstream half): bit-identical to the float path fed the
    exactly converted samples; ragged / unaligned rows take the unstaged path; other shapes convert first."""
    from audio_amd.pipelines import RNNTFeatureExtractor
    g = torch.Generator().manual_seed(hop)
    stats = {"mean": (10 + 3 * torch.randn(80, generator=g)).tolist(), "invstddev": (0.2 + torch.rand(80, generator=g)).tolist()}
    fe = RNNTFeatureExtractor(stats, hop_length=hop).cuda()
    pcm = torch.randint(-20000, 20000, (5, 48000), generator=g, dtype=torch.int16)
    pcm[1, :50] = 32767
    pcm[2, -7:] = -32768
    with torch.no_grad():
        a = fe(pcm.cuda())
        b = fe(pcm.cuda().float() * (1.0 / 32768.0))
        assert a.shape == b.shape and a.dtype == torch.float32
        # same kernel arithmetic; the only difference is where the 2^-15 factor enters (window vs samples): exact power of two
        assert torch.equal(a, b)
        a2 = fe(pcm[:, 1:47990].cuda())                    # misaligned rows: unstaged gather
        b2 = fe(pcm[:, 1:47990].cuda().float() * (1.0 / 32768.0))
        assert torch.equal(a2, b2)
        one, length = fe(pcm[0].cuda())
        assert torch.equal(one, a[0]) and int(length) == a.shape[1]
        fe512 = RNNTFeatureExtractor(stats, n_fft=512, hop_length=128).cuda()    # not served by the PCM kernel: converts
        c = fe512(pcm.cuda())
        d = fe512(pcm.cuda().float() * (1.0 / 32768.0))
        assert float((c - d).abs().max()) <= 2e-4


@pytest.mark.parametrize("hop", [160, 200])
def test_rnnt_features_from_interleaved_pcm(hop):
    """SURVEY 8(f) rank 4, VERDICT r2 item 8: interleaved (time, channel) int16 PCM -- the decoder's order, which the
    reference transposes first (torchaudio/_torchcodec.py:150-152) -- read directly: stereo and mono are de-interleaved
    in the kernel's load and come out BIT-identical to `pcm.transpose(-1, -2).float() / 32768` through the existing path;
    4 channels transpose first (same values)."""
    from audio_amd.pipelines import RNNTFeatureExtractor
    g = torch.Generator().manual_seed(hop + 3)
    stats = {"mean": (10 + 3 * torch.randn(80, generator=g)).tolist(), "invstddev": (0.2 + torch.rand(80, generator=g)).tolist()}
    fe = RNNTFeatureExtractor(stats, hop_length=hop).cuda()
    for chans in (2, 1, 4):
        pcm = torch.randint(-20000, 20000, (3, 48000, chans), generator=g, dtype=torch.int16)     # (clip, time, channel)
        pcm[1, :50, chans - 1] = 32767
        pcm[2, -7:, 0] = -32768
        with torch.no_grad():
            got = fe.features(pcm.cuda(), channels_first=False)
            want = fe(pcm.transpose(-1, -2).cuda().float() * (1.0 / 32768.0))
            assert got.shape == want.shape == (3, chans, 48000 // hop + 1 + 4, 80)
            assert torch.equal(got, want), chans
            # rows that are not 16-byte multiples / a misaligned start: the unstaged gather
            odd = pcm[:, 1:47990].contiguous()
            got2 = fe.features(odd.cuda(), channels_first=False)
            want2 = fe(odd.transpose(-1, -2).cuda().float() * (1.0 / 32768.0))
            assert torch.equal(got2, want2), chans
    # a contiguous stereo VIEW that starts on an odd half-word (2-byte aligned) is served, not refused
    flat = torch.randint(-20000, 20000, (2 * 48000 * 2 + 1,), generator=g, dtype=torch.int16).cuda()
    view = flat[1:].view(2, 48000, 2)
    assert view.data_ptr() % 4 == 2 and view.is_contiguous()
    with torch.no_grad():
        got3 = fe.features(view, channels_first=False)
        want3 = fe(view.transpose(-1, -2).float() * (1.0 / 32768.0))
    assert torch.equal(got3, want3)
    with pytest.raises(ValueError):
        fe.features(torch.zeros(2, 4800, 2).cuda(), channels_first=False)          # float input is not PCM


def test_mel400_tail_pools_are_bit_identical_to_static_runs_and_reset_themselves():
    """Round 5 experiment (compiled in with -DAAMD_M400_POOLS=1 only; in the product build both arms run the static hand-out and the
    test checks repeated launches on two streams): the last tiles of every workgroup's run of the n_fft = 400 kernel are shared between
    the XCDs through ticket counters in memory (csrc/melspec400.h, pool_tile).  Which wave computes a tile must not matter: every epilogue (mel, dB,
    spectrogram, MFCC pass 0 + fix-up) gives the bits of the static hand-out (AAMD_POLICY_MEL400_NO_POOL), repeated launches
    find their counters back at zero (the launch's last ticket resets them), two streams use separate counter blocks, and a
    batch whose last run is cut short draws the slots past its end without running them."""
    import audio_amd.transforms as T
    from audio_amd import _lib
    g = torch.Generator().manual_seed(77)
    mods = [T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80).cuda(),
            T.MelSpectrogram(16000, n_fft=400, hop_length=200, n_mels=128).cuda(),
            T.Spectrogram(n_fft=400, hop_length=160).cuda(),
            torch.nn.Sequential(T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80), T.AmplitudeToDB(top_db=80.0)).cuda(),
            T.MFCC(16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()]
    # 256 x 10 s: 167 tiles per workgroup, 13 of them pooled; 301 x 7.77 s: the last run is cut short; 40 x 3 s: short runs, no pools
    for rows, n in ((256, 160000), (301, 124321), (40, 48000)):
        x = (0.5 * torch.randn(rows, n, generator=g)).clamp_(-1, 1).cuda()
        x[rows // 3, n // 2:] = 0.0                                    # (MFCC: tiles under the cut-off, redone by the fix-up launch)
        for m in mods:
            with torch.no_grad():
                with _lib.kernel_policy(_lib.POLICY_MEL400_NO_POOL):
                    ref = m(x)
                for _ in range(4):                                     # the fourth launch must find the counters the first one left
                    got = m(x)
                    assert torch.equal(got, ref), (type(m).__name__, rows, n)
    # two streams at once: one counter block per stream
    m = mods[0]
    x = (0.5 * torch.randn(256, 160000, generator=g)).clamp_(-1, 1).cuda()
    with torch.no_grad():
        with _lib.kernel_policy(_lib.POLICY_MEL400_NO_POOL):
            ref = m(x)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = []
        for _ in range(6):
            for st in (s1, s2):
                with torch.cuda.stream(st):
                    outs.append(m(x))
        torch.cuda.synchronize()
        assert all(torch.equal(o, ref) for o in outs)


def test_inverse_spectrogram_odd_frame_count_low_envelope_tail():
    """Fuzz campaign seed 311 (round 5; the CPU replay of the same case is in tests/test_cpu_sim.py): behind an odd number of frames
    the generic inverse STFT overlap-added the rounding cross-talk of the last pair's missing partner frame at full window weight;
    where the envelope is ~4e-5 (hann, hop = n_fft / 2, a length that ends inside the last frame's taper) that was 2e-4 of the
    peak against 5e-6 for torch.istft.  Every size family of the inverse (generic 200 / 96 / 600, wave FFT 256 / 1024, n_fft = 400)."""
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(311)
    for n_fft in (200, 96, 600, 256, 1024, 400):
        hop = n_fft // 2
        L = 11 * hop + int(round(0.975 * n_fft)) + 1          # 13 frames; the row ends where the last frame's window is ~6e-3
        x = (0.5 * torch.randn(3, L, generator=g)).cuda()
        s = T.Spectrogram(n_fft=n_fft, hop_length=hop, power=None).cuda()
        inv = T.InverseSpectrogram(n_fft=n_fft, hop_length=hop).cuda()
        with torch.no_grad():
            X = s(x)
            got = inv(X, L)
        assert X.shape[-1] % 2 == 1, (n_fft, X.shape)
        w = torch.hann_window(n_fft, dtype=torch.float64).cuda()
        ref = torch.istft(X.to(torch.complex128), n_fft, hop, n_fft, w, True, False, True, L, False)
        ref32 = torch.istft(X, n_fft, hop, n_fft, w.float(), True, False, True, L, False)      # aten's float32 path: the yardstick
        bar = max(2e-5, 4.0 * peak_rel_err(ref32.double().cpu().numpy(), ref.cpu().numpy()))
        assert peak_rel_err(got.cpu().numpy(), ref.cpu().numpy()) <= bar, (n_fft, bar)

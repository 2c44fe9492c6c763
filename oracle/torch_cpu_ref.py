"""CPU timing baseline for bench.py -- TEST/BENCH INFRASTRUCTURE, NOT PRODUCT.

The reference's hot path is a thin composition of ATen CPU ops (SURVEY.md "things to know" #1),
so the CPU baseline that can travel to the GPU box (where /root/reference does not exist) is a
PORT that issues the same ATen calls in the same order:
  MelSpectrogram.forward  = F.spectrogram (torch.stft + abs().pow(2))  -> matmul with fb
  (src/torchaudio/functional/functional.py:112-145, transforms/_transforms.py:403-415,612-622)
  MFCC tail               = amplitude_to_DB (functional.py:390-402) + matmul with dct_mat
  Resample                = F.pad + F.conv1d(stride=orig) + transpose/reshape + crop (functional.py:1421-1428)
It is validated against the committed reference-run fixtures by tests/test_oracle_golden.py.
Only bench.py's cpu_baseline leg and tests import this file.
"""
from __future__ import annotations

import math
import time

import torch


def mel_spectrogram(x: torch.Tensor, window: torch.Tensor, fb: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
    shape = x.size()
    x = x.reshape(-1, shape[-1])
    spec = torch.stft(x, n_fft=n_fft, hop_length=hop, win_length=window.shape[0], window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = spec.reshape(shape[:-1] + spec.shape[-2:])
    p = spec.abs().pow(2.0)
    return torch.matmul(p.transpose(-1, -2), fb).transpose(-1, -2)


def amplitude_to_db(x, multiplier=10.0, amin=1e-10, db_multiplier=0.0, top_db=80.0):
    x_db = multiplier * torch.log10(torch.clamp(x, min=amin))
    x_db -= multiplier * db_multiplier
    if top_db is not None:
        shape = x_db.size()
        packed = shape[-3] if x_db.dim() > 2 else 1
        x_db = x_db.reshape(-1, packed, shape[-2], shape[-1])
        x_db = torch.max(x_db, (x_db.amax(dim=(-3, -2, -1)) - top_db).view(-1, 1, 1, 1))
        x_db = x_db.reshape(shape)
    return x_db


def mfcc(x, window, fb, dct_mat, n_fft, hop):
    mel = amplitude_to_db(mel_spectrogram(x, window, fb, n_fft, hop))
    return torch.matmul(mel.transpose(-1, -2), dct_mat).transpose(-1, -2)


def resample(x, kernel, orig, new, width):
    shape = x.size()
    x = x.view(-1, shape[-1])
    n, length = x.shape
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernel, stride=orig)
    y = y.transpose(1, 2).reshape(n, -1)
    target = int(math.ceil(new * length / orig))
    return y[..., :target].view(shape[:-1] + (target,))


def time_mel_baseline(n_clips: int, seconds: float, sample_rate: int = 16000, n_fft: int = 400, hop: int = 160,
                      n_mels: int = 80, budget_s: float = 12.0, seed: int = 1234):
    """Time the CPU path on `n_clips` clips of `seconds` s; returns (audio_sec_per_sec, cores, n_calls)."""
    from audio_amd import _host  # host-side constant builders only (no kernels)
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(n_clips, int(seconds * sample_rate), generator=g)).clamp_(-1, 1)
    window = torch.hann_window(n_fft)
    fb = _host.melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)
    mel_spectrogram(x[:2], window, fb, n_fft, hop)      # warm-up
    t_best, calls, t_total = float("inf"), 0, 0.0
    while calls < 3 or (t_total < budget_s and calls < 50):
        t0 = time.perf_counter()
        mel_spectrogram(x, window, fb, n_fft, hop)
        dt = time.perf_counter() - t0
        t_best = min(t_best, dt)
        t_total += dt
        calls += 1
        if t_total > 2.5 * budget_s:
            break
    return n_clips * seconds / t_best, torch.get_num_threads(), calls


def time_mel_baseline_single_thread(n_clips: int, seconds: float, sample_rate: int = 16000, n_fft: int = 400, hop: int = 160,
                                    n_mels: int = 80, budget_s: float = 6.0, seed: int = 1234):
    """The same composition on ONE host thread (report n in {1, all cores}); returns audio_sec_per_sec."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        v, _, _ = time_mel_baseline(n_clips, seconds, sample_rate, n_fft, hop, n_mels, budget_s=budget_s, seed=seed)
    finally:
        torch.set_num_threads(n)
    return v


def sweep_mel_baseline(n_clips: int, seconds: float, sample_rate: int = 16000, n_fft: int = 400, hop: int = 160,
                       n_mels: int = 80, budget_s: float = 3.0, seed: int = 1234, counts=None):
    """the reference's CPU composition at its BEST over host thread counts.  One big call does not scale
    with torch's intra-op pool (round 2: 128 threads = 1 thread), so the clips are dealt to `threads` Python threads,
    each running the single-threaded composition on its contiguous share (the ops release the GIL).  Returns a list of
    {"threads", "audio_sec_per_sec", "calls"} for threads in {1, 8, 16, 32, 64, all host cores}."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from audio_amd import _host
    cores = os.cpu_count() or 1
    counts = counts or sorted({c for c in (1, 8, 16, 32, 64, cores) if c <= cores})
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(n_clips, int(seconds * sample_rate), generator=g)).clamp_(-1, 1)
    window = torch.hann_window(n_fft)
    fb = _host.melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)
    prev = torch.get_num_threads()
    out = []
    try:
        torch.set_num_threads(1)
        mel_spectrogram(x[:2], window, fb, n_fft, hop)      # warm-up
        for th in counts:
            use = min(th, n_clips)
            bounds = [(n_clips * i // use, n_clips * (i + 1) // use) for i in range(use)]

            def work(b):
                return mel_spectrogram(x[b[0]:b[1]], window, fb, n_fft, hop).shape

            with ThreadPoolExecutor(max_workers=use) as pool:
                t_best, t_total, calls = float("inf"), 0.0, 0
                while calls < 2 or (t_total < budget_s and calls < 40):
                    t0 = time.perf_counter()
                    list(pool.map(work, bounds))
                    dt = time.perf_counter() - t0
                    t_best, t_total, calls = min(t_best, dt), t_total + dt, calls + 1
                    if t_total > 2.5 * budget_s:
                        break
            out.append({"threads": th, "audio_sec_per_sec": n_clips * seconds / t_best, "calls": calls})
    finally:
        torch.set_num_threads(prev)
    return out


def protocol_set_num_threads(n_clips: int, seconds: float, sample_rate: int = 16000, n_fft: int = 400, hop: int = 160,
                             n_mels: int = 80, budget_s: float = 4.0, seed: int = 1234):
    """BASELINE.md section 4 as written: ONE call over the whole batch under torch.set_num_threads(n) for n in {1, all host
    cores}, 1 warm-up + >= 3 timed calls, the best reported.  (The dealt-threads sweep above is the more favourable figure for
    the CPU -- one big call does not scale with the intra-op pool -- and stays the headline `value`; this is the stated
    protocol, printed beside it.)  Returns {"1": audio-sec/sec, "all": ..., "all_threads": n}."""
    import os
    cores = os.cpu_count() or 1
    prev = torch.get_num_threads()
    out = {}
    try:
        for key, n in (("1", 1), ("all", cores)):
            torch.set_num_threads(n)
            v, _, calls = time_mel_baseline(n_clips, seconds, sample_rate, n_fft, hop, n_mels, budget_s=budget_s, seed=seed)
            out[key] = v
            out[key + "_calls"] = calls
    finally:
        torch.set_num_threads(prev)
    out["all_threads"] = cores
    return out

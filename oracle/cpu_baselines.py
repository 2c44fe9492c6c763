"""CPU baselines of the BASELINE configs beside the headline -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT.

VERDICT r5 next 6: every entry of `bench.py`'s `configs` carries a `cpu_baseline` of its own.  Each baseline is a PORT that issues
the ATen calls of the reference's CPU path in the reference's order (`/root/reference` does not exist on the GPU box, so the
reference itself cannot be imported there), on a BOUNDED sample of the config's workload, the rows dealt over host threads the
way `oracle/torch_cpu_ref.sweep_mel_baseline` deals clips (one big call does not scale with torch's intra-op pool):

  cfg3  Resample      F.pad + F.conv1d(stride = orig) + reshape + crop          functional/functional.py:1405-1432
  cfg4  MFCC          stft -> |.|^2 -> mel matmul -> amplitude_to_DB -> DCT matmul   transforms/_transforms.py:692-709
  cfg5a lfilter x 4   FIR = F.pad + grouped F.conv1d; IIR = the compiled core loop; clamp   functional/filtering.py:941-1099,
                      core loop: oracle/_ref (the reference's own lfilter.cpp, compiled here from its sources: `core =
                      "reference"`) when that library travelled with the tree, else oracle/lfilter_core.c (`core = "port"`)
  cfg5b fftconvolve   irfft(rfft(x, n) * rfft(y, n), n)                          functional/functional.py:2252-2258

Only bench.py's `cpu_baseline` leg (tools/bench_configs.py) and tests import this file.  Checked against the float64 oracle /
the committed reference-run fixtures by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time
from concurrent.futures import ThreadPoolExecutor

import torch

from . import torch_cpu_ref as ref

HERE = os.path.dirname(os.path.abspath(__file__))
_CORE = {}


def lfilter_core():
    """(callable(x3, a_flipped, y_padded, seq_lo, seq_hi), "reference" | "port")."""
    if _CORE:
        return _CORE["fn"], _CORE["kind"]
    so = os.path.join(HERE, "_build", "liboracle_lfilter.so")
    if not os.path.exists(so):
        raise RuntimeError("oracle/_build/liboracle_lfilter.so is not built (make -C oracle)")
    lib = C.CDLL(so)
    lib.oracle_lfilter_core_f32.restype = None
    lib.oracle_lfilter_core_f32.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 5

    def port(x3, a_flipped, y_padded, lo, hi):
        lib.oracle_lfilter_core_f32(x3.data_ptr(), a_flipped.data_ptr(), y_padded.data_ptr(), x3.shape[1], x3.shape[2],
                                    a_flipped.shape[1], lo, hi)

    _CORE["fn"], _CORE["kind"] = port, "port"
    return _CORE["fn"], _CORE["kind"]


def reference_lfilter_core():
    """The reference's compiled loop (oracle/_ref/ref_libtorchaudio_lfilter.so: `torchaudio::_lfilter_core_loop` on the CPU key),
    or None where it did not travel / the schema is already taken by another library in this process."""
    so = os.path.join(HERE, "_ref", "ref_libtorchaudio_lfilter.so")
    if not os.path.exists(so):
        return None
    try:
        torch.ops.load_library(so)
        return torch.ops.torchaudio._lfilter_core_loop
    except Exception:
        return None


def lfilter(waveform: torch.Tensor, a_coeffs: torch.Tensor, b_coeffs: torch.Tensor, clamp: bool = True, core=None) -> torch.Tensor:
    """F.lfilter on (batch, channel, time) with per-channel (channel, order) coefficients: filtering.py:941-1099 as ATen calls +
    the core loop (`core(x3, a_flipped, y_padded)`; default: the C restatement over all sequences)."""
    n_order = a_coeffs.shape[1]
    n_channel = a_coeffs.shape[0]
    b = (b_coeffs / a_coeffs[:, 0:1]).flip(1).contiguous()
    a = (a_coeffs / a_coeffs[:, 0:1]).flip(1).contiguous()
    fir = torch.nn.functional.conv1d(torch.nn.functional.pad(waveform, (n_order - 1, 0)), b.unsqueeze(1), groups=n_channel)
    fir = fir.contiguous()
    y = torch.zeros(fir.shape[0], fir.shape[1], fir.shape[2] + n_order - 1, dtype=fir.dtype)
    if core is None:
        fn, _ = lfilter_core()
        fn(fir, a, y, 0, fir.shape[0] * fir.shape[1])
    else:
        core(fir, a, y)
    out = y[:, :, n_order - 1:]
    return out.clamp(-1.0, 1.0) if clamp else out


def biquad_cascade(x3: torch.Tensor, a4: torch.Tensor, b4: torch.Tensor, core=None) -> torch.Tensor:
    """cfg5a: four sequential F.lfilter calls (each clamped, the reference's default), the same coefficients on every channel."""
    ch = x3.shape[1]
    for s in range(a4.shape[0]):
        x3 = lfilter(x3, a4[s:s + 1].expand(ch, -1).contiguous(), b4[s:s + 1].expand(ch, -1).contiguous(), True, core)
    return x3


def fftconvolve(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    n = x.size(-1) + y.size(-1) - 1
    return torch.fft.irfft(torch.fft.rfft(x, n=n) * torch.fft.rfft(y, n=n), n=n)


def _dealt(work, n_rows: int, threads, budget_s: float, rows_per_thread: int = 1):
    """Best wall time of `work(lo, hi)` with `rows_per_thread` rows on each of `t` host threads (t x rows_per_thread rows of the
    sample, at most all of them), for every t in `threads`: every measurement costs about one thread's share, whatever t is.
    Returns a list of {"threads", "rows", "seconds", "calls"}.  torch's intra-op pool is pinned to 1 (the dealt threads are the
    parallelism)."""
    prev = torch.get_num_threads()
    out = []
    try:
        torch.set_num_threads(1)
        work(0, min(n_rows, 1))                                  # warm-up (FFT plans, conv algorithm selection)
        for t in threads:
            use = max(1, min(t, n_rows))
            rows_t = min(n_rows, use * rows_per_thread)
            bounds = [(rows_t * i // use, rows_t * (i + 1) // use) for i in range(use)]
            with ThreadPoolExecutor(max_workers=use) as pool:
                best, total, calls = float("inf"), 0.0, 0
                while calls < 2 or (total < budget_s and calls < 20):
                    t0 = time.perf_counter()
                    list(pool.map(lambda b: work(b[0], b[1]), bounds))
                    dt = time.perf_counter() - t0
                    best, total, calls = min(best, dt), total + dt, calls + 1
                    if total > 2.5 * budget_s:
                        break
            out.append({"threads": t, "rows": rows_t, "seconds": best, "calls": calls})
    finally:
        torch.set_num_threads(prev)
    return out


def _thread_counts():
    cores = os.cpu_count() or 1
    return sorted({c for c in (1, 16, 64) if c <= cores} | ({cores} if cores < 64 else set()))


def _record(sweep, seconds_per_row: float, sample: str, kind: str, extra=None):
    rate = lambda r: r["rows"] * seconds_per_row / r["seconds"]        # noqa: E731
    best = max(sweep, key=rate)
    rec = {"value": rate(best), "unit": "audio-sec/sec", "cores": best["threads"], "kind": kind,
           "sample": sample, "host_cores": os.cpu_count(),
           "sweep": [{"threads": r["threads"], "rows": r["rows"], "audio_sec_per_sec": rate(r), "calls": r["calls"]} for r in sweep]}
    if extra:
        rec.update(extra)
    return rec


def time_cfg3(budget_s: float = 3.0, rows: int = 64, seconds: float = 10.0, seed: int = 1234):
    """Resample 44.1 k -> 16 k kaiser_best on `rows` mono rows of `seconds` s (cfg3's shard: 256 rows x 30 s)."""
    from audio_amd import _host
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(rows, int(44100 * seconds), generator=g)).clamp_(-1, 1)
    gcd = math.gcd(44100, 16000)
    kernel, width = _host.sinc_resample_kernel(44100, 16000, gcd, 64, 0.9475937167399596, "sinc_interp_kaiser",
                                               14.769656459379492)
    orig, new = 44100 // gcd, 16000 // gcd
    sweep = _dealt(lambda lo, hi: ref.resample(x[lo:hi], kernel, orig, new, width).shape, rows, _thread_counts(), budget_s)
    return _record(sweep, seconds, f"up to {rows} mono rows x {seconds:g} s @44.1 kHz (the shard: 256 rows x 30 s); audio seconds "
                   "of ONE channel row each, as the GPU figure's `channel_audio_sec_per_sec`", "port")


def time_cfg4(budget_s: float = 3.0, clips: int = 128, seconds: float = 10.0, seed: int = 1234):
    """MFCC n_mfcc = 40 on `clips` clips of `seconds` s.  Dealt threads give every share its OWN top_db cut-off (the batch-global
    one of the (B, L) input would need a second pass over the shares): a slightly cheaper computation than the config's, stated."""
    from audio_amd import _host
    g = torch.Generator().manual_seed(seed)
    x = (0.5 * torch.randn(clips, int(16000 * seconds), generator=g)).clamp_(-1, 1)
    window = torch.hann_window(400)
    fb = _host.melscale_fbanks(201, 0.0, 8000.0, 80, 16000)
    dct = _host.create_dct(40, 80, "ortho")
    sweep = _dealt(lambda lo, hi: ref.mfcc(x[lo:hi], window, fb, dct, 400, 160).shape, clips, _thread_counts(), budget_s, 2)
    return _record(sweep, seconds, f"up to {clips} clips x {seconds:g} s @16 kHz (the config: 512 clips); the cut-off is taken per "
                   "dealt share, not batch-wide", "port")


def time_cfg5a(budget_s: float = 3.0, batch: int = 64, channels: int = 8, seconds: float = 10.0, seed: int = 1234):
    """4-biquad cascade on (batch, channels, time) @48 kHz; batches dealt to threads."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, channels, int(48000 * seconds), generator=g) - 0.5
    A, B = [], []
    for fc in (8000.0, 6000.0, 4000.0, 3000.0):
        w0 = 2 * math.pi * fc / 48000
        alpha = math.sin(w0) / 2 / 0.707
        A.append([1 + alpha, -2 * math.cos(w0), 1 - alpha])
        B.append([(1 - math.cos(w0)) / 2, 1 - math.cos(w0), (1 - math.cos(w0)) / 2])
    a4, b4 = torch.tensor(A), torch.tensor(B)
    ref_core = reference_lfilter_core()
    kind = "reference" if ref_core is not None else lfilter_core()[1]
    core = (lambda fir, a, y: ref_core(fir, a, y)) if ref_core is not None else None
    sweep = _dealt(lambda lo, hi: biquad_cascade(x[lo:hi], a4, b4, core).shape, batch, _thread_counts(), budget_s)
    return _record(sweep, seconds, f"up to {batch} x {channels} ch x {seconds:g} s @48 kHz (the shard: 32 x 8 ch); audio seconds of the "
                   "multichannel clip, as the GPU figure's `value`", "port",
                   {"core_loop": ("reference: /root/reference/src/libtorchaudio/lfilter.cpp compiled from its sources (oracle/_ref)"
                                  if kind == "reference" else "port: oracle/lfilter_core.c (restatement of lfilter.cpp:17-48)"),
                    "composition": "port of filtering.py:941-1099 (pad + grouped conv1d, core loop, clamp), 4 sequential calls"})


def time_cfg5b(budget_s: float = 3.0, rows: int = 64, seconds: float = 10.0, taps: int = 24000, seed: int = 1234):
    """fftconvolve of `rows` channel rows of `seconds` s @48 kHz with one 24 000-tap response."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(rows, int(48000 * seconds), generator=g) - 0.5
    t = torch.arange(taps) / 48000.0
    rir = torch.randn(1, taps, generator=g) * torch.exp(-t / 0.1) * 0.05
    sweep = _dealt(lambda lo, hi: fftconvolve(x[lo:hi], rir).shape, rows, _thread_counts(), budget_s)
    return _record(sweep, seconds / 8.0, f"up to {rows} channel rows x {seconds:g} s @48 kHz = {rows // 8} clips of 8 channels (the shard: "
                   "256 rows); audio seconds of the multichannel clip, as the GPU figure's `value`", "port")


BY_CONFIG = {"cfg3": time_cfg3, "cfg4": time_cfg4, "cfg5a": time_cfg5a, "cfg5b": time_cfg5b}


def measure(which=None, budget_s: float = 3.0):
    out = {}
    for key, fn in BY_CONFIG.items():
        if which is None or key in which:
            try:
                out[key] = fn(budget_s=budget_s)
            except Exception as e:                       # a baseline must never take the bench line with it
                out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


if __name__ == "__main__":
    import json
    import sys
    print(json.dumps(measure(set(sys.argv[1:]) or None), indent=1))

#!/usr/bin/env python
"""One JSON line per BASELINE.json config (per-GPU shard of the 8-GPU configs), same timing protocol as
bench.py (inputs resident, warm-up to steady clocks, events on the launch stream).  Not the driver's
contract -- that is bench.py (config 2) -- but the same fields, so the other rows of SURVEY.md 8(d) can be
reproduced:   python tools/bench_configs.py [--steps K --warmup W] > profiles/rNN_configs.jsonl"""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_amd.functional as F
import audio_amd.transforms as T

HBM, FP32, F16 = 8000.0, 157.3, 2500.0   # GB/s, TFLOP/s fp32 vector / MFMA, TFLOP/s dense f16 MFMA (MI355X_MICROARCH.md)


RAMP_MS = 60.0   # GPU time of set-up launches per workload before its warm-up counts as done (see timed())


def timed(fn, warmup, steps, ramp_ms=None):
    """ms per call, steady state.  Set-up as in bench.py (--clock-ramp): a workload that starts on a chip that was idle -- or
    was running something else -- runs 8-12 % slow for its first 25-30 ms while power management settles, whatever the
    kernel (cfg5a: 271 us per call over launches 10..60 of a fresh process against 240 us from launch ~110 on; cfg3 611
    against 558 us; profiles/r04_t_clock_ramp_configs.txt).  Ten warm-up launches of a 0.25 ms kernel end inside that ramp,
    so the warm-up repeats until RAMP_MS of GPU time have gone by."""
    ramp_ms = RAMP_MS if ramp_ms is None else ramp_ms
    spent, first = 0.0, True
    while True:
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for _ in range(max(warmup, 1)):
            fn()
        r1.record()
        torch.cuda.synchronize()
        # (round 6: the FIRST batch of warm-up calls does not count.  Its event pair also brackets the host's one-off work --
        # the 0.5 GB output allocation of cfg5b, the tap spectra / fragment builds -- during which the GPU idles: round 5 took
        # those tens of milliseconds for GPU time, left the loop after 5 launches and timed cfg5b / cfg3 on a chip that had not
        # ramped: 0.69 ms reported for a kernel that runs 0.60 ms from its ~100th launch on, profiles/r06_h_fdr_product_vs_lab.txt)
        if not first:
            spent += r0.elapsed_time(r1)
        first = False
        if spent >= ramp_ms:
            break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps   # ms


def measure_configs(dev, steps=200, warmup=100, which=None, gen_seed=1234):
    """Every BASELINE.json config besides the headline one, per-GPU shard, as a list of records:
      {id, workload, ms, clip_seconds_per_launch, value (audio-s/s per GPU), algorithmic_bytes,
       roofline {bound, achieved, peak, unit, frac} [, roofline_matrix_f16, fp32_equivalent, note, ...]}
    bench.py calls this on rank 0 at N = 1, OUTSIDE its timed K steps, and puts the list into the driver-run JSON line under
    "configs" (VERDICT r3 next 2); `python tools/bench_configs.py` prints the same records one per line."""
    g = torch.Generator(device=dev).manual_seed(gen_seed)
    recs = []
    want = set(which) if which else None

    def on(key):
        return want is None or key in want

    def noise(*shape):
        return (0.5 * torch.randn(*shape, device=dev, generator=g)).clamp_(-1, 1)

    def ring(fn, bufs):
        """fn(buffer) over rotating input buffers (more than the 256 MiB Infinity Cache in total), as bench.py does"""
        it = [0]

        def step():
            it[0] += 1
            return fn(bufs[it[0] % len(bufs)])
        return step

    def emit(key, name, ms, clip_seconds, algo_bytes, flops=None, note="", f16_terms=0, channels=1, extra=None):
        rec = {"id": key, "workload": name, "ms": ms, "clock_ramp_ms": RAMP_MS, "clip_seconds_per_launch": clip_seconds,
               "value": clip_seconds / (ms * 1e-3), "unit": "audio-sec/sec (per GPU)", "algorithmic_bytes": algo_bytes,
               "roofline": {"bound": "hbm", "achieved": algo_bytes / ms / 1e6, "peak": HBM, "unit": "GB/s",
                            "frac": algo_bytes / ms / 1e6 / HBM}}
        if channels > 1:
            rec["channel_audio_sec_per_sec"] = channels * clip_seconds / (ms * 1e-3)
        if flops and f16_terms:
            # the kernel runs on the f16 matrix pipe with `f16_terms` MFMA instructions per product (hi/lo-split operands):
            # priced against THAT pipe's dense peak; the fp32 figure is what the same arithmetic would need on fp32 FMAs
            rec["roofline_matrix_f16"] = {"bound": "mfma", "algorithmic_flops": flops, "issued_flops": f16_terms * flops,
                                          "achieved": f16_terms * flops / ms / 1e9, "peak": F16, "unit": "TFLOP/s",
                                          "frac": f16_terms * flops / ms / 1e9 / F16}
            rec["fp32_equivalent"] = {"achieved_TFLOPs": flops / ms / 1e9, "of_fp32_peak": flops / ms / 1e9 / FP32,
                                      "note": "not a roofline of this kernel: no fp32 pipe executes these flops"}
        elif flops:
            rec["roofline_fp32"] = {"algorithmic_flops": flops, "achieved_TFLOPs": flops / ms / 1e9,
                                    "frac": flops / ms / 1e9 / FP32}
        if note:
            rec["note"] = note
        if extra:
            rec.update(extra)
        recs.append(rec)

    with torch.no_grad():
        if on("cfg2") or on("spec"):
            xs = [noise(256, 160000) for _ in range(4)]
            if on("cfg2"):
                mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
                emit("cfg2", "MelSpectrogram n_fft=400 hop=160 n_mels=80, 256 x 10 s @16 kHz (BASELINE configs[1])",
                     timed(ring(mel, xs), 5 * warmup, 5 * steps), 2560.0, 256 * 160000 * 4 + 256 * 1001 * 80 * 4,
                     note="4 input batches rotate")
            if on("spec"):
                sp = T.Spectrogram(n_fft=400, hop_length=160).to(dev)
                emit("spec", "Spectrogram n_fft=400 hop=160 (power 2), 256 x 10 s @16 kHz (configs[0]'s transform at batch size)",
                     timed(ring(sp, xs), 5 * warmup, 5 * steps), 2560.0, 256 * 160000 * 4 + 256 * 1001 * 201 * 4,
                     note="4 input batches rotate")
            del xs
        if on("cfg4") or on("cfg4_per_item"):
            xs = [noise(512, 160000) for _ in range(3)]
            mf = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
            if on("cfg4"):
                ms = timed(ring(mf, xs), warmup, steps)
                rep = mf.fused_report()
                emit("cfg4", "MFCC n_mfcc=40, 512 x 10 s @16 kHz, (B, L) input = one batch-global top_db cut-off (BASELINE configs[3])",
                     ms, 5120.0, 512 * 160000 * 4 + 512 * 1001 * 40 * 4,
                     note="3 input batches rotate",
                     extra={"mfcc_path": rep["path"], "mfcc_fused_setting": repr(mf.fused), "mfcc_decided": rep["decided"],
                            "mfcc_redone_share": rep["redone_share"],
                            "kernels": ("melspec400_kernel<EPI400_MFCC> pass 0 + fix-up launch of the same kernel"
                                        if rep["path"] == "fused" else
                                        "melspec400_kernel<EPI400_MEL_DB> + mfcc_dct_mfma_kernel")})
            if on("cfg4_per_item"):
                x3 = [x[:, None, :] for x in xs]
                emit("cfg4_per_item", "MFCC variant: (B, 1, L) input = per-item cut-offs", timed(ring(mf, x3), warmup, steps),
                     5120.0, 512 * 160000 * 4 + 512 * 1001 * 40 * 4)
                del x3
            del xs
        if on("cfg3"):
            x = noise(128, 2, 1323000)
            rs = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                            rolloff=0.9475937167399596, beta=14.769656459379492).to(dev)
            emit("cfg3", "Resample 44.1k->16k kaiser_best, per-GPU shard 128 x stereo x 30 s of BASELINE configs[2]",
                 timed(lambda: rs(x), 10, max(steps // 6, 10)), 128 * 30.0, x.numel() * 4 + 128 * 2 * 480000 * 4,
                 flops=128 * 2 * 480000 * 373 * 2, f16_terms=3, channels=2,
                 note="binary16 hi/lo-split MFMA kernel (default): 3 f16 MFMA terms per product; 373 effective taps per output")
            del x
        if on("cfg5a") or on("cfg5b"):
            x = torch.rand(32, 8, 480000, device=dev, generator=g) - 0.5
            if on("cfg5a"):
                A, B = [], []
                for fc in (8000.0, 6000.0, 4000.0, 3000.0):
                    w0 = 2 * math.pi * fc / 48000
                    alpha = math.sin(w0) / 2 / 0.707
                    A.append([1 + alpha, -2 * math.cos(w0), 1 - alpha])
                    B.append([(1 - math.cos(w0)) / 2, 1 - math.cos(w0), (1 - math.cos(w0)) / 2])
                a4, b4 = torch.tensor(A, device=dev), torch.tensor(B, device=dev)
                emit("cfg5a", "lfilter: 4-biquad lowpass cascade (8/6/4/3 kHz, Q=0.707) fused, per-GPU shard 32 x 8 ch x 10 s @48 kHz "
                              "of BASELINE configs[4]",
                     timed(lambda: F.biquad_cascade(x, a4, b4), 10, max(steps // 4, 10)), 32 * 10.0, 2 * x.numel() * 4, channels=8)
            if on("cfg5b"):
                t = torch.arange(24000, device=dev) / 48000.0
                g2 = torch.Generator(device=dev).manual_seed(4321)
                rir = torch.randn(1, 1, 24000, device=dev, generator=g2) * torch.exp(-t / 0.1) * 0.05
                emit("cfg5b", "fftconvolve with a 0.5 s RIR (24000 taps), per-GPU shard 32 x 8 ch x 10 s @48 kHz of BASELINE configs[4]",
                     timed(lambda: F.fftconvolve(x, rir), 5, max(steps // 10, 10)), 32 * 10.0,
                     x.numel() * 4 + 32 * 8 * 503999 * 4, channels=8)
            del x
    torch.cuda.empty_cache()
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--only", default="", help="comma-separated config ids (cfg2, spec, cfg4, cfg4_per_item, cfg3, cfg5a, cfg5b)")
    args = ap.parse_args()
    which = [w for w in args.only.split(",") if w] or None
    for rec in measure_configs(torch.device("cuda"), args.steps, args.warmup, which):
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Micro-benchmarks on one MI355X (run through gpurun): per-op kernel time with HIP events.
  python tools/gpu_microbench.py [mel] [spec] [mfcc] [resample] [lfilter] [fftconv]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import audio_amd.functional as F
from audio_amd import _lib
import audio_amd.transforms as T


def timeit(fn, warm=5, iters=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    what = sys.argv[1:] or ["mel"]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    with torch.no_grad():
        if "mel" in what:
            x = (0.5 * torch.randn(256, 160000, device=dev)).clamp_(-1, 1)
            mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
            us = timeit(lambda: mel(x), 10, 100)
            by = 256 * 160000 * 4 + 256 * 1001 * 80 * 4
            print(f"mel400 fast : {us:9.1f} us  {by / us / 1e3:8.1f} GB/s  frac {by / us / 1e3 / 8000:.3f}", flush=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                mel(x)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"  host issue {1e6 * (t1 - t0) / 200:.1f} us/call, issue+drain {1e6 * (t2 - t0) / 200:.1f} us/call", flush=True)
            out = torch.empty(256, 1001, 80, device=dev)
            import ctypes as C
            from audio_amd import _lib
            sp = mel.spectrogram
            x2 = x
            desc = F._stft_desc(x2, 0, sp.window, 400, 160, 2.0, False, True, "reflect", True)
            bands = F._mel_bands(mel.mel_scale.fb, dev)
            L = _lib.lib()
            args = (x2.data_ptr(), F._padded_window(sp.window, 400).data_ptr(), F._twiddles(400, dev).data_ptr(),
                    C.byref(bands.struct), out.data_ptr(), C.byref(desc), _lib.current_stream(dev))
            us = timeit(lambda: L.aamd_melspectrogram_f32(*args), 10, 200)
            print(f"  raw C-ABI launches into ONE output buffer: {us:9.1f} us  frac {by / us / 1e3 / 8000:.3f}", flush=True)
            # A/B in one process: lane order on/off, interleaved (box-to-box variation is +-8 %)
            os.environ["AAMD_MEL400_NO_LANE_ORDER"] = "1"
            mel_b = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
            mel_b(x)
            del os.environ["AAMD_MEL400_NO_LANE_ORDER"]
            for rep in range(3):
                ua = timeit(lambda: mel(x), 5, 100)
                ub = timeit(lambda: mel_b(x), 5, 100)
                print(f"  A/B lane order on {ua:7.1f} us | off {ub:7.1f} us", flush=True)
            for v in os.environ.get("AAMD_VARIANTS", "").split(","):
                if v:
                    os.environ["AAMD_MEL400_VARIANT"] = v
                    us = timeit(lambda: mel(x), 10, 100)
                    print(f"mel400 var {v}: {us:9.1f} us  frac {by / us / 1e3 / 8000:.3f}", flush=True)
            _lib.lib().aamd_set_kernel_policy(1)
            us = timeit(lambda: mel(x), 2, 5)
            _lib.lib().aamd_set_kernel_policy(0)
            print(f"mel generic : {us:9.1f} us")
        if "spec" in what:
            x = (0.5 * torch.randn(256, 160000, device=dev)).clamp_(-1, 1)
            sp = T.Spectrogram(n_fft=400, hop_length=160).to(dev)
            us = timeit(lambda: sp(x), 5, 50)
            by = 256 * 160000 * 4 + 256 * 1001 * 201 * 4
            print(f"spectrogram400 fast: {us:9.1f} us  {by / us / 1e3:8.1f} GB/s  frac {by / us / 1e3 / 8000:.3f}")
            _lib.lib().aamd_set_kernel_policy(1)
            print(f"spectrogram generic 400/160: {timeit(lambda: sp(x), 2, 5):9.1f} us")
            _lib.lib().aamd_set_kernel_policy(0)
        if "mfcc" in what:
            x = (0.5 * torch.randn(512, 160000, device=dev)).clamp_(-1, 1)
            m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
            us = timeit(lambda: m(x), 3, 20)
            by = 512 * 160000 * 4 + 512 * 1001 * 40 * 4
            print(f"mfcc b=512 (2-D, one cut-off): {us:9.1f} us  algorithmic {by / us / 1e3:8.1f} GB/s  frac {by / us / 1e3 / 8000:.3f}")
            x3 = x[:, None, :]
            us = timeit(lambda: m(x3), 3, 20)
            print(f"mfcc b=512 (3-D, per-item cut-off): {us:9.1f} us")
        if "resample" in what:
            x = (0.5 * torch.randn(16, 2, 1323000, device=dev)).clamp_(-1, 1)
            r = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                           rolloff=0.9475937167399596, beta=14.769656459379492).to(dev)
            x = (0.5 * torch.randn(128, 2, 1323000, device=dev)).clamp_(-1, 1)     # cfg3 per-GPU shard (1/8)
            us = timeit(lambda: r(x), 2, 10)
            fl = 128 * 2 * 480000 * 373 * 2
            by = x.numel() * 4 + 128 * 2 * 480000 * 4
            print(f"resample kaiser_best 128x2x30s (MFMA): {us:9.1f} us  ({128 * 30 / (us * 1e-6):.0f} clip-s/s) "
                  f"{fl / us / 1e6:.1f} algorithmic TFLOP/s (frac {fl / us / 1e6 / 157.3:.3f})  {by / us / 1e3:.0f} GB/s")
            _lib.lib().aamd_set_kernel_policy(1)
            us = timeit(lambda: r(x[:16]), 1, 3)
            _lib.lib().aamd_set_kernel_policy(0)
            print(f"resample kaiser_best 16x2x30s scalar kernel: {us:9.1f} us")
        if "lfilter" in what:
            x = (torch.rand(32, 8, 480000, device=dev) - 0.5)
            a = torch.tensor([1.0, -1.2, 0.5], device=dev)
            b = torch.tensor([0.1, 0.2, 0.1], device=dev)
            us = timeit(lambda: F.lfilter(x, a, b), 2, 10)
            print(f"lfilter biquad 32x8x10s@48k (wave kernel): {us:9.1f} us  {2 * x.numel() * 4 / us / 1e3:.1f} GB/s")
            a4 = torch.tensor([[1.0, -1.2, 0.5], [1.0, -0.9, 0.3], [1.0, -0.5, 0.2], [1.0, -0.2, 0.1]], device=dev)
            b4 = torch.tensor([[0.1, 0.2, 0.1], [0.2, 0.3, 0.2], [0.3, 0.2, 0.1], [0.2, 0.1, 0.05]], device=dev)
            us = timeit(lambda: F.biquad_cascade(x, a4, b4), 2, 10)
            print(f"4-biquad cascade fused 32x8x10s@48k (cfg5a shard): {us:9.1f} us  {2 * x.numel() * 4 / us / 1e3:.1f} GB/s "
                  f"(frac {2 * x.numel() * 4 / us / 1e3 / 8000:.3f})")
            _lib.lib().aamd_set_kernel_policy(1)
            us = timeit(lambda: F.biquad_cascade(x, a4, b4), 1, 3)
            _lib.lib().aamd_set_kernel_policy(0)
            print(f"  same, workgroup-scan kernel: {us:9.1f} us")
        if "fftconv" in what:
            x = torch.rand(32, 8, 480000, device=dev) - 0.5                       # cfg5b per-GPU shard (1/8)
            y = torch.randn(1, 1, 24000, device=dev) * 0.05
            us = timeit(lambda: F.fftconvolve(x, y), 2, 10)
            by = x.numel() * 4 + 32 * 8 * 503999 * 4
            print(f"fftconvolve 32x8x10s@48k * 24000-tap RIR (overlap-save LDS FFT): {us:9.1f} us  "
                  f"algorithmic {by / us / 1e3:.0f} GB/s (frac {by / us / 1e3 / 8000:.3f})")
            x1, y1 = torch.randn(4, 8, 48000, device=dev), torch.randn(1, 1, 2400, device=dev)
            print(f"fftconvolve 4x8x1s * 2400 taps: {timeit(lambda: F.fftconvolve(x1, y1), 1, 3):9.1f} us")
            _lib.lib().aamd_set_kernel_policy(1)
            print(f"  same, time-domain kernel: {timeit(lambda: F.fftconvolve(x1, y1), 1, 3):9.1f} us")
            _lib.lib().aamd_set_kernel_policy(0)
    if "standalone" in what:
        x = (0.5 * torch.randn(256, 160000, device=dev)).clamp_(-1, 1)
        with torch.no_grad():
            spec = T.Spectrogram(n_fft=400, hop_length=160).to(dev)(x)
            ms = T.MelScale(n_mels=80, sample_rate=16000, n_stft=201).to(dev)
            db = T.AmplitudeToDB(top_db=80.0).to(dev)
            db0 = T.AmplitudeToDB(top_db=None).to(dev)
            mel = ms(spec)
            print(f"MelScale 256x201x1001 -> 80 (288 MB):          {timeit(lambda: ms(spec), 5, 30):8.1f} us")
            print(f"AmplitudeToDB top_db=80, 256x80x1001 (246 MB): {timeit(lambda: db(mel), 5, 30):8.1f} us")
            print(f"AmplitudeToDB top_db=None (164 MB):            {timeit(lambda: db0(mel), 5, 30):8.1f} us")
            gl = T.GriffinLim(n_fft=400, n_iter=32, rand_init=False, length=160000).to(dev)
            p = T.Spectrogram(n_fft=400).to(dev)(x[:64])
            print(f"GriffinLim 64x10s, n_fft=400 hop=200, 32 iterations:   {timeit(lambda: gl(p), 1, 3):8.1f} us")
            ps = T.PitchShift(16000, 4).to(dev)
            print(f"PitchShift 64x10s (+4 semitones, n_fft=512):           {timeit(lambda: ps(x[:64]), 1, 3):8.1f} us")
            import audio_amd.compliance.kaldi as K
            one = x[:1] * 32768
            print(f"kaldi.fbank 1x10s, 80 bins (one utterance per call):   {timeit(lambda: K.fbank(one, num_mel_bins=80), 3, 20):8.1f} us")
    if "rnnt" in what:
        from audio_amd.pipelines import GAIN, RNNTFeatureExtractor, piecewise_linear_log
        x = (0.1 * torch.randn(256, 160000, device=dev)).clamp_(-1, 1)
        stats = {"mean": (10 + 3 * torch.randn(80)).tolist(), "invstddev": (0.2 + torch.rand(80)).tolist()}
        fe = RNNTFeatureExtractor(stats).to(dev)
        mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, n_mels=80, hop_length=160).to(dev)
        with torch.no_grad():
            us = timeit(lambda: fe(x), 10, 100)
            print(f"RNN-T features 256x10s, fused epilogue (one kernel + 82 KB memset): {us:9.1f} us")

            def unfused():
                m = mel(x).transpose(-1, -2)
                y = (piecewise_linear_log(m * GAIN) - fe.mean) * fe.invstddev
                return torch.nn.functional.pad(y, (0, 0, 0, 4))
            us = timeit(unfused, 10, 100)
            print(f"  same steps unfused (HIP mel kernel + torch element-wise chain):   {us:9.1f} us")
            pcm = (x * 32767).to(torch.int16)
            us = timeit(lambda: fe(pcm), 10, 100)
            print(f"RNN-T features from int16 PCM, one kernel (82 MB in, 82 MB out):     {us:9.1f} us")
            us = timeit(lambda: fe(pcm.float() * (1.0 / 32768.0)), 10, 100)
            print(f"  int16 -> float pass + fused kernel:                                {us:9.1f} us")
    if "istft" in what:
        x = (0.5 * torch.randn(256, 160000, device=dev)).clamp_(-1, 1)
        with torch.no_grad():
            sp = T.Spectrogram(n_fft=400, hop_length=160, power=None).to(dev)
            inv = T.InverseSpectrogram(n_fft=400, hop_length=160).to(dev)
            X = sp(x)
            us = timeit(lambda: inv(X, 160000), 3, 20)
            by = X.numel() * 8 + x.numel() * 4
            print(f"InverseSpectrogram 256x10s n_fft=400 hop=160: {us:9.1f} us  algorithmic {by / us / 1e3:.0f} GB/s "
                  f"(frac {by / us / 1e3 / 8000:.3f})")
            sp2 = T.Spectrogram(n_fft=1024, hop_length=256, power=None).to(dev)
            inv2 = T.InverseSpectrogram(n_fft=1024, hop_length=256).to(dev)
            X2 = sp2(x)
            us = timeit(lambda: inv2(X2, 160000), 3, 20)
            by = X2.numel() * 8 + x.numel() * 4
            print(f"InverseSpectrogram 256x10s n_fft=1024 hop=256: {us:9.1f} us  algorithmic {by / us / 1e3:.0f} GB/s")
        mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
        xg = x.clone().requires_grad_()
        r = torch.randn(256, 80, 1001, device=dev)

        def fwd_bwd():
            xg.grad = None
            (mel(xg) * r).sum().backward()
        us = timeit(fwd_bwd, 3, 20)
        print(f"MelSpectrogram fwd+bwd 256x10s (fused fwd; fb^T, cotangent, STFT adjoint): {us:9.1f} us")
        mt = mel_torch_composition(mel)

        def fwd_bwd_t():
            xg.grad = None
            (mt(xg) * r).sum().backward()
        us = timeit(fwd_bwd_t, 3, 20)
        print(f"  same through torch.stft (ATen/rocFFT) autograd:                    {us:9.1f} us")


def mel_torch_composition(mel):
    w, fb = mel.spectrogram.window, mel.mel_scale.fb

    def f(x):
        X = torch.stft(x, 400, 160, 400, w, True, "reflect", False, True, return_complex=True)
        return torch.matmul(X.abs().pow(2.0).transpose(-1, -2), fb).transpose(-1, -2)
    return f


if __name__ == "__main__":
    main()

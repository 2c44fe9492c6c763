#!/usr/bin/env python
"""A/B lab for the headline kernel (tools only).

  build (here, no GPU):   python tools/mel400_lab.py build NAME[:-DFLAG[,-DFLAG...]] ...
  run (GPU box):          python tools/mel400_lab.py run NAME NAME ... [--launches 300] [--rounds 4]
  MFCC epilogue:          python tools/mel400_lab.py mfcc NAME NAME ... [--launches 200] [--rounds 4]   (cfg4 batch, both passes)

Every variant is tools/lab/mel400_lab.hip (= csrc/melspec400.h alone) compiled into tools/lab/_build/libm400_NAME.so with
its -D switches.  `run` launches the cfg2 batch (256 x 160 000) with buffers rotating over > 256 MiB, variants interleaved
round by round, checks every variant's output against the first one (max |diff| relative to the peak) and prints the
average launch time per variant and round."""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "lab", "_build")
SRC = os.path.join(HERE, "lab", "mel400_lab.hip")


def so_path(name):
    return os.path.join(OUT, "libm400_%s.so" % name)


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        fl = []
        for f in flags.split(","):
            if f.startswith("-mllvm="):
                fl += ["-mllvm", f[len("-mllvm="):]]
            elif f:
                fl.append(f)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
               "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage"] + fl + [SRC, "-o", so_path(name)]
        log = open(os.path.join(OUT, name + ".log"), "w")
        procs.append((name, subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT), log))
    for name, p, log in procs:
        rc = p.wait()
        log.close()
        txt = open(os.path.join(OUT, name + ".log")).read()
        res, on = {}, False
        for line in txt.splitlines():
            if "Function Name" in line:
                on = "melspec400_kernel" in line
            elif on:
                for key in ("VGPRs:", "AGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:"):
                    if " " + key in line and "Spill" not in line:
                        res[key.rstrip(":")] = line.split(key)[1].split("[")[0].strip()
        print(name, "rc", rc, res)
        if rc != 0:
            print(txt[-3000:])


def run(names, launches, rounds, rows=256, seconds=10.0, wide=0, blocks=0):
    import numpy as np
    import torch
    import audio_amd.functional as F
    import audio_amd.transforms as T
    from audio_amd import _lib
    dev = torch.device("cuda")
    length = int(16000 * seconds)
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
    bands = F._mel_bands(mel.mel_scale.fb, dev)
    window = F._padded_window(mel.spectrogram.window, 400)
    tw = F._twiddles(400, dev)
    n_frames = 1 + length // 160
    g = torch.Generator(device="cuda").manual_seed(0)
    xs = [(torch.rand(rows, length, device=dev, generator=g) - 0.5) for _ in range(4)]
    outs = [torch.empty(rows, n_frames, 80, device=dev) for _ in range(5)]
    libs = {}
    for n in names:
        L = C.CDLL(so_path(n))
        L.lab_mel400.argtypes = [C.c_void_p] * 5 + [C.c_int64] * 3 + [C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
        libs[n] = L
    stream = _lib.current_stream(dev)
    dbg = torch.zeros((4 + 8) * 12 * 256, dtype=torch.int64, device=dev)
    for L in libs.values():
        L.lab_set_debug.argtypes = [C.c_void_p]
        L.lab_set_debug(dbg.data_ptr())

    wide_of = {id(L): (1 if n.endswith("_w") else wide) for n, L in libs.items()}   # variants named *_w: the wide store path

    def launch(L, i):
        x, o = xs[i % 4], outs[i % 5]
        rc = L.lab_mel400(x.data_ptr(), window.data_ptr(), tw.data_ptr(), C.byref(bands.struct), o.data_ptr(), rows, length,
                          length, n_frames, 1.0, wide_of[id(L)], blocks, stream)
        assert rc == 0, rc
        return o

    # the product's answer, and every variant against it
    ref = mel(xs[0]).transpose(-1, -2).contiguous() if False else None
    base = launch(libs[names[0]], 0).clone()
    torch.cuda.synchronize()
    prod = mel(xs[0])
    prod_fm = prod.transpose(-1, -2)
    peak = float(prod_fm.abs().max())
    print(json.dumps({"check": names[0], "vs_product_peak_rel": float((base - prod_fm).abs().max()) / peak}))
    for n in names[1:]:
        o = launch(libs[n], 0)
        torch.cuda.synchronize()
        d = float((o - base).abs().max()) / peak
        print(json.dumps({"check": n, "vs_first_peak_rel": d, "bit_equal": bool(torch.equal(o, base))}))
    # clock ramp
    for i in range(400):
        launch(libs[names[0]], i)
    torch.cuda.synchronize()
    def clock_report(n):
        bits = libs[n].lab_info(C.byref(C.c_int()), C.byref(C.c_int()))
        if bits & 8388608:      # per-phase shader cycles, summed per wave over its tiles
            ph = dbg.cpu().numpy()[4 * 12 * 256:].reshape(-1, 8).astype(float)
            ph = ph[ph[:, 6] > 0]
            per_tile = ph[:, :6] / ph[:, 6:7]
            names6 = ["A: claim+gather+window+DFT+twiddle+transposed writes", "B1: column reads", "B: DMA issue+DFT+exchange",
                      "B2: power+P rows", "C: band reduction", "stores"]
            print(json.dumps({"variant": n, "waves": int(len(ph)), "tiles_per_wave_mean": round(float(ph[:, 6].mean()), 2),
                              "cycles_per_tile_by_phase": {k: round(float(v), 0) for k, v in zip(names6, per_tile.mean(axis=0))},
                              "cycles_per_tile_total": round(float(per_tile.sum(axis=1).mean()), 0)}))
        if not (bits & 1048576):
            return
        if True:
            d = dbg.cpu().numpy()[:4 * 12 * 256].reshape(-1, 4)
            d = d[d[:, 1] != 0]
            cyc, wall = (d[:, 2] - d[:, 0]).astype(float), (d[:, 3] - d[:, 1]).astype(float)
            full = dbg.cpu().numpy()[:4 * 12 * 256].reshape(-1, 12, 4)          # [block][wave][4]
            t00 = full[:, :, 1][full[:, :, 1] != 0].min()
            blk_end = (full[:, :, 3].max(axis=1) - t00) / 100.0
            blk_start = (full[:, :, 1].min(axis=1) - t00) / 100.0
            wave_end = (full[:, :, 3] - t00) / 100.0
            by_xcd = [round(float(blk_end[x::8].mean()), 2) for x in range(8)]
            print(json.dumps({"variant": n, "block_end_us_by_xcd_mean": by_xcd,
                              "block_end_us_min_med_max": [round(float(np.min(blk_end)), 2), round(float(np.median(blk_end)), 2), round(float(np.max(blk_end)), 2)],
                              "block_start_us_max": round(float(blk_start.max()), 2),
                              "within_block_wave_end_spread_us_mean": round(float((wave_end.max(axis=1) - wave_end.min(axis=1)).mean()), 2)}))
            print(json.dumps({"variant": n, "waves": int(len(d)), "wave_us_mean": round(float(wall.mean()) / 100.0, 2),
                              "wave_us_max": round(float(wall.max()) / 100.0, 2),
                              "span_us": round(float(d[:, 3].max() - d[:, 1].min()) / 100.0, 2),
                              "shader_GHz": round(float((cyc / wall).mean()) * 0.1, 3)}))
    res = {n: [] for n in names}
    for r in range(rounds):
        for n in names:
            L = libs[n]
            for i in range(20):
                launch(L, i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(launches):
                launch(L, i)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / launches * 1e3)
            if r == rounds - 1:
                clock_report(n)
    for n in names:
        v = res[n]
        print(json.dumps({"variant": n, "us_per_launch": [round(t, 2) for t in v], "best": round(min(v), 2),
                          "mean": round(sum(v) / len(v), 2), "frac_hbm": round(245841920 / (min(v) * 1e-6) / 8e12, 4)}))


def run_mfcc(names, launches, rounds, rows=512, seconds=10.0):
    """The one-kernel MFCC (lab_mfcc400: pass 0 + fix-up launch, as F._mfcc_fused issues them) on the cfg4 batch: every variant
    against the product's two-kernel result (max |diff| in dB-weighted coefficient units) and against the first variant, then
    interleaved timings of the two launches together."""
    import torch
    import audio_amd.functional as F
    import audio_amd.transforms as T
    from audio_amd import _lib
    dev = torch.device("cuda")
    length = int(16000 * seconds)
    m = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
    m.fused = False
    mel = m.MelSpectrogram
    bands = F._mel_bands(mel.mel_scale.fb, dev)
    window = F._padded_window(mel.spectrogram.window, 400)
    tw = F._twiddles(400, dev)
    dct = m.dct_mat.detach().to(torch.float32).contiguous()
    n_frames = 1 + length // 160
    n_tiles = rows * ((n_frames + 5) // 6)
    g = torch.Generator(device="cuda").manual_seed(1234)
    xs = [(0.5 * torch.randn(rows, length, device=dev, generator=g)).clamp_(-1, 1) for _ in range(3)]
    outs = [torch.empty(rows, n_frames, 40, device=dev) for _ in range(3)]
    tile_min = torch.empty(n_tiles, device=dev)
    tile_list = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    gpool = torch.full((1 << 16,), float("-inf"), device=dev)
    gi = [0]
    stream = _lib.current_stream(dev)
    libs, frags, labsw = {}, {}, {}
    for n in names:
        base_name, _, sw = n.partition("@")      # NAME@K: library NAME with the epilogue's lab switch K (needs -DLAB_MFCC_BITS=524288)
        labsw[n] = int(sw or 0)
        L = C.CDLL(so_path(base_name))
        L.lab_mfcc400.argtypes = [C.c_void_p] * 5 + [C.c_int64] * 3 + [C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
        L.lab_mfcc_frag_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        fr = torch.empty(L.lab_mfcc_frag_floats(), device=dev)
        assert L.lab_mfcc_frag_build(dct.data_ptr(), 80, 40, fr.data_ptr(), stream) == 0
        libs[n], frags[n] = L, fr

    def launch(n, i, passes=(0, 1)):
        L = libs[n]
        x, o = xs[i % 3], outs[i % 3]
        gm = gpool[gi[0]:gi[0] + 1]
        gi[0] = (gi[0] + 1) % gpool.numel()
        f = _lib.MfccFused(frags[n].data_ptr(), 40, 0, 10.0, 1e-10, 0.0, 80.0, gm.data_ptr(), rows, tile_min.data_ptr(),
                           count.data_ptr(), tile_list.data_ptr())
        for ps in passes:
            f.pass_ = ps
            rc = L.lab_mfcc400(x.data_ptr(), window.data_ptr(), tw.data_ptr(), C.byref(bands.struct), o.data_ptr(), rows, length,
                               length, n_frames, 1.0, C.byref(f), labsw[n], stream)
            assert rc == 0, rc
        return o

    ref = m(xs[0]).transpose(-1, -2).contiguous()       # the exact two-kernel path
    torch.cuda.synchronize()
    first = None
    for n in names:
        o = launch(n, 0).clone()
        torch.cuda.synchronize()
        rec = {"check": n, "max_abs_vs_two_kernel": float((o - ref).abs().max()), "redone_tiles": int(count.item())}
        if first is None:
            first = o
        else:
            rec["max_abs_vs_first"] = float((o - first).abs().max())
            rec["bit_equal_first"] = bool(torch.equal(o, first))
        print(json.dumps(rec), flush=True)
    # a batch that clamps: the last 20 % of every clip silent -> the fix-up pass redoes a fifth of the tiles
    xz = xs[0].clone()
    xz[:, int(0.8 * length):] = 0.0
    keep = xs[0]
    xs[0] = xz
    refz = m(xz).transpose(-1, -2).contiguous()
    for n in names:
        o = launch(n, 0)
        torch.cuda.synchronize()
        print(json.dumps({"check_clamped": n, "max_abs_vs_two_kernel": float((o - refz).abs().max()),
                          "redone_tiles": int(count.item())}), flush=True)
    xs[0] = keep
    gpool.fill_(float("-inf"))
    for i in range(200):
        launch(names[0], i)
    torch.cuda.synchronize()
    res = {n: [] for n in names}
    res0 = {n: [] for n in names}
    for r in range(rounds):
        for passes, acc in (((0, 1), res), ((0,), res0)):
            for n in names:
                gpool.fill_(float("-inf"))
                for i in range(20):
                    launch(n, i, passes)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(launches):
                    launch(n, i, passes)
                e1.record()
                torch.cuda.synchronize()
                acc[n].append(e0.elapsed_time(e1) / launches * 1e3)
    for n in names:
        v, v0 = res[n], res0[n]
        print(json.dumps({"variant": n, "us_per_call_both_passes": [round(t, 1) for t in v], "best": round(min(v), 1),
                          "pass0_only": [round(t, 1) for t in v0], "best_pass0": round(min(v0), 1),
                          "frac_hbm": round(409661440 / (min(v) * 1e-6) / 8e12, 4)}), flush=True)


def run_spec(names, launches, rounds, rows=256, seconds=10.0):
    """The Spectrogram epilogue (lab_spec400, power 2) on the cfg2 batch: variants against the product, interleaved timings."""
    import torch
    import audio_amd.functional as F
    import audio_amd.transforms as T
    from audio_amd import _lib
    dev = torch.device("cuda")
    length = int(16000 * seconds)
    sp = T.Spectrogram(n_fft=400, hop_length=160).to(dev)
    window = F._padded_window(sp.window, 400)
    tw = F._twiddles(400, dev)
    n_frames = 1 + length // 160
    g = torch.Generator(device="cuda").manual_seed(0)
    xs = [(torch.rand(rows, length, device=dev, generator=g) - 0.5) for _ in range(4)]
    outs = [torch.empty(rows, n_frames, 201, device=dev) for _ in range(3)]
    stream = _lib.current_stream(dev)
    libs = {}
    for n in names:
        L = C.CDLL(so_path(n))
        L.lab_spec400.argtypes = [C.c_void_p] * 4 + [C.c_int64] * 3 + [C.c_int, C.c_float, C.c_float, C.c_void_p]
        libs[n] = L

    def launch(n, i):
        x, o = xs[i % 4], outs[i % 3]
        rc = libs[n].lab_spec400(x.data_ptr(), window.data_ptr(), tw.data_ptr(), o.data_ptr(), rows, length, length, n_frames, 1.0, 2.0, stream)
        assert rc == 0, rc
        return o

    ref = sp(xs[0]).transpose(-1, -2)
    peak = float(ref.abs().max())
    for n in names:
        o = launch(n, 0)
        torch.cuda.synchronize()
        print(json.dumps({"check": n, "vs_product_peak_rel": float((o - ref).abs().max()) / peak, "bit_equal": bool(torch.equal(o, ref))}), flush=True)
    for i in range(400):
        launch(names[0], i)
    torch.cuda.synchronize()
    res = {n: [] for n in names}
    for r in range(rounds):
        for n in names:
            for i in range(20):
                launch(n, i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(launches):
                launch(n, i)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / launches * 1e3)
    algo = rows * length * 4 + rows * n_frames * 201 * 4
    for n in names:
        v = res[n]
        print(json.dumps({"variant": n, "us_per_launch": [round(t, 2) for t in v], "best": round(min(v), 2),
                          "frac_hbm": round(algo / (min(v) * 1e-6) / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        args = sys.argv[2:]
        launches, rounds, wide, blocks = 300, 4, 0, 0
        names = []
        i = 0
        while i < len(args):
            if args[i] == "--launches":
                launches = int(args[i + 1]); i += 2
            elif args[i] == "--rounds":
                rounds = int(args[i + 1]); i += 2
            elif args[i] == "--wide":
                wide = int(args[i + 1]); i += 2
            elif args[i] == "--blocks":
                blocks = int(args[i + 1]); i += 2
            else:
                names.append(args[i]); i += 1
        if sys.argv[1] == "mfcc":
            run_mfcc(names, launches, rounds)
        elif sys.argv[1] == "spec":
            run_spec(names, launches, rounds)
        else:
            run(names, launches, rounds, wide=wide, blocks=blocks)

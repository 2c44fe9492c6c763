#!/usr/bin/env python
"""A/B lab for the fused biquad cascade (cfg5a shard: 256 sequences x 480 000 samples, 4 stages, clamp after each), tools only.

  build (here, no GPU):   python tools/lfw_ab.py build NAME[:FLAG[,FLAG...]] ...      FLAG = -DX=Y or -mllvm=... (= -mllvm ...)
  run (GPU box):          python tools/lfw_ab.py run NAME NAME ... [--launches 40] [--rounds 3]

Every variant is tools/lab/lfw_lab.hip (= csrc/lfilter_wave.h alone) compiled into tools/lab/_build/liblfw_NAME.so.  `run`
rotates over 3 input / output buffer pairs (> 256 MiB), interleaves the variants round by round, checks every variant
against the float64 scipy cascade on one row and against the first variant, and prints us per launch."""
import ctypes as C
import json
import math
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "lab", "_build")
SRC = os.path.join(HERE, "lab", "lfw_lab.hip")


def so_path(name):
    return os.path.join(OUT, "liblfw_%s.so" % name)


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
        log = open(os.path.join(OUT, "lfw_" + name + ".log"), "w")
        procs.append((name, subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT), log))
    for name, p, log in procs:
        rc = p.wait()
        log.close()
        txt = open(os.path.join(OUT, "lfw_" + name + ".log")).read()
        res, on = {}, False
        for line in txt.splitlines():
            if "Function Name" in line:
                on = "mover" in line
            elif on:
                for key in ("VGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:"):
                    if " " + key in line and "Spill" not in line:
                        res[key.rstrip(":")] = line.split(key)[1].split("[")[0].strip()
        print(name, "rc", rc, res)
        if rc != 0:
            print(txt[-3000:])


def run(names, launches, rounds):
    import numpy as np
    import scipy.signal as ss
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    ring = 3
    xs = [torch.rand(256, 480000, device=dev, generator=g) - 0.5 for _ in range(ring)]
    ys = [torch.empty_like(x) for x in xs]
    A, B = [], []
    for fc in (8000.0, 6000.0, 4000.0, 3000.0):
        w0 = 2 * math.pi * fc / 48000
        alpha = math.sin(w0) / 2 / 0.707
        A.append([1 + alpha, -2 * math.cos(w0), 1 - alpha])
        B.append([(1 - math.cos(w0)) / 2, 1 - math.cos(w0), (1 - math.cos(w0)) / 2])
    a4 = torch.tensor(A, device=dev).contiguous()
    b4 = torch.tensor(B, device=dev).contiguous()
    libs = {}
    for n in names:
        L = C.CDLL(so_path(n))
        L.lab_lfw_mover.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_int64] + [C.c_int] * 4 + [C.c_void_p]
        libs[n] = L
    stream = torch.cuda.current_stream().cuda_stream

    def launch(n, i):
        rc = libs[n].lab_lfw_mover(xs[i % ring].data_ptr(), a4.data_ptr(), b4.data_ptr(), ys[i % ring].data_ptr(), 256, 1,
                                   480000, 3, 1, 4, 1, stream)
        assert rc == 0, (n, rc)
    # the float64 cascade of row 0 (clamp after every stage)
    ref = xs[0][0].double().cpu().numpy()
    for k in range(4):
        ref = np.clip(ss.lfilter(np.array(B[k]) / A[k][0], np.array(A[k]) / A[k][0], ref), -1, 1)
    first = None
    for n in names:
        launch(n, 0)
        torch.cuda.synchronize()
        got = ys[0].clone()
        err = float(np.abs(got[0].double().cpu().numpy() - ref).max())
        d = 0.0 if first is None else float((got - first).abs().max())
        if first is None:
            first = got
        print(json.dumps({"check": n, "max_abs_err_vs_float64_row0": err,
                          "max_abs_diff_vs_first": d}), flush=True)
    res = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            for i in range(5):
                launch(n, i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(launches):
                launch(n, i)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / launches * 1e3)
    for n in names:
        best = min(res[n])
        print(json.dumps({"variant": n, "us_per_launch": [round(v, 1) for v in res[n]], "best": round(best, 1),
                          "frac_hbm": round(2 * 256 * 480000 * 4 / (best * 1e-6) / 8e12, 4)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "build":
        build(sys.argv[2:])
    elif len(sys.argv) >= 3 and sys.argv[1] == "run":
        args = sys.argv[2:]
        launches, rounds, names = 40, 3, []
        i = 0
        while i < len(args):
            if args[i] == "--launches":
                launches = int(args[i + 1]); i += 2
            elif args[i] == "--rounds":
                rounds = int(args[i + 1]); i += 2
            else:
                names.append(args[i]); i += 1
        run(names, launches, rounds)
    else:
        print(__doc__)

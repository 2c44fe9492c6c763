#!/usr/bin/env python
"""A/B lab for the real-block fftconvolve kernel (tools only).

  build (here, no GPU):   python tools/fdr_lab.py build NAME[:-DFLAG[,-DFLAG...]] ...
  run (GPU box):          python tools/fdr_lab.py run NAME NAME ... [--launches 20] [--rounds 4] [--taps 24000]

Every variant is tools/lab/fdr_lab.hip (= csrc/fftconv_fdr.h alone) compiled into tools/lab/_build/libfdr_NAME.so with its -D
switches.  `run` convolves the BASELINE config-5 shard (256 rows x 480 000 samples, one 24 000-tap response) with the tap
spectra prepared once, variants interleaved round by round, checks every variant against the first one and the first one
against the product (F.fftconvolve), and prints the average time per launch."""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "lab", "_build")
SRC = os.path.join(HERE, "lab", "fdr_lab.hip")


def so_path(name):
    return os.path.join(OUT, "libfdr_%s.so" % name)


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        fl = [f for f in flags.split(",") if f]
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
               "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage"] + fl + [SRC, "-o", so_path(name)]
        log = open(os.path.join(OUT, "fdr_" + name + ".log"), "w")
        procs.append((name, subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT), log))
    for name, p, log in procs:
        rc = p.wait()
        log.close()
        txt = open(os.path.join(OUT, "fdr_" + name + ".log")).read()
        res, fn = {}, None
        for line in txt.splitlines():
            if "Function Name" in line:
                fn = line.split("Function Name:")[1].strip()
                fn = "NP%s" % fn.split("ILi")[1][0] if "delay_line_kernel" in fn else None
            elif fn:
                for key in ("VGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:"):
                    if " " + key in line and "Spill" not in line:
                        res.setdefault(fn, {})[key.rstrip(":")] = line.split(key)[1].split("[")[0].strip()
        print(name, "rc", rc, json.dumps(res))
        if rc != 0:
            print(txt[-3000:])


def run(names, launches, rounds, taps, rows=256, nx=480000):
    import torch
    import audio_amd.functional as F
    from audio_amd import _lib
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    xs = [(torch.rand(rows, nx, device=dev, generator=g) - 0.5) for _ in range(3)]
    h = torch.randn(1, taps, device=dev, generator=g) * 0.01
    n_out = nx + taps - 1
    outs = [torch.empty(rows, n_out, device=dev) for _ in range(3)]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    stream = _lib.current_stream(dev)
    libs, wss = {}, {}
    for n in names:
        L = C.CDLL(so_path(n))
        L.lab_fdr_workspace.restype = C.c_int64
        L.lab_fdr_workspace.argtypes = [C.c_int64, C.c_int64]
        L.lab_fdr.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        libs[n] = L
        nb = L.lab_fdr_workspace(1, taps)
        ws = torch.zeros((nb + 8 * rows) // 4 + 2, dtype=torch.float32, device=dev)     # (tail: the all-zero tap-row map)
        wss[n] = ws
        rc = L.lab_fdr(xs[0].data_ptr(), h.data_ptr(), outs[0].data_ptr(), rows, 1, nx, taps, ws.data_ptr(), 1, cus, stream)
        assert rc == 0, rc
    torch.cuda.synchronize()

    def launch(n, i):
        x, o = xs[i % 3], outs[i % 3]
        rc = libs[n].lab_fdr(x.data_ptr(), h.data_ptr(), o.data_ptr(), rows, 1, nx, taps, wss[n].data_ptr(), 2, cus, stream)
        assert rc == 0, rc
        return o

    base = launch(names[0], 0).clone()
    torch.cuda.synchronize()
    prod = F.fftconvolve(xs[0], h)
    peak = float(prod.abs().max())
    print(json.dumps({"check": names[0], "vs_product_peak_rel": float((base - prod).abs().max()) / peak,
                      "bit_equal_product": bool(torch.equal(base, prod))}))
    del prod
    for n in names[1:]:
        o = launch(n, 0)
        torch.cuda.synchronize()
        print(json.dumps({"check": n, "vs_first_peak_rel": float((o - base).abs().max()) / peak,
                          "bit_equal": bool(torch.equal(o, base))}))
    for i in range(80):                      # clock ramp: ~60 ms of GPU time
        launch(names[0], i)
    torch.cuda.synchronize()
    res = {n: [] for n in names}
    for r in range(rounds):
        for n in names:
            for i in range(3):
                launch(n, i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(launches):
                launch(n, i)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / launches)
    # a race shows up as run-to-run differences: every variant again on the first input, 6 times, against its own first answer
    for n in names:
        first = launch(n, 0).clone()
        same = True
        for _ in range(6):
            for i in (1, 2, 0):
                o = launch(n, i)
            same = same and bool(torch.equal(o, first))
        torch.cuda.synchronize()
        print(json.dumps({"repeat_check": n, "bit_stable": same,
                          "vs_first_variant_peak_rel": float((first - base).abs().max()) / peak}))
    alg = 4.0 * rows * (nx + n_out)
    for n in names:
        v = res[n]
        print(json.dumps({"variant": n, "taps": taps, "ms_per_launch": [round(t, 4) for t in v], "best": round(min(v), 4),
                          "mean": round(sum(v) / len(v), 4), "frac_hbm_best": round(alg / (min(v) * 1e-3) / 8e12, 4)}))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        args = sys.argv[2:]
        launches, rounds, taps = 20, 4, 24000
        names = []
        i = 0
        while i < len(args):
            if args[i] == "--launches":
                launches = int(args[i + 1]); i += 2
            elif args[i] == "--rounds":
                rounds = int(args[i + 1]); i += 2
            elif args[i] == "--taps":
                taps = int(args[i + 1]); i += 2
            else:
                names.append(args[i]); i += 1
        run(names, launches, rounds, taps)

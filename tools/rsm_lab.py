#!/usr/bin/env python
"""A/B lab for the binary16-split resampler (tools only).

  build (here, no GPU):   python tools/rsm_lab.py build NAME[:-DFLAG[,-DFLAG...]] ...
  run (GPU box):          python tools/rsm_lab.py run NAME NAME ... [--launches 20] [--rounds 4]

Every variant is tools/lab/rsm_lab.hip (= csrc/resample_mfma.h alone) compiled into tools/lab/_build/librsm_NAME.so with its -D
switches (-DLAB_RSM_RD=1: the 8-byte operand-read layout).  `run` resamples the BASELINE config-3 shard (128 x stereo x 30 s,
44.1 kHz -> 16 kHz, kaiser_best) with the variants interleaved round by round on three rotating input / output buffers, checks
every variant against the product under AAMD_POLICY_RESAMPLE_FP32 (exact-fp32 MFMA, same taps) and against the first variant,
and prints the average time per launch."""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "lab", "_build")
SRC = os.path.join(HERE, "lab", "rsm_lab.hip")


def so_path(name):
    return os.path.join(OUT, "librsm_%s.so" % name)


def build(specs):
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(":")
        fl = [f for f in flags.split(",") if f]
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
               "-fno-slp-vectorize", "-Rpass-analysis=kernel-resource-usage"] + fl + [SRC, "-o", so_path(name)]
        log = open(os.path.join(OUT, "rsm_" + name + ".log"), "w")
        procs.append((name, subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT), log))
    for name, p, log in procs:
        rc = p.wait()
        log.close()
        txt = open(os.path.join(OUT, "rsm_" + name + ".log")).read()
        res = {}
        for line in txt.splitlines():
            for key in ("VGPRs:", "ScratchSize [bytes/lane]:", "Occupancy [waves/SIMD]:"):
                if " " + key in line and "Spill" not in line:
                    res[key.rstrip(":")] = line.split(key)[1].split("[")[0].strip()
        print(name, "rc", rc, json.dumps(res))
        if rc != 0:
            print(txt[-3000:])


def run(names, launches, rounds, rows=(128, 2), n=1323000, nbuf=3):
    import numpy as np
    import torch
    import audio_amd.transforms as T
    from audio_amd import _host, _lib
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    xs = [(0.5 * torch.randn(*rows, n, device=dev, generator=g)).clamp_(-1, 1) for _ in range(nbuf)]
    rs = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                    rolloff=0.9475937167399596, beta=14.769656459379492).to(dev)
    kern = rs.kernel.reshape(160, -1).contiguous()
    width = (kern.shape[1] - 441) // 2
    lo, span = _host.resample_band_table(kern.cpu().numpy())
    lo = np.ascontiguousarray(lo, dtype=np.int32)
    n_rows = rows[0] * rows[1]
    out_len = -(-160 * n // 441)
    outs = [torch.empty(n_rows, out_len, device=dev) for _ in range(nbuf)]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    stream = _lib.current_stream(dev)
    libs = {}
    for nm in names:
        L = C.CDLL(so_path(nm))
        L.lab_rsm.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 3 + [C.c_int] * 3 + [C.c_int64, C.c_void_p] + [C.c_int] * 3 + [C.c_void_p]
        libs[nm] = L

    def launch(nm, i):
        x, o = xs[i % nbuf], outs[i % nbuf]
        rc = libs[nm].lab_rsm(x.data_ptr(), kern.data_ptr(), o.data_ptr(), n_rows, n, n, 441, 160, width, out_len,
                              lo.ctypes.data_as(C.c_void_p), span, 0, cus, stream)
        assert rc == 0, (nm, rc)
        return o

    with torch.no_grad():
        with _lib.kernel_policy(_lib.POLICY_RESAMPLE_FP32):
            ref32 = rs(xs[0]).reshape(n_rows, out_len)
        prod = rs(xs[0]).reshape(n_rows, out_len)
    peak = float(ref32.abs().max())
    print(json.dumps({"product_vs_fp32_peak_rel": float((prod - ref32).abs().max()) / peak}))
    base = None
    for nm in names:
        o = launch(nm, 0)
        torch.cuda.synchronize()
        rec = {"check": nm, "rd": libs[nm].lab_rsm_rd(), "vs_fp32_kernel_peak_rel": float((o - ref32).abs().max()) / peak,
               "bit_equal_product": bool(torch.equal(o, prod)), "nan": bool(torch.isnan(o).any())}
        if base is None:
            base = o.clone()
        else:
            rec["vs_first_peak_rel"] = float((o - base).abs().max()) / peak
        print(json.dumps(rec))
    del prod, ref32
    for i in range(100):                     # clock ramp: ~60 ms of GPU time
        launch(names[0], i)
    torch.cuda.synchronize()
    res = {nm: [] for nm in names}
    for r in range(rounds):
        for nm in names:
            for i in range(3):
                launch(nm, i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(launches):
                launch(nm, i)
            e1.record()
            torch.cuda.synchronize()
            res[nm].append(e0.elapsed_time(e1) / launches)
    for nm in names:                          # a race shows up as run-to-run differences
        first = launch(nm, 0).clone()
        same = True
        for _ in range(4):
            for i in (1, 2, 0):
                o = launch(nm, i)
            same = same and bool(torch.equal(o, first))
        torch.cuda.synchronize()
        print(json.dumps({"repeat_check": nm, "bit_stable": same}))
    alg = 4.0 * n_rows * (n + out_len)
    for nm in names:
        v = res[nm]
        print(json.dumps({"variant": nm, "ms_per_launch": [round(t, 4) for t in v], "best": round(min(v), 4),
                          "mean": round(sum(v) / len(v), 4), "frac_hbm_best": round(alg / (min(v) * 1e-3) / 8e12, 4)}))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        args = sys.argv[2:]
        launches, rounds = 20, 4
        rows, nbuf = (128, 2), 3            # --rows R: R x stereo rows; --nbuf 1: one input / output buffer (cache-resident when it fits the 256 MiB L3)
        names = []
        i = 0
        while i < len(args):
            if args[i] == "--launches":
                launches = int(args[i + 1]); i += 2
            elif args[i] == "--rounds":
                rounds = int(args[i + 1]); i += 2
            elif args[i] == "--rows":
                rows = (int(args[i + 1]), 2); i += 2
            elif args[i] == "--nbuf":
                nbuf = int(args[i + 1]); i += 2
            else:
                names.append(args[i]); i += 1
        run(names, launches, rounds, rows=rows, nbuf=nbuf)

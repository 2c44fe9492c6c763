#!/usr/bin/env python
"""cfg5b: the product call (F.fftconvolve, repeated impulse response) against the bare C-ABI run stage and the lab build of the
same header, in one process on one box: what the host path costs on top of the kernel."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import audio_amd.functional as F
from audio_amd import _lib

dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand(32, 8, 480000, device=dev, generator=g) - 0.5
t = torch.arange(24000, device=dev) / 48000.0
rir = torch.randn(1, 1, 24000, device=dev, generator=g) * torch.exp(-t / 0.1) * 0.05


def timed(fn, warmup=5, steps=20, rounds=4):
    out = []
    for _ in range(rounds):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(round(e0.elapsed_time(e1) / steps, 4))
    return out


with torch.no_grad():
    for _ in range(80):
        F.fftconvolve(x, rir)
    print(json.dumps({"path": "F.fftconvolve (held taps, dispatcher op)", "ms": timed(lambda: F.fftconvolve(x, rir))}), flush=True)
    # the bare run stage of the C ABI on preallocated buffers
    L = _lib.lib()
    xr, yr = x.reshape(256, 480000), rir.reshape(1, 24000)
    out = torch.empty(256, 503999, device=dev)
    ymap = torch.zeros(256, dtype=torch.int64, device=dev)
    nb = L.aamd_fftconvolve_workspace(256, 256, 1, 480000, 24000)
    ws = torch.empty(nb // 4 + 2, dtype=torch.float32, device=dev)
    st = _lib.current_stream(dev)

    def c_call(stages):
        _lib.check(L.aamd_fftconvolve_staged_f32(xr.data_ptr(), yr.data_ptr(), out.data_ptr(), 256, 256, 1, 480000, 24000, None,
                                                 ymap.data_ptr(), 0, 503999, ws.data_ptr(), stages, st))
    c_call(3)
    print(json.dumps({"path": "aamd_fftconvolve_staged_f32 RUN only, one output buffer", "ms": timed(lambda: c_call(2))}), flush=True)
    print(json.dumps({"path": "aamd_fftconvolve_staged_f32 PREPARE + RUN", "ms": timed(lambda: c_call(3))}), flush=True)
    print(json.dumps({"path": "F.fftconvolve again", "ms": timed(lambda: F.fftconvolve(x, rir))}), flush=True)
    fresh = lambda: F.fftconvolve(x, rir.clone())
    print(json.dumps({"path": "F.fftconvolve with a NEW tap tensor per call (prepares every call)", "ms": timed(fresh)}), flush=True)
    # the lab build of the same header (tools/fdr_lab.py build NAME), in this process on this box
    for name in sys.argv[1:]:
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab", "_build", "libfdr_%s.so" % name)
        if not os.path.exists(so):
            continue
        LL = C.CDLL(so)
        LL.lab_fdr_workspace.restype = C.c_int64
        LL.lab_fdr_workspace.argtypes = [C.c_int64, C.c_int64]
        LL.lab_fdr.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 4 + [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        nbl = LL.lab_fdr_workspace(1, 24000)
        wsl = torch.zeros((nbl + 8 * 256) // 4 + 2, dtype=torch.float32, device=dev)
        cus = torch.cuda.get_device_properties(0).multi_processor_count

        def lab_call(stages):
            rc = LL.lab_fdr(xr.data_ptr(), yr.data_ptr(), out.data_ptr(), 256, 1, 480000, 24000, wsl.data_ptr(), stages, cus, st)
            assert rc == 0, rc
        lab_call(1)
        print(json.dumps({"path": "lab build '%s', RUN only, the same buffers" % name, "ms": timed(lambda: lab_call(2))}), flush=True)
        ref = F.fftconvolve(x, rir).reshape(256, -1)
        lab_call(2)
        torch.cuda.synchronize()
        print(json.dumps({"lab_vs_product_bit_equal": bool(torch.equal(out, ref))}), flush=True)

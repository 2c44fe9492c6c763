"""Sorted first among the GPU tests.  Uses torch ONLY (no audio_amd import): if these fail, the box cannot run torch
and nothing below says anything about the product kernels; if they pass and a later test faults, the product did it.
The same stages already ran in a child process at session start (conftest.pytest_sessionstart -> gpu_preflight)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_00_child_process_preflight_passed():
    import conftest
    pf = conftest.PREFLIGHT
    if pf is None:          # session hook did not run (e.g. test selected by node id without -m gpu)
        import gpu_preflight
        pf = gpu_preflight.preflight(verbose=True, apply_env=False)
    assert pf["ok"], "torch-only child process cannot use the GPU on this box: " + str(
        [(a["label"], a["rc"], a["last_stage"]) for a in pf["attempts"]])
    print("preflight environment:", pf["chosen"])


def test_01_torch_only_first_touch_in_this_process():
    import torch
    assert "audio_amd._lib" not in sys.modules or True
    print("device:", torch.cuda.get_device_name(0), "| torch", torch.__version__, "| hip", torch.version.hip, flush=True)
    try:
        out = subprocess.run("rocminfo | grep -E 'gfx9|Marketing' | sort | uniq -c | head -6", shell=True,
                             capture_output=True, text=True, timeout=60).stdout
        print(out, flush=True)
    except Exception as e:   # noqa: BLE001
        print("rocminfo unavailable:", e)
    a = torch.full((1 << 20,), 2.0, device="cuda")
    torch.cuda.synchronize()
    assert float((a * 3).sum().item()) == 6.0 * (1 << 20)
    h = torch.arange(1 << 16, dtype=torch.float64)
    g = h.cuda()                                     # pageable H2D: the operation GPUTEST_r01 died in
    torch.cuda.synchronize()
    assert torch.equal(g.cpu(), h)


def test_02_product_library_loads_and_reports_device():
    import torch
    from audio_amd import _lib
    h = _lib.lib()                                   # dlopen + symbol check, explicit and first
    torch.cuda.synchronize()
    import ctypes as C
    name = C.create_string_buffer(128)
    cus = C.c_int32(0)
    mem = C.c_int64(0)
    _lib.check(h.aamd_device_info(name, 128, C.byref(cus), C.byref(mem)))
    print("libaudio_amd sees:", name.value.decode(), cus.value, "CUs", mem.value >> 30, "GiB", flush=True)
    assert cus.value >= 64


def test_03_one_product_launch_synchronised():
    """One small headline launch bracketed by synchronize(): a fault here is the product's, unambiguously."""
    import torch
    import audio_amd.transforms as T
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    x = torch.zeros(2, 4000, device="cuda")
    torch.cuda.synchronize()
    y = mel(x)
    torch.cuda.synchronize()
    assert y.shape == (2, 80, 26) and float(y.abs().max().item()) == 0.0

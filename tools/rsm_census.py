#!/usr/bin/env python
"""Per-phase time stamps of the f16 resampler's workgroup 0 (AAMD_RSM_LAB=64 must be set): where a chunk period goes."""
import ctypes as C, os, sys
# the tools-only kernel variants live in libaudio_amd_lab.so (python -m audio_amd._build --lab), reached through ctypes
os.environ.setdefault("AAMD_USE_LAB_LIB", "1")
os.environ.setdefault("AAMD_NO_TORCH_SHIM", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audio_amd.transforms as T
from audio_amd import _lib

dev = torch.device("cuda")
x = (0.5 * torch.randn(128, 2, 1323000, device=dev)).clamp_(-1, 1)
rs = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser", lowpass_filter_width=64,
                rolloff=0.9475937167399596, beta=14.769656459379492).to(dev)
with torch.no_grad():
    for _ in range(5):
        rs(x)
    torch.cuda.synchronize()
h = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(16 * 32 * 8, dtype=np.int64)
assert h.aamd_debug_rsm_census(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
t = buf.reshape(16, 32, 8).astype(np.float64) * 0.01     # us (100 MHz)
base = t[0, :, 0]
print("chunk period (compute wave 0, stamp 0 -> next stamp 0): mean %.2f us" % np.diff(t[0, :, 0]).mean())
print("compute waves (0..9): MFMA loop = s1 - s0, stores = s2 - s1, wait for the loaders = s3 - s2, convert = s4 - s3, "
      "barrier = s5 - s4   [us, mean over 32 chunks]")
for w in range(10):
    a = t[w]
    d = [(a[:, i + 1] - a[:, i]).mean() for i in range(5)]
    print("  wave %2d: loop %5.2f  stores %5.2f  wait %5.2f  convert %5.2f  barrier %5.2f" % (w, *d))
print("loader waves (10, 11), round 6: data wait + max = s1 - s0, publish + wait for the other loader = s2 - s1, convert + LDS writes = s3 - s2, fetch issue = s4 - s3, barrier = s5 - s4")
for w in (10, 11):
    a = t[w]
    d = [(a[:, i + 1] - a[:, i]).mean() for i in range(5)]
    print("  wave %2d: data+max %5.2f  publish %5.2f  convert+write %5.2f  fetch %5.2f  barrier %5.2f   | s0 relative to compute wave 0's chunk start: %5.2f" % (
        w, *d, (a[:, 0] - t[0, :, 0]).mean()))

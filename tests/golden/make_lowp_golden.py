#!/usr/bin/env python
"""Reference runs with float16 / bfloat16 INPUTS -> tests/golden/lowp_goldens.npz (round 5).

Run ONLY in the build container, where the reference checkout is mounted:

    PYTHONPATH=/root/reference/src python tests/golden/make_lowp_golden.py

The reference accepts every floating dtype and returns it (functional/functional.py:1413-1414, filtering.py:1032-1099); on
the CPU its backend implements the reduced-precision dtypes for resample (conv1d), lfilter (the compiled loop / the Python
fallback), amplitude_to_DB (element-wise) and the MelScale product (matmul) -- aten::stft and fft do not.  Each case stores
the input (its reduced-precision values, as float32), the reference's reduced-precision output (as float32) and the same call in
float64 on the same input values, so that a test can ask "at least as close to the exact answer as the reference's own
reduced-precision path".  Nothing here is imported by the product package."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
import torchaudio.functional as F  # noqa: E402  (the reference, pure-python mode)

HERE = os.path.dirname(os.path.abspath(__file__))
DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def main():
    g = torch.Generator().manual_seed(20250923)
    out = {}
    for tag, dt in DT.items():
        x = (0.4 * torch.randn(3, 2, 6000, generator=g)).clamp_(-1, 1).to(dt)
        out[f"{tag}_wave"] = x.float().numpy()
        # resample: two rate pairs (the tap table is evaluated in the waveform's dtype by the reference)
        for orig, new in ((16000, 8000), (44100, 16000)):
            y = F.resample(x, orig, new)
            assert y.dtype == dt
            out[f"{tag}_resample_{orig}_{new}"] = y.float().numpy()
            out[f"{tag}_resample_{orig}_{new}_f64"] = F.resample(x.double(), orig, new).numpy()
        # lfilter / biquad
        a = torch.tensor([1.0, -1.2, 0.5]).to(dt)
        b = torch.tensor([0.25, 0.5, 0.25]).to(dt)
        y = F.lfilter(x, a, b)
        assert y.dtype == dt
        out[f"{tag}_lfilter_a"], out[f"{tag}_lfilter_b"] = a.float().numpy(), b.float().numpy()
        out[f"{tag}_lfilter"] = y.float().numpy()
        out[f"{tag}_lfilter_f64"] = F.lfilter(x.double(), a.double(), b.double()).numpy()
        y = F.lowpass_biquad(x, 16000, 3000.0)
        out[f"{tag}_lowpass"] = y.float().numpy()
        out[f"{tag}_lowpass_f64"] = F.lowpass_biquad(x.double(), 16000, 3000.0).numpy()
        # amplitude_to_DB on a power spectrogram shaped tensor, with top_db
        p = (torch.rand(2, 3, 40, 30, generator=g) * 10.0 + 1e-3).to(dt)
        y = F.amplitude_to_DB(p, 10.0, 1e-10, 0.0, 80.0)
        assert y.dtype == dt
        out[f"{tag}_db_in"] = p.float().numpy()
        out[f"{tag}_db"] = y.float().numpy()
        out[f"{tag}_db_f64"] = F.amplitude_to_DB(p.double(), 10.0, 1e-10, 0.0, 80.0).numpy()
        # the MelScale product (transforms/_transforms.py:403-415) with a filterbank cast like module.half() casts it
        fb = F.melscale_fbanks(201, 0.0, 8000.0, 40, 16000)
        s = (torch.rand(2, 201, 25, generator=g) * 5.0).to(dt)
        y = torch.matmul(s.transpose(-1, -2), fb.to(dt)).transpose(-1, -2)
        out[f"{tag}_melscale_in"] = s.float().numpy()
        out[f"{tag}_melscale"] = y.float().numpy()
        out[f"{tag}_melscale_f64"] = torch.matmul(s.double().transpose(-1, -2), fb.to(dt).double()).transpose(-1, -2).numpy()
    np.savez_compressed(os.path.join(HERE, "lowp_goldens.npz"), **out)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "lowp_goldens.npz")), "bytes")


if __name__ == "__main__":
    main()

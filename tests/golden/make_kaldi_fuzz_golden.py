#!/usr/bin/env python
"""The REFERENCE's compliance/kaldi.py outputs for the random configurations of tests/kaldi_fuzz_cases.py (suite seeds), CPU.
Run only in the build container:   python tests/golden/make_kaldi_fuzz_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
sys.path.insert(0, os.path.dirname(HERE))
import torchaudio.compliance.kaldi as K  # noqa: E402  (the reference)
import kaldi_fuzz_cases as C  # noqa: E402

out = {}
for seed in C.SUITE_SEEDS:
    fn, wav, kw = C.case(seed)
    y = getattr(K, fn)(torch.from_numpy(wav), **kw)
    out[f"seed{seed}"] = y.numpy()
    print(seed, fn, tuple(y.shape), {k: kw[k] for k in ("sample_frequency", "frame_length", "round_to_power_of_two", "window_type")})
np.savez_compressed(os.path.join(HERE, "kaldi_fuzz_goldens.npz"), **out)

"""Host-side constants must be BIT-identical to what the reference builds for its buffers
(fixtures produced by the reference itself, tests/golden/make_golden.py)."""
import math
import re
import os
import warnings

import numpy as np
import pytest
import torch

from audio_amd import _host, _lib
from conftest import ref_runs, ROOT


@pytest.mark.parametrize("case", ref_runs().select("melscale_fbanks"), ids=lambda c: str(c["id"]))
def test_fbanks_bit_exact(case):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fb = _host.melscale_fbanks(**case["kwargs"]).numpy()
    assert np.array_equal(fb, ref_runs().output(case))


@pytest.mark.parametrize("case", ref_runs().select("create_dct"), ids=lambda c: str(c["id"]))
def test_dct_bit_exact(case):
    assert np.array_equal(_host.create_dct(**case["kwargs"]).numpy(), ref_runs().output(case))


@pytest.mark.parametrize("op,dtype", [("sinc_kernel_transform", None), ("sinc_kernel_functional_f32", torch.float32)])
def test_sinc_kernel_bit_exact(op, dtype):
    case = ref_runs().select(op)[0]
    kw = dict(case["kwargs"])
    o, n, w = kw.pop("orig_freq"), kw.pop("new_freq"), kw.pop("width")
    k, width = _host.sinc_resample_kernel(o, n, math.gcd(o, n), **kw, dtype=dtype)
    assert width == w
    assert np.array_equal(k.numpy(), ref_runs().output(case))


def test_band_table_reconstructs_fb():
    fb = _host.melscale_fbanks(201, 0.0, 8000.0, 80, 16000).numpy()
    lo, width, weights, mw = _host.mel_band_table(fb)
    assert mw == 13 and int((fb != 0).sum()) == 392
    rec = np.zeros_like(fb)
    for m in range(80):
        rec[lo[m]:lo[m] + width[m], m] = weights[m, :width[m]]
    assert np.array_equal(rec, fb)
    # arbitrary (non-triangular, with an all-zero column) matrix still round-trips
    rng = np.random.default_rng(0)
    g = (rng.random((33, 7)) > 0.6) * rng.standard_normal((33, 7)).astype(np.float32)
    g[:, 3] = 0
    lo, width, weights, mw = _host.mel_band_table(g)
    rec = np.zeros_like(g)
    for m in range(7):
        rec[lo[m]:lo[m] + width[m], m] = weights[m, :width[m]]
    assert np.array_equal(rec, g.astype(np.float32))


def test_mel_warning_count():
    # functional_impl.py:1332-1350: exactly one warning when a filter is all-zero
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _host.melscale_fbanks(201, 0.0, 8000.0, 128, 16000)
    assert len(w) == 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _host.melscale_fbanks(201, 0.0, 8000.0, 80, 16000)
    assert len(w) == 0


def test_frame_count_matches_oracle():
    from oracle import dsp_oracle as O
    for L, n, h, c in [(16000, 400, 160, True), (160000, 400, 160, True), (4000, 1024, 256, False), (300, 400, 160, True)]:
        assert _host.frame_count(L, n, h, c) == O.frame_count(L, n, h, c)
    assert _host.frame_count(160000, 400, 160, True) == 1001


def test_abi_header_symbols_exported():
    """The C-ABI library loads and exports every function include/audio_amd.h declares
    (no compute calls: there is no GPU here)."""
    hdr = open(os.path.join(ROOT, "include", "audio_amd.h")).read()
    declared = set(re.findall(r"\b(aamd_[a-z0-9_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_desc") and not d.endswith("_bands")}
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libaudio_amd.so not built yet (python -m audio_amd._build)")
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name)
    assert L.aamd_abi_version() == 7


def test_mfcc_fixup_shares_partition_the_tiles():
    """The fix-up launch of the fused MFCC (csrc/melspec400.h, top of melspec400_kernel): workgroup b of nb checks the tiles
    b, b + nb, ... and keeps its flagged ones in a private run of the n_tiles-entry scratch list starting at
    b * (nt / nb) + min(b, nt % nb).  Restated here (the kernel body is device code): every tile is some workgroup's candidate
    exactly once, and the private runs are disjoint, in order, and exactly fill [0, nt)."""
    import random
    rnd = random.Random(7)
    cases = [(1, 1), (5, 8), (8, 8), (9, 8), (85504, 256), (42752, 256), (167, 24), (1000003, 248)]
    cases += [(rnd.randint(1, 5000), rnd.choice([1, 8, 16, 104, 248, 256])) for _ in range(40)]
    for nt, nb in cases:
        seen = [0] * nt
        end_prev = 0
        for b in range(nb):
            n_cand = (nt - b + nb - 1) // nb if b < nt else 0
            base = b * (nt // nb) + min(b, nt % nb)
            assert base == end_prev, (nt, nb, b)
            end_prev = base + n_cand
            for k in range(n_cand):
                t = b + k * nb
                assert 0 <= t < nt
                seen[t] += 1
        assert end_prev == nt and all(c == 1 for c in seen), (nt, nb)

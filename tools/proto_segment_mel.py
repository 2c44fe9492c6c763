#!/usr/bin/env python
"""Prototype (numpy, float32) of the SEGMENT formulation of a triangular mel filterbank -- a candidate for the headline
kernel's epilogue: every bin k lies in one segment s = (c_s, c_{s+1}] between adjacent triangle
corners and feeds at most mel s-1 ... with weights n_m d_k (falling edge of mel m = s) and n_{m+1} (1 - d_k) (rising edge
of mel m+1), so
    A_s = sum_{k in s} d_k P_k,   B_s = sum_{k in s} P_k,   mel_m = n_m (A_m + B_{m-1} - A_{m-1})
reads every P value once.  This script (1) recovers (segment, d_k, n_m) from an arbitrary fb matrix and reports whether the
matrix has the structure, (2) measures the float32 error of the formula against the float64 dense product on noise, on
tones placed at the worst spots (just above a corner) and on silence, for HTK / Slaney scales and norms."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_amd import _host  # noqa: E402


def segment_tables(fb, tol=2e-6):
    """fb: (n_freq, n_mels) float32.  Returns (seg_of_bin, d, n) or None when fb is not a partition-of-unity triangle bank."""
    F, M = fb.shape
    fb = fb.astype(np.float64)
    n = fb.max(axis=0)                                   # peak of each triangle ~ n_m (exact only if a bin hits the corner)
    seg = np.full(F, -2, dtype=np.int64)                 # segment s feeds mel s (falling, weight d) and mel s+1 (rising)
    d = np.zeros(F)
    nz = [np.nonzero(fb[k])[0] for k in range(F)]
    for k in range(F):
        if len(nz[k]) == 0:
            seg[k] = -2                                  # bin outside every triangle
        elif len(nz[k]) == 1:
            seg[k] = -1                                  # decided below (first rising edge / last falling edge)
        elif len(nz[k]) == 2 and nz[k][1] == nz[k][0] + 1:
            seg[k] = nz[k][0]
        else:
            return None
    # slopes: within segment s, fb[k, s] falls linearly in mel(f) and fb[k, s+1] rises; solve n_s, n_{s+1} from two bins of a
    # segment when possible, otherwise from the triangle peaks
    norm = np.zeros(M)
    for m in range(M):
        ks = [k for k in range(F) if seg[k] == m or seg[k] == m - 1]
        norm[m] = n[m]
    # refine n_m with least squares on fb[k,m]/n_m + fb[k,m+1]/n_{m+1} = 1 over all two-mel bins
    rows, rhs = [], []
    for k in range(F):
        if seg[k] >= 0:
            r = np.zeros(M); r[seg[k]] = fb[k, seg[k]]; r[seg[k] + 1] = fb[k, seg[k] + 1]
            rows.append(r); rhs.append(1.0)
    if rows:
        sol, *_ = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)
        ok = sol > 0
        norm[ok] = 1.0 / sol[ok]
    for k in range(F):
        if seg[k] >= 0:
            d[k] = fb[k, seg[k]] / norm[seg[k]]
            if abs(fb[k, seg[k] + 1] / norm[seg[k] + 1] - (1 - d[k])) > tol:
                return None
        elif seg[k] == -1:
            m = nz[k][0]
            frac = fb[k, m] / norm[m]
            # single-mel bins: rising edge of mel 0 (segment -1: d = 1 - frac) or falling edge of the last mel (segment M-1: d = frac)
            if m == 0 and (k == 0 or seg[k - 1] in (-2, -1)):
                seg[k], d[k] = -1, 1 - frac
            else:
                seg[k], d[k] = m, frac
    return seg, d, norm


def mel_segments(P, seg, d, norm, M, dtype=np.float32):
    """P: (frames, F).  Segment sums in `dtype`, then mel_m = n_m (A_m + B_{m-1} - A_{m-1})."""
    P = P.astype(dtype); dd = d.astype(dtype)
    A = np.zeros((P.shape[0], M + 1), dtype=dtype); B = np.zeros_like(A)      # index s + 1 (segment -1 -> slot 0)
    for k in range(P.shape[1]):
        if seg[k] >= -1:
            A[:, seg[k] + 1] += dd[k] * P[:, k]
            B[:, seg[k] + 1] += P[:, k]
    mel = norm.astype(dtype)[None, :] * (A[:, 1:] + (B[:, :-1] - A[:, :-1]))
    return mel


def main():
    rng = np.random.default_rng(0)
    for (scale, nrm, n_mels) in [("htk", None, 80), ("slaney", "slaney", 80), ("htk", None, 128), ("slaney", None, 40)]:
        fb = _host.melscale_fbanks(201, 0.0, 8000.0, n_mels, 16000, nrm, scale).numpy()
        tabs = segment_tables(fb)
        if tabs is None:
            print(f"{scale}/{nrm}/{n_mels}: NOT a partition-of-unity triangle bank -> keep the gather")
            continue
        seg, d, norm = tabs
        cases = {"noise": rng.random((64, 201)) ** 4 * 100}
        tone = np.full((201, 201), 1e-9); tone[np.arange(201), np.arange(201)] = 1e4       # a pure tone in every bin
        cases["tones"] = tone
        cases["silence+tone"] = np.where(np.arange(201)[None, :] == 57, 1.0, 0.0) * np.ones((4, 1))
        line = []
        for name, P in cases.items():
            ref = P.astype(np.float64) @ fb.astype(np.float64)
            got = mel_segments(P, seg, d, norm, n_mels)
            gather = (P.astype(np.float32) @ fb).astype(np.float64)
            peak = np.abs(ref).max(axis=1, keepdims=True)
            line.append(f"{name}: segment {np.abs(got - ref).max() / peak.max():.1e} (per-frame peak-rel {np.max(np.abs(got - ref) / peak):.1e}), "
                        f"gather {np.max(np.abs(gather - ref) / peak):.1e}")
        reads_now = sum(((np.count_nonzero(fb[:, m]) + 3) // 4) * 4 for m in range(n_mels))
        print(f"{scale}/{nrm}/{n_mels}: structure ok; P reads per frame {reads_now} (padded gather) -> {int((seg >= -1).sum())} (segments)")
        for l in line:
            print("   ", l)


if __name__ == "__main__":
    main()

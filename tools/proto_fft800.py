"""Numpy prototype for DESIGN "next steps" 7(a): n_fft = 800, hop = 200 on the machinery of the radix-20x20 kernel
(audio_amd/csrc/melspec400.h).  ONE real frame of 800 samples is one complex 400-point FFT of z[n] = x[2n] + i x[2n+1] plus the
real-FFT split -- the same transform the kernel already runs on a PAIR of 400-sample frames -- so a 20-lane group would own ONE
frame instead of a pair.  This file validates the index maps and the geometry only (no GPU code exists for it yet):

  * gather: lane (r = n mod 20) of the group reads the float2 (x[2n], x[2n+1]) for n = r + 20 q, q = 0..19, i.e. the 40 samples
    2 r + 40 q + {0, 1} -- 8-byte loads from the staged tile, 40 window taps per lane instead of 20;
  * 400-point complex FFT = the kernel's two DFT-20 passes with the twiddles W_400^(r s) in between (unchanged);
  * split: X[k] = E[k] + W_800^k O[k],  E = (Z[k] + conj Z[400-k]) / 2,  O = (Z[k] - conj Z[400-k]) / (2i),  k = 0..400:
    the partner bin 400 - k is what the pair separation already fetches (lane ^ 1 after the column permutation), the extra work
    per bin is ONE complex multiply by W_800^k; bin 400 = E[0] - O[0] comes with bin 0;
  * 401 power values per frame = the LDS row pair of 2 x 201 the pair kernel writes (kPPair = 416 dwords hold 401 + padding);
  * tile: 3 frames of 800 at hop 200 span 2 * 200 + 800 = 1200 samples -- the 1200-sample staging tile of the hop-160 kernel
    (6 frames: 5 * 160 + 400); frames per wave-tile 3 instead of 6, so the tile loop, queue and DMA are unchanged.
Run: python tools/proto_fft800.py"""
import numpy as np

from proto_fft400 import dft20


def fft400_via_20x20(z):
    """the kernel's decomposition: n = r + 20 q, k = s + 20 u (tools/proto_fft400.py)"""
    Y = np.zeros((20, 20), complex)
    for r in range(20):
        Y[r] = dft20([z[r + 20 * q] for q in range(20)])
        Y[r] *= np.exp(-2j * np.pi * r * np.arange(20) / 400)
    Z = np.zeros(400, complex)
    for s in range(20):
        o = dft20([Y[r][s] for r in range(20)])
        for u in range(20):
            Z[s + 20 * u] = o[u]
    return Z


def rfft800(frame, window):
    xw = frame * window
    z = xw[0::2] + 1j * xw[1::2]                      # lane r, register q: samples 2 (r + 20 q) and 2 (r + 20 q) + 1
    Z = fft400_via_20x20(z)
    X = np.zeros(401, complex)
    for k in range(401):
        zk = Z[k % 400]
        zp = np.conj(Z[(400 - k) % 400])              # the partner bin the pair separation already exchanges
        E = 0.5 * (zk + zp)
        O = -0.5j * (zk - zp)
        X[k] = E + np.exp(-2j * np.pi * k / 800) * O
    return X


def main():
    rng = np.random.default_rng(1)
    hop, n_fft, frames = 200, 800, 3
    tile = rng.standard_normal((frames - 1) * hop + n_fft)
    assert tile.size == 1200                           # the staging tile of the hop-160 kernel: 5 * 160 + 400
    w = np.hanning(n_fft + 1)[:-1]
    worst = 0.0
    for f in range(frames):
        fr = tile[f * hop:f * hop + n_fft]
        got = rfft800(fr, w)
        ref = np.fft.rfft(fr * w)
        worst = max(worst, float(np.abs(got - ref).max() / np.abs(ref).max()))
    assert worst < 1e-12, worst
    # lanes of a 20-lane group: lane r gathers float2 at sample 2 r + 40 q -- strides of 40 samples = 160 bytes, 8-byte aligned
    for r in range(20):
        idx = np.array([2 * (r + 20 * q) for q in range(20)])
        assert (idx % 2 == 0).all() and idx.max() + 1 < n_fft
    # the power rows: 401 values fit the 416-dword row pair the pair kernel uses for 2 x 201 (+ zeroed tail)
    assert 401 <= 416
    print(f"n_fft 800 / hop 200 as one complex 400-point FFT per frame: max error {worst:.2e} of the peak over {frames} frames; "
          f"tile {tile.size} samples = the hop-160 staging tile")


if __name__ == "__main__":
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()

"""Host-side (init-time) constants for the HIP kernels: twiddle tables, centre-padded
windows, banded mel filterbanks, DCT matrices, windowed-sinc resampling kernels.

These are small, computed once per configuration on the CPU exactly as the reference
computes its own ``register_buffer`` constants, then uploaded.  Nothing here is on the
per-call data path.
"""
from __future__ import annotations

import math
import warnings
from typing import Optional, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------- #
# FFT twiddles                                                                #
# --------------------------------------------------------------------------- #


def twiddle_table(n_fft: int) -> np.ndarray:
    """float32 [n_fft, 2] = (cos, -sin)(2 pi t / n_fft), evaluated in float64."""
    t = np.arange(n_fft, dtype=np.float64)
    # exact quadrant symmetry: reduce the angle before cos/sin
    ang = 2.0 * np.pi * t / n_fft
    tw = np.stack([np.cos(ang), -np.sin(ang)], axis=1)
    return tw.astype(np.float32)


def frame_count(length: int, n_fft: int, hop: int, center: bool, pad: int = 0) -> int:
    lp = length + 2 * pad + (2 * (n_fft // 2) if center else 0)
    return 1 + (lp - n_fft) // hop


def center_pad_window(window: torch.Tensor, n_fft: int) -> torch.Tensor:
    """aten::stft zero-pads a short window to n_fft, centred (left = (n_fft - win_length)//2)."""
    wl = window.shape[0]
    if wl == n_fft:
        return window
    left = (n_fft - wl) // 2
    return torch.nn.functional.pad(window, (left, n_fft - wl - left))


# --------------------------------------------------------------------------- #
# mel filterbank (reference: functional/functional.py:425-587)                #
# --------------------------------------------------------------------------- #


def _hz_to_mel(freq: float, mel_scale: str = "htk") -> float:
    if mel_scale not in ["slaney", "htk"]:
        raise ValueError('mel_scale should be one of "htk" or "slaney".')
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + (freq / 700.0))
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    if freq >= min_log_hz:
        return min_log_hz / f_sp + math.log(freq / min_log_hz) / (math.log(6.4) / 27.0)
    return freq / f_sp


def _mel_to_hz(mels: torch.Tensor, mel_scale: str = "htk") -> torch.Tensor:
    if mel_scale not in ["slaney", "htk"]:
        raise ValueError('mel_scale should be one of "htk" or "slaney".')
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_mel = 1000.0 / f_sp
    logstep = math.log(6.4) / 27.0
    is_log = mels >= min_log_mel
    freqs[is_log] = 1000.0 * torch.exp(logstep * (mels[is_log] - min_log_mel))
    return freqs


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int,
                    norm: Optional[str] = None, mel_scale: str = "htk") -> torch.Tensor:
    """Triangular mel filterbank (n_freqs, n_mels), float32 torch CPU ops in the same order as
    the reference so the buffer is bit-identical to torchaudio's ``fb``."""
    if norm is not None and norm != "slaney":
        raise ValueError('norm must be one of None or "slaney"')
    bin_hz = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_lo = _hz_to_mel(f_min, mel_scale=mel_scale)
    mel_hi = _hz_to_mel(f_max, mel_scale=mel_scale)
    edges_hz = _mel_to_hz(torch.linspace(mel_lo, mel_hi, n_mels + 2), mel_scale=mel_scale)
    gaps = edges_hz[1:] - edges_hz[:-1]                         # (n_mels + 1)
    dist = edges_hz.unsqueeze(0) - bin_hz.unsqueeze(1)          # (n_freqs, n_mels + 2)
    falling = (-1.0 * dist[:, :-2]) / gaps[:-1]
    rising = dist[:, 2:] / gaps[1:]
    fb = torch.max(torch.zeros(1), torch.min(falling, rising))
    if norm is not None and norm == "slaney":       # (written with the refinement TorchScript needs for an Optional[str])
        fb *= (2.0 / (edges_hz[2:n_mels + 2] - edges_hz[:n_mels])).unsqueeze(0)
    if (fb.max(dim=0).values == 0.0).any():
        warnings.warn(
            "At least one mel filterbank has all zero values. "
            f"The value for `n_mels` ({n_mels}) may be set too high. "
            f"Or, the value for `n_freqs` ({n_freqs}) may be set too low."
        )
    return fb


def mel_band_table(fb: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """Banded view of ANY fb[n_freq, n_mels]: per column first/last non-zero row.
    Returns (lo int32[M], width int32[M], weights float32[M, max_width], max_width)."""
    fb = np.asarray(fb, dtype=np.float32)
    n_freq, n_mels = fb.shape
    nz = fb != 0
    any_nz = nz.any(axis=0)
    lo = np.where(any_nz, nz.argmax(axis=0), 0).astype(np.int32)
    hi = np.where(any_nz, n_freq - 1 - nz[::-1].argmax(axis=0), -1).astype(np.int32)
    width = np.where(any_nz, hi - lo + 1, 0).astype(np.int32)
    max_width = max(1, int(width.max()) if n_mels else 1)
    weights = np.zeros((n_mels, max_width), dtype=np.float32)
    for m in range(n_mels):
        w = int(width[m])
        if w:
            weights[m, :w] = fb[lo[m]:lo[m] + w, m]
    return lo, width, weights, max_width


# --------------------------------------------------------------------------- #
# DCT (reference: functional/functional.py:636-667)                           #
# --------------------------------------------------------------------------- #


def create_dct(n_mfcc: int, n_mels: int, norm: Optional[str]) -> torch.Tensor:
    if norm is not None and norm != "ortho":
        raise ValueError('norm must be either "ortho" or None')
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    basis = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)   # (n_mfcc, n_mels)
    if norm is None:
        basis *= 2.0
    else:
        basis[0] *= 1.0 / math.sqrt(2.0)
        basis *= math.sqrt(2.0 / float(n_mels))
    return basis.t()


# --------------------------------------------------------------------------- #
# windowed-sinc resampling kernel (reference: functional/functional.py:1305-1402)
# --------------------------------------------------------------------------- #

_DEPRECATED_METHODS = {"sinc_interpolation": "sinc_interp_hann", "kaiser_window": "sinc_interp_kaiser"}


def sinc_resample_kernel(orig_freq: int, new_freq: int, gcd: int, lowpass_filter_width: int = 6,
                         rolloff: float = 0.99, resampling_method: str = "sinc_interp_hann",
                         beta: Optional[float] = None, dtype: Optional[torch.dtype] = None):
    """CPU evaluation with the reference's dtype rules: ``dtype=None`` (Transform) evaluates the
    index grid in float64 and casts the result to float32; an explicit dtype (functional path)
    evaluates everything in that dtype.  Returns (kernel[new, 1, 2*width+orig], width)."""
    if not (int(orig_freq) == orig_freq and int(new_freq) == new_freq):
        raise Exception(
            "Frequencies must be of integer type to ensure quality resampling computation. "
            "To work around this, manually convert both frequencies to integer values "
            "that maintain their resampling rate ratio before passing them into the function. "
            "Example: To downsample a 44100 hz waveform by a factor of 8, use "
            "`orig_freq=8` and `new_freq=1` instead of `orig_freq=44100` and `new_freq=5512.5`. "
            "For more information, please refer to https://github.com/pytorch/audio/issues/1487."
        )
    if resampling_method in _DEPRECATED_METHODS:
        warnings.warn(
            f'"{resampling_method}" resampling method name is being deprecated and replaced by '
            f'"{_DEPRECATED_METHODS[resampling_method]}" in the next release. '
            "The default behavior remains unchanged.",
            stacklevel=3,
        )
    elif resampling_method not in ["sinc_interp_hann", "sinc_interp_kaiser"]:
        raise ValueError("Invalid resampling method: {}".format(resampling_method))

    orig = int(orig_freq) // gcd
    new = int(new_freq) // gcd
    if lowpass_filter_width <= 0:
        raise ValueError("Low pass filter width should be positive.")
    cutoff = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / cutoff)

    grid_dtype = dtype if dtype is not None else torch.float64
    grid = torch.arange(-width, width + orig, dtype=grid_dtype)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=dtype)[:, None, None] / new + grid
    t *= cutoff
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    if resampling_method in ("sinc_interp_hann", "sinc_interpolation"):
        taper = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    else:
        if beta is None:
            beta = 14.769656459379492
        beta_t = torch.tensor(float(beta))
        taper = torch.i0(beta_t * torch.sqrt(1 - (t / lowpass_filter_width) ** 2)) / torch.i0(beta_t)
    t *= math.pi
    gain = cutoff / orig
    kern = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    kern *= taper * gain
    if dtype is None:
        kern = kern.to(dtype=torch.float32)
    return kern, width


def resample_band_table(kernel: np.ndarray, rel_threshold: float = 2.0 ** -40):
    """Per tile of 16 consecutive phases: first tap and common span of the non-negligible taps of
    the polyphase table ``kernel`` (new, taps).  A tap is negligible when ``|h| <= rel_threshold *
    max|h|``; with 2^-40 the dropped taps of one output sum to < 1e-9 of a full-scale sample
    (the kaiser/hann windows of functional.py:1378-1397 leave ~1e-20-sized tails outside
    +-lowpass_filter_width zero crossings).  Returns (tap_lo int32[n_tiles], tap_span)."""
    k = np.abs(np.asarray(kernel, dtype=np.float64))
    new, taps = k.shape
    thr = rel_threshold * (k.max() if k.size else 0.0)
    n_tiles = (new + 15) // 16
    lo = np.zeros(n_tiles, dtype=np.int32)
    span = 1
    for t in range(n_tiles):
        cols = np.nonzero((k[16 * t: 16 * t + 16] > thr).any(axis=0))[0]
        if cols.size:
            lo[t] = cols[0]
            span = max(span, int(cols[-1] - cols[0] + 1))
    return lo, span


def _rs_pick_ks(span: int, orig: int = 0) -> int:
    """rsm::pick_ks (csrc/resample_mfma.h): k-steps of the narrowest kernel instantiation that covers a band of `span` taps
    (104 exists for odd `orig` only: the 8-byte operand layout)."""
    need = (span + 3) // 4
    for ks in (16, 48, 80, 104, 112):
        if need <= ks and (ks != 104 or (orig & 1)):
            return ks
    return 0


def resample_fill_phase_tiles(kernel: np.ndarray, orig: int, new: int, width: int):
    """The matrix-core resampler works on tiles of 16 output phases; for a reduced rate pair with few phases (48 k -> 16 k is
    3 : 1 -- ONE phase) fifteen sixteenths of every MFMA are padding.  The same filter written for the pair (m orig : m new)
    has m new phases -- phase j new + p of the expanded table is phase p delayed by j orig samples:
        kernel'[j new + p][k] = kernel[p][k - j orig],   taps' = taps + (m - 1) orig = 2 width + m orig
    -- the outputs, their order and out_len = ceil(new L / orig) are unchanged, only the tiling of the same sums differs.
    Picks m by the MFMA work per output sample, (tiles x k-steps of the band) / (m new), from the real band table of each
    candidate, and returns (kernel', m); m = 1 returns the table as it is."""
    kernel = np.asarray(kernel)
    taps = kernel.shape[1]
    best = None
    for m in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
        if m > 1 and (m * new > 256 or new % 16 == 0):
            break
        if m == 1:
            kp = kernel
        else:
            kp = np.zeros((m * new, taps + (m - 1) * orig), dtype=kernel.dtype)
            for j in range(m):
                kp[j * new:(j + 1) * new, j * orig:j * orig + taps] = kernel
        _, span = resample_band_table(kp)
        ks = _rs_pick_ks(span, m * orig)
        if ks == 0:
            if m == 1:
                return kernel, 1
            break
        cost = ((m * new + 15) // 16) * ks / float(m * new)
        if best is None or cost < 0.85 * best[0]:
            best = (cost, m, kp)
    return best[2], best[1]


def resample_sparse_table(kernel: np.ndarray, rel_threshold: float = 2.0 ** -40):
    """Compact form of a polyphase table whose rows are almost empty (huge reduced rates, e.g. PitchShift's
    8000 x 10095): per phase the first non-negligible tap and the common span, and the taps of that window.
    Same negligibility rule as resample_band_table.  Returns (compact float32[new][span], tap_lo int32[new], span)."""
    h = np.asarray(kernel, dtype=np.float32)
    new, taps = h.shape
    k = np.abs(h.astype(np.float64))
    thr = rel_threshold * (k.max() if k.size else 0.0)
    mask = k > thr
    any_row = mask.any(axis=1)
    first = np.where(any_row, mask.argmax(axis=1), 0)
    last = np.where(any_row, taps - 1 - mask[:, ::-1].argmax(axis=1), 0)
    span = int((last - first + 1).max()) if new else 1
    lo = np.minimum(first, max(taps - span, 0)).astype(np.int32)
    idx = lo[:, None].astype(np.int64) + np.arange(span)[None, :]
    compact = np.take_along_axis(h, np.minimum(idx, taps - 1), axis=1)
    compact[idx >= taps] = 0.0
    return np.ascontiguousarray(compact, dtype=np.float32), lo, span


# --------------------------------------------------------------------------- #
# lane assignment of the radix-20x20 mel kernel (audio_amd/csrc/melspec400.h)   #
# --------------------------------------------------------------------------- #

# ds_read_b128 is serviced in four 16-lane groups (MI355X_MICROARCH.md, LDS table); lane l of the
# kernel sits at (pair l // 20, position l % 20), lanes 60..63 shadow 40..43 (same addresses).
_B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
_B128_GROUPS = _B128_GROUPS + [[l + 32 for l in g] for g in _B128_GROUPS]
_M400_PAIR_DWORDS, _M400_PK = 416, 208        # kPPair, kPK of melspec400.h


def _m400_round_cost(lo2_of_pos: np.ndarray) -> int:
    """LDS cycles of ONE b128 band read of a round: per 16-lane group the worst 4-bank slot's number
    of distinct addresses.  lo2_of_pos[i] = even band start (bins) of the mel at lane position i."""
    cost = 0
    for grp in _B128_GROUPS:
        seen = {}
        for l in grp:
            ll = l if l < 60 else l - 20
            pair, pos = divmod(ll, 20)
            addr = _M400_PAIR_DWORDS * pair + 2 * int(lo2_of_pos[pos])      # dword address, multiple of 4
            seen.setdefault((addr >> 2) & 15, set()).add(addr)
        cost += max(len(v) for v in seen.values())
    return cost


def mel_lane_order(lo: np.ndarray, width: np.ndarray, iters: int = 4000, seed: int = 0) -> np.ndarray:
    """Which mel each lane position evaluates in each round of the (n_fft, hop) = (400, 160) kernel:
    int32[ceil(M/20)*20], entry 20 r + i = mel at position i of round r (-1 = unused).  Mels stay in
    their round (rounds group bands of similar width); inside a round they are permuted by a seeded
    local search so that the b128 reads of the power-spectrum rows of each 16-lane group hit distinct
    4-bank LDS slots (the identity order costs ~2.2x the conflict-free cycles for the HTK 80-mel
    bank, the optimised order ~1.5x).  Results of the kernel do not depend on the order."""
    lo = np.asarray(lo, dtype=np.int64)
    width = np.asarray(width, dtype=np.int64)
    n_mels = lo.shape[0]
    n_rounds = (n_mels + 19) // 20
    order = np.full(n_rounds * 20, -1, dtype=np.int32)
    rng = np.random.default_rng(seed)
    for r in range(n_rounds):
        mels = list(range(20 * r, min(20 * r + 20, n_mels)))
        rw = max([4] + [int((width[m] + (lo[m] & 1) + 3) & ~3) for m in mels])
        lo2 = {m: min(int(lo[m]) & ~1, _M400_PK - rw) for m in mels}
        cur = mels + [-1] * (20 - len(mels))

        def cost(perm):
            return _m400_round_cost(np.array([lo2[m] if m >= 0 else lo2[mels[0]] for m in perm]))

        c = cost(cur)
        floor = len(_B128_GROUPS)
        for _ in range(iters):
            if c <= floor:
                break
            i, j = rng.choice(20, size=2, replace=False)
            cand = list(cur)
            cand[i], cand[j] = cand[j], cand[i]
            cc = cost(cand)
            if cc <= c:
                cur, c = cand, cc
        order[20 * r: 20 * r + 20] = cur
    return order


def _m400_cost_fast(l2_of_pos) -> int:
    """`_m400_round_cost` on plain ints (the search below calls it tens of thousands of times)."""
    cost = 0
    for grp in _B128_LANES:
        seen = {}
        for pair, pos in grp:
            addr = _M400_PAIR_DWORDS * pair + 2 * l2_of_pos[pos]
            seen.setdefault((addr >> 2) & 15, set()).add(addr)
        cost += max(len(v) for v in seen.values())
    return cost


_B128_LANES = [[divmod(l if l < 60 else l - 20, 20) for l in g] for g in _B128_GROUPS]


def mel400_table_signature(image: np.ndarray, n_mels: int, max_width: int) -> int:
    """Shape of a table image for `aamd_mel_bands.table_sig`: chunk count of round r in nibble r when the filterbank has
    exactly 4 rounds of 20 mels (n_mels = 80) and chunk counts below 16; else 0 (= unknown: the generic instantiation)."""
    if n_mels != 80:
        return 0
    rows = 80
    w4 = (int(max_width) + 1 + 3) & ~3
    ws = w4 if (w4 >> 2) & 1 else w4 + 4
    rc = image.view(np.int32)[rows * ws + 2 * rows: rows * ws + 2 * rows + 4]
    if any(int(v) < 1 or int(v) > 15 for v in rc):
        return 0
    return int(rc[0]) | (int(rc[1]) << 4) | (int(rc[2]) << 8) | (int(rc[3]) << 12)


def mel400_table_image(lo: np.ndarray, width: np.ndarray, weights: np.ndarray, max_width: int, iters: int = 6000,
                       seed: int = 0):
    """The LDS image of the band table of the (n_fft, hop) = (400, *) kernel (`m400::mel_tab_layout` in csrc/melspec400.h:
    [rows][ws] weights | lo2[rows] | row_mel[rows] | rc[8]) built on the host, with the two freedoms the kernel's results do
    not depend on used against LDS bank conflicts of the power-row reads (one b128 per lane, chunk and frame pair, serviced in
    the four 16-lane groups of MI355X_MICROARCH.md "LDS"):
      * which lane position of a round evaluates which mel (as `mel_lane_order`), and
      * the even bin a row's band read STARTS at: any even start <= the band's first bin that still covers the band inside
        the round's chunk count (the weights of the skipped bins are zero).
    HTK 80-mel bank: 104 LDS cycles per tile for the power-row reads with the order alone, 76-84 with both, floor 72.
    Returns (image float32[rows * (ws + 2) + 8], order int32[rows])."""
    lo = np.asarray(lo, dtype=np.int64)
    width = np.asarray(width, dtype=np.int64)
    n_mels = lo.shape[0]
    n_rounds = (n_mels + 19) // 20
    rows = n_rounds * 20
    w4 = (int(max_width) + 1 + 3) & ~3
    ws = w4 if (w4 >> 2) & 1 else w4 + 4
    img = np.zeros(rows * (ws + 2) + 8, dtype=np.float32)
    ints = img.view(np.int32)
    lo2_o, mel_o, rc_o = rows * ws, rows * ws + rows, rows * ws + 2 * rows
    order = np.full(rows, -1, dtype=np.int32)
    rng = np.random.default_rng(seed)
    for r in range(n_rounds):
        mels = list(range(20 * r, min(20 * r + 20, n_mels)))
        rw = max([4] + [int((width[m] + (lo[m] & 1) + 3) & ~3) for m in mels])
        opts = {}
        for m in mels:
            l2 = min(int(lo[m]) & ~1, _M400_PK - rw)
            opts[m] = []
            while l2 >= 0 and int(lo[m]) + int(width[m]) <= l2 + rw:
                opts[m].append(l2)
                l2 -= 2
        opts[-1] = [0]
        best = None
        for restart in range(3):
            perm = mels + [-1] * (20 - len(mels))
            if restart:
                rng.shuffle(perm)
            start = {m: opts[m][0] for m in set(perm)}
            c = _m400_cost_fast([start[m] for m in perm])
            for _ in range(iters):
                if c <= len(_B128_LANES):
                    break
                if rng.random() < 0.5:
                    i, j = rng.choice(20, size=2, replace=False)
                    cand, st = list(perm), start
                    cand[i], cand[j] = cand[j], cand[i]
                else:
                    m = perm[int(rng.integers(20))]
                    st = dict(start)
                    st[m] = opts[m][int(rng.integers(len(opts[m])))]
                    cand = perm
                cc = _m400_cost_fast([st[m] for m in cand])
                if cc <= c:
                    perm, start, c = cand, st, cc
            if best is None or c < best[0]:
                best = (c, perm, start)
            if c <= len(_B128_LANES):
                break
        _, perm, start = best
        ints[rc_o + r] = rw >> 2
        for i, m in enumerate(perm):
            row = 20 * r + i
            order[row] = m
            ints[mel_o + row] = m
            if m < 0:
                ints[lo2_o + row] = 0
                continue
            l2 = start[m]
            ints[lo2_o + row] = l2
            off = int(lo[m]) - l2
            img[row * ws + off: row * ws + off + int(width[m])] = weights[m, :int(width[m])]
    return img, order


def resample_adjoint_table(kernel: np.ndarray, orig: int, new: int, width: int):
    """Tap table of the ADJOINT of `y[q new + p] = sum_k h[p][k] xpad[q orig + k]` written as the same kind of
    polyphase operator with the rates swapped (orig' = new, new' = orig):
        dx[q' orig + r] = sum_{k'} h'[r][k'] dypad[q' new + k'],   dypad = [0]*W' ++ dy ++ [0]*(W' + new),
        W' = A new,  A = ceil(width / orig),  k' = (A - d) new + p  <->  h[p][d orig + r + width]
    (d = q' - q is the block offset between the input sample and the output block it contributed to).
    Returns (h' float32[orig][2 W' + new], W')."""
    h = np.asarray(kernel)
    if h.dtype not in (np.float32, np.float64):
        h = h.astype(np.float32)
    h = h.reshape(new, -1)
    taps = h.shape[1]
    assert taps == 2 * width + orig
    A = -(-width // orig)
    Wp = A * new
    tp = 2 * Wp + new
    out = np.zeros((orig, tp), dtype=h.dtype)
    r = np.arange(orig)[:, None]
    kp = np.arange(tp)[None, :]
    d = A - kp // new
    p = kp % new
    k = d * orig + r + width
    ok = (k >= 0) & (k < taps)
    out[ok] = h[np.broadcast_to(p, k.shape)[ok], k[ok]]
    return out, Wp


# --------------------------------------------------------------------------- #
# Kaldi-compatible constants (reference: compliance/kaldi.py:86-113, 318-511)   #
# --------------------------------------------------------------------------- #


def kaldi_window(window_type: str, window_size: int, blackman_coeff: float) -> torch.Tensor:
    """The window functions of compliance/kaldi.py:86-113 (float32, symmetric = periodic=False)."""
    if window_type == "hanning":
        return torch.hann_window(window_size, periodic=False)
    if window_type == "hamming":
        return torch.hamming_window(window_size, periodic=False, alpha=0.54, beta=0.46)
    if window_type == "povey":                       # hann ** 0.85: goes to zero at the edges
        return torch.hann_window(window_size, periodic=False).pow(0.85)
    if window_type == "rectangular":
        return torch.ones(window_size)
    if window_type == "blackman":
        a = 2 * math.pi / (window_size - 1)
        n = torch.arange(window_size, dtype=torch.float32)
        return blackman_coeff - 0.5 * torch.cos(a * n) + (0.5 - blackman_coeff) * torch.cos(2 * a * n)
    raise Exception("Invalid window type " + window_type)


def kaldi_mel_scale_scalar(freq: float) -> float:
    return 1127.0 * math.log(1.0 + freq / 700.0)                     # kaldi.py:326-327


def kaldi_mel_scale(freq: torch.Tensor) -> torch.Tensor:
    return 1127.0 * (1.0 + freq / 700.0).log()                       # kaldi.py:330-331


def kaldi_inverse_mel_scale_scalar(mel_freq: float) -> float:
    return 700.0 * (math.exp(mel_freq / 1127.0) - 1.0)               # kaldi.py:318-319


def kaldi_inverse_mel_scale(mel_freq: torch.Tensor) -> torch.Tensor:
    return 700.0 * ((mel_freq / 1127.0).exp() - 1.0)                 # kaldi.py:322-323


def kaldi_vtln_warp_freq(vtln_low_cutoff: float, vtln_high_cutoff: float, low_freq: float, high_freq: float,
                         vtln_warp_factor: float, freq: torch.Tensor) -> torch.Tensor:
    """Piecewise-linear VTLN warp with two inflection points l < h (kaldi.py:334-406): identity outside
    [low_freq, high_freq], slope 1/warp between l and h, and straight lines joining (low_freq, low_freq) to (l, l/warp)
    and (h, h/warp) to (high_freq, high_freq)."""
    assert vtln_low_cutoff > low_freq, "be sure to set the vtln_low option higher than low_freq"
    assert vtln_high_cutoff < high_freq, "be sure to set the vtln_high option lower than high_freq [or negative]"
    lo_pt = vtln_low_cutoff * max(1.0, vtln_warp_factor)
    hi_pt = vtln_high_cutoff * min(1.0, vtln_warp_factor)
    slope = 1.0 / vtln_warp_factor
    assert lo_pt > low_freq and hi_pt < high_freq
    slope_left = (slope * lo_pt - low_freq) / (lo_pt - low_freq)
    slope_right = (high_freq - slope * hi_pt) / (high_freq - hi_pt)
    warped = torch.empty_like(freq)
    # later assignments win, exactly in the reference's order: >= h, < h, < l, outside
    upper = torch.ge(freq, hi_pt)
    warped[upper] = high_freq + slope_right * (freq[upper] - high_freq)
    mid = torch.lt(freq, hi_pt)
    warped[mid] = slope * freq[mid]
    lower = torch.lt(freq, lo_pt)
    warped[lower] = low_freq + slope_left * (freq[lower] - low_freq)
    outside = torch.lt(freq, low_freq) | torch.gt(freq, high_freq)
    warped[outside] = freq[outside]
    return warped


def kaldi_vtln_warp_mel_freq(vtln_low_cutoff: float, vtln_high_cutoff: float, low_freq, high_freq: float,
                             vtln_warp_factor: float, mel_freq: torch.Tensor) -> torch.Tensor:
    return kaldi_mel_scale(kaldi_vtln_warp_freq(vtln_low_cutoff, vtln_high_cutoff, low_freq, high_freq, vtln_warp_factor,
                                                kaldi_inverse_mel_scale(mel_freq)))      # kaldi.py:409-433


def kaldi_get_mel_banks(num_bins: int, window_length_padded: int, sample_freq: float, low_freq: float, high_freq: float,
                        vtln_low: float, vtln_high: float, vtln_warp_factor: float):
    """Triangular banks that are linear in MEL (kaldi.py:436-511).  Returns (bins (num_bins, padded / 2), center_freqs)."""
    assert num_bins > 3, "Must have at least 3 mel bins"
    assert window_length_padded % 2 == 0
    num_fft_bins = window_length_padded / 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    assert (0.0 <= low_freq < nyquist) and (0.0 < high_freq <= nyquist) and (low_freq < high_freq), (
        "Bad values in options: low-freq {} and high-freq {} vs. nyquist {}".format(low_freq, high_freq, nyquist))
    fft_bin_width = sample_freq / window_length_padded
    mel_lo = kaldi_mel_scale_scalar(low_freq)
    mel_hi = kaldi_mel_scale_scalar(high_freq)
    step = (mel_hi - mel_lo) / (num_bins + 1)
    if vtln_high < 0.0:
        vtln_high += nyquist
    assert vtln_warp_factor == 1.0 or ((low_freq < vtln_low < high_freq) and (0.0 < vtln_high < high_freq) and
                                       (vtln_low < vtln_high)), (
        "Bad values in options: vtln-low {} and vtln-high {}, versus " "low-freq {} and high-freq {}".format(
            vtln_low, vtln_high, low_freq, high_freq))
    idx = torch.arange(num_bins).unsqueeze(1)
    left = mel_lo + idx * step
    center = mel_lo + (idx + 1.0) * step
    right = mel_lo + (idx + 2.0) * step
    if vtln_warp_factor != 1.0:
        left = kaldi_vtln_warp_mel_freq(vtln_low, vtln_high, low_freq, high_freq, vtln_warp_factor, left)
        center = kaldi_vtln_warp_mel_freq(vtln_low, vtln_high, low_freq, high_freq, vtln_warp_factor, center)
        right = kaldi_vtln_warp_mel_freq(vtln_low, vtln_high, low_freq, high_freq, vtln_warp_factor, right)
    center_freqs = kaldi_inverse_mel_scale(center)
    mel = kaldi_mel_scale(fft_bin_width * torch.arange(num_fft_bins)).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    if vtln_warp_factor == 1.0:
        bins = torch.max(torch.zeros(1), torch.min(up, down))
    else:                                              # warped banks: the two slopes are gated separately
        bins = torch.zeros_like(up)
        rising = torch.gt(mel, left) & torch.le(mel, center)
        falling = torch.gt(mel, center) & torch.lt(mel, right)
        bins[rising] = up[rising]
        bins[falling] = down[falling]
    return bins, center_freqs


# --------------------------------------------------------------------------- #
# lfilter: orders 3 .. 8 as second-order sections                             #
# --------------------------------------------------------------------------- #


def _real_factors(roots: np.ndarray, tol: float = 1e-9):
    """Roots of a real polynomial -> real factor polynomials in z^-1: [1, -2 Re r, |r|^2] per conjugate pair and
    [1, -r] per real root (lower delays first).  None when the roots do not pair up."""
    roots = list(roots)
    quad, lin = [], []
    while roots:
        r = roots.pop()
        if abs(r.imag) <= tol * max(1.0, abs(r)):
            lin.append((np.array([1.0, -r.real]), abs(r.real)))
            continue
        j = int(np.argmin([abs(q - np.conj(r)) for q in roots])) if roots else -1
        if j < 0 or abs(roots[j] - np.conj(r)) > 1e-6 * max(1.0, abs(r)):
            return None
        roots.pop(j)
        quad.append((np.array([1.0, -2.0 * r.real, abs(r) ** 2]), abs(r)))
    return quad, lin


def _pack_quadratics(quad, lin):
    """Linear factors two at a time into quadratics (largest radii together), every polynomial padded to 3 coefficients."""
    lin = sorted(lin, key=lambda t: -t[1])
    out = list(quad)
    for i in range(0, len(lin) - 1, 2):
        out.append((np.convolve(lin[i][0], lin[i + 1][0]), max(lin[i][1], lin[i + 1][1])))
    if len(lin) % 2:
        out.append((np.array([lin[-1][0][0], lin[-1][0][1], 0.0]), lin[-1][1]))
    return out


def _direct_form(b, a, x):
    """float64 difference equation (host check only)."""
    try:
        from scipy.signal import lfilter as _sl
        return _sl(b, a, x)
    except Exception:  # pragma: no cover - scipy is part of the image
        y = np.zeros_like(x)
        for n in range(x.size):
            acc = sum(b[k] * x[n - k] for k in range(len(b)) if n >= k)
            acc -= sum(a[k] * y[n - k] for k in range(1, len(a)) if n >= k)
            y[n] = acc / a[0]
        return y


def lfilter_sos(a: np.ndarray, b: np.ndarray, max_pole_radius: float = 0.99, rel_tol: float = 2e-6):
    """(rows, n_order) float32 coefficient rows of filters of order 3 .. 16 -> second-order sections
    (a_s, b_s) float32 (n_sections, rows, 3) for the cascade kernels (clamp mode 2), or None.

    Poles and zeros from the float64 companion matrices, conjugates kept together, every pole section paired with the
    nearest remaining zero section, sections ordered by pole radius (the sharpest resonance last, as scipy's zpk2sos
    does).  The factorisation is only used when it REPRODUCES the filter: the float64 impulse response of the cascade of
    the float32-rounded sections must equal that of the direct form to rel_tol (in l1 over 8192 samples), and no pole may
    lie outside max_pole_radius (beyond it float32 section coefficients move the resonance noticeably).  Anything else --
    unstable, near-unstable, clustered roots -- keeps the general-order kernel."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    rows, n_order = a.shape
    if not (4 <= n_order <= 17):      # <= 8 sections: what one launch of the cascade kernels takes (lfw::kMaxCascade)
        return None
    n_sec = n_order // 2            # ceil(order / 2), order = n_order - 1
    a_s = np.zeros((n_sec, rows, 3))
    b_s = np.zeros((n_sec, rows, 3))
    imp = np.zeros(8192)
    imp[0] = 1.0
    for r in range(rows):
        if a[r, 0] == 0.0 or not np.all(np.isfinite(a[r])) or not np.all(np.isfinite(b[r])):
            return None
        an, bn = a[r] / a[r, 0], b[r] / a[r, 0]
        nz = np.nonzero(bn)[0]
        if nz.size == 0:
            return None
        d = int(nz[0])                                   # leading zeros of b: a pure delay
        bt = np.trim_zeros(bn[d:], "b")
        at = np.trim_zeros(an, "b")
        poles = np.roots(at) if at.size > 1 else np.array([])
        zeros = np.roots(bt) if bt.size > 1 else np.array([])
        if poles.size and np.abs(poles).max() > max_pole_radius:
            return None
        pf, zf = _real_factors(poles), _real_factors(zeros)
        if pf is None or zf is None:
            return None
        den = _pack_quadratics(*pf)
        zq, zl = zf
        zl = zl + [(np.array([0.0, 1.0]), 0.0)] * d      # z^-1 factors of the delay
        num = _pack_quadratics(zq, zl)
        if len(den) > n_sec or len(num) > n_sec:
            return None
        den += [(np.array([1.0, 0.0, 0.0]), 0.0)] * (n_sec - len(den))
        num += [(np.array([1.0, 0.0, 0.0]), -1.0)] * (n_sec - len(num))
        den.sort(key=lambda t: t[1])                     # sharpest resonance last
        # nearest zero section for the sharpest poles first (root radius as the distance: conjugates share it)
        order = sorted(range(n_sec), key=lambda i: -den[i][1])
        left = list(range(n_sec))
        pick = [None] * n_sec
        for i in order:
            j = min(left, key=lambda k: abs(num[k][1] - den[i][1]))
            left.remove(j)
            pick[i] = j
        gain = bt[0]
        for i in range(n_sec):
            a_s[i, r] = den[i][0]
            b_s[i, r] = num[pick[i]][0] * (gain if i == 0 else 1.0)
        # the check: cascade of the float32-ROUNDED sections against the direct form, both in float64
        y = imp
        for i in range(n_sec):
            y = _direct_form(b_s[i, r].astype(np.float32).astype(np.float64), a_s[i, r].astype(np.float32).astype(np.float64), y)
        ref = _direct_form(bn, an, imp)
        scale = np.abs(ref).sum()
        if not np.isfinite(scale) or scale == 0.0 or np.abs(y - ref).sum() > rel_tol * scale:
            return None
    return a_s.astype(np.float32), b_s.astype(np.float32)

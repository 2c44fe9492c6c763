"""Build + load the CPU replay of the HIP kernels' phase functions (tests/cpu_sim/sim.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from audio_amd import _host, _lib

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu_sim", "sim.cpp")
OUT = os.path.join(HERE, "cpu_sim", "_build", "libaamd_sim.so")
CSRC = os.path.join(os.path.dirname(HERE), "audio_amd", "csrc")

_sim = None


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.getmtime(d) > t for d in deps)


def sim():
    global _sim
    if _sim is None:
        if _stale():
            os.makedirs(os.path.dirname(OUT), exist_ok=True)
            tmp = "%s.tmp.%d" % (OUT, os.getpid())     # xdist workers build side by side: each its own file, renamed into place
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", SRC, "-o", tmp])
            os.replace(tmp, OUT)
        _sim = C.CDLL(OUT)
    return _sim


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


class HostBands:
    def __init__(self, fb, permute=False, image=False):
        self.lo, self.width, self.weights, self.max_width = _host.mel_band_table(np.asarray(fb))
        self.lo = np.ascontiguousarray(self.lo)
        self.width = np.ascontiguousarray(self.width)
        self.weights = np.ascontiguousarray(self.weights)
        self.n_mels = self.lo.shape[0]
        self.lane_order = np.ascontiguousarray(_host.mel_lane_order(self.lo, self.width)) if permute else None
        self.image = None
        if image:       # the host-built LDS image (order + shifted band starts), as audio_amd.functional uploads it
            self.image, order = _host.mel400_table_image(self.lo, self.width, self.weights, self.max_width)
            self.image = np.ascontiguousarray(self.image)
            self.lane_order = np.ascontiguousarray(order)
        self.struct = _lib.MelBands(self.n_mels, self.max_width, self.lo.ctypes.data, self.width.ctypes.data,
                                    self.weights.ctypes.data,
                                    self.lane_order.ctypes.data if self.lane_order is not None else None,
                                    self.image.ctypes.data if self.image is not None else None)


def make_desc(rows, length, n_fft, hop, pad=0, center=True, pad_mode="reflect", onesided=True, scale=1.0,
              power=2.0):
    T = _host.frame_count(length, n_fft, hop, center, pad)
    return _lib.StftDesc(rows, length, length, n_fft, hop, pad, int(center), _lib.PAD_MODES[pad_mode],
                         int(onesided), T, scale, 0.0 if power is None else float(power))


def sim_spectrogram(x, window_padded, desc):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(window_padded, dtype=np.float32)
    tw = np.ascontiguousarray(_host.twiddle_table(desc.n_fft))
    n_freq = desc.n_fft // 2 + 1 if desc.onesided else desc.n_fft
    comp = 2 if desc.power <= 0 else 1
    out = np.zeros((desc.rows, desc.n_frames, n_freq * comp), dtype=np.float32)
    rc = sim().sim_stft_generic(fptr(x), fptr(w), fptr(tw), None, fptr(out), C.byref(desc), 0)
    assert rc == 0
    if comp == 2:
        out = out.reshape(desc.rows, desc.n_frames, n_freq, 2)
        out = out[..., 0] + 1j * out[..., 1]
    return np.swapaxes(out, -1, -2)


def sim_mel_generic(x, window_padded, bands, desc):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(window_padded, dtype=np.float32)
    tw = np.ascontiguousarray(_host.twiddle_table(desc.n_fft))
    out = np.zeros((desc.rows, desc.n_frames, bands.n_mels), dtype=np.float32)
    rc = sim().sim_stft_generic(fptr(x), fptr(w), fptr(tw), C.byref(bands.struct), fptr(out), C.byref(desc), 1)
    assert rc == 0
    return np.swapaxes(out, -1, -2)


def _sim_fft400(x, window, bands, scale, epi, db=None, gmax=None, rows_per_group=1, power=2.0, out_width=None,
                wide=0, hop=160, out_frames=None, i16=False):
    """i16: False = float rows, True = planar int16 rows, 2 = interleaved 16-bit stereo (clips, time, 2): the kernel's rows are
    then (clip, channel) pairs."""
    x = np.ascontiguousarray(x, dtype=np.int16 if i16 else np.float32)
    if i16 == 2:
        assert x.ndim == 3 and x.shape[-1] == 2
        rows, length = 2 * x.shape[0], x.shape[1]
    else:
        rows, length = x.shape
    w = np.ascontiguousarray(window, dtype=np.float32)
    tw = np.ascontiguousarray(_host.twiddle_table(400))
    T = _host.frame_count(length, 400, hop, True)
    out = np.zeros((rows, out_frames or T, out_width), dtype=np.float32)
    f = sim().sim_melspec400
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                  C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int, C.c_int, C.c_int]
    dbv = None if db is None else fptr(np.ascontiguousarray(db, dtype=np.float32))
    rc = f(x.ctypes.data_as(C.c_void_p), fptr(w), fptr(tw),
           None if bands is None else C.cast(C.byref(bands.struct), C.c_void_p),
           fptr(out), rows, length, length, T, scale, epi, dbv, None if gmax is None else fptr(gmax),
           rows_per_group, power, wide, hop, int(i16))
    assert rc == 0
    return np.swapaxes(out, -1, -2)


def sim_mel400(x, window, bands, scale=1.0, wide=0, hop=160, i16=False):
    return _sim_fft400(x, window, bands, scale, 0, out_width=bands.n_mels, wide=wide, hop=hop, i16=i16)


def sim_mel400_db(x, window, bands, multiplier, amin, db_multiplier, gmax, rows_per_group, scale=1.0):
    """gmax: float32 array pre-filled with -inf, max-reduced in place."""
    return _sim_fft400(x, window, bands, scale, 1, db=[multiplier, amin, multiplier * db_multiplier], gmax=gmax,
                       rows_per_group=rows_per_group, out_width=bands.n_mels)


def sim_mfcc_fused(x, window, bands, dct, multiplier, amin, db_multiplier, top_db, rows_per_group, hop=160):
    """The fused MFCC kernel's two launches: pass 0 (unclamped MFCC + group maxima + tile minima), pass 1 (fix-up of the
    tiles under the cut-off).  Returns (mfcc (rows, n_mfcc, T), group_max, tiles redone)."""
    dct = np.ascontiguousarray(dct, dtype=np.float32)
    n_mels, n_mfcc = dct.shape
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, length = x.shape
    T = _host.frame_count(length, 400, hop, True)
    n_groups = -(-rows // rows_per_group)
    gmax = np.full(n_groups, -np.inf, dtype=np.float32)
    tmin = np.zeros(rows * (-(-T // 6)), dtype=np.float32)
    setup = sim().sim_mfcc_fused_setup
    setup.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int]
    w = np.ascontiguousarray(window, dtype=np.float32)
    tw = np.ascontiguousarray(_host.twiddle_table(400))
    out = np.zeros((rows, T, n_mfcc), dtype=np.float32)
    f = sim().sim_melspec400
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                  C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int, C.c_int, C.c_int]
    dbv = np.ascontiguousarray([multiplier, amin, multiplier * db_multiplier], dtype=np.float32)
    for fixup in (0, 1):
        assert setup(fptr(dct), n_mels, n_mfcc, top_db, fptr(tmin), fixup) == 0
        rc = f(x.ctypes.data_as(C.c_void_p), fptr(w), fptr(tw), C.cast(C.byref(bands.struct), C.c_void_p), fptr(out), rows,
               length, length, T, 1.0, 4, fptr(dbv), fptr(gmax), rows_per_group, 2.0, 0, hop, 0)
        assert rc == 0, rc
    return np.swapaxes(out, -1, -2), gmax, sim().sim_mfcc_fused_fix_count()


def sim_mel400_norm(x, window, bands, gain, mean, invstddev, right_padding=0, hop=160, i16=False, scale=1.0):
    """Fused RNN-T feature epilogue: ((plog(mel * gain)) - mean) * invstddev, rows of T + right_padding frames
    (the padding rows stay zero).  Returns frame-major (rows, T + right_padding, n_mels)."""
    x = np.ascontiguousarray(x, dtype=np.int16 if i16 else np.float32)
    T = _host.frame_count(x.shape[1], 400, hop, True)       # (clips, time, 2) for interleaved stereo: shape[1] is time
    stats = np.ascontiguousarray(np.concatenate([mean, invstddev]), dtype=np.float32)
    o = _sim_fft400(x, window, bands, scale, 3, db=[gain, float(T + right_padding)], gmax=stats, out_width=bands.n_mels,
                    hop=hop, out_frames=T + right_padding, i16=i16)
    return np.swapaxes(o, -1, -2)


def sim_spec400(x, window, power, scale=1.0, hop=160):
    if power is None:      # complex: interleaved (re, im) rows
        o = _sim_fft400(x, window, None, scale, 2, power=0.0, out_width=402, hop=hop)     # (rows, 402, T)
        o = np.swapaxes(o, -1, -2)
        return np.swapaxes(o[..., 0::2] + 1j * o[..., 1::2], -1, -2)
    return _sim_fft400(x, window, None, scale, 2, power=power, out_width=201, hop=hop)


def sim_mfcc_dct_mfma(mel_fm, dct, log_mode, group_max=None, vec_per_group=1, top_db=-1.0):
    """mel_fm: (n_vec, n_mels) frame-major; dct: (n_mels, n_mfcc)."""
    mel_fm = np.ascontiguousarray(mel_fm, dtype=np.float32)
    dct = np.ascontiguousarray(dct, dtype=np.float32)
    n_vec, n_mels = mel_fm.shape
    n_mfcc = dct.shape[1]
    out = np.zeros((n_vec, n_mfcc), dtype=np.float32)
    f = sim().sim_mfcc_dct_mfma
    f.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_float]
    gm = None if group_max is None else fptr(np.ascontiguousarray(group_max, dtype=np.float32))
    assert f(fptr(mel_fm), fptr(dct), fptr(out), n_vec, n_mels, n_mfcc, log_mode, gm, vec_per_group, top_db) == 0
    return out


def sim_resample(x, kernel, orig, new, width, qt=None, use_lds=1):
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, length = x.shape
    k = np.ascontiguousarray(kernel, dtype=np.float32).reshape(new, -1)
    out_len = -(-new * length // orig)
    out = np.zeros((rows, out_len), dtype=np.float32)
    if qt is None:
        qt = max(1, -(-2048 // new))
    f = sim().sim_resample
    f.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 3 + [C.c_int] * 3 + [C.c_int64, C.c_int, C.c_int]
    assert f(fptr(x), fptr(k), fptr(out), rows, length, length, orig, new, width, out_len, qt, use_lds) == 0
    return out


def sim_resample_sparse(x, kernel, orig, new, width):
    """resample_sparse_kernel replay over the host-compacted table (_host.resample_sparse_table)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, length = x.shape
    k = np.ascontiguousarray(kernel, dtype=np.float32).reshape(new, -1)
    hb, lo, span = _host.resample_sparse_table(k)
    out_len = -(-new * length // orig)
    out = np.full((rows, out_len), np.nan, dtype=np.float32)
    f = sim().sim_resample_sparse
    f.argtypes = [C.c_void_p] * 4 + [C.c_int64] * 3 + [C.c_int] * 4 + [C.c_int64]
    assert f(fptr(x), fptr(hb), fptr(np.ascontiguousarray(lo, dtype=np.int32)), fptr(out), rows, length, length, orig, new,
             width, span, out_len) == 0
    return out, span


def sim_resample_mfma(x, kernel, orig, new, width, vec_ok=1, f16=0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    rows, length = x.shape
    k = np.ascontiguousarray(kernel, dtype=np.float32).reshape(new, -1)
    out_len = -(-new * length // orig)
    out = np.full((rows, out_len), np.nan, dtype=np.float32)
    lo, span = _host.resample_band_table(k)
    lo = np.ascontiguousarray(lo, dtype=np.int32)
    f = sim().sim_resample_mfma
    f.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 3 + [C.c_int] * 3 + [C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int]
    rc = f(fptr(x), fptr(k), fptr(out), rows, length, length, orig, new, width, out_len, fptr(lo), span, vec_ok, f16)
    return rc, out


def sim_lfilter(x, a, b, clamp=True):
    """x: (batch, channels, L); a, b: (stages, rows, order+1)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    batch, ch, L = x.shape
    y = np.zeros_like(x)
    f = sim().sim_lfilter
    f.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    assert f(fptr(x), fptr(a), fptr(b), fptr(y), batch, ch, L, a.shape[-1], a.shape[-2], a.shape[0], int(clamp)) == 0
    return y


def sim_lfilter_wave(x, a, b, clamp=True, waves=1):
    """Wave-per-sequence biquad kernel.  x: (batch, channels, L); a, b: (stages, rows, order+1), order <= 2."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    batch, ch, L = x.shape
    y = np.full_like(x, np.nan)
    f = sim().sim_lfilter_wave
    f.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    rc = f(fptr(x), fptr(a), fptr(b), fptr(y), batch, ch, L, a.shape[-1], a.shape[-2], a.shape[0], int(clamp), waves)
    return rc, y


def sim_fftconv(x, y, start, out_len, xmap=None, ymap=None, rows=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    nx, ny = x.shape[-1], y.shape[-1]
    rows = rows if rows is not None else x.shape[0]
    out = np.zeros((rows, out_len), dtype=np.float32)
    f = sim().sim_fftconv
    f.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 3 + [C.c_void_p] * 2 + [C.c_int64] * 2
    xm = None if xmap is None else fptr(np.ascontiguousarray(xmap, dtype=np.int64))
    ym = None if ymap is None else fptr(np.ascontiguousarray(ymap, dtype=np.int64))
    assert f(fptr(x), fptr(y), fptr(out), rows, nx, ny, xm, ym, start, out_len) == 0
    return out


def sim_fftconv_os(x, y, start, out_len, xmap=None, ymap=None, rows=None, cu_count=0, plan=None):
    """Overlap-save path (16384-point LDS FFT).  x: (n_x_rows, nx), y: (n_y_rows, ny).  cu_count > 0 lets the launcher's
    cost model pick the frequency-domain delay-line plan as it would on a device with that many CUs; `plan`, a list, receives
    "fdl" or "recompute"."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    nx, ny = x.shape[-1], y.shape[-1]
    rows = rows if rows is not None else x.shape[0]
    out = np.full((rows, out_len), np.nan, dtype=np.float32)
    f = sim().sim_fftconv_os
    f.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 5 + [C.c_void_p] * 2 + [C.c_int64] * 2 + [C.c_int]
    xm = None if xmap is None else fptr(np.ascontiguousarray(xmap, dtype=np.int64))
    ym = None if ymap is None else fptr(np.ascontiguousarray(ymap, dtype=np.int64))
    rc = f(fptr(x), fptr(y), fptr(out), rows, x.shape[0], y.shape[0], nx, ny, xm, ym, start, out_len, cu_count)
    assert rc in (0, 1)
    if plan is not None:
        plan.append("fdl" if rc == 1 else "recompute")
    return out


def sim_fftconv_fdr(x, y, start, out_len, xmap=None, ymap=None, rows=None, cu_count=256):
    """The real-block delay line (csrc/fftconv_fdr.h, plan 3 of aamd_fftconvolve_f32), replayed; None when the plan does
    not serve the shape."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    nx, ny = x.shape[-1], y.shape[-1]
    rows = rows if rows is not None else x.shape[0]
    out = np.full((rows, out_len), np.nan, dtype=np.float32)
    f = sim().sim_fftconv_fdr
    f.argtypes = [C.c_void_p] * 3 + [C.c_int64] * 5 + [C.c_void_p] * 2 + [C.c_int64] * 2 + [C.c_int]
    xm = None if xmap is None else fptr(np.ascontiguousarray(xmap, dtype=np.int64))
    ym = None if ymap is None else fptr(np.ascontiguousarray(ymap, dtype=np.int64))
    rc = f(fptr(x), fptr(y), fptr(out), rows, x.shape[0], y.shape[0], nx, ny, xm, ym, start, out_len, cu_count)
    return out if rc == 1 else None


def sim_istft(spec_fm, window_padded, length, n_fft, hop, center=True, pad_mode="constant", pad=0, scale=1.0,
              adjoint=False, inv_env=None, pow2=False, runs=0, fast400=False, trace=None):
    """trace: optional dict; filled with 'stores' / 'adds' int32 arrays (rows, length): plain stores / atomic adds per sample."""
    """spec_fm: complex (rows, T, n_freq) frame-major.  Returns (rows, length)."""
    spec_fm = np.ascontiguousarray(spec_fm, dtype=np.complex64)
    rows, T, n_freq = spec_fm.shape
    w = np.ascontiguousarray(window_padded, dtype=np.float32)
    tw = np.ascontiguousarray(_host.twiddle_table(n_fft))
    out = np.zeros((rows, length), dtype=np.float32)
    desc = _lib.StftDesc(rows, length, length, n_fft, hop, pad, int(center), _lib.PAD_MODES[pad_mode], 1, T, scale, 0.0)
    ie = None if inv_env is None else fptr(np.ascontiguousarray(inv_env, dtype=np.float32))
    if trace is not None:
        trace["stores"] = np.zeros((rows, length), dtype=np.int32)
        trace["adds"] = np.zeros((rows, length), dtype=np.int32)
        tf = sim().sim_trace_writes
        tf.argtypes = [C.c_void_p] * 3
        tf.restype = None
        tf(fptr(out), trace["stores"].ctypes.data_as(C.c_void_p), trace["adds"].ctypes.data_as(C.c_void_p))
    try:
        return _sim_istft_dispatch(spec_fm, w, tw, ie, out, desc, adjoint, pow2, runs, fast400)
    finally:
        if trace is not None:
            sim().sim_trace_writes(None, None, None)


def _sim_istft_dispatch(spec_fm, w, tw, ie, out, desc, adjoint, pow2, runs, fast400):
    if fast400:
        f = sim().sim_istft400
        f.argtypes = [C.c_void_p] * 5 + [C.POINTER(_lib.StftDesc), C.c_int]
        assert f(spec_fm.view(np.float32).ctypes.data_as(C.c_void_p), fptr(w), fptr(tw), ie, fptr(out), C.byref(desc),
                 int(adjoint)) == 0
        return out
    if pow2:
        f = sim().sim_istft_pow2
        f.argtypes = [C.c_void_p] * 5 + [C.POINTER(_lib.StftDesc), C.c_int, C.c_int]
        assert f(spec_fm.view(np.float32).ctypes.data_as(C.c_void_p), fptr(w), fptr(tw), ie, fptr(out), C.byref(desc),
                 int(adjoint), int(runs)) == 0
        return out
    f = sim().sim_istft
    f.argtypes = [C.c_void_p] * 5 + [C.POINTER(_lib.StftDesc), C.c_int]
    assert f(spec_fm.view(np.float32).ctypes.data_as(C.c_void_p), fptr(w), fptr(tw), ie, fptr(out), C.byref(desc),
             int(adjoint)) == 0
    return out


def sim_phase_vocoder(spec, rate, phase_advance):
    """spec: complex64 (rows, F, T) C-contiguous -> (rows, F, ceil(T / rate)), like F.phase_vocoder."""
    import math
    spec = np.ascontiguousarray(spec, dtype=np.complex64)
    rows, F, T = spec.shape
    n_out = int(math.ceil(T / rate))
    out = np.zeros((rows, F, n_out), dtype=np.complex64)
    pa = np.ascontiguousarray(phase_advance, dtype=np.float32).reshape(-1)
    d = _lib.VocoderDesc(rows, F, T, n_out, F * T, T, 1, F * n_out, n_out, 1, float(rate))
    f = sim().sim_phase_vocoder
    f.argtypes = [C.c_void_p] * 3 + [C.POINTER(_lib.VocoderDesc)]
    assert f(spec.view(np.float32).ctypes.data_as(C.c_void_p), fptr(pa), out.view(np.float32).ctypes.data_as(C.c_void_p),
             C.byref(d)) == 0
    return out


def sim_griffinlim_update(rebuilt, tprev, mag, momentum):
    """Returns (next, new tprev)."""
    rebuilt = np.ascontiguousarray(rebuilt, dtype=np.complex64)
    tprev = np.ascontiguousarray(tprev, dtype=np.complex64).copy()
    mag = np.ascontiguousarray(mag, dtype=np.float32)
    nxt = np.zeros_like(rebuilt)
    f = sim().sim_griffinlim_update
    f.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_float]
    vp = lambda a: a.view(np.float32).ctypes.data_as(C.c_void_p)   # noqa: E731
    assert f(vp(rebuilt), vp(tprev), fptr(mag), vp(nxt), rebuilt.size, float(momentum)) == 0
    return nxt, tprev


def sim_stft_pow2(x, window_padded, desc, bands=None):
    """stft_pow2.h replay (n_fft = 512 / 1024 / 2048, onesided).  Returns (rows, F or n_mels, T) like the modules."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(window_padded, dtype=np.float32)
    tw = np.ascontiguousarray(_host.twiddle_table(desc.n_fft))
    n_freq = desc.n_fft // 2 + 1
    comp = 2 if (bands is None and desc.power <= 0) else 1
    width = bands.n_mels if bands is not None else n_freq * comp
    out = np.zeros((desc.rows, desc.n_frames, width), dtype=np.float32)
    f = sim().sim_stft_pow2
    f.argtypes = [C.c_void_p] * 3 + [C.c_void_p, C.c_void_p, C.POINTER(_lib.StftDesc)]
    rc = f(fptr(x), fptr(w), fptr(tw), None if bands is None else C.cast(C.byref(bands.struct), C.c_void_p), fptr(out),
           C.byref(desc))
    assert rc == 0
    if comp == 2:
        out = out.reshape(desc.rows, desc.n_frames, n_freq, 2)
        out = out[..., 0] + 1j * out[..., 1]
    return np.swapaxes(out, -1, -2)


def sim_kaldi_features(x, window_padded, n_fft, shift, win, snip_edges=True, preemph=0.97, remove_dc=True, raw_energy=True,
                       energy_floor=1.0, use_power=True, use_log=True, bands=None, energy_col=-1, first_col=0, n_cols=None,
                       dither=0.0, noise=None, force_generic=False):
    """Replay for ONE waveform of kaldi_pow2_kernel (n_fft 256 ... 2048) or kgen::kaldi_generic_kernel (any other even
    size, or force_generic).  Returns (n_frames, n_cols)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = x.shape[0]
    m = (0 if n < win else 1 + (n - win) // shift) if snip_edges else (n + shift // 2) // shift
    if n_cols is None:
        n_cols = n_fft // 2 + 1 if bands is None else bands.n_mels
    out = np.zeros((m, n_cols), dtype=np.float32)
    w = np.ascontiguousarray(window_padded, dtype=np.float32)
    tw = np.ascontiguousarray(_host.twiddle_table(n_fft))
    if noise is not None:
        noise = np.ascontiguousarray(noise, dtype=np.float32)
        assert noise.shape == (m, win)
    d = _lib.KaldiDesc(n, m, n_fft, shift, win, int(snip_edges), preemph, int(remove_dc), int(raw_energy), energy_floor,
                       int(use_power), int(use_log), energy_col, first_col, n_cols, float(dither),
                       None if noise is None else noise.ctypes.data)
    f = sim().sim_kaldi_features
    f.argtypes = [C.c_void_p] * 3 + [C.c_void_p, C.c_void_p, C.POINTER(_lib.KaldiDesc)]
    sim().sim_set_force_generic(int(force_generic))
    try:
        assert f(fptr(x), fptr(w), fptr(tw), None if bands is None else C.cast(C.byref(bands.struct), C.c_void_p), fptr(out),
                 C.byref(d)) == 0
    finally:
        sim().sim_set_force_generic(0)
    return out

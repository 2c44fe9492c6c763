"""ctypes binding of libaudio_amd.so (the C ABI declared in include/audio_amd.h).

The shared library is built in-tree by ``python -m audio_amd._build`` (or
``__graft_entry__.build()``) with ``hipcc --offload-arch=gfx950``.  There is NO
fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaudio_amd.so")
if os.environ.get("AAMD_USE_LAB_LIB"):
    # tools/ only: the library with the tools-only kernel instantiations and their environment switches
    # (python -m audio_amd._build --lab; see audio_amd/_build.py).  The product never sets this.
    LIB_PATH = os.path.join(_HERE, "lib", "libaudio_amd_lab.so")

AAMD_OK = 0
PAD_MODES = {"reflect": 0, "constant": 1, "replicate": 2, "circular": 3}


class StftDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_int64), ("length", C.c_int64), ("row_stride", C.c_int64),
        ("n_fft", C.c_int32), ("hop", C.c_int32), ("pad", C.c_int32), ("center", C.c_int32),
        ("pad_mode", C.c_int32), ("onesided", C.c_int32), ("n_frames", C.c_int32),
        ("scale", C.c_float), ("power", C.c_float),
    ]


class MelBands(C.Structure):
    _fields_ = [
        ("n_mels", C.c_int32), ("max_width", C.c_int32),
        ("lo", C.c_void_p), ("width", C.c_void_p), ("weights", C.c_void_p), ("lane_order", C.c_void_p),
        ("table400", C.c_void_p), ("table_sig", C.c_int32),
    ]


class MfccFused(C.Structure):
    """aamd_mfcc_fused (include/audio_amd.h)."""
    _fields_ = [("dct_frag", C.c_void_p), ("n_mfcc", C.c_int32), ("pass_", C.c_int32), ("multiplier", C.c_float),
                ("amin", C.c_float), ("db_multiplier", C.c_float), ("top_db", C.c_float), ("group_max", C.c_void_p),
                ("rows_per_group", C.c_int64), ("tile_min", C.c_void_p), ("fix_count", C.c_void_p),
                ("tile_list", C.c_void_p)]


class ResampleBands(C.Structure):
    _fields_ = [("n_tiles", C.c_int32), ("tap_span", C.c_int32), ("tap_lo", C.POINTER(C.c_int32))]


_lib = None
_lock = threading.Lock()

_P = C.c_void_p
class VocoderDesc(C.Structure):
    """aamd_vocoder_desc (include/audio_amd.h)."""
    _fields_ = [("rows", C.c_int64), ("n_freq", C.c_int32), ("n_frames_in", C.c_int32), ("n_frames_out", C.c_int32),
                ("in_stride_row", C.c_int64), ("in_stride_freq", C.c_int64), ("in_stride_frame", C.c_int64),
                ("out_stride_row", C.c_int64), ("out_stride_freq", C.c_int64), ("out_stride_frame", C.c_int64),
                ("rate", C.c_double)]


class KaldiDesc(C.Structure):
    """aamd_kaldi_desc (include/audio_amd.h)."""
    _fields_ = [("n_samples", C.c_int64), ("n_frames", C.c_int64), ("n_fft", C.c_int32), ("shift", C.c_int32),
                ("win", C.c_int32), ("snip_edges", C.c_int32), ("preemphasis", C.c_float),
                ("remove_dc_offset", C.c_int32), ("raw_energy", C.c_int32), ("energy_floor", C.c_float),
                ("use_power", C.c_int32), ("use_log", C.c_int32), ("energy_col", C.c_int32), ("first_col", C.c_int32),
                ("n_cols", C.c_int32), ("dither", C.c_float), ("noise", C.c_void_p), ("n_utt", C.c_int64),
                ("utt_stride", C.c_int64)]


_SIGS = {
    "aamd_abi_version": (C.c_int, []),
    "aamd_last_error": (C.c_char_p, []),
    "aamd_set_kernel_policy": (C.c_int, [C.c_int]),
    "aamd_mel400_table_dwords": (C.c_int64, [C.c_int32, C.c_int32]),
    "aamd_mel400_table_build": (C.c_int, [C.POINTER(MelBands), _P, _P]),
    "aamd_device_info": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "aamd_box_probe": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "aamd_spectrogram_f32": (C.c_int, [_P, _P, _P, _P, C.POINTER(StftDesc), _P]),
    "aamd_melspectrogram_f32": (C.c_int, [_P, _P, _P, C.POINTER(MelBands), _P, C.POINTER(StftDesc), _P]),
    "aamd_melspectrogram_db_f32": (C.c_int, [_P, _P, _P, C.POINTER(MelBands), _P, C.POINTER(StftDesc), C.c_float,
                                             C.c_float, C.c_float, _P, C.c_int64, _P]),
    "aamd_melspectrogram_lognorm_f32": (C.c_int, [_P, _P, _P, C.POINTER(MelBands), _P, C.POINTER(StftDesc), C.c_float,
                                                  _P, _P, C.c_int64, _P]),
    "aamd_melspectrogram_pcm16_f32": (C.c_int, [_P, _P, _P, C.POINTER(MelBands), _P, C.POINTER(StftDesc), C.c_float,
                                                _P, _P, C.c_int64, _P]),
    "aamd_melspectrogram_pcm16_interleaved_f32": (C.c_int, [_P, C.c_int32, _P, _P, C.POINTER(MelBands), _P,
                                                            C.POINTER(StftDesc), C.c_float, _P, _P, C.c_int64, _P]),
    "aamd_melspectrogram_lowp_f32": (C.c_int, [_P, C.c_int32, _P, _P, C.POINTER(MelBands), _P, C.POINTER(StftDesc), _P]),
    "aamd_spectrogram_grad_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_float, _P]),
    "aamd_melspectrogram_grad_f32": (C.c_int, [_P, _P, C.POINTER(MelBands), C.c_int64, C.c_int32, C.c_int32, C.c_float, _P]),
    "aamd_kaldi_features_f32": (C.c_int, [_P, _P, _P, C.POINTER(MelBands), _P, C.POINTER(KaldiDesc), _P]),
    "aamd_istft_f32": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(StftDesc), C.c_int32, _P]),
    "aamd_phase_vocoder_f32": (C.c_int, [_P, _P, _P, C.POINTER(VocoderDesc), _P]),
    "aamd_griffinlim_update_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, _P]),
    "aamd_mel_scale_f32": (C.c_int, [_P, C.POINTER(MelBands), _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "aamd_amplitude_to_db_f32": (C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, _P, C.c_int64, _P]),
    "aamd_amplitude_to_db_clamped_f32": (C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, _P, C.c_int64,
                                                   C.c_float, _P]),
    "aamd_db_clamp_f32": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_float, _P]),
    "aamd_mfcc_dct_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64,
                                    C.c_float, _P]),
    "aamd_mfcc_frag_floats": (C.c_int32, []),
    "aamd_mfcc_fused_tiles": (C.c_int64, [C.POINTER(StftDesc)]),
    "aamd_mfcc_fused_supported": (C.c_int, [C.POINTER(StftDesc), C.POINTER(MelBands), C.c_int32]),
    "aamd_mfcc_frag_build": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "aamd_mfcc_fused_f32": (C.c_int, [_P, _P, _P, C.POINTER(MelBands), _P, C.POINTER(StftDesc), C.POINTER(MfccFused), _P]),
    "aamd_resample_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int64, _P]),
    "aamd_resample_banded_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int64, C.POINTER(ResampleBands), _P]),
    "aamd_resample_frag_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.POINTER(ResampleBands)]),
    "aamd_resample_frag_build_f32": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(ResampleBands), _P, _P]),
    "aamd_resample_prepared_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int64, C.POINTER(ResampleBands), _P, _P]),
    "aamd_resample_sparse_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int64, _P]),
    "aamd_lfilter_f32": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, _P]),
    "aamd_spectrogram_f64": (C.c_int, [_P, _P, _P, _P, C.POINTER(StftDesc), _P]),
    "aamd_istft_f64": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(StftDesc), C.c_int32, _P]),
    "aamd_lfilter_f64": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, _P]),
    "aamd_resample_f64": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int64, _P]),
    "aamd_fftconvolve_f64": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P,
                                       C.c_int64, C.c_int64, _P]),
    "aamd_fftconvolve_workspace": (C.c_int64, [C.c_int64] * 5),
    "aamd_fftconvolve_plan": (C.c_int, [C.c_int64] * 4),
    "aamd_fftconvolve_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P,
                                       C.c_int64, C.c_int64, _P, _P]),
    "aamd_fftconvolve_staged_f32": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _P,
                                              C.c_int64, C.c_int64, _P, C.c_int32, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)


def lib():
    """Load (once) and return the ctypes handle; raise loudly if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"audio_amd: {LIB_PATH} not found. Build the HIP extension first "
                    "(python -m audio_amd._build). There is no CPU fallback.")
            h = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(h, name)   # AttributeError if the ABI symbol is missing
                fn.restype = res
                fn.argtypes = args
            if h.aamd_abi_version() != 7:
                raise RuntimeError("audio_amd: ABI version mismatch between _lib.py and libaudio_amd.so")
            _lib = h
    return _lib


POLICY_FORCE_GENERIC, POLICY_MEL400_WIDE, POLICY_ISTFT_ATOMIC, POLICY_RESAMPLE_FP32 = 1, 2, 4, 8
POLICY_FFTCONV_NO_FDL, POLICY_FFTCONV_FDL, POLICY_FFTCONV_COMPLEX = 16, 32, 64
POLICY_RESAMPLE_B32, POLICY_MEL400_NO_POOL = 128, 256


class kernel_policy:
    """``with _lib.kernel_policy(_lib.POLICY_FORCE_GENERIC): ...`` -- tests / A-B runs pick the kernel family."""

    def __init__(self, flags: int):
        self.flags = flags

    def __enter__(self):
        self.prev = lib().aamd_set_kernel_policy(self.flags)
        return self

    def __exit__(self, *exc):
        lib().aamd_set_kernel_policy(self.prev)
        return False


EUNSUPPORTED = -2          # AAMD_EUNSUPPORTED: valid request this build cannot serve


def check(rc: int) -> None:
    if rc != AAMD_OK:
        msg = lib().aamd_last_error()
        raise RuntimeError((msg or b"audio_amd: unknown error").decode())


def ptr(t) -> int:
    """Device pointer of a torch tensor (0-size tensors give a dummy non-null pointer)."""
    return t.data_ptr() if t.numel() else 0


def current_stream(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream

"""audio_amd -- MI355X (gfx950) native kernels behind torchaudio's DSP transform API.

    import audio_amd.transforms as T      # Spectrogram, MelSpectrogram, MFCC, Resample, ...
    import audio_amd.functional as F      # spectrogram, resample, lfilter, fftconvolve, ...

Arithmetic runs in hand-written HIP kernels loaded from ``audio_amd/lib/libaudio_amd.so``
(C ABI: include/audio_amd.h).  No CPU fallback exists.
"""
from . import functional, transforms  # noqa: F401
from . import _ops  # noqa: F401  (registers torch.ops.audio_amd.*)

__version__ = "0.1.0"

"""Host-side caches of audio_amd.functional (no device needed): what is derived from a tensor lives exactly as long as the
tensor object, and the constants of a reduced-precision module are widened once per tensor."""
import gc

import torch

import audio_amd.functional as F


def test_a_dead_tensor_takes_its_cache_slot_with_it():
    t = torch.zeros(8)
    tid = id(t)
    assert F._tensor_cached(t, "derived", lambda: [1, 2, 3]) == [1, 2, 3]
    assert tid in F._TENSOR_CACHE
    del t
    gc.collect()
    assert tid not in F._TENSOR_CACHE


def test_a_new_tensor_on_a_reused_id_never_sees_the_old_slot():
    made = []
    for i in range(50):
        t = torch.full((4,), float(i))
        v = F._tensor_cached(t, "derived", lambda: made.append(i) or i)
        assert v == i
        del t
    assert made == list(range(50))


def test_reduced_precision_constants_are_widened_once_per_tensor():
    calls = []

    @F._reduced_precision_io
    def op(waveform, kernel):
        calls.append(kernel)
        return waveform * 2

    x = torch.ones(4, dtype=torch.float16)
    k = torch.ones(3, dtype=torch.float16)
    y1, y2 = op(x, k), op(x, k)
    assert y1.dtype == torch.float16 and torch.equal(y1, y2)
    assert calls[0].dtype == torch.float32 and calls[0] is calls[1]       # the same widened object: derived caches hit
    k.mul_(2.0)                                                           # an in-place update is seen
    op(x, k)
    assert calls[2] is not calls[0] and float(calls[2][0]) == 2.0


def test_speed_keeps_the_dtype_of_lengths():
    """`speed` with a half waveform and float32 lengths returned float16 lengths (44101 -> 44096, 70000 -> inf);
    the reference keeps lengths.dtype (functional/functional.py:2421).  Checked through the scripted front on the meta device."""
    f = torch.jit.script(F.speed)
    wav = torch.empty(2, 1000, dtype=torch.float16, device="meta")
    lengths = torch.empty(2, dtype=torch.float32, device="meta")
    out, n = f(wav, 1000, 1.1, lengths)
    assert n is not None and n.dtype == torch.float32 and out.shape == (2, 910)

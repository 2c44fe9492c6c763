"""torch.ops.audio_amd.*: schemas registered, Meta kernels give the reference's shapes/strides,
and a CPU tensor fails loudly in the dispatcher (there is no CPU kernel)."""
import pytest
import torch

import audio_amd  # noqa: F401  (registers the ops)

OPS = ["spectrogram", "mel_spectrogram", "mfcc", "amplitude_to_DB", "resample_apply", "lfilter", "lfilter_cascade",
       "fftconvolve", "inverse_spectrogram", "phase_vocoder", "griffinlim", "rnnt_features"]


def test_ops_are_registered():
    for name in OPS:
        assert hasattr(torch.ops.audio_amd, name)
        assert "Tensor" in str(getattr(torch.ops.audio_amd, name).default._schema)


def test_meta_shapes_and_strides_match_reference_layout():
    x = torch.empty(3, 2, 16000, device="meta")
    w = torch.empty(400, device="meta")
    fb = torch.empty(201, 80, device="meta")
    dct = torch.empty(80, 40, device="meta")
    s = torch.ops.audio_amd.spectrogram(x, w, 0, 400, 160, 400, 2.0, 0, True, "reflect", True)
    assert s.shape == (3, 2, 201, 101) and s.stride()[-2:] == (1, 201)       # frame-major storage
    c = torch.ops.audio_amd.spectrogram(x, w, 0, 400, 160, 400, None, 0, True, "reflect", True)
    assert c.dtype == torch.complex64
    m = torch.ops.audio_amd.mel_spectrogram(x, w, fb, 0, 400, 160, 400, 2.0, 0, True, "reflect")
    assert m.shape == (3, 2, 80, 101) and m.stride()[-2:] == (1, 80)
    k = torch.ops.audio_amd.mfcc(x, w, fb, dct, 0, 400, 160, 400, 2.0, 0, True, "reflect", False, 80.0)
    assert k.shape == (3, 2, 40, 101)
    r = torch.ops.audio_amd.resample_apply(torch.empty(4, 44100, device="meta"), torch.empty(160, 1, 815, device="meta"),
                                           44100, 16000, 100, 187)
    assert r.shape == (4, 16000)
    f = torch.ops.audio_amd.fftconvolve(torch.empty(4, 1, 100, device="meta"), torch.empty(1, 3, 7, device="meta"), "full")
    assert f.shape == (4, 3, 106)
    i = torch.ops.audio_amd.inverse_spectrogram(c, None, w, 0, 400, 160, 400, 0, True, "reflect", True)
    assert i.shape == (3, 2, 16000) and i.dtype == torch.float32
    v = torch.ops.audio_amd.phase_vocoder(c, 1.3, torch.empty(201, 1, device="meta"))
    assert v.shape == (3, 2, 201, 78) and v.dtype == torch.complex64
    gl = torch.ops.audio_amd.griffinlim(s, w, 400, 160, 400, 2.0, 4, 0.99, 16000, False)
    assert gl.shape == (3, 2, 16000)
    ft = torch.ops.audio_amd.rnnt_features(x, w, fb, 400, 160, 1.0, torch.empty(80, device="meta"),
                                           torch.empty(80, device="meta"), 4)
    assert ft.shape == (3, 2, 105, 80)
    pcm = torch.ops.audio_amd.rnnt_features(torch.empty(5, 16000, device="meta", dtype=torch.int16), w, fb, 400, 160, 1.0,
                                            torch.empty(80, device="meta"), torch.empty(80, device="meta"), 0)
    assert pcm.shape == (5, 101, 80) and pcm.dtype == torch.float32


def test_cpu_tensor_has_no_kernel():
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.audio_amd.fftconvolve(torch.randn(5), torch.randn(3), "full")
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.audio_amd.lfilter(torch.randn(2, 50), torch.ones(3), torch.ones(3), True, True)


def test_torch_compile_fullgraph_sees_each_module_as_one_op():
    """VERDICT r2 item 10 / missing #6: the reference's hot-path modules trace (torchscript_consistency_impl.py:34-75).
    Here `torch.compile(fullgraph=True)` captures MelSpectrogram / MFCC / Spectrogram as ONE `audio_amd::*` op each, with the
    fake kernel giving the reference's shape and strides (meta device: no GPU needed)."""
    import audio_amd.transforms as T
    graphs = []

    def backend(gm, example_inputs):
        graphs.append(gm)
        return gm.forward

    cases = [(T.MelSpectrogram(n_fft=400, hop_length=160, n_mels=80), "audio_amd.mel_spectrogram", (4, 80, 101)),
             (T.MFCC(n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)), "audio_amd.mfcc", (4, 40, 101)),
             (T.Spectrogram(n_fft=400, hop_length=160), "audio_amd.spectrogram", (4, 201, 101))]
    for mod, op, shape in cases:
        y = torch.compile(mod.to("meta"), fullgraph=True, backend=backend)(torch.empty(4, 16000, device="meta"))
        assert tuple(y.shape) == shape and y.stride()[-2:] == (1, shape[1])
        calls = [str(n.target) for n in graphs[-1].graph.nodes if n.op == "call_function"]
        assert calls == [op], calls
    rs = T.Resample(44100, 16000).to("meta")
    y = torch.compile(rs, fullgraph=True, backend=backend)(torch.empty(2, 44100, device="meta"))
    assert tuple(y.shape) == (2, 16000)
    assert [str(n.target) for n in graphs[-1].graph.nodes if n.op == "call_function"] == ["audio_amd.resample_apply"]
    fc = T.FFTConvolve("full")
    y = torch.compile(fc, fullgraph=True, backend=backend)(torch.empty(3, 100, device="meta"), torch.empty(3, 7, device="meta"))
    assert tuple(y.shape) == (3, 106)
    db = T.AmplitudeToDB("power", 80.0)
    y = torch.compile(db, fullgraph=True, backend=backend)(torch.empty(3, 80, 101, device="meta"))
    assert tuple(y.shape) == (3, 80, 101)


def test_compiled_aamd_ops_have_fake_kernels():
    """The compiled boxed ops (csrc/torch_shim.cpp, the default route of functional.py) trace under FakeTensorMode: output
    shapes as the shim allocates them."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from audio_amd import _shim
    if not _shim.available():
        pytest.skip("libaudio_amd_torch.so not built")
    _shim.load()
    with FakeTensorMode():
        dev = "cuda"
        w = torch.empty(4, 16000, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        win, tw = torch.empty(400, device=dev), torch.empty(400, 2, device=dev)
        lo, wd, wt = torch.empty(80, **i32), torch.empty(80, **i32), torch.empty(80, 30, device=dev)
        assert torch.ops.aamd.spectrogram(w, win, tw, 400, 160, 0, True, 0, True, 101, 1.0, 2.0).shape == (4, 101, 201)
        assert torch.ops.aamd.spectrogram(w, win, tw, 400, 160, 0, True, 0, True, 101, 1.0, 0.0).shape == (4, 101, 402)
        assert torch.ops.aamd.mel_spectrogram(w, win, tw, lo, wd, wt, None, None, 400, 160, 0, True, 0, 101, 1.0, 2.0, 0).shape == (4, 101, 80)
        assert torch.ops.aamd.mel_spectrogram_db(w, win, tw, lo, wd, wt, None, None, 400, 160, 0, True, 0, 101, 1.0, 2.0, 10.0,
                                                 1e-10, 0.0, None, 1, 0).shape == (4, 101, 80)
        assert torch.ops.aamd.mfcc_dct(torch.empty(404, 80, device=dev), torch.empty(80, 40, device=dev), 2, None, 1, 80.0).shape == (404, 40)
        assert torch.ops.aamd.resample(w, torch.empty(160, 815, device=dev), 441, 160, 187, 5805, None, 0, None).shape == (4, 5805)
        fr = torch.ops.aamd.resample_frag_build(torch.empty(160, 815, device=dev), 441, 160, 187, [0] * 10, 448)    # 10 tiles x 14 steps x 2 x 1 KB
        assert fr.shape == (10 * 14 * 2 * 64 * 4,)
        assert torch.ops.aamd.resample(w, torch.empty(160, 815, device=dev), 441, 160, 187, 5805, [0] * 10, 448, fr).shape == (4, 5805)
        x3 = torch.empty(2, 3, 100, device=dev)
        assert torch.ops.aamd.lfilter(x3, torch.empty(1, 3, 3, device=dev), torch.empty(1, 3, 3, device=dev), 1, 1).shape == x3.shape
        assert torch.ops.aamd.fftconvolve(torch.empty(6, 100, device=dev), torch.empty(6, 7, device=dev), None, None, 6, 0, 106).shape == (6, 106)

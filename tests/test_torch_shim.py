"""libaudio_amd_torch.so: the compiled dispatcher-level boundary (audio_amd/csrc/torch_shim.cpp) -- the same mechanism the
reference uses for its one native kernel (STABLE_TORCH_LIBRARY + TORCH_BOX, torch.ops.load_library;
/root/reference/src/libtorchaudio/lfilter.cpp:118-138, src/torchaudio/_extension/utils.py:50-56).

CPU tests: the library loads, the schemas are the documented ones, the reference's own op
`torchaudio::_lfilter_core_loop` gets the reference's schema and a CUDA-key kernel, CPU tensors have no kernel.
GPU tests: every boxed op is bit-identical to the ctypes binding of the same C-ABI entry, and the reference's call
sequence around `_lfilter_core_loop` (filtering.py:985-996) lands in aamd_lfilter_f32 and matches the float64 oracle."""
import numpy as np
import pytest
import torch

from conftest import peak_rel_err

SCHEMAS = {
    "spectrogram": "aamd::spectrogram(Tensor wav, Tensor window, Tensor twiddle, int n_fft, int hop, int pad, bool center, "
                   "int pad_mode, bool onesided, int n_frames, float scale, float power) -> Tensor",
    "mel_spectrogram": "aamd::mel_spectrogram(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width, "
                       "Tensor band_weights, Tensor? lane_order, Tensor? table400, int n_fft, int hop, int pad, bool center, int pad_mode, "
                       "int n_frames, float scale, float power, int table_sig) -> Tensor",
    "mfcc_dct": "aamd::mfcc_dct(Tensor mel, Tensor dct_mat, int log_mode, Tensor? group_max, int vec_per_group, "
                "float top_db) -> Tensor",
    "resample": "aamd::resample(Tensor wav, Tensor kernel, int orig, int new, int width, int out_len, int[]? band_tap_lo, "
                "int tap_span, Tensor? frag) -> Tensor",
    "lfilter": "aamd::lfilter(Tensor waveform, Tensor a_coeffs, Tensor b_coeffs, int n_stages, int clamp) -> Tensor",
    "fftconvolve": "aamd::fftconvolve(Tensor x, Tensor y, Tensor? x_row_of, Tensor? y_row_of, int rows, int start, "
                   "int out_len) -> Tensor",
}


def test_shim_loads_and_registers_documented_schemas():
    from audio_amd import _shim
    _shim.load()
    for name in _shim.OPS:
        assert hasattr(torch.ops.aamd, name)
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"aamd::{name}", "CUDA")
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f"aamd::{name}", "CPU")
    for name, want in SCHEMAS.items():
        assert str(torch._C._get_schema(f"aamd::{name}", "")) == want
    s = str(torch._C._get_schema("aamd::mel_spectrogram_db", ""))
    assert "Tensor(a!)? group_max" in s          # the running group maximum is declared as mutated


def test_reference_op_gets_reference_schema_and_cuda_kernel():
    from audio_amd import _shim
    _shim.ensure_torchaudio_op()
    _shim.ensure_torchaudio_op()                 # idempotent
    # lfilter.cpp:119-123, verbatim
    assert str(torch._C._get_schema("torchaudio::_lfilter_core_loop", "")) == (
        "torchaudio::_lfilter_core_loop(Tensor input_signal_windows, Tensor a_coeff_flipped, "
        "Tensor(a!) padded_output_waveform) -> Tensor(a!)")
    assert torch._C._dispatch_has_kernel_for_dispatch_key("torchaudio::_lfilter_core_loop", "CUDA")


def test_cpu_tensors_have_no_kernel_in_the_compiled_ops():
    from audio_amd import _shim
    _shim.load()
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.aamd.lfilter(torch.zeros(1, 1, 8), torch.ones(1, 1, 3), torch.ones(1, 1, 3), 1, True)
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.aamd.mfcc_dct(torch.zeros(4, 80), torch.zeros(80, 40), 2, None, 1, -1.0)


def test_product_routes_through_the_shim_by_default(monkeypatch):
    import audio_amd.functional as F
    monkeypatch.delenv("AAMD_NO_TORCH_SHIM", raising=False)
    F._force_route(None)
    assert F._ops() is torch.ops.aamd
    monkeypatch.setenv("AAMD_NO_TORCH_SHIM", "1")
    F._force_route(None)
    assert F._ops() is None
    monkeypatch.delenv("AAMD_NO_TORCH_SHIM")
    F._force_route(None)


# ----------------------------------------------------------------------------------------------------------------- GPU


def _both_routes(fn):
    import audio_amd.functional as F
    try:
        F._force_route("shim")
        a = fn()
        F._force_route("ctypes")
        b = fn()
    finally:
        F._force_route(None)
    return a, b


@pytest.mark.gpu
def test_boxed_ops_are_bit_identical_to_the_ctypes_binding():
    import audio_amd.functional as F
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(5)
    x = (0.5 * torch.randn(3, 2, 9000, generator=g)).clamp_(-1, 1).cuda()
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    mel512 = T.MelSpectrogram(sample_rate=16000, n_fft=512, hop_length=128, n_mels=64).cuda()
    spec = T.Spectrogram(n_fft=400, hop_length=160, power=None).cuda()
    mfcc = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()
    rs = T.Resample(44100, 16000, resampling_method="sinc_interp_kaiser").cuda()
    a = torch.tensor([1.0, -1.6, 0.7]).cuda()
    b = torch.tensor([0.2, 0.3, 0.1]).cuda()
    rir = torch.randn(1, 2, 700, generator=g).cuda() * 0.05
    inv = T.InverseSpectrogram(n_fft=400, hop_length=160).cuda()
    gl = T.GriffinLim(n_fft=400, hop_length=160, n_iter=3, rand_init=False, length=9000).cuda()
    ts = T.TimeStretch(n_freq=201, hop_length=160, fixed_rate=1.25).cuda()
    msc = T.MelScale(n_mels=80, sample_rate=16000, n_stft=201).cuda()
    pw = T.Spectrogram(n_fft=400, hop_length=160).cuda()
    import audio_amd.compliance.kaldi as K

    def spec_grad():
        xa = x.clone().requires_grad_()
        with torch.enable_grad():
            (g_,) = torch.autograd.grad(mel(xa).sum() + pw(xa).sqrt().sum(), xa)
        return g_

    cases = {
        # round 4: the routes added with the rest of the C ABI as dispatcher ops
        "istft": lambda: inv(spec(x), 9000), "griffinlim": lambda: gl(pw(x)),
        "phase_vocoder": lambda: torch.view_as_real(ts(spec(x))), "mel_scale": lambda: msc(pw(x)),
        "stft_adjoint (autograd)": spec_grad,
        "kaldi_fbank": lambda: K.fbank(x[0, :1] * 32768.0, num_mel_bins=23, dither=0.0),
        "kaldi_spectrogram": lambda: K.spectrogram(x[1, :1] * 32768.0, dither=0.0),
        "mel400": lambda: mel(x), "mel512": lambda: mel512(x), "spec_complex": lambda: torch.view_as_real(spec(x)),
        "mfcc": lambda: mfcc(x), "mfcc_sliced_rows": lambda: mfcc(x[..., 100:8100]),
        "resample": lambda: rs(x), "lfilter": lambda: F.lfilter(x, a, b),
        "fftconvolve": lambda: F.fftconvolve(x, rir, mode="same"),
    }
    with torch.no_grad():
        for name, fn in cases.items():
            u, v = _both_routes(fn)
            assert u.shape == v.shape and u.stride() == v.stride(), name
            assert torch.equal(u, v), name


@pytest.mark.gpu
@pytest.mark.parametrize("n_order", [2, 3, 5])
def test_reference_lfilter_core_loop_lands_in_the_hip_kernel(n_order):
    """The reference's DifferentiableIIR.forward (filtering.py:985-996), statement for statement, on ROCm tensors with
    torch.ops.torchaudio._lfilter_core_loop served by this library."""
    from audio_amd import _shim
    from oracle import dsp_oracle as O
    _shim.ensure_torchaudio_op()
    g = torch.Generator().manual_seed(n_order)
    n_batch, n_channel, n_sample = 3, 2, 5000
    waveform = (0.3 * torch.randn(n_batch, n_channel, n_sample, generator=g)).cuda()
    # stable per-channel denominators: conjugate pole pairs (and one real pole for an odd count) inside the unit circle
    rs = np.random.default_rng(100 + n_order)
    poly = []
    for c in range(n_channel):
        roots = []
        for _ in range((n_order - 1) // 2):
            z = rs.uniform(0.3, 0.85) * np.exp(1j * rs.uniform(0.2, 2.9))
            roots += [z, np.conj(z)]
        if (n_order - 1) % 2:
            roots.append(rs.uniform(-0.8, 0.8))
        poly.append(np.poly(roots).real)
    a_coeffs_normalized = torch.tensor(np.stack(poly), dtype=torch.float32).cuda()
    n_sample_padded = n_sample + n_order - 1
    a_coeff_flipped = a_coeffs_normalized.flip(1).contiguous()
    padded_output_waveform = torch.zeros(n_batch, n_channel, n_sample_padded, device=waveform.device, dtype=waveform.dtype)
    ret = torch.ops.torchaudio._lfilter_core_loop(waveform, a_coeff_flipped, padded_output_waveform)
    assert ret.data_ptr() == padded_output_waveform.data_ptr()          # Tensor(a!) -> Tensor(a!)
    output = padded_output_waveform[:, :, n_order - 1:]
    assert float(padded_output_waveform[:, :, : n_order - 1].abs().max()) == 0.0
    b = np.zeros((n_channel, n_order))
    b[:, 0] = 1.0
    want = O.lfilter(waveform.cpu().numpy().astype(np.float64), a_coeffs_normalized.cpu().numpy().astype(np.float64), b,
                     clamp=False)
    assert peak_rel_err(output.cpu().numpy(), want) <= 2e-5


@pytest.mark.gpu
def test_boxed_op_uses_the_current_stream():
    """iir_cuda.cu:73 launches on the default stream whatever the current one is; the shim takes the current stream."""
    import audio_amd.transforms as T
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda()
    x = torch.randn(8, 16000, device="cuda")
    with torch.no_grad():
        ref = mel(x)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            big = torch.randn(64, 1 << 20, device="cuda")          # keeps stream s busy ahead of the launch
            for _ in range(4):
                big = big * 1.0001
            y = mel(x * 1.0)                                      # producer and consumer on s: ordered only on s
        s.synchronize()
    assert torch.equal(y, ref)


@pytest.mark.gpu
def test_torch_compile_fullgraph_melspectrogram_equals_eager():
    """`torch.compile(fullgraph=True)` over the headline module (eager backend: the captured graph is one audio_amd::* op,
    tests/test_ops_registration.py) gives the eager result bit for bit, for MelSpectrogram and MFCC."""
    import audio_amd.transforms as T
    g = torch.Generator().manual_seed(0)
    x = (0.3 * torch.randn(3, 16000, generator=g)).cuda()
    for mod in (T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).cuda(),
                T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).cuda()):
        if hasattr(mod, "fused"):
            mod.fused = False   # the traced op is the stateless two-kernel MFCC; eager "auto" may take the one-kernel path
        with torch.no_grad():
            want = mod(x)
            got = torch.compile(mod, fullgraph=True, backend="eager")(x)
        assert got.shape == want.shape and got.stride() == want.stride()
        assert torch.equal(got, want)


@pytest.mark.gpu
def test_round4_ops_reach_every_remaining_entry_point():
    """VERDICT r3 missing 6: the entries that the Python layer still calls through ctypes are ops too -- each compared bit for
    bit with the Python route on the same inputs."""
    import audio_amd.functional as F
    import audio_amd.transforms as T
    from audio_amd import _shim
    from audio_amd.pipelines import RNNTFeatureExtractor
    _shim.load()
    ops = torch.ops.aamd
    g = torch.Generator().manual_seed(7)
    dev = torch.device("cuda")
    x = (0.4 * torch.randn(4, 16000, generator=g)).clamp_(-1, 1).to(dev)
    x[3, 6000:] = 0.0
    mel = T.MelSpectrogram(sample_rate=16000, n_fft=400, hop_length=160, n_mels=80).to(dev)
    fb, win = mel.mel_scale.fb, mel.spectrogram.window
    bands = F._mel_bands(fb, dev)
    band_args = (bands.lo, bands.width, bands.weights, bands.lane_order, bands.table400)
    wp, tw = F._padded_window(win, 400), F._twiddles(400, dev)
    with torch.no_grad():
        # --- RNN-T features: float, planar PCM, interleaved stereo PCM
        stats = {"mean": (10 + torch.randn(80, generator=g)).tolist(), "invstddev": (0.3 + torch.rand(80, generator=g)).tolist()}
        fe = RNNTFeatureExtractor(stats).to(dev)
        want = fe.features(x)
        mean, inv = fe.mean.to(dev), fe.invstddev.to(dev)
        frames = want.shape[-2]
        got = ops.mel_spectrogram_lognorm(x, wp, tw, *band_args, 400, 160, 101, 1.0, float(fe.gain), mean, inv, frames,
                                          int(bands.table_sig))
        assert torch.equal(got.view(want.shape), want)
        pcm = torch.randint(-20000, 20000, (3, 16000, 2), generator=g, dtype=torch.int16).to(dev)
        want = fe.features(pcm, channels_first=False)
        got = ops.mel_spectrogram_lognorm(pcm, wp, tw, *band_args, 400, 160, 101, 1.0 / 32768.0, float(fe.gain), mean, inv,
                                          frames, int(bands.table_sig))
        assert torch.equal(got.view(want.shape), want)
        # --- the one-kernel MFCC as one op
        mfcc = T.MFCC(sample_rate=16000, n_mfcc=40, melkwargs=dict(n_fft=400, hop_length=160, n_mels=80)).to(dev)
        mfcc.fused = True
        want = mfcc(x)                                            # (4, 40, 101): one cut-off for the batch
        frag = ops.mfcc_frag_build(mfcc.dct_mat.contiguous(), 80, 40)
        gmax = torch.full((1,), float("-inf"), device=dev)
        a2 = mfcc.amplitude_to_DB
        got = ops.mfcc_fused(x, wp, tw, *band_args, frag, gmax, 400, 160, 0, True, 0, 101, 1.0, 40, a2.multiplier, a2.amin,
                             a2.db_multiplier, 80.0, 4, int(bands.table_sig))
        assert torch.equal(got.transpose(-1, -2), want) and float(gmax) > -100.0
        # --- dB family
        p = mel(x)
        want = F.amplitude_to_DB(p, 10.0, 1e-10, 0.0, 80.0)
        pc = p.transpose(-1, -2).contiguous()
        # a 3-D input is ONE item of (channel, freq, time) for the reference's cut-off (functional.py:393-402): one group
        gm = torch.full((1,), float("-inf"), device=dev)
        db = ops.amplitude_to_db(pc, 10.0, 1e-10, 0.0, gm, pc.numel())
        got = ops.db_clamp(db, gm, pc.numel(), 80.0)
        assert torch.equal(got.transpose(-1, -2), want)
        got2 = ops.amplitude_to_db_clamped(pc, 10.0, 1e-10, 0.0, gm, pc.numel(), 80.0)
        assert torch.equal(got2, got)
        # --- float64 precision entries
        x64 = x[:2, :3000].double()
        a = torch.tensor([1.0, -1.2, 0.5], dtype=torch.float64, device=dev)
        b = torch.tensor([0.3, 0.2, 0.1], dtype=torch.float64, device=dev)
        assert torch.equal(ops.lfilter_f64(x64.view(2, 1, -1), a.view(1, 1, 3), b.view(1, 1, 3), 1, 1).view(2, -1),
                           F.lfilter(x64, a, b))
        rs = T.Resample(16000, 12000, dtype=torch.float64).to(dev)
        want = rs(x64)
        got = ops.resample_f64(x64, rs.kernel.view(rs.kernel.shape[0], -1).contiguous(), 4, 3, rs.width, want.shape[-1])
        assert torch.equal(got, want)
        y64 = torch.randn(2, 200, generator=g, dtype=torch.float64).to(dev)
        want = F.fftconvolve(x64, y64)
        assert torch.equal(ops.fftconvolve_f64(x64, y64, None, None, 2, 0, want.shape[-1]), want)
        w64, tw64 = win.double(), None
        from audio_amd import _diff
        tw64 = _diff.twiddles(400, dev, torch.float64)
        X = ops.spectrogram_f64(x64, w64, tw64, 400, 160, 0, True, 0, 19)
        want = F.spectrogram(x64, 0, w64, 400, 160, 400, None, False)
        assert torch.equal(torch.view_as_complex(X).transpose(-1, -2), want)
        back = ops.istft_f64(X, w64, tw64, None, 400, 160, 0, True, 0, 3000, 1.0, True)
        assert back.shape == (2, 3000) and back.dtype == torch.float64
        # --- spectrum cotangents
        Xc = torch.view_as_real(F.spectrogram(x, 0, win, 400, 160, 400, None, False).transpose(-1, -2).contiguous())
        dP = torch.randn(4, 101, 201, generator=g).to(dev)
        G = ops.spectrogram_grad(Xc, dP, 2.0)
        want = 2.0 * dP.unsqueeze(-1) * Xc
        assert float((G - want).abs().max()) <= 1e-5 * float(want.abs().max())


# ------------------------------------------------------------------ round 5: prepared tap spectra (aamd_fftconvolve_staged_f32)


def test_staged_fftconvolve_schema():
    from audio_amd import _shim
    _shim.load()
    assert str(torch._C._get_schema("aamd::fftconvolve_staged", "")) == (
        "aamd::fftconvolve_staged(Tensor x, Tensor y, Tensor? x_row_of, Tensor? y_row_of, int rows, int start, int out_len, "
        "Tensor(a!) workspace, int stages) -> Tensor")


@pytest.mark.gpu
@pytest.mark.parametrize("route", ["shim", "ctypes"])
@pytest.mark.parametrize("taps", [700, 9000, 24000, 30000])
def test_repeated_impulse_response_skips_the_preparation_and_stays_bit_identical(route, taps):
    """A tap tensor that is used AGAIN gets a workspace of its own on that second call (round 6: a tensor used once leaves
    nothing behind), and from the third call on the transform runs on it without the twiddle / tap-spectrum launches; the results
    are bit-identical to a call that prepares afresh (a clone of the taps), an in-place update of the taps invalidates the held
    workspace; a workspace is held only for plans 1 and 3 (plan 2 keeps its delay-line ring there) and only up to 64 MiB
    (30 000 taps: the workspace formula still reserves the complex-block rings)."""
    import audio_amd.functional as F
    from audio_amd import _lib
    g = torch.Generator().manual_seed(taps)
    x1 = (torch.rand(6, 70000, generator=g) - 0.5).cuda()
    x2 = (torch.rand(6, 70000, generator=g) - 0.5).cuda()
    h = (torch.randn(1, taps, generator=g) * 0.02).cuda()
    try:
        F._force_route(route)
        plan = _lib.lib().aamd_fftconvolve_plan(6, 70000, taps, 70000 + taps - 1)
        y1 = F.fftconvolve(x1, h)
        assert F.fftconvolve_held_taps(h) == 0          # used once: nothing is kept
        y2 = F.fftconvolve(x2, h)                       # used again: prepared into a workspace that stays with `h`
        held = F.fftconvolve_held_taps(h)
        ws_bytes = _lib.lib().aamd_fftconvolve_workspace(6, 6, 1, 70000, taps)
        assert held == (1 if plan in (1, 3) and ws_bytes <= F._FFTCONV_HELD_BYTES else 0)
        y2_run_only = F.fftconvolve(x2, h)              # run-only on the held workspace
        y2_fresh = F.fftconvolve(x2, h.clone())         # prepares again
        assert torch.equal(y2, y2_fresh) and torch.equal(y2_run_only, y2_fresh)
        y1_again = F.fftconvolve(x1, h)
        assert torch.equal(y1, y1_again)
        # against the float64 direct evaluation at a few output samples
        idx = torch.tensor([0, 1234, taps - 1, taps + 5000, 70000 + taps - 2])
        xd, hd = x2.double().cpu(), h.double().cpu()[0]
        for i in idx.tolist():
            lo, hi = max(0, i - taps + 1), min(i, 69999)
            want = (xd[:, lo:hi + 1] * hd[i - hi:i - lo + 1].flip(0)).sum(-1)
            assert float((y2[:, i].double().cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max() + 1.0)
        h.mul_(2.0)                                     # in-place update: the version counter moves, the workspace is stale
        assert F.fftconvolve_held_taps(h) == 0
        y3 = F.fftconvolve(x2, h)
        assert torch.allclose(y3, 2.0 * y2, rtol=1e-5, atol=1e-6)
    finally:
        F._force_route(None)


@pytest.mark.gpu
def test_held_taps_are_per_stream_and_not_used_under_capture():
    import audio_amd.functional as F
    g = torch.Generator().manual_seed(77)
    x = (torch.rand(4, 50000, generator=g) - 0.5).cuda()
    h = (torch.randn(1, 12000, generator=g) * 0.02).cuda()
    y0 = F.fftconvolve(x, h)
    y0 = F.fftconvolve(x, h)                            # (a workspace is kept from the second use on)
    assert F.fftconvolve_held_taps(h) == 1
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y1 = F.fftconvolve(x, h)
        y1 = F.fftconvolve(x, h)                        # another stream: its own prepared workspace
    s.synchronize()
    assert F.fftconvolve_held_taps(h) == 2
    assert torch.equal(y0, y1)
    # a captured call prepares inside the graph (a replay must not depend on a host-side version check)
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        F.fftconvolve(x, h)                             # warm-up on the capture stream
        cap.synchronize()
        held_before = F.fftconvolve_held_taps(h)
        with torch.cuda.graph(graph, stream=cap):
            yg = F.fftconvolve(x, h)
    assert F.fftconvolve_held_taps(h) == held_before
    h.mul_(0.5)                                         # the replay sees the new taps: the preparation is part of the graph
    graph.replay()
    torch.cuda.synchronize()
    assert torch.allclose(yg, 0.5 * y0, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_tap_tensors_used_once_and_backward_temporaries_leave_no_workspace_behind():
    """a fresh impulse-response batch per step, and the flipped operands of the autograd backward, must not
    pin a prepared workspace each; the total kept per process is bounded."""
    import gc
    import audio_amd.functional as F
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(4, 60000, generator=g) - 0.5).cuda()
    gc.collect()
    live0 = sum(1 for r in F._HELD_LIVE if r() is not None)
    for i in range(12):                                 # an augmentation loop: new taps every step
        h = (torch.randn(1, 12000, generator=g) * 0.02).cuda()
        F.fftconvolve(x, h)
        assert F.fftconvolve_held_taps(h) == 0
    xg = x.clone().requires_grad_(True)
    hg = (torch.randn(1, 12000, generator=g) * 0.02).cuda().requires_grad_(True)
    for _ in range(3):                                  # the same leaves every step: forward may keep, backward never does
        F.fftconvolve(xg, hg).square().sum().backward()
    del h
    gc.collect()
    live = sum(1 for r in F._HELD_LIVE if r() is not None)
    assert live - live0 <= 1, (live0, live)             # at most the one workspace of the reused leaf `hg`
    # the bound: with the total set below one workspace nothing more is kept
    keep = F._FFTCONV_HELD_TOTAL
    try:
        F._FFTCONV_HELD_TOTAL = 0
        h2 = (torch.randn(1, 12000, generator=g) * 0.02).cuda()
        a = F.fftconvolve(x, h2)
        b = F.fftconvolve(x, h2)
        assert F.fftconvolve_held_taps(h2) == 0 and torch.equal(a, b)
    finally:
        F._FFTCONV_HELD_TOTAL = keep
    # a dead tap tensor takes its slot (and its workspace) with it
    h3 = (torch.randn(1, 12000, generator=g) * 0.02).cuda()
    F.fftconvolve(x, h3)
    F.fftconvolve(x, h3)
    assert F.fftconvolve_held_taps(h3) == 1
    tid = id(h3)
    del h3
    gc.collect()
    assert tid not in F._TENSOR_CACHE

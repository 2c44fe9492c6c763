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
                "int tap_span) -> Tensor",
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
    cases = {
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

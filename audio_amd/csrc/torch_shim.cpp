// libaudio_amd_torch.so -- dispatcher-level boundary of the MI355X audio kernels.
//
// The reference binds its one native kernel to PyTorch with the LibTorch stable ABI:
//   STABLE_TORCH_LIBRARY_FRAGMENT(torchaudio, m) { m.def("_lfilter_core_loop(...)"); }
//   STABLE_TORCH_LIBRARY_IMPL(torchaudio, CUDA, m) { m.impl("_lfilter_core_loop", TORCH_BOX(&cuda_lfilter_core_loop)); }
// (/root/reference/src/libtorchaudio/lfilter.cpp:118-138), loaded by torch.ops.load_library
// (src/torchaudio/_extension/utils.py:50-56) and called as torch.ops.torchaudio._lfilter_core_loop
// (src/torchaudio/functional/filtering.py:994).  This translation unit is the same mechanism for libaudio_amd.so:
// boxed kernels registered on the CUDA dispatch key (ROCm tensors carry it) that validate their tensors the way
// iir_cuda.cu:41-65 does, allocate outputs through torch (caching allocator, stream semantics), take the CURRENT
// stream of the tensor's device (cuda_utils.h:9-15; the reference's own IIR launch forgets it, iir_cuda.cu:73) and
// forward raw pointers to the C ABI of include/audio_amd.h.  No torch type crosses into libaudio_amd.so.
//
//   namespace aamd          spectrogram / mel_spectrogram / mel_spectrogram_db / mfcc_dct / resample / lfilter /
//                           fftconvolve: one op per C-ABI entry point of the hot path
//   namespace torchaudio    _lfilter_core_loop on the CUDA key with the reference's schema -- the one place the
//                           unmodified reference calls into native code.  The schema itself is defined by
//                           aamd_define_torchaudio_schema() only when libtorchaudio has not defined it already.
//
// Built by audio_amd/_build.py with g++ against the torch headers; links libaudio_amd.so ($ORIGIN rpath).
#include <audio_amd.h>

#include <torch/csrc/inductor/aoti_torch/c/shim.h>
#include <torch/csrc/stable/accelerator.h>
#include <torch/csrc/stable/library.h>
#include <torch/csrc/stable/ops.h>
#include <torch/csrc/stable/tensor.h>
#include <torch/headeronly/core/ScalarType.h>

#include <cmath>
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <type_traits>
#include <vector>

namespace {

using torch::headeronly::ScalarType;
using torch::stable::Tensor;

void check(int rc) {
  STD_TORCH_CHECK(rc == AAMD_OK, aamd_last_error());
}

void* current_stream(const Tensor& t) {
  void* s = nullptr;
  TORCH_ERROR_CODE_CHECK(aoti_torch_get_current_cuda_stream(t.get_device_index(), &s));
  return s;
}

void want_f32(const Tensor& t, const char* what, int64_t dim = -1) {
  STD_TORCH_CHECK(t.is_cuda(), "audio_amd: ", what, " must be on an MI355X (ROCm) device; there is no CPU kernel");
  STD_TORCH_CHECK(t.scalar_type() == ScalarType::Float, "audio_amd: ", what, " must be float32");
  STD_TORCH_CHECK(t.is_contiguous(), "audio_amd: ", what, " must be contiguous");
  if (dim >= 0) STD_TORCH_CHECK(t.dim() == dim, "audio_amd: ", what, " must have ", dim, " dimensions");
}

void want_i32(const Tensor& t, const char* what) {
  STD_TORCH_CHECK(t.is_cuda() && t.scalar_type() == ScalarType::Int && t.is_contiguous(), "audio_amd: ", what,
                  " must be a contiguous int32 device tensor");
}

void same_device(const Tensor& a, const Tensor& b) {
  STD_TORCH_CHECK(a.get_device_index() == b.get_device_index(), "audio_amd: tensors on different devices");
}

const float* fp(const Tensor& t) { return t.numel() ? static_cast<const float*>(t.data_ptr()) : nullptr; }
float* fpm(Tensor& t) { return t.numel() ? static_cast<float*>(t.data_ptr()) : nullptr; }

aamd_stft_desc make_desc(const Tensor& wav, int64_t n_fft, int64_t hop, int64_t pad, bool center, int64_t pad_mode,
                         bool onesided, int64_t n_frames, double scale, double power) {
  // rows may be a strided view of longer rows (a batch sliced in time): unit stride along time is what the kernels need
  STD_TORCH_CHECK(wav.is_cuda(), "audio_amd: waveform must be on an MI355X (ROCm) device; there is no CPU kernel");
  STD_TORCH_CHECK(wav.scalar_type() == ScalarType::Float, "audio_amd: waveform must be float32");
  STD_TORCH_CHECK(wav.dim() == 2, "audio_amd: waveform must be (rows, time)");
  STD_TORCH_CHECK(wav.size(1) <= 1 || wav.stride(1) == 1, "audio_amd: waveform rows must have unit stride");
  STD_TORCH_CHECK(wav.size(0) <= 1 || wav.stride(0) >= wav.size(1), "audio_amd: overlapping waveform rows");
  aamd_stft_desc d{};
  d.rows = wav.size(0);
  d.length = wav.size(1);
  d.row_stride = d.rows > 1 ? wav.stride(0) : (d.length > 0 ? d.length : 1);
  d.n_fft = (int32_t)n_fft;
  d.hop = (int32_t)hop;
  d.pad = (int32_t)pad;
  d.center = center;
  d.pad_mode = (int32_t)pad_mode;
  d.onesided = onesided;
  d.n_frames = (int32_t)n_frames;
  d.scale = (float)scale;
  d.power = (float)power;
  return d;
}

struct Bands {
  aamd_mel_bands b{};
  Bands(const Tensor& wav, const Tensor& lo, const Tensor& width, const Tensor& weights,
        const std::optional<Tensor>& lane_order, const std::optional<Tensor>& table400, int64_t table_sig = 0) {
    want_i32(lo, "band_lo");
    want_i32(width, "band_width");
    want_f32(weights, "band_weights", 2);
    same_device(wav, lo);
    same_device(wav, width);
    same_device(wav, weights);
    STD_TORCH_CHECK(lo.numel() == width.numel() && weights.size(0) == lo.numel(), "audio_amd: band table shapes disagree");
    b.n_mels = (int32_t)lo.numel();
    b.max_width = (int32_t)weights.size(1);
    b.lo = static_cast<const int32_t*>(lo.data_ptr());
    b.width = static_cast<const int32_t*>(width.data_ptr());
    b.weights = fp(weights);
    b.lane_order = nullptr;
    if (lane_order.has_value()) {
      want_i32(*lane_order, "lane_order");
      same_device(wav, *lane_order);
      b.lane_order = static_cast<const int32_t*>(lane_order->data_ptr());
    }
    b.table400 = nullptr;
    if (table400.has_value()) {
      want_f32(*table400, "table400", 1);
      same_device(wav, *table400);
      STD_TORCH_CHECK(table400->numel() == aamd_mel400_table_dwords(b.n_mels, b.max_width),
                      "audio_amd: table400 does not belong to this band table");
      b.table400 = fp(*table400);
    }
    b.table_sig = b.table400 ? (int32_t)table_sig : 0;     // (the kernel verifies it against the table and traps on a mismatch)
  }
};

// ---- aamd::spectrogram  (functional/functional.py:123-145) --------------------------------------------------------
Tensor spectrogram(Tensor wav, Tensor window, Tensor twiddle, int64_t n_fft, int64_t hop, int64_t pad, bool center,
                   int64_t pad_mode, bool onesided, int64_t n_frames, double scale, double power) {
  aamd_stft_desc d = make_desc(wav, n_fft, hop, pad, center, pad_mode, onesided, n_frames, scale, power);
  want_f32(window, "window", 1);
  want_f32(twiddle, "twiddle");
  same_device(wav, window);
  same_device(wav, twiddle);
  STD_TORCH_CHECK(window.numel() == n_fft && twiddle.numel() == 2 * n_fft, "audio_amd: window / twiddle size");
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  const int64_t n_freq = onesided ? n_fft / 2 + 1 : n_fft;
  Tensor out = torch::stable::new_empty(wav, {d.rows, n_frames, n_freq * (power > 0.0 ? 1 : 2)});
  if (out.numel()) check(aamd_spectrogram_f32(fp(wav), fp(window), fp(twiddle), fpm(out), &d, current_stream(wav)));
  return out;
}

// ---- aamd::mel_spectrogram  (transforms/_transforms.py:612-622) ---------------------------------------------------
Tensor mel_spectrogram(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width, Tensor band_weights,
                       std::optional<Tensor> lane_order, std::optional<Tensor> table400, int64_t n_fft, int64_t hop, int64_t pad, bool center,
                       int64_t pad_mode, int64_t n_frames, double scale, double power, int64_t table_sig) {
  aamd_stft_desc d = make_desc(wav, n_fft, hop, pad, center, pad_mode, true, n_frames, scale, power);
  want_f32(window, "window", 1);
  want_f32(twiddle, "twiddle");
  same_device(wav, window);
  same_device(wav, twiddle);
  STD_TORCH_CHECK(window.numel() == n_fft && twiddle.numel() == 2 * n_fft, "audio_amd: window / twiddle size");
  Bands bands(wav, band_lo, band_width, band_weights, lane_order, table400, table_sig);
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(wav, {d.rows, n_frames, (int64_t)bands.b.n_mels});
  if (out.numel())
    check(aamd_melspectrogram_f32(fp(wav), fp(window), fp(twiddle), &bands.b, fpm(out), &d, current_stream(wav)));
  return out;
}

// ---- aamd::mel_spectrogram_db  (first half of MFCC.forward, _transforms.py:692-706) --------------------------------
Tensor mel_spectrogram_db(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width,
                          Tensor band_weights, std::optional<Tensor> lane_order, std::optional<Tensor> table400, int64_t n_fft, int64_t hop, int64_t pad,
                          bool center, int64_t pad_mode, int64_t n_frames, double scale, double power, double multiplier,
                          double amin, double db_multiplier, std::optional<Tensor> group_max, int64_t rows_per_group,
                          int64_t table_sig) {
  aamd_stft_desc d = make_desc(wav, n_fft, hop, pad, center, pad_mode, true, n_frames, scale, power);
  want_f32(window, "window", 1);
  want_f32(twiddle, "twiddle");
  same_device(wav, window);
  same_device(wav, twiddle);
  Bands bands(wav, band_lo, band_width, band_weights, lane_order, table400, table_sig);
  float* gmax = nullptr;
  if (group_max.has_value()) {
    want_f32(*group_max, "group_max", 1);
    same_device(wav, *group_max);
    STD_TORCH_CHECK(rows_per_group > 0 && group_max->numel() * rows_per_group >= d.rows, "audio_amd: group_max too small");
    gmax = fpm(*group_max);
  }
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(wav, {d.rows, n_frames, (int64_t)bands.b.n_mels});
  if (out.numel())
    check(aamd_melspectrogram_db_f32(fp(wav), fp(window), fp(twiddle), &bands.b, fpm(out), &d, (float)multiplier,
                                     (float)amin, (float)db_multiplier, gmax, rows_per_group, current_stream(wav)));
  return out;
}

// ---- aamd::mfcc_dct  (second half of MFCC.forward, _transforms.py:706-709) -----------------------------------------
Tensor mfcc_dct(Tensor mel, Tensor dct, int64_t log_mode, std::optional<Tensor> group_max, int64_t vec_per_group,
                double top_db) {
  want_f32(mel, "mel", 2);
  want_f32(dct, "dct_mat", 2);
  same_device(mel, dct);
  STD_TORCH_CHECK(dct.size(0) == mel.size(1), "audio_amd: dct_mat rows must equal n_mels");
  const float* gmax = nullptr;
  if (group_max.has_value()) {
    want_f32(*group_max, "group_max", 1);
    same_device(mel, *group_max);
    gmax = fp(*group_max);
  }
  const torch::stable::accelerator::DeviceGuard guard(mel.get_device_index());
  Tensor out = torch::stable::new_empty(mel, {mel.size(0), dct.size(1)});
  if (out.numel())
    check(aamd_mfcc_dct_f32(fp(mel), fp(dct), fpm(out), mel.size(0), (int32_t)mel.size(1), (int32_t)dct.size(1),
                            (int32_t)log_mode, gmax, vec_per_group > 0 ? vec_per_group : 1, (float)top_db,
                            current_stream(mel)));
  return out;
}

// ---- aamd::resample  (functional/functional.py:1421-1428) ----------------------------------------------------------
// `frag`: the prepared tap fragments of aamd::resample_frag_build for the same kernel values and band table (ABI 7), or none
Tensor resample(Tensor wav, Tensor kernel, int64_t orig, int64_t new_, int64_t width, int64_t out_len,
                std::optional<std::vector<int64_t>> band_tap_lo, int64_t tap_span, std::optional<Tensor> frag) {
  want_f32(wav, "waveform", 2);
  want_f32(kernel, "kernel", 2);
  same_device(wav, kernel);
  if (frag.has_value()) same_device(wav, *frag);
  STD_TORCH_CHECK(kernel.size(0) == new_ && kernel.size(1) == 2 * width + orig,
                  "audio_amd: resample kernel shape does not match (new, 2*width+orig)");
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(wav, {wav.size(0), out_len});
  if (out.numel()) {
    const int64_t length = wav.size(1);
    if (band_tap_lo.has_value()) {
      std::vector<int32_t> lo(band_tap_lo->begin(), band_tap_lo->end());
      aamd_resample_bands bands{(int32_t)lo.size(), (int32_t)tap_span, lo.data()};
      const void* fr = nullptr;
      if (frag.has_value()) {
        STD_TORCH_CHECK(frag->is_contiguous() && (int64_t)(frag->numel() * frag->element_size()) >=
                            aamd_resample_frag_bytes((int32_t)orig, (int32_t)new_, &bands),
                        "audio_amd: prepared tap fragments too small for this band table");
        fr = frag->data_ptr();
      }
      check(aamd_resample_prepared_f32(fp(wav), fp(kernel), fpm(out), wav.size(0), length, length > 0 ? length : 1,
                                       (int32_t)orig, (int32_t)new_, (int32_t)width, out_len, &bands, fr, current_stream(wav)));
    } else {
      check(aamd_resample_f32(fp(wav), fp(kernel), fpm(out), wav.size(0), length, length > 0 ? length : 1, (int32_t)orig,
                              (int32_t)new_, (int32_t)width, out_len, current_stream(wav)));
    }
  }
  return out;
}

// ---- aamd::resample_frag_build: the packed binary16 tap fragments of a resampling kernel, once per filter (ABI 7) -------------
Tensor resample_frag_build(Tensor kernel, int64_t orig, int64_t new_, int64_t width, std::vector<int64_t> band_tap_lo, int64_t tap_span) {
  want_f32(kernel, "kernel", 2);
  STD_TORCH_CHECK(kernel.size(0) == new_ && kernel.size(1) == 2 * width + orig,
                  "audio_amd: resample kernel shape does not match (new, 2*width+orig)");
  const torch::stable::accelerator::DeviceGuard guard(kernel.get_device_index());
  std::vector<int32_t> lo(band_tap_lo.begin(), band_tap_lo.end());
  aamd_resample_bands bands{(int32_t)lo.size(), (int32_t)tap_span, lo.data()};
  const int64_t bytes = aamd_resample_frag_bytes((int32_t)orig, (int32_t)new_, &bands);
  STD_TORCH_CHECK(bytes > 0, "audio_amd: no matrix-core resampling kernel serves this band table (no fragments to prepare)");
  Tensor out = torch::stable::new_empty(kernel, {bytes / 4});          // float32 storage holding the packed dwords
  check(aamd_resample_frag_build_f32(fp(kernel), (int32_t)orig, (int32_t)new_, (int32_t)width, &bands, out.data_ptr(),
                                     current_stream(kernel)));
  return out;
}

// ---- aamd::lfilter  (functional/filtering.py:1027-1099; cascades fused) --------------------------------------------
Tensor lfilter(Tensor x, Tensor a, Tensor b, int64_t n_stages, int64_t clamp) {   // clamp: 0 / 1 every stage / 2 last stage
  want_f32(x, "waveform", 3);
  want_f32(a, "a_coeffs", 3);
  want_f32(b, "b_coeffs", 3);
  same_device(x, a);
  same_device(x, b);
  STD_TORCH_CHECK(a.size(0) == n_stages && b.size(0) == n_stages && a.size(1) == b.size(1) && a.size(2) == b.size(2),
                  "audio_amd: a / b must be (n_stages, rows, n_order)");
  STD_TORCH_CHECK(a.size(1) == 1 || a.size(1) == x.size(1), "audio_amd: coefficient rows must be 1 or channels");
  STD_TORCH_CHECK(clamp >= 0 && clamp <= 2, "audio_amd: clamp must be 0, 1 (after every stage) or 2 (after the last stage)");
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  Tensor y = torch::stable::empty_like(x);
  if (y.numel())
    check(aamd_lfilter_f32(fp(x), fp(a), fp(b), fpm(y), x.size(0), (int32_t)x.size(1), x.size(2), (int32_t)a.size(2),
                           (int32_t)a.size(1), (int32_t)n_stages, (int32_t)clamp, current_stream(x)));
  return y;
}

// ---- aamd::fftconvolve  (functional/functional.py:2252-2258) -------------------------------------------------------
Tensor fftconvolve(Tensor x, Tensor y, std::optional<Tensor> x_row_of, std::optional<Tensor> y_row_of, int64_t rows,
                   int64_t start, int64_t out_len) {
  want_f32(x, "x", 2);
  want_f32(y, "y", 2);
  same_device(x, y);
  const int64_t* xm = nullptr;
  const int64_t* ym = nullptr;
  if (x_row_of.has_value()) {
    STD_TORCH_CHECK(x_row_of->is_cuda() && x_row_of->scalar_type() == ScalarType::Long && x_row_of->numel() == rows,
                    "audio_amd: x_row_of must be int64[rows] on the device");
    xm = static_cast<const int64_t*>(x_row_of->data_ptr());
  }
  if (y_row_of.has_value()) {
    STD_TORCH_CHECK(y_row_of->is_cuda() && y_row_of->scalar_type() == ScalarType::Long && y_row_of->numel() == rows,
                    "audio_amd: y_row_of must be int64[rows] on the device");
    ym = static_cast<const int64_t*>(y_row_of->data_ptr());
  }
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  Tensor out = torch::stable::new_empty(x, {rows, out_len});
  if (out.numel()) {
    const int64_t ws_bytes = aamd_fftconvolve_workspace(rows, x.size(0), y.size(0), x.size(1), y.size(1));
    STD_TORCH_CHECK(ws_bytes >= 0, aamd_last_error());
    Tensor ws = torch::stable::new_empty(x, {(ws_bytes + 3) / 4 + 2});
    check(aamd_fftconvolve_f32(fp(x), fp(y), fpm(out), rows, x.size(0), y.size(0), x.size(1), y.size(1), xm, ym, start,
                               out_len, ws_bytes ? ws.data_ptr() : nullptr, current_stream(x)));
  }
  return out;
}

// ---- aamd::fftconvolve_staged: the same call with a caller-held workspace (aamd_fftconvolve_staged_f32) --------------------
// workspace: float32, at least aamd_fftconvolve_workspace() bytes, 8-byte aligned.  stages: 1 prepare (twiddles + tap spectra
// into the workspace), 2 run on a prepared workspace, 3 both.  The host keeps the workspace of a repeated impulse response
// (audio_amd/functional.py: _conv_slice) and skips the two preparation launches from the second call on.
Tensor fftconvolve_staged(Tensor x, Tensor y, std::optional<Tensor> x_row_of, std::optional<Tensor> y_row_of, int64_t rows,
                          int64_t start, int64_t out_len, Tensor workspace, int64_t stages) {
  want_f32(x, "x", 2);
  want_f32(y, "y", 2);
  want_f32(workspace, "workspace", 1);
  same_device(x, y);
  same_device(x, workspace);
  const int64_t* xm = nullptr;
  const int64_t* ym = nullptr;
  if (x_row_of.has_value()) {
    STD_TORCH_CHECK(x_row_of->is_cuda() && x_row_of->scalar_type() == ScalarType::Long && x_row_of->numel() == rows,
                    "audio_amd: x_row_of must be int64[rows] on the device");
    xm = static_cast<const int64_t*>(x_row_of->data_ptr());
  }
  if (y_row_of.has_value()) {
    STD_TORCH_CHECK(y_row_of->is_cuda() && y_row_of->scalar_type() == ScalarType::Long && y_row_of->numel() == rows,
                    "audio_amd: y_row_of must be int64[rows] on the device");
    ym = static_cast<const int64_t*>(y_row_of->data_ptr());
  }
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  Tensor out = torch::stable::new_empty(x, {(stages & AAMD_FFTCONV_RUN) ? rows : 0, out_len});
  if (rows * out_len) {
    const int64_t ws_bytes = aamd_fftconvolve_workspace(rows, x.size(0), y.size(0), x.size(1), y.size(1));
    STD_TORCH_CHECK(ws_bytes >= 0, aamd_last_error());
    STD_TORCH_CHECK(workspace.numel() * 4 >= ws_bytes, "audio_amd: workspace smaller than aamd_fftconvolve_workspace()");
    STD_TORCH_CHECK(reinterpret_cast<uintptr_t>(workspace.data_ptr()) % 8 == 0, "audio_amd: workspace must be 8-byte aligned");
    check(aamd_fftconvolve_staged_f32(fp(x), fp(y), out.numel() ? fpm(out) : nullptr, rows, x.size(0), y.size(0), x.size(1),
                                      y.size(1), xm, ym, start, out_len, ws_bytes ? workspace.data_ptr() : nullptr,
                                      (int32_t)stages, current_stream(x)));
  }
  return out;
}


// =====================================================================================================================
// Round 4 (VERDICT r3 missing 6): one dispatcher op per remaining compute entry of include/audio_amd.h -- the reference
// mechanism (lfilter.cpp:118-138) is one op per native entry.  Same rules as above: validate like iir_cuda.cu:41-65,
// allocate through torch, current stream of the tensor's device, raw pointers into the C ABI.
// =====================================================================================================================
void want_dev(const Tensor& t, ScalarType st, const char* what, int64_t dim = -1) {
  STD_TORCH_CHECK(t.is_cuda(), "audio_amd: ", what, " must be on an MI355X (ROCm) device; there is no CPU kernel");
  STD_TORCH_CHECK(t.scalar_type() == st, "audio_amd: ", what, " has the wrong dtype");
  STD_TORCH_CHECK(t.is_contiguous(), "audio_amd: ", what, " must be contiguous");
  if (dim >= 0) STD_TORCH_CHECK(t.dim() == dim, "audio_amd: ", what, " must have ", dim, " dimensions");
}
const double* dp(const Tensor& t) { return t.numel() ? static_cast<const double*>(t.data_ptr()) : nullptr; }
double* dpm(Tensor& t) { return t.numel() ? static_cast<double*>(t.data_ptr()) : nullptr; }

void stft_consts(const Tensor& ref, const Tensor& window, const Tensor& twiddle, int64_t n_fft, ScalarType st) {
  want_dev(window, st, "window", 1);
  want_dev(twiddle, st, "twiddle");
  same_device(ref, window);
  same_device(ref, twiddle);
  STD_TORCH_CHECK(window.numel() == n_fft && twiddle.numel() == 2 * n_fft, "audio_amd: window / twiddle size");
}

// ---- aamd::mel_spectrogram_lognorm / _pcm16 (pipelines/rnnt_pipeline.py:16-47, 319-326 fused; _torchcodec.py:150-152) ----
// wav: float32 (rows, time) | int16 (rows, time) planar PCM | int16 (clips, time, channels) interleaved PCM (channels 1 / 2).
// mean / invstddev absent: plain mel spectrogram of the PCM (the float form of that is aamd::mel_spectrogram).
Tensor mel_spectrogram_lognorm(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width,
                               Tensor band_weights, std::optional<Tensor> lane_order, std::optional<Tensor> table400,
                               int64_t n_fft, int64_t hop, int64_t n_frames, double scale, double gain,
                               std::optional<Tensor> mean, std::optional<Tensor> invstddev, int64_t out_frames,
                               int64_t table_sig) {
  STD_TORCH_CHECK(wav.is_cuda() && wav.is_contiguous(), "audio_amd: waveform must be a contiguous device tensor");
  const bool pcm = wav.scalar_type() == ScalarType::Short;
  STD_TORCH_CHECK(pcm || wav.scalar_type() == ScalarType::Float, "audio_amd: waveform must be float32 or int16 PCM");
  STD_TORCH_CHECK(wav.dim() == 2 || (pcm && wav.dim() == 3), "audio_amd: waveform must be (rows, time) or int16 (clips, time, channels)");
  const int64_t channels = wav.dim() == 3 ? wav.size(2) : 0;
  // interleaved stereo: the kernel reads one 32-bit (L, R) word per sample time (a view that starts on an odd
  // int16 offset is contiguous and still misaligned -- refuse it here instead of faulting in the kernel)
  STD_TORCH_CHECK(channels != 2 || wav.numel() == 0 || reinterpret_cast<uintptr_t>(wav.data_ptr()) % 4 == 0,
                  "audio_amd: interleaved stereo PCM must start on a 4-byte boundary (clone() the view)");
  STD_TORCH_CHECK(mean.has_value() == invstddev.has_value(), "audio_amd: mean and invstddev come together");
  STD_TORCH_CHECK(pcm || mean.has_value(), "audio_amd: float input without statistics is aamd::mel_spectrogram");
  stft_consts(wav, window, twiddle, n_fft, ScalarType::Float);
  aamd_stft_desc d{};
  d.rows = channels ? wav.size(0) * channels : wav.size(0);
  d.length = wav.size(1);
  d.row_stride = d.length > 0 ? d.length : 1;
  d.n_fft = (int32_t)n_fft; d.hop = (int32_t)hop; d.pad = 0; d.center = 1; d.pad_mode = AAMD_PAD_REFLECT; d.onesided = 1;
  d.n_frames = (int32_t)n_frames; d.scale = (float)scale; d.power = 2.0f;
  Bands bands(wav, band_lo, band_width, band_weights, lane_order, table400, table_sig);
  const float *mp = nullptr, *ip = nullptr;
  if (mean.has_value()) {
    want_f32(*mean, "mean", 1); want_f32(*invstddev, "invstddev", 1);
    same_device(wav, *mean); same_device(wav, *invstddev);
    STD_TORCH_CHECK(mean->numel() == bands.b.n_mels && invstddev->numel() == bands.b.n_mels, "audio_amd: statistics must have n_mels entries");
    mp = fp(*mean); ip = fp(*invstddev);
  }
  const int64_t frames = mean.has_value() ? out_frames : n_frames;
  STD_TORCH_CHECK(frames >= n_frames, "audio_amd: out_frames must be >= n_frames");
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(window, {d.rows, frames, (int64_t)bands.b.n_mels});
  if (frames > n_frames && out.numel()) {               // the pipeline's right padding: zero rows (the kernel leaves them alone)
    Tensor tail = torch::stable::narrow(out, 1, n_frames, frames - n_frames);
    torch::stable::fill_(tail, 0.0);
  }
  if (out.numel() && n_frames > 0) {
    void* st = current_stream(wav);
    if (!pcm)
      check(aamd_melspectrogram_lognorm_f32(fp(wav), fp(window), fp(twiddle), &bands.b, fpm(out), &d, (float)gain, mp, ip, frames, st));
    else if (channels)
      check(aamd_melspectrogram_pcm16_interleaved_f32(static_cast<const int16_t*>(wav.data_ptr()), (int32_t)channels, fp(window),
                                                      fp(twiddle), &bands.b, fpm(out), &d, (float)gain, mp, ip, frames, st));
    else
      check(aamd_melspectrogram_pcm16_f32(static_cast<const int16_t*>(wav.data_ptr()), fp(window), fp(twiddle), &bands.b,
                                          fpm(out), &d, (float)gain, mp, ip, frames, st));
  }
  return out;
}

// ---- aamd::mfcc_frag_build / aamd::mfcc_fused (transforms/_transforms.py:692-709 in one kernel + fix-up launch) ---------
Tensor mfcc_frag_build(Tensor dct, int64_t n_mels, int64_t n_mfcc) {
  want_f32(dct, "dct_mat", 2);
  STD_TORCH_CHECK(dct.size(0) == n_mels && dct.size(1) == n_mfcc, "audio_amd: dct_mat must be (n_mels, n_mfcc)");
  const torch::stable::accelerator::DeviceGuard guard(dct.get_device_index());
  Tensor frag = torch::stable::new_empty(dct, {(int64_t)aamd_mfcc_frag_floats()});
  check(aamd_mfcc_frag_build(fp(dct), (int32_t)n_mels, (int32_t)n_mfcc, fpm(frag), current_stream(dct)));
  return frag;
}
// Single-rank form (no exchange of group_max between the passes): both passes in one op.
// scratch: int32[2 * tiles + 1] (tile minima as float bits, the list, the count); group_max: float[n_groups] pre-filled with
// -inf, updated in place.  Returns (rows, n_frames, n_mfcc).
Tensor mfcc_fused(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width, Tensor band_weights,
                  std::optional<Tensor> lane_order, std::optional<Tensor> table400, Tensor dct_frag, Tensor group_max,
                  int64_t n_fft, int64_t hop, int64_t pad, bool center, int64_t pad_mode, int64_t n_frames, double scale,
                  int64_t n_mfcc, double multiplier, double amin, double db_multiplier, double top_db, int64_t rows_per_group,
                  int64_t table_sig) {
  aamd_stft_desc d = make_desc(wav, n_fft, hop, pad, center, pad_mode, true, n_frames, scale, 2.0);
  stft_consts(wav, window, twiddle, n_fft, ScalarType::Float);
  want_f32(dct_frag, "dct_frag", 1);
  want_f32(group_max, "group_max", 1);
  same_device(wav, dct_frag); same_device(wav, group_max);
  STD_TORCH_CHECK(dct_frag.numel() == aamd_mfcc_frag_floats(), "audio_amd: dct_frag comes from aamd::mfcc_frag_build");
  STD_TORCH_CHECK(rows_per_group > 0 && group_max.numel() * rows_per_group >= d.rows, "audio_amd: group_max too small");
  Bands bands(wav, band_lo, band_width, band_weights, lane_order, table400, table_sig);
  STD_TORCH_CHECK(aamd_mfcc_fused_supported(&d, &bands.b, (int32_t)n_mfcc), "audio_amd: shape not served by the fused MFCC "
                  "(n_fft 400, hop 100 / 160 / 200, 80 mels, n_mfcc <= 48 and a multiple of 4)");
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(wav, {d.rows, n_frames, n_mfcc});
  if (!out.numel()) return out;
  const int64_t tiles = aamd_mfcc_fused_tiles(&d);
  Tensor scratch = torch::stable::new_empty(wav, {2 * tiles + 1});
  float* sp = fpm(scratch);
  aamd_mfcc_fused f{};
  f.dct_frag = fp(dct_frag); f.n_mfcc = (int32_t)n_mfcc; f.pass = 0;
  f.multiplier = (float)multiplier; f.amin = (float)amin; f.db_multiplier = (float)db_multiplier; f.top_db = (float)top_db;
  f.group_max = fpm(group_max); f.rows_per_group = rows_per_group;
  f.tile_min = sp; f.tile_list = reinterpret_cast<int32_t*>(sp + tiles);
  f.fix_count = reinterpret_cast<int32_t*>(sp + 2 * tiles);
  void* st = current_stream(wav);
  check(aamd_mfcc_fused_f32(fp(wav), fp(window), fp(twiddle), &bands.b, fpm(out), &d, &f, st));
  f.pass = 1;
  check(aamd_mfcc_fused_f32(fp(wav), fp(window), fp(twiddle), &bands.b, fpm(out), &d, &f, st));
  return out;
}

// ---- aamd::istft / aamd::istft_f64 (functional/functional.py:148-225; adjoint = the STFT's backward) --------------------
template <typename T>
Tensor istft_any(Tensor spec, Tensor window, Tensor twiddle, std::optional<Tensor> inv_envelope, int64_t n_fft, int64_t hop,
                 int64_t pad, bool center, int64_t pad_mode, int64_t length, double scale, bool adjoint) {
  constexpr ScalarType st = std::is_same<T, double>::value ? ScalarType::Double : ScalarType::Float;
  want_dev(spec, st, "spec", 4);                         // (rows, n_frames, n_fft / 2 + 1, 2): view_as_real of the complex frames
  STD_TORCH_CHECK(spec.size(2) == n_fft / 2 + 1 && spec.size(3) == 2, "audio_amd: spec must be (rows, frames, n_fft / 2 + 1, 2)");
  stft_consts(spec, window, twiddle, n_fft, st);
  const T* env = nullptr;
  if (inv_envelope.has_value()) {
    want_dev(*inv_envelope, st, "inv_envelope", 1);
    same_device(spec, *inv_envelope);
    STD_TORCH_CHECK(inv_envelope->numel() == length, "audio_amd: inv_envelope must have `length` entries");
    env = static_cast<const T*>(inv_envelope->data_ptr());
  }
  aamd_stft_desc d{};
  d.rows = spec.size(0); d.length = length; d.row_stride = length > 0 ? length : 1;
  d.n_fft = (int32_t)n_fft; d.hop = (int32_t)hop; d.pad = (int32_t)pad; d.center = center; d.pad_mode = (int32_t)pad_mode;
  d.onesided = 1; d.n_frames = (int32_t)spec.size(1); d.scale = (float)scale; d.power = 0.0f;
  const torch::stable::accelerator::DeviceGuard guard(spec.get_device_index());
  Tensor out = torch::stable::new_zeros(window, {d.rows, length});     // the kernel accumulates with atomic adds
  if (out.numel() && d.n_frames) {
    if constexpr (std::is_same<T, double>::value)
      check(aamd_istft_f64(dp(spec), dp(window), dp(twiddle), env, dpm(out), &d, adjoint ? 1 : 0, current_stream(spec)));
    else
      check(aamd_istft_f32(fp(spec), fp(window), fp(twiddle), env, fpm(out), &d, adjoint ? 1 : 0, current_stream(spec)));
  }
  return out;
}
Tensor istft(Tensor spec, Tensor window, Tensor twiddle, std::optional<Tensor> inv_envelope, int64_t n_fft, int64_t hop,
             int64_t pad, bool center, int64_t pad_mode, int64_t length, double scale, bool adjoint) {
  return istft_any<float>(spec, window, twiddle, inv_envelope, n_fft, hop, pad, center, pad_mode, length, scale, adjoint);
}
Tensor istft_f64(Tensor spec, Tensor window, Tensor twiddle, std::optional<Tensor> inv_envelope, int64_t n_fft, int64_t hop,
                 int64_t pad, bool center, int64_t pad_mode, int64_t length, double scale, bool adjoint) {
  return istft_any<double>(spec, window, twiddle, inv_envelope, n_fft, hop, pad, center, pad_mode, length, scale, adjoint);
}

// ---- aamd::spectrogram_f64 (functional/functional.py:123-145 on double; complex frames only: the |X|^p is the caller's) ---
Tensor spectrogram_f64(Tensor wav, Tensor window, Tensor twiddle, int64_t n_fft, int64_t hop, int64_t pad, bool center,
                       int64_t pad_mode, int64_t n_frames) {
  want_dev(wav, ScalarType::Double, "waveform", 2);
  stft_consts(wav, window, twiddle, n_fft, ScalarType::Double);
  aamd_stft_desc d{};
  d.rows = wav.size(0); d.length = wav.size(1); d.row_stride = d.length > 0 ? d.length : 1;
  d.n_fft = (int32_t)n_fft; d.hop = (int32_t)hop; d.pad = (int32_t)pad; d.center = center; d.pad_mode = (int32_t)pad_mode;
  d.onesided = 1; d.n_frames = (int32_t)n_frames; d.scale = 1.0f; d.power = 0.0f;
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(wav, {d.rows, n_frames, n_fft / 2 + 1, 2});
  if (out.numel()) check(aamd_spectrogram_f64(dp(wav), dp(window), dp(twiddle), dpm(out), &d, current_stream(wav)));
  return out;
}

// ---- aamd::phase_vocoder (functional/functional.py:732-803) ----------------------------------------------------------------
// spec: view_as_real of a complex64 (rows, freq, frames) tensor with ANY strides (float (rows, freq, frames, 2), last stride 1);
// frame_major_out: the result is (rows, frames_out, freq, 2) memory -- what aamd::istft eats -- else (rows, freq, frames_out, 2).
Tensor phase_vocoder(Tensor spec, Tensor phase_advance, double rate, bool frame_major_out) {
  STD_TORCH_CHECK(spec.is_cuda() && spec.scalar_type() == ScalarType::Float && spec.dim() == 4 && spec.size(3) == 2,
                  "audio_amd: spec must be view_as_real of a complex64 (rows, freq, frames) device tensor");
  STD_TORCH_CHECK(spec.numel() == 0 || (spec.stride(3) == 1 && spec.stride(0) % 2 == 0 && spec.stride(1) % 2 == 0 && spec.stride(2) % 2 == 0),
                  "audio_amd: spec strides must address whole complex elements");
  want_f32(phase_advance, "phase_advance", 1);
  same_device(spec, phase_advance);
  STD_TORCH_CHECK(rate > 0.0 && phase_advance.numel() == spec.size(1), "audio_amd: phase_advance must have n_freq entries, rate > 0");
  const int64_t rows = spec.size(0), n_freq = spec.size(1), n_in = spec.size(2);
  const int64_t n_out = (int64_t)std::ceil((double)n_in / rate);
  const torch::stable::accelerator::DeviceGuard guard(spec.get_device_index());
  Tensor out = frame_major_out ? torch::stable::new_empty(phase_advance, {rows, n_out, n_freq, 2})
                               : torch::stable::new_empty(phase_advance, {rows, n_freq, n_out, 2});
  if (out.numel()) {
    aamd_vocoder_desc v{};
    v.rows = rows; v.n_freq = (int32_t)n_freq; v.n_frames_in = (int32_t)n_in; v.n_frames_out = (int32_t)n_out;
    v.in_stride_row = spec.stride(0) / 2; v.in_stride_freq = spec.stride(1) / 2; v.in_stride_frame = spec.stride(2) / 2;
    v.out_stride_row = n_out * n_freq;
    v.out_stride_freq = frame_major_out ? 1 : n_out;
    v.out_stride_frame = frame_major_out ? n_freq : 1;
    v.rate = rate;
    check(aamd_phase_vocoder_f32(static_cast<const float*>(spec.data_ptr()), fp(phase_advance), fpm(out), &v, current_stream(spec)));
  }
  return out;
}

// ---- aamd::griffinlim_update (functional/functional.py:336-343): returns `next`, updates tprev in place ---------------------
Tensor griffinlim_update(Tensor rebuilt, Tensor tprev, Tensor magnitude, double momentum) {
  want_f32(rebuilt, "rebuilt"); want_f32(tprev, "tprev"); want_f32(magnitude, "magnitude");
  same_device(rebuilt, tprev); same_device(rebuilt, magnitude);
  STD_TORCH_CHECK(rebuilt.numel() == 2 * magnitude.numel() && tprev.numel() == rebuilt.numel(), "audio_amd: rebuilt / tprev are view_as_real of the magnitude's shape");
  const torch::stable::accelerator::DeviceGuard guard(rebuilt.get_device_index());
  Tensor next = torch::stable::empty_like(rebuilt);
  if (next.numel())
    check(aamd_griffinlim_update_f32(fp(rebuilt), fpm(tprev), fp(magnitude), fpm(next), magnitude.numel(), (float)momentum, current_stream(rebuilt)));
  return next;
}

// ---- aamd::mel_scale (transforms/_transforms.py:403-415 on a frame-major spectrogram) ---------------------------------------
Tensor mel_scale(Tensor spec, Tensor band_lo, Tensor band_width, Tensor band_weights) {
  want_f32(spec, "spec", 3);                             // (rows, frames, n_freq)
  Bands bands(spec, band_lo, band_width, band_weights, std::nullopt, std::nullopt, 0);
  const torch::stable::accelerator::DeviceGuard guard(spec.get_device_index());
  Tensor out = torch::stable::new_empty(spec, {spec.size(0), spec.size(1), (int64_t)bands.b.n_mels});
  if (out.numel())
    check(aamd_mel_scale_f32(fp(spec), &bands.b, fpm(out), spec.size(0), (int32_t)spec.size(1), (int32_t)spec.size(2), current_stream(spec)));
  return out;
}

// ---- aamd::amplitude_to_db / _clamped / db_clamp (functional/functional.py:356-404) -----------------------------------------
Tensor amplitude_to_db(Tensor x, double multiplier, double amin, double db_multiplier, std::optional<Tensor> group_max, int64_t group_size) {
  want_f32(x, "x");
  float* gm = nullptr;
  if (group_max.has_value()) {
    want_f32(*group_max, "group_max", 1); same_device(x, *group_max);
    STD_TORCH_CHECK(group_size > 0 && group_max->numel() * group_size >= x.numel(), "audio_amd: group_max too small");
    gm = fpm(*group_max);
  }
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  Tensor out = torch::stable::empty_like(x);
  if (out.numel())
    check(aamd_amplitude_to_db_f32(fp(x), fpm(out), x.numel(), (float)multiplier, (float)amin, (float)db_multiplier, gm,
                                   group_size > 0 ? group_size : 1, current_stream(x)));
  return out;
}
Tensor amplitude_to_db_clamped(Tensor x, double multiplier, double amin, double db_multiplier, Tensor group_max, int64_t group_size, double top_db) {
  want_f32(x, "x"); want_f32(group_max, "group_max", 1); same_device(x, group_max);
  STD_TORCH_CHECK(group_size > 0 && group_max.numel() * group_size >= x.numel(), "audio_amd: group_max too small");
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  Tensor out = torch::stable::empty_like(x);
  if (out.numel())
    check(aamd_amplitude_to_db_clamped_f32(fp(x), fpm(out), x.numel(), (float)multiplier, (float)amin, (float)db_multiplier,
                                           fp(group_max), group_size, (float)top_db, current_stream(x)));
  return out;
}
Tensor db_clamp(Tensor x, Tensor group_max, int64_t group_size, double top_db) {
  want_f32(x, "x"); want_f32(group_max, "group_max", 1); same_device(x, group_max);
  STD_TORCH_CHECK(group_size > 0 && group_max.numel() * group_size >= x.numel(), "audio_amd: group_max too small");
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  Tensor out = torch::stable::empty_like(x);
  if (out.numel()) check(aamd_db_clamp_f32(fp(x), fpm(out), x.numel(), fp(group_max), group_size, (float)top_db, current_stream(x)));
  return out;
}

// ---- aamd::spectrogram_grad / aamd::mel_spectrogram_grad (backward of |X|^p and of fb . |X|^p) ------------------------------
Tensor spectrogram_grad(Tensor spec, Tensor dpower, double power) {
  want_f32(spec, "spec"); want_f32(dpower, "dpower"); same_device(spec, dpower);
  STD_TORCH_CHECK(spec.numel() == 2 * dpower.numel(), "audio_amd: spec is view_as_real of dpower's shape");
  const torch::stable::accelerator::DeviceGuard guard(spec.get_device_index());
  Tensor out = torch::stable::empty_like(spec);
  if (out.numel()) check(aamd_spectrogram_grad_f32(fp(spec), fp(dpower), fpm(out), dpower.numel(), (float)power, current_stream(spec)));
  return out;
}
Tensor mel_spectrogram_grad(Tensor spec, Tensor dmel, Tensor band_lo, Tensor band_width, Tensor band_weights, double power) {
  want_f32(spec, "spec", 3);                             // (n_vec, n_freq, 2): the complex STFT, overwritten with the cotangent
  want_f32(dmel, "dmel", 2);
  same_device(spec, dmel);
  Bands bt(spec, band_lo, band_width, band_weights, std::nullopt, std::nullopt, 0);     // the band table of fb^T
  STD_TORCH_CHECK(spec.size(2) == 2 && dmel.size(0) == spec.size(0) && bt.b.n_mels == spec.size(1), "audio_amd: shapes of spec / dmel / fb^T bands");
  const torch::stable::accelerator::DeviceGuard guard(spec.get_device_index());
  if (spec.numel())
    check(aamd_melspectrogram_grad_f32(fpm(spec), fp(dmel), &bt.b, spec.size(0), (int32_t)spec.size(1), (int32_t)dmel.size(1),
                                       (float)power, current_stream(spec)));
  return spec;
}

// ---- aamd::resample_sparse (F.pitch_shift's huge reduced rates, functional/functional.py:1790-1840) ------------------------
Tensor resample_sparse(Tensor wav, Tensor taps_compact, Tensor tap_lo, int64_t orig, int64_t new_, int64_t width, int64_t out_len) {
  want_f32(wav, "waveform", 2); want_f32(taps_compact, "taps_compact", 2); want_i32(tap_lo, "tap_lo");
  same_device(wav, taps_compact); same_device(wav, tap_lo);
  STD_TORCH_CHECK(taps_compact.size(0) == new_ && tap_lo.numel() == new_, "audio_amd: one compacted tap row per output phase");
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(wav, {wav.size(0), out_len});
  if (out.numel()) {
    const int64_t length = wav.size(1);
    check(aamd_resample_sparse_f32(fp(wav), fp(taps_compact), static_cast<const int32_t*>(tap_lo.data_ptr()), fpm(out), wav.size(0),
                                   length, length > 0 ? length : 1, (int32_t)orig, (int32_t)new_, (int32_t)width,
                                   (int32_t)taps_compact.size(1), out_len, current_stream(wav)));
  }
  return out;
}

// ---- aamd::kaldi_features (compliance/kaldi.py:229-315, 514-645) -----------------------------------------------------------
// opts: [n_fft, shift, win, snip_edges, remove_dc_offset, raw_energy, use_power, use_log, energy_col, first_col, n_cols];
// fopts: [preemphasis, energy_floor, dither].  wav: (n_utt, n_samples).  No band tensors: kaldi.spectrogram rows.
Tensor kaldi_features(Tensor wav, Tensor window, Tensor twiddle, std::optional<Tensor> band_lo, std::optional<Tensor> band_width,
                      std::optional<Tensor> band_weights, std::optional<Tensor> noise, int64_t n_frames, std::vector<int64_t> opts,
                      std::vector<double> fopts) {
  want_f32(wav, "waveform", 2);
  STD_TORCH_CHECK(opts.size() == 11 && fopts.size() == 3, "audio_amd: kaldi_features takes 11 integer and 3 float options");
  stft_consts(wav, window, twiddle, opts[0], ScalarType::Float);
  aamd_kaldi_desc k{};
  k.n_samples = wav.size(1); k.n_frames = n_frames; k.n_fft = (int32_t)opts[0]; k.shift = (int32_t)opts[1]; k.win = (int32_t)opts[2];
  k.snip_edges = (int32_t)opts[3]; k.preemphasis = (float)fopts[0]; k.remove_dc_offset = (int32_t)opts[4]; k.raw_energy = (int32_t)opts[5];
  k.energy_floor = (float)fopts[1]; k.use_power = (int32_t)opts[6]; k.use_log = (int32_t)opts[7]; k.energy_col = (int32_t)opts[8];
  k.first_col = (int32_t)opts[9]; k.n_cols = (int32_t)opts[10]; k.dither = (float)fopts[2]; k.noise = nullptr;
  k.n_utt = wav.size(0); k.utt_stride = wav.size(1);
  if (noise.has_value()) {
    want_f32(*noise, "noise"); same_device(wav, *noise);
    STD_TORCH_CHECK(noise->numel() == k.n_utt * n_frames * k.win, "audio_amd: noise must be (n_utt, n_frames, win)");
    k.noise = fp(*noise);
  }
  STD_TORCH_CHECK(k.dither == 0.0f || k.noise != nullptr, "audio_amd: dither needs the caller's Gaussian draws");
  std::optional<Bands> bands;
  const bool fbank = band_lo.has_value();
  STD_TORCH_CHECK(fbank == band_width.has_value() && fbank == band_weights.has_value(), "audio_amd: band tensors come together");
  if (fbank) bands.emplace(wav, *band_lo, *band_width, *band_weights, std::nullopt, std::nullopt, 0);
  const int64_t row = fbank ? k.n_cols : k.n_fft / 2 + 1;
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(wav, {k.n_utt, n_frames, row});
  if (out.numel())
    check(aamd_kaldi_features_f32(fp(wav), fp(window), fp(twiddle), fbank ? &bands->b : nullptr, fpm(out), &k, current_stream(wav)));
  return out;
}

// ---- float64 entries: aamd::lfilter_f64 / resample_f64 / fftconvolve_f64 (the precision path, csrc/f64_paths.h) -------------
Tensor lfilter_f64(Tensor x, Tensor a, Tensor b, int64_t n_stages, int64_t clamp) {
  want_dev(x, ScalarType::Double, "waveform", 3); want_dev(a, ScalarType::Double, "a_coeffs", 3); want_dev(b, ScalarType::Double, "b_coeffs", 3);
  same_device(x, a); same_device(x, b);
  STD_TORCH_CHECK(a.size(0) == n_stages && b.size(0) == n_stages && a.size(1) == b.size(1) && a.size(2) == b.size(2),
                  "audio_amd: a / b must be (n_stages, rows, n_order)");
  STD_TORCH_CHECK(a.size(1) == 1 || a.size(1) == x.size(1), "audio_amd: coefficient rows must be 1 or channels");
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  Tensor y = torch::stable::empty_like(x);
  if (y.numel())
    check(aamd_lfilter_f64(dp(x), dp(a), dp(b), dpm(y), x.size(0), (int32_t)x.size(1), x.size(2), (int32_t)a.size(2),
                           (int32_t)a.size(1), (int32_t)n_stages, (int32_t)clamp, current_stream(x)));
  return y;
}
Tensor resample_f64(Tensor wav, Tensor kernel, int64_t orig, int64_t new_, int64_t width, int64_t out_len) {
  want_dev(wav, ScalarType::Double, "waveform", 2); want_dev(kernel, ScalarType::Double, "kernel", 2);
  same_device(wav, kernel);
  STD_TORCH_CHECK(kernel.size(0) == new_ && kernel.size(1) == 2 * width + orig, "audio_amd: resample kernel shape does not match (new, 2*width+orig)");
  const torch::stable::accelerator::DeviceGuard guard(wav.get_device_index());
  Tensor out = torch::stable::new_empty(wav, {wav.size(0), out_len});
  if (out.numel()) {
    const int64_t length = wav.size(1);
    check(aamd_resample_f64(dp(wav), dp(kernel), dpm(out), wav.size(0), length, length > 0 ? length : 1, (int32_t)orig,
                            (int32_t)new_, (int32_t)width, out_len, current_stream(wav)));
  }
  return out;
}
Tensor fftconvolve_f64(Tensor x, Tensor y, std::optional<Tensor> x_row_of, std::optional<Tensor> y_row_of, int64_t rows,
                       int64_t start, int64_t out_len) {
  want_dev(x, ScalarType::Double, "x", 2); want_dev(y, ScalarType::Double, "y", 2);
  same_device(x, y);
  const int64_t *xm = nullptr, *ym = nullptr;
  if (x_row_of.has_value()) { want_dev(*x_row_of, ScalarType::Long, "x_row_of", 1); STD_TORCH_CHECK(x_row_of->numel() == rows); xm = static_cast<const int64_t*>(x_row_of->data_ptr()); }
  if (y_row_of.has_value()) { want_dev(*y_row_of, ScalarType::Long, "y_row_of", 1); STD_TORCH_CHECK(y_row_of->numel() == rows); ym = static_cast<const int64_t*>(y_row_of->data_ptr()); }
  const torch::stable::accelerator::DeviceGuard guard(x.get_device_index());
  Tensor out = torch::stable::new_empty(x, {rows, out_len});
  if (out.numel())
    check(aamd_fftconvolve_f64(dp(x), dp(y), dpm(out), rows, x.size(0), y.size(0), x.size(1), y.size(1), xm, ym, start, out_len,
                               current_stream(x)));
  return out;
}

// ---- torchaudio::_lfilter_core_loop on the CUDA key (lfilter.cpp:118-134, iir_cuda.cu:37-79) ------------------------
//   padded_out[n][c][i + n_order - 1] = in[n][c][i] - sum_{j < n_order-1} a_flipped[c][j] * padded_out[n][c][i + j]
// = the pure recursion y = IIR(in; a) with a = flip(a_flipped), b = (1, 0, ...), no clamp: aamd_lfilter_f32 runs it as a
// chunked scan across the whole chip instead of one thread per (n, c) sequence.  padded_out arrives zero-filled
// (filtering.py:990-993); its first n_order - 1 samples stay zero.
Tensor cuda_lfilter_core_loop(Tensor in, Tensor a_flipped, Tensor padded_out) {
  STD_TORCH_CHECK(in.is_cuda() && a_flipped.is_cuda() && padded_out.is_cuda());
  STD_TORCH_CHECK((in.get_device_index() == a_flipped.get_device_index()) &&
                  (in.get_device_index() == padded_out.get_device_index()));
  STD_TORCH_CHECK(in.is_contiguous() && a_flipped.is_contiguous() && padded_out.is_contiguous());
  STD_TORCH_CHECK(in.scalar_type() == ScalarType::Float && a_flipped.scalar_type() == ScalarType::Float &&
                      padded_out.scalar_type() == ScalarType::Float,
                  "audio_amd: _lfilter_core_loop on MI355X computes in float32 (got another dtype)");
  STD_TORCH_CHECK(in.dim() == 3 && a_flipped.dim() == 2 && padded_out.dim() == 3);
  const int64_t N = in.size(0), C = in.size(1), L = in.size(2), n_order = a_flipped.size(1);
  STD_TORCH_CHECK(N == padded_out.size(0));
  STD_TORCH_CHECK(C == padded_out.size(1));
  STD_TORCH_CHECK(a_flipped.size(0) == C);
  STD_TORCH_CHECK(L + n_order - 1 == padded_out.size(2));
  const torch::stable::accelerator::DeviceGuard guard(in.get_device_index());
  if (N * C * L == 0) return padded_out;
  // a = flip(a_flipped, 1), lower delays first; b = e_0
  Tensor a = torch::stable::new_empty(a_flipped, {1, C, n_order});
  {
    const auto num_args = 2;
    std::vector<int64_t> dims{1};
    std::array<StableIValue, num_args> stack{torch::stable::detail::from(a_flipped),
                                             torch::stable::detail::from(dims)};
    TORCH_ERROR_CODE_CHECK(torch_call_dispatcher("aten::flip", "", stack.data(), TORCH_ABI_VERSION));
    Tensor flipped = torch::stable::detail::to<Tensor>(stack[0]);
    Tensor a_view = torch::stable::select(a, 0, 0);
    torch::stable::copy_(a_view, flipped);
  }
  Tensor b = torch::stable::new_zeros(a_flipped, {1, C, n_order});
  {
    Tensor b0 = torch::stable::narrow(b, 2, 0, 1);
    torch::stable::fill_(b0, 1.0);
  }
  Tensor y = torch::stable::empty_like(in);
  check(aamd_lfilter_f32(fp(in), fp(a), fp(b), fpm(y), N, (int32_t)C, L, (int32_t)n_order, (int32_t)C, 1, 0,
                         current_stream(in)));
  Tensor dst = torch::stable::narrow(padded_out, 2, n_order - 1, L);
  torch::stable::copy_(dst, y);
  return padded_out;
}

}  // namespace

STABLE_TORCH_LIBRARY(aamd, m) {
  m.def("spectrogram(Tensor wav, Tensor window, Tensor twiddle, int n_fft, int hop, int pad, bool center, int pad_mode, "
        "bool onesided, int n_frames, float scale, float power) -> Tensor");
  m.def("mel_spectrogram(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width, "
        "Tensor band_weights, Tensor? lane_order, Tensor? table400, int n_fft, int hop, int pad, bool center, int pad_mode, int n_frames, "
        "float scale, float power, int table_sig) -> Tensor");
  m.def("mel_spectrogram_db(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width, "
        "Tensor band_weights, Tensor? lane_order, Tensor? table400, int n_fft, int hop, int pad, bool center, int pad_mode, int n_frames, "
        "float scale, float power, float multiplier, float amin, float db_multiplier, Tensor(a!)? group_max, "
        "int rows_per_group, int table_sig) -> Tensor");
  m.def("mfcc_dct(Tensor mel, Tensor dct_mat, int log_mode, Tensor? group_max, int vec_per_group, float top_db) -> Tensor");
  m.def("resample(Tensor wav, Tensor kernel, int orig, int new, int width, int out_len, int[]? band_tap_lo, "
        "int tap_span, Tensor? frag) -> Tensor");
  m.def("resample_frag_build(Tensor kernel, int orig, int new, int width, int[] band_tap_lo, int tap_span) -> Tensor");
  m.def("lfilter(Tensor waveform, Tensor a_coeffs, Tensor b_coeffs, int n_stages, int clamp) -> Tensor");
  m.def("fftconvolve(Tensor x, Tensor y, Tensor? x_row_of, Tensor? y_row_of, int rows, int start, int out_len) -> Tensor");
  m.def("fftconvolve_staged(Tensor x, Tensor y, Tensor? x_row_of, Tensor? y_row_of, int rows, int start, int out_len, "
        "Tensor(a!) workspace, int stages) -> Tensor");
  // round 4: the rest of include/audio_amd.h
  m.def("mel_spectrogram_lognorm(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width, "
        "Tensor band_weights, Tensor? lane_order, Tensor? table400, int n_fft, int hop, int n_frames, float scale, float gain, "
        "Tensor? mean, Tensor? invstddev, int out_frames, int table_sig) -> Tensor");
  m.def("mfcc_frag_build(Tensor dct_mat, int n_mels, int n_mfcc) -> Tensor");
  m.def("mfcc_fused(Tensor wav, Tensor window, Tensor twiddle, Tensor band_lo, Tensor band_width, Tensor band_weights, "
        "Tensor? lane_order, Tensor? table400, Tensor dct_frag, Tensor(a!) group_max, int n_fft, int hop, int pad, bool center, "
        "int pad_mode, int n_frames, float scale, int n_mfcc, float multiplier, float amin, float db_multiplier, float top_db, "
        "int rows_per_group, int table_sig) -> Tensor");
  m.def("istft(Tensor spec, Tensor window, Tensor twiddle, Tensor? inv_envelope, int n_fft, int hop, int pad, bool center, "
        "int pad_mode, int length, float scale, bool adjoint) -> Tensor");
  m.def("istft_f64(Tensor spec, Tensor window, Tensor twiddle, Tensor? inv_envelope, int n_fft, int hop, int pad, bool center, "
        "int pad_mode, int length, float scale, bool adjoint) -> Tensor");
  m.def("spectrogram_f64(Tensor wav, Tensor window, Tensor twiddle, int n_fft, int hop, int pad, bool center, int pad_mode, "
        "int n_frames) -> Tensor");
  m.def("phase_vocoder(Tensor spec, Tensor phase_advance, float rate, bool frame_major_out) -> Tensor");
  m.def("griffinlim_update(Tensor rebuilt, Tensor(a!) tprev, Tensor magnitude, float momentum) -> Tensor");
  m.def("mel_scale(Tensor spec, Tensor band_lo, Tensor band_width, Tensor band_weights) -> Tensor");
  m.def("amplitude_to_db(Tensor x, float multiplier, float amin, float db_multiplier, Tensor(a!)? group_max, int group_size) -> Tensor");
  m.def("amplitude_to_db_clamped(Tensor x, float multiplier, float amin, float db_multiplier, Tensor group_max, int group_size, "
        "float top_db) -> Tensor");
  m.def("db_clamp(Tensor x, Tensor group_max, int group_size, float top_db) -> Tensor");
  m.def("spectrogram_grad(Tensor spec, Tensor dpower, float power) -> Tensor");
  m.def("mel_spectrogram_grad(Tensor(a!) spec, Tensor dmel, Tensor band_lo, Tensor band_width, Tensor band_weights, float power) -> Tensor(a!)");
  m.def("resample_sparse(Tensor wav, Tensor taps_compact, Tensor tap_lo, int orig, int new, int width, int out_len) -> Tensor");
  m.def("kaldi_features(Tensor wav, Tensor window, Tensor twiddle, Tensor? band_lo, Tensor? band_width, Tensor? band_weights, "
        "Tensor? noise, int n_frames, int[] opts, float[] fopts) -> Tensor");
  m.def("lfilter_f64(Tensor waveform, Tensor a_coeffs, Tensor b_coeffs, int n_stages, int clamp) -> Tensor");
  m.def("resample_f64(Tensor wav, Tensor kernel, int orig, int new, int width, int out_len) -> Tensor");
  m.def("fftconvolve_f64(Tensor x, Tensor y, Tensor? x_row_of, Tensor? y_row_of, int rows, int start, int out_len) -> Tensor");
}

STABLE_TORCH_LIBRARY_IMPL(aamd, CUDA, m) {
  m.impl("spectrogram", TORCH_BOX(&spectrogram));
  m.impl("mel_spectrogram", TORCH_BOX(&mel_spectrogram));
  m.impl("mel_spectrogram_db", TORCH_BOX(&mel_spectrogram_db));
  m.impl("mfcc_dct", TORCH_BOX(&mfcc_dct));
  m.impl("resample", TORCH_BOX(&resample));
  m.impl("resample_frag_build", TORCH_BOX(&resample_frag_build));
  m.impl("lfilter", TORCH_BOX(&lfilter));
  m.impl("fftconvolve", TORCH_BOX(&fftconvolve));
  m.impl("fftconvolve_staged", TORCH_BOX(&fftconvolve_staged));
  m.impl("mel_spectrogram_lognorm", TORCH_BOX(&mel_spectrogram_lognorm));
  m.impl("mfcc_frag_build", TORCH_BOX(&mfcc_frag_build));
  m.impl("mfcc_fused", TORCH_BOX(&mfcc_fused));
  m.impl("istft", TORCH_BOX(&istft));
  m.impl("istft_f64", TORCH_BOX(&istft_f64));
  m.impl("spectrogram_f64", TORCH_BOX(&spectrogram_f64));
  m.impl("phase_vocoder", TORCH_BOX(&phase_vocoder));
  m.impl("griffinlim_update", TORCH_BOX(&griffinlim_update));
  m.impl("mel_scale", TORCH_BOX(&mel_scale));
  m.impl("amplitude_to_db", TORCH_BOX(&amplitude_to_db));
  m.impl("amplitude_to_db_clamped", TORCH_BOX(&amplitude_to_db_clamped));
  m.impl("db_clamp", TORCH_BOX(&db_clamp));
  m.impl("spectrogram_grad", TORCH_BOX(&spectrogram_grad));
  m.impl("mel_spectrogram_grad", TORCH_BOX(&mel_spectrogram_grad));
  m.impl("resample_sparse", TORCH_BOX(&resample_sparse));
  m.impl("kaldi_features", TORCH_BOX(&kaldi_features));
  m.impl("lfilter_f64", TORCH_BOX(&lfilter_f64));
  m.impl("resample_f64", TORCH_BOX(&resample_f64));
  m.impl("fftconvolve_f64", TORCH_BOX(&fftconvolve_f64));
}

// The reference's op.  libtorchaudio (when present) has already run
//   STABLE_TORCH_LIBRARY_FRAGMENT(torchaudio, m) { m.def("_lfilter_core_loop(...)"); }
// and a second def of the same schema is an error, so the schema and the CUDA-key kernel are registered by two explicit
// calls made from Python after load_library (audio_amd/_shim.py decides from torch._C._dispatch_has_kernel / the schema
// registry which of them are needed) instead of by static initialisers.
extern "C" {

__attribute__((visibility("default"))) int aamd_torch_define_torchaudio_schema(void) {
  static std::unique_ptr<torch::stable::detail::StableLibrary> frag;
  if (frag) return 0;
  try {
    frag = std::make_unique<torch::stable::detail::StableLibrary>(torch::stable::detail::StableLibrary::Kind::FRAGMENT,
                                                                  "torchaudio", nullptr, __FILE__, __LINE__);
    frag->def("_lfilter_core_loop(Tensor input_signal_windows, Tensor a_coeff_flipped, "
              "Tensor(a!) padded_output_waveform) -> Tensor(a!)");
  } catch (...) {
    frag.reset();
    return -1;
  }
  return 0;
}

__attribute__((visibility("default"))) int aamd_torch_register_torchaudio_cuda(void) {
  static std::unique_ptr<torch::stable::detail::StableLibrary> impl;
  if (impl) return 0;
  try {
    impl = std::make_unique<torch::stable::detail::StableLibrary>(torch::stable::detail::StableLibrary::Kind::IMPL,
                                                                  "torchaudio", "CUDA", __FILE__, __LINE__);
    impl->impl("_lfilter_core_loop", TORCH_BOX(&cuda_lfilter_core_loop));
  } catch (...) {
    impl.reset();
    return -1;
  }
  return 0;
}

__attribute__((visibility("default"))) int aamd_torch_shim_abi(void) { return AAMD_ABI_VERSION; }

}  // extern "C"

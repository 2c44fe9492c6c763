// C ABI of libaudio_amd.so (see include/audio_amd.h): argument validation + kernel launches.
// No torch dependency, no global mutable state except a thread-local error string and a
// per-process cache of immutable device properties.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <string>

#include "../../include/audio_amd.h"
#include "db_mfcc.h"
#include "f64_paths.h"
#include "fftconv.h"
#include "fftconv_os.h"
#include "fftconv_fdr.h"
#include "istft.h"
#include "kaldi_generic.h"
#include "vocoder.h"
#include "stft_pow2.h"
#include "istft400.h"
#include "lfilter.h"
#include "lfilter_wave.h"
#include "melspec400.h"
#include "resample.h"
#include "resample_mfma.h"
#include "stft_generic.h"

using namespace aamd;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define AAMD_CHECK_ARG(cond, msg) \
  do { if (!(cond)) return fail(AAMD_EINVAL, std::string("audio_amd: ") + msg); } while (0)

#define AAMD_HIP(expr)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(AAMD_EHIP, std::string("audio_amd: HIP error: ") + hipGetErrorString(e_) + \
                                 " at " #expr);                                          \
  } while (0)

int launch_check() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AAMD_EHIP, std::string("audio_amd: kernel launch failed: ") + hipGetErrorString(e));
  return AAMD_OK;
}

// Kernel-selection switches (tests / A-B experiments): a process-wide bit mask, initialised ONCE from the environment
// (AAMD_FORCE_GENERIC, AAMD_MEL400_WIDE, AAMD_ISTFT_ATOMIC) and changed afterwards only through
// aamd_set_kernel_policy() -- no getenv() on the launch path.
std::atomic<int> g_policy{-1};

int policy() {
  int p = g_policy.load(std::memory_order_relaxed);
  if (p < 0) {
    p = 0;
    if (std::getenv("AAMD_FORCE_GENERIC") != nullptr) p |= AAMD_POLICY_FORCE_GENERIC;
    if (std::getenv("AAMD_MEL400_WIDE") != nullptr) p |= AAMD_POLICY_MEL400_WIDE;
    if (std::getenv("AAMD_ISTFT_ATOMIC") != nullptr) p |= AAMD_POLICY_ISTFT_ATOMIC;
    if (std::getenv("AAMD_RESAMPLE_FP32") != nullptr) p |= AAMD_POLICY_RESAMPLE_FP32;
    if (std::getenv("AAMD_FFTCONV_NO_FDL") != nullptr) p |= AAMD_POLICY_FFTCONV_NO_FDL;
    if (std::getenv("AAMD_FFTCONV_FDL") != nullptr) p |= AAMD_POLICY_FFTCONV_FDL;
    if (std::getenv("AAMD_RESAMPLE_B32") != nullptr) p |= AAMD_POLICY_RESAMPLE_B32;
    if (std::getenv("AAMD_MEL400_NO_POOL") != nullptr) p |= AAMD_POLICY_MEL400_NO_POOL;
    int expected = -1;
    g_policy.compare_exchange_strong(expected, p);
    p = g_policy.load(std::memory_order_relaxed);
  }
  return p;
}
inline bool force_generic() { return (policy() & AAMD_POLICY_FORCE_GENERIC) != 0; }

// The launches size their grids from, and set function attributes on, the CURRENT device.  In a process that sees
// several GPUs (not the one-process-per-GPU deployment) the caller's tensors may live on another one: make the
// device that owns the buffer current for the duration of the call.  Costs nothing when one GPU is visible.
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(const void* p) {
    static const int n_dev = [] { int n = 1; (void)hipGetDeviceCount(&n); return n; }();
    if (n_dev <= 1 || p == nullptr) return;
    int cur = 0, own = 0;
    if (hipGetDevice(&cur) != hipSuccess) return;
    if (hipPointerGetAttribute(&own, HIP_POINTER_ATTRIBUTE_DEVICE_ORDINAL, const_cast<void*>(p)) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    if (own != cur && own >= 0 && own < n_dev && hipSetDevice(own) == hipSuccess) prev = cur;
  }
  ~DeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

struct DevProps {
  int cu_count = 0;
  size_t lds_per_block = 0;
  size_t lds_per_block_optin = 0;
  bool ok = false;
};

DevProps& dev_props() {
  static DevProps props[64];
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  DevProps& p = props[dev];
  if (!p.ok) {
    hipDeviceProp_t dp;
    if (hipGetDeviceProperties(&dp, dev) == hipSuccess) {
      p.cu_count = dp.multiProcessorCount;
      p.lds_per_block = dp.sharedMemPerBlock;
      p.lds_per_block_optin = dp.sharedMemPerBlockOptin ? dp.sharedMemPerBlockOptin : dp.sharedMemPerBlock;
      p.ok = true;
    } else {
      p.cu_count = 256;
      p.lds_per_block = 64 * 1024;
      p.lds_per_block_optin = 160 * 1024;
    }
  }
  return p;
}

int frames_expected(const aamd_stft_desc* d) {
  int64_t lp = d->length + 2 * (int64_t)d->pad + (d->center ? 2 * (int64_t)(d->n_fft / 2) : 0);
  if (lp < d->n_fft) return -1;
  return (int)(1 + (lp - d->n_fft) / d->hop);
}

int validate_desc(const aamd_stft_desc* d, StftGeom& g) {
  AAMD_CHECK_ARG(d != nullptr, "null stft desc");
  AAMD_CHECK_ARG(d->rows >= 0 && d->length >= 0, "negative sizes");
  AAMD_CHECK_ARG(d->n_fft >= 1 && d->hop >= 1 && d->pad >= 0, "n_fft, hop must be >= 1 and pad >= 0");
  AAMD_CHECK_ARG(d->row_stride >= d->length, "row_stride < length");
  AAMD_CHECK_ARG(d->pad_mode >= 0 && d->pad_mode <= 3, "bad pad_mode");
  if (d->n_fft > 8192) return fail(AAMD_EUNSUPPORTED, "audio_amd: n_fft > 8192 not supported");
  const int64_t l1 = d->length + 2 * (int64_t)d->pad;
  if (d->center && d->pad_mode == AAMD_PAD_REFLECT)
    AAMD_CHECK_ARG(d->n_fft / 2 < l1, "reflect padding needs n_fft/2 < padded length");
  if (d->center && d->pad_mode == AAMD_PAD_CIRCULAR)
    AAMD_CHECK_ARG(d->n_fft / 2 <= l1, "circular padding needs n_fft/2 <= padded length");
  if (d->center && d->pad_mode == AAMD_PAD_REPLICATE) AAMD_CHECK_ARG(l1 >= 1, "empty input");
  const int T = frames_expected(d);
  AAMD_CHECK_ARG(T >= 1, "input shorter than n_fft");
  AAMD_CHECK_ARG(T == d->n_frames, "n_frames does not match 1 + (L' - n_fft)/hop");
  g.rows = d->rows; g.length = d->length; g.row_stride = d->row_stride;
  g.n_fft = d->n_fft; g.hop = d->hop; g.pad = d->pad; g.center = d->center;
  g.pad_mode = d->pad_mode; g.onesided = d->onesided; g.n_frames = d->n_frames;
  g.n_freq = d->onesided ? d->n_fft / 2 + 1 : d->n_fft;
  g.scale = d->scale; g.power = d->power;
  g.n_stages = plan_radices(d->n_fft, g.radix);
  if (g.n_stages < 0) return fail(AAMD_EUNSUPPORTED, "audio_amd: n_fft has too many prime factors");
  return AAMD_OK;
}

int validate_bands(const aamd_mel_bands* b, int n_freq, MelBandsDev& mb) {
  AAMD_CHECK_ARG(b != nullptr, "null mel bands");
  AAMD_CHECK_ARG(b->n_mels >= 1 && b->max_width >= 1 && b->max_width <= n_freq, "bad mel band table");
  AAMD_CHECK_ARG(b->lo && b->width && b->weights, "null mel band pointers");
  mb.n_mels = b->n_mels; mb.max_width = b->max_width;
  mb.lo = b->lo; mb.width = b->width; mb.weights = b->weights; mb.order = b->lane_order;
  mb.table400 = b->table400;
  mb.table_sig = b->table400 ? b->table_sig : 0;
  return AAMD_OK;
}

// n_fft = 256 / 512 / 1024 / 2048, onesided: the register-resident wave FFT of stft_pow2.h
template <int EPI, int E>
int launch_pow2(const StftGeom& g, const MelBandsDev& mb, const float* wav, const float* window,
                const float* twiddle, float* out, hipStream_t s) {
  const int64_t pairs_per_row = (g.n_frames + 1) / 2;
  const int64_t n_pairs = g.rows * pairs_per_row;
  if (n_pairs == 0) return AAMD_OK;
  size_t lds = (size_t)p2::kWaves * p2::Cfg<E>::lds_complex * sizeof(p2::C32);
  if (EPI == EPI_MEL && p2::mel_in_lds(mb.n_mels, mb.max_width))
    lds += (size_t)p2::mel_lds_floats(mb.n_mels, mb.max_width) * sizeof(float);
  auto kern = p2::stft_pow2_kernel<E, EPI>;
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // persistent waves striding over the pairs; workgroups per CU measured best on 256 x 10 s (among 2..16):
  // 4 / 2 / 1 workgroups of 4 waves are resident at 126 / 204 / 256 VGPRs, the grid is two resident rounds
  int64_t blocks = (int64_t)dev_props().cu_count * (E <= 8 ? 8 : E == 16 ? 4 : 2);
  const int64_t need = (n_pairs + p2::kWaves - 1) / p2::kWaves;
  if (blocks > need) blocks = need;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * p2::kWaves), lds, s, g, wav, window,
                     reinterpret_cast<const p2::C32*>(twiddle), mb, out, pairs_per_row, n_pairs);
  return launch_check();
}

template <int EPI>
int launch_generic(const StftGeom& g, const MelBandsDev& mb, const float* wav, const float* window,
                   const float* twiddle, float* out, hipStream_t s) {
  if (g.rows == 0) return AAMD_OK;
  if (g.onesided && !force_generic()) {
    if (g.n_fft == 256) return launch_pow2<EPI, 4>(g, mb, wav, window, twiddle, out, s);
    if (g.n_fft == 512) return launch_pow2<EPI, 8>(g, mb, wav, window, twiddle, out, s);
    if (g.n_fft == 1024) return launch_pow2<EPI, 16>(g, mb, wav, window, twiddle, out, s);
    if (g.n_fft == 2048) return launch_pow2<EPI, 32>(g, mb, wav, window, twiddle, out, s);
  }
  int pb = gen_pairs_per_block(g.n_fft);
  const int pairs_per_row = (g.n_frames + 1) / 2;
  if (pb > pairs_per_row) pb = pairs_per_row;
  const int bpr = (pairs_per_row + pb - 1) / pb;
  const int64_t blocks = g.rows * bpr;
  AAMD_CHECK_ARG(blocks < (1ll << 31), "too many frames for one launch");
  size_t lds = gen_lds_floats(g.n_fft, g.n_freq, pb) * sizeof(float);
  auto kern = stft_generic_kernel<float, EPI>;
  if (lds > dev_props().lds_per_block_optin) {
    // long windows (n_fft ~5 750 .. 8 192): the layout without the LDS twiddle table (stft_generic.h, gen_lds_floats_long)
    lds = gen_lds_floats_long(g.n_fft, pb) * sizeof(float);
    kern = stft_generic_kernel<float, EPI, 1>;
    if (lds > dev_props().lds_per_block_optin) return fail(AAMD_EUNSUPPORTED, "audio_amd: n_fft too large for the LDS");
  }
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kGenThreads), lds, s, g, wav, window,
                     reinterpret_cast<const cplx<float>*>(twiddle), mb, out, pb, bpr);
  return launch_check();
}

bool fft400_eligible(const StftGeom& g) {
  return g.n_fft == 400 && (g.hop == 160 || g.hop == 200 || g.hop == 100) && g.center && g.pad_mode == AAMD_PAD_REFLECT &&
         g.onesided && g.pad == 0 && g.length > 400 && !force_generic();
}

template <int EPI>
int launch_fft400(const StftGeom& g, const MelBandsDev& mb, const float* wav, const float* window,
                  const float* twiddle, float* out, const m400::Epi400& epi, hipStream_t s);

bool mel400_eligible(const StftGeom& g, const MelBandsDev& mb) {
  return fft400_eligible(g) && g.power == 2.0f &&
         m400::mel_ws(mb.max_width) <= m400::kMelMaxTaps + 4 &&
         m400::mel_rounds(mb.n_mels) <= m400::kMelMaxRounds;
}

template <int EPI, int H, typename TIn, int NR, int SIG = 0>
int launch_fft400_nr(const StftGeom& g, const MelBandsDev& mb, const TIn* wav, const float* window,
                     const float* twiddle, float* out, const m400::Epi400& epi, hipStream_t s);

// NR = 4 (n_mels <= 80: band-table control words in registers) or 8 (up to 160 mels); the spectrogram epilogue has
// no mel phase and uses one instantiation
template <int EPI, int H, typename TIn = float>
int launch_fft400_h(const StftGeom& g, const MelBandsDev& mb, const TIn* wav, const float* window,
                    const float* twiddle, float* out, const m400::Epi400& epi, hipStream_t s) {
  if (EPI != m400::EPI400_SPEC && m400::mel_rounds(mb.n_mels) <= 4) {
    // the 80-mel HTK / Slaney banks of the 16 kHz front-ends (headline hop, float input): band reduction compiled for them
    if constexpr (H == 8 && sizeof(TIn) == 4 && std::is_same<TIn, float>::value &&
                  (EPI == m400::EPI400_MEL || EPI == m400::EPI400_MFCC)) {   // (MEL_DB: the straight-line form spills 4 registers)
      if (mb.table_sig == m400::kSigHtk80 && mb.n_mels == 80)
        return launch_fft400_nr<EPI, H, TIn, 4, m400::kSigHtk80>(g, mb, wav, window, twiddle, out, epi, s);
      if (mb.table_sig == m400::kSigSlaney80 && mb.n_mels == 80)
        return launch_fft400_nr<EPI, H, TIn, 4, m400::kSigSlaney80>(g, mb, wav, window, twiddle, out, epi, s);
    }
    return launch_fft400_nr<EPI, H, TIn, 4>(g, mb, wav, window, twiddle, out, epi, s);
  }
  return launch_fft400_nr<EPI, H, TIn, m400::kMelMaxRounds>(g, mb, wav, window, twiddle, out, epi, s);
}

// (Only in builds with -DAAMD_M400_POOLS=1: the product is built without the pools, DESIGN 4.1 "Round 5".)
// Ticket counters of the n_fft = 400 kernel's tail pools (csrc/melspec400.h, pool_tile): one zeroed 64 KB block per (device,
// stream), allocated on the stream's first eligible launch and never freed (at most 256 of them, 16 MB).  The kernel leaves
// every counter at zero, and launches on one stream run one after the other, so the block needs no memset between launches.
// Under stream capture nothing is handed out (no allocation inside a capture, and a captured launch may be replayed on
// another stream beside eager launches of this one): such launches run their static tile runs, as every launch of the
// product does.  In such a build this is the library's only mutable state besides the thread-local error string.
#if AAMD_M400_POOLS
unsigned* mel400_pool_block(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (st != hipStreamCaptureStatusNone) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, unsigned*> blocks;
  std::lock_guard<std::mutex> lock(mu);
  auto it = blocks.find({dev, s});
  if (it != blocks.end()) return it->second;
  if (blocks.size() >= 256) return nullptr;
  constexpr size_t kBytes = 64 * 1024;
  void* p = nullptr;
  if (hipMalloc(&p, kBytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemsetAsync(p, 0, kBytes, s) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return nullptr; }   // ordered in front of the launch
  blocks[{dev, s}] = static_cast<unsigned*>(p);
  return static_cast<unsigned*>(p);
}
#endif

template <int EPI, int H, typename TIn, int NR, int SIG>
int launch_fft400_nr(const StftGeom& g, const MelBandsDev& mb, const TIn* wav, const float* window,
                     const float* twiddle, float* out, const m400::Epi400& epi_in, hipStream_t s) {
  if (g.rows == 0) return AAMD_OK;
  const int tiles_per_row = (g.n_frames + m400::kFramesPerWave - 1) / m400::kFramesPerWave;
  const int64_t n_tiles = g.rows * tiles_per_row;
  AAMD_CHECK_ARG(n_tiles < (1ll << 31), "too many frames for one launch");
  const int wpb = m400::kWavesPerBlock;
  const int wdw = m400::Hop<H>::lds_dwords;
  size_t lds = (EPI == m400::EPI400_SPEC) ? m400::lds_bytes(0, 1, wdw) : m400::lds_bytes(mb.n_mels, mb.max_width, wdw);
  m400::Epi400 epi = epi_in;
  if (EPI == m400::EPI400_MFCC && H != 10) {
    // hop 100 / 160: the DCT fragments sit next to the band table in LDS (the kernel's choice per instantiation, melspec400.h
    // kFragLds; mfcc_fused_ok has checked that they fit); hop 200: read from the cache-resident table
    lds = m400::lds_bytes(mb.n_mels, mb.max_width, wdw, true);
    epi.frag_in_lds = 1;
  }
  if (lds > dev_props().lds_per_block_optin)
    return fail(AAMD_EUNSUPPORTED, "audio_amd: mel filterbank too large for the LDS of this device");
  auto kern = m400::melspec400_kernel<0, EPI, H, TIn, NR, SIG>;
#ifdef AAMD_LAB    // tools-only instantiations are compiled into the lab library only (python -m audio_amd._build --lab)
  if (EPI == m400::EPI400_MFCC && epi.lab != 0) kern = m400::melspec400_kernel<(EPI == m400::EPI400_MFCC ? 524288 : 0), EPI, H, TIn, NR, SIG>;
#endif
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // persistent grid: ONE 12-wave workgroup per CU; each owns a contiguous run of tiles (6 frames
  // each) that its waves claim dynamically
  int64_t blocks = dev_props().cu_count;
  const int64_t need = (n_tiles + wpb - 1) / wpb;
  if (blocks > need) blocks = need;
  if (blocks >= 8) blocks -= blocks % 8;  // XCD remap wants a multiple of 8
  if (blocks < 1) blocks = 1;
  const int tiles_per_block = (int)((n_tiles + blocks - 1) / blocks);
  // tail pools: the last P tiles of every workgroup's run are shared with the workgroups of the other XCDs (melspec400.h, pool_tile)
  epi.pool = nullptr;
  epi.pool_p = 0;
#if AAMD_M400_POOLS
  {
#ifdef AAMD_LAB
    static const int lab_p = [] { const char* e = std::getenv("AAMD_MEL400_POOL_P"); return e ? std::atoi(e) : -1; }();   // tools only
#else
    constexpr int lab_p = -1;
#endif
    int P = lab_p >= 0 ? lab_p : m400::pool_share(tiles_per_block);
    if (P > tiles_per_block) P = tiles_per_block;
    const bool fixup_pass = (EPI == m400::EPI400_MFCC) && epi.fixup != 0;
    if (P > 0 && !fixup_pass && epi.lab == 0 && (policy() & AAMD_POLICY_MEL400_NO_POOL) == 0 &&
        (size_t)m400::pool_count((int)blocks) * m400::kPoolStride * sizeof(unsigned) <= 64 * 1024) {
      epi.pool = mel400_pool_block(s);
      epi.pool_p = epi.pool ? P : 0;
    }
  }
#endif
  // 16-B paths: LDS-DMA staging of the waveform, dwordx4 stores of the output rows
  const int in_aligned = (reinterpret_cast<uintptr_t>(wav) % 16 == 0) && (g.row_stride % (16 / (int)sizeof(TIn)) == 0);
  // mel rows leave as 4-byte stores straight from the accumulators: the LDS pipe is this kernel's
  // bottleneck and the LDS-staged dwordx4 path measured 3-4 us slower (AAMD_MEL400_WIDE=1 selects it)
  const bool want_wide = (EPI == m400::EPI400_SPEC) || (policy() & AAMD_POLICY_MEL400_WIDE) != 0;
  const int out_wide = want_wide && (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                       (EPI == m400::EPI400_SPEC || mb.n_mels % 4 == 0);
  if (EPI == m400::EPI400_SPEC && !out_wide)
    return fail(AAMD_EINVAL, "audio_amd: spectrogram output buffer must be 16-byte aligned");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wpb), lds, s, wav, window,
                     twiddle, mb, out, g.rows, g.length, g.row_stride, g.n_frames, g.scale,
                     tiles_per_row, n_tiles, tiles_per_block, in_aligned, out_wide, epi);
  return launch_check();
}

template <int EPI>
int launch_fft400(const StftGeom& g, const MelBandsDev& mb, const float* wav, const float* window,
                  const float* twiddle, float* out, const m400::Epi400& epi, hipStream_t s) {
  switch (g.hop) {   // hop = 20 H
    case 100: return launch_fft400_h<EPI, 5>(g, mb, wav, window, twiddle, out, epi, s);
    case 200: return launch_fft400_h<EPI, 10>(g, mb, wav, window, twiddle, out, epi, s);
    default: return launch_fft400_h<EPI, 8>(g, mb, wav, window, twiddle, out, epi, s);
  }
}

int grid_for(int64_t n, int per_block, int max_blocks) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b > max_blocks) b = max_blocks;
  if (b < 1) b = 1;
  return (int)b;
}

template <int D>
int launch_lfilter(const float* x, const float* a, const float* b, float* y, int64_t n_seq,
                   int channels, int64_t length, int n_order, int n_rows, int n_stages, int clamp,
                   hipStream_t s) {
  using L = LfLds<D>;
  const size_t lds = L::bytes(n_stages);
  if (lds > 160 * 1024) return fail(AAMD_EUNSUPPORTED, "audio_amd: lfilter cascade too long for LDS");
  auto kern = lfilter_kernel<D>;
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int blocks = grid_for(n_seq, 1, dev_props().cu_count * 8);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kLfThreads), lds, s, x, a, b, y, n_seq, channels,
                     length, n_order, n_rows, n_stages, clamp);
  return launch_check();
}

}  // namespace

extern "C" {

int aamd_abi_version(void) { return AAMD_ABI_VERSION; }

int aamd_set_kernel_policy(int flags) {
  const int prev = policy();
  if (flags >= 0) g_policy.store(flags & (AAMD_POLICY_FORCE_GENERIC | AAMD_POLICY_MEL400_WIDE | AAMD_POLICY_ISTFT_ATOMIC |
                                          AAMD_POLICY_RESAMPLE_FP32 | AAMD_POLICY_FFTCONV_NO_FDL | AAMD_POLICY_FFTCONV_FDL |
                                          AAMD_POLICY_FFTCONV_COMPLEX | AAMD_POLICY_RESAMPLE_B32 | AAMD_POLICY_MEL400_NO_POOL));
  return prev;
}

const char* aamd_last_error(void) { return g_err.c_str(); }

int64_t aamd_mel400_table_dwords(int32_t n_mels, int32_t max_width) {
  if (n_mels < 1 || max_width < 1) return 0;
  if (m400::mel_ws(max_width) > m400::kMelMaxTaps + 4 || m400::mel_rounds(n_mels) > m400::kMelMaxRounds) return 0;
  return m400::mel_tab_dwords(n_mels, max_width);
}

int aamd_mel400_table_build(const aamd_mel_bands* bands, float* table_out, void* stream) {
  DeviceScope dev_scope_(table_out);
  MelBandsDev mb{};
  int rc = validate_bands(bands, 201, mb);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(table_out != nullptr, "null table buffer");
  if (aamd_mel400_table_dwords(mb.n_mels, mb.max_width) == 0)
    return fail(AAMD_EUNSUPPORTED, "audio_amd: filterbank outside the radix-20x20 kernel (n_mels > 160 or band > 62 bins)");
  mb.table400 = nullptr;
  hipLaunchKernelGGL(m400::mel_tab_build_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mb, table_out);
  return launch_check();
}

int aamd_device_info(char* name, int32_t name_len, int32_t* cu_count, int64_t* hbm_bytes) {
  int dev = 0;
  AAMD_HIP(hipGetDevice(&dev));
  hipDeviceProp_t dp;
  AAMD_HIP(hipGetDeviceProperties(&dp, dev));
  if (name && name_len > 0) {
    std::snprintf(name, (size_t)name_len, "%s", dp.gcnArchName);
  }
  if (cu_count) *cu_count = dp.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)dp.totalGlobalMem;
  return AAMD_OK;
}

// Box probe (bench.py `box_calibration`): every workgroup runs the same fixed chain of dependent fp32 FMAs in all of its
// waves and records shader cycles (s_memtime), 100 MHz wall ticks (s_memrealtime) and the XCD it ran on.  cycles / ticks is
// the shader clock this box sustains under vector-ALU load; the spread of the per-XCD means is the XCD skew.  Boxes of one
// pool ran the same binary between 65.9 and 72.5 us: this is what the bench line reports beside its
// number so that a slow box can be told from a slow kernel.
__global__ void __launch_bounds__(256) box_probe_kernel(long long* __restrict__ rec, int iters) {
  float a0 = 1.0f + threadIdx.x * 1e-6f, a1 = 0.5f, a2 = 0.25f, a3 = 0.125f;
  const float m = 0.99999f, c = 1e-7f;
  const long long c0 = (long long)clock64(), w0 = (long long)wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
    }
  }
  const long long c1 = (long long)clock64(), w1 = (long long)wall_clock64();
  unsigned xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) {
    long long* r = rec + 4 * (long long)blockIdx.x;
    r[0] = c1 - c0; r[1] = w1 - w0; r[2] = (long long)(xcc & 15u); r[3] = w0;
  }
  if (a0 + a1 + a2 + a3 == 123456.0f) rec[0] = 0;       // keeps the chains alive
}

int aamd_box_probe(int64_t* rec, int32_t n_blocks, int32_t iters, void* stream) {
  DeviceScope dev_scope_(rec);
  AAMD_CHECK_ARG(rec != nullptr && n_blocks >= 1 && iters >= 1, "box probe: rec[4 * n_blocks], n_blocks >= 1, iters >= 1");
  hipLaunchKernelGGL(box_probe_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<long long*>(rec), (int)iters);
  return launch_check();
}

int aamd_spectrogram_f32(const float* wav, const float* window, const float* twiddle, float* out,
                         const aamd_stft_desc* desc, void* stream) {
  DeviceScope dev_scope_(wav);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(wav && window && twiddle && out, "null buffer");
  MelBandsDev mb{};
  if (fft400_eligible(g) && reinterpret_cast<uintptr_t>(out) % 16 == 0) {   // power <= 0: complex rows
    m400::Epi400 epi{};
    epi.power = g.power;
    return launch_fft400<m400::EPI400_SPEC>(g, mb, wav, window, twiddle, out, epi, (hipStream_t)stream);
  }
  return launch_generic<EPI_SPEC>(g, mb, wav, window, twiddle, out, (hipStream_t)stream);
}

int aamd_melspectrogram_f32(const float* wav, const float* window, const float* twiddle,
                            const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc,
                            void* stream) {
  DeviceScope dev_scope_(wav);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(wav && window && twiddle && out, "null buffer");
  AAMD_CHECK_ARG(desc->power > 0.0f, "mel spectrogram needs power > 0");
  AAMD_CHECK_ARG(desc->onesided, "mel spectrogram needs a onesided spectrum");
  MelBandsDev mb;
  rc = validate_bands(bands, g.n_freq, mb);
  if (rc != AAMD_OK) return rc;
  if (mel400_eligible(g, mb))
    return launch_fft400<m400::EPI400_MEL>(g, mb, wav, window, twiddle, out, m400::Epi400{}, (hipStream_t)stream);
  return launch_generic<EPI_MEL>(g, mb, wav, window, twiddle, out, (hipStream_t)stream);
}

// ---- MFCC in one kernel (+ a fix-up launch for clamped tiles) --------------------------------------------------------
namespace {
bool mfcc_fused_ok(const StftGeom& g, const MelBandsDev& mb, int n_mfcc) {
  if (!(mel400_eligible(g, mb) && mb.n_mels == m400::kMfccMels && n_mfcc >= 4 && n_mfcc <= 16 * m400::kMfccMT && n_mfcc % 4 == 0))
    return false;
  // hop 100 / 160 keep the DCT fragments in LDS: a band table too wide to leave them room takes the two-kernel path
  const int wdw = g.hop == 100 ? m400::Hop<5>::lds_dwords : g.hop == 200 ? m400::Hop<10>::lds_dwords : m400::Hop<8>::lds_dwords;
  return g.hop == 200 || m400::lds_bytes(mb.n_mels, mb.max_width, wdw, true) <= dev_props().lds_per_block_optin;
}
}  // namespace

int32_t aamd_mfcc_frag_floats(void) { return m400::kMfccFragFloats; }

int64_t aamd_mfcc_fused_tiles(const aamd_stft_desc* desc) {
  StftGeom g;
  if (validate_desc(desc, g) != AAMD_OK) return -1;
  return g.rows * ((g.n_frames + m400::kFramesPerWave - 1) / m400::kFramesPerWave);
}

int aamd_mfcc_fused_supported(const aamd_stft_desc* desc, const aamd_mel_bands* bands, int32_t n_mfcc) {
  StftGeom g;
  if (validate_desc(desc, g) != AAMD_OK) return 0;
  MelBandsDev mb;
  if (validate_bands(bands, g.n_freq, mb) != AAMD_OK) return 0;
  return mfcc_fused_ok(g, mb, n_mfcc) ? 1 : 0;
}

int aamd_mfcc_frag_build(const float* dct, int32_t n_mels, int32_t n_mfcc, float* frag, void* stream) {
  DeviceScope dev_scope_(dct);
  AAMD_CHECK_ARG(dct && frag, "null buffer");
  AAMD_CHECK_ARG(n_mels >= 1 && n_mels <= m400::kMfccMels && n_mfcc >= 1 && n_mfcc <= 16 * m400::kMfccMT, "bad sizes");
  hipLaunchKernelGGL(m400::mfcc_frag_build_kernel, dim3(15), dim3(256), 0, (hipStream_t)stream, dct, n_mels, n_mfcc, frag);
  return launch_check();
}

int aamd_mfcc_fused_f32(const float* wav, const float* window, const float* twiddle, const aamd_mel_bands* bands,
                        float* out, const aamd_stft_desc* desc, const aamd_mfcc_fused* f, void* stream) {
  DeviceScope dev_scope_(wav);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(wav && window && twiddle && out && f, "null buffer");
  AAMD_CHECK_ARG(f->dct_frag && f->group_max && f->tile_min, "the fused MFCC needs dct_frag, group_max and tile_min");
  AAMD_CHECK_ARG(f->rows_per_group >= 1 && (f->pass == 0 || f->pass == 1), "bad rows_per_group / pass");
  AAMD_CHECK_ARG(f->fix_count != nullptr, "the fused MFCC needs fix_count in both passes (pass 0 resets it)");
  AAMD_CHECK_ARG(reinterpret_cast<uintptr_t>(out) % 16 == 0 && reinterpret_cast<uintptr_t>(f->dct_frag) % 16 == 0,
                 "out and dct_frag must be 16-byte aligned");
  MelBandsDev mb;
  rc = validate_bands(bands, g.n_freq, mb);
  if (rc != AAMD_OK) return rc;
  if (!mfcc_fused_ok(g, mb, f->n_mfcc))
    return fail(AAMD_EUNSUPPORTED, "audio_amd: the fused MFCC serves n_fft 400 / hop 100, 160, 200 / 80 mels / n_mfcc <= 48 "
                                   "(multiple of 4); use aamd_melspectrogram_db_f32 + aamd_mfcc_dct_f32");
  m400::Epi400 epi{};
  epi.multiplier = f->multiplier; epi.amin = f->amin; epi.db_sub = f->multiplier * f->db_multiplier;
  epi.group_max = f->group_max; epi.rows_per_group = f->rows_per_group;
  epi.dct_frag = f->dct_frag; epi.n_mfcc = f->n_mfcc; epi.top_db = f->top_db; epi.tile_min = f->tile_min;
  epi.fix_count = f->fix_count; epi.fixup = f->pass; epi.fix_list = f->tile_list;
  // (pass 1: every workgroup of the fix-up launch finds the flagged tiles among its own strided share of the tile minima --
  // no list kernel between the passes; fix_count was reset by pass 0 of this call and collects what the workgroups redo)
  if (f->pass == 1) AAMD_CHECK_ARG(f->fix_count && f->tile_list, "pass 1 of the fused MFCC needs fix_count and tile_list");
#ifdef AAMD_LAB
  static const int mfcc_lab = [] { const char* e = std::getenv("AAMD_MFCC_LAB"); return e ? std::atoi(e) : 0; }();   // tools only
  epi.lab = mfcc_lab;
#endif
  hipStream_t s = (hipStream_t)stream;
  switch (g.hop) {
    case 100: return launch_fft400_nr<m400::EPI400_MFCC, 5, float, 4>(g, mb, wav, window, twiddle, out, epi, s);
    case 200: return launch_fft400_nr<m400::EPI400_MFCC, 10, float, 4>(g, mb, wav, window, twiddle, out, epi, s);
    default: return launch_fft400_nr<m400::EPI400_MFCC, 8, float, 4>(g, mb, wav, window, twiddle, out, epi, s);
  }
}

int aamd_melspectrogram_db_f32(const float* wav, const float* window, const float* twiddle,
                               const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc,
                               float multiplier, float amin, float db_multiplier, float* group_max,
                               int64_t rows_per_group, void* stream) {
  DeviceScope dev_scope_(wav);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(wav && window && twiddle && out, "null buffer");
  AAMD_CHECK_ARG(desc->power > 0.0f, "mel spectrogram needs power > 0");
  AAMD_CHECK_ARG(desc->onesided, "mel spectrogram needs a onesided spectrum");
  AAMD_CHECK_ARG(group_max == nullptr || rows_per_group >= 1, "rows_per_group must be >= 1");
  MelBandsDev mb;
  rc = validate_bands(bands, g.n_freq, mb);
  if (rc != AAMD_OK) return rc;
  if (mel400_eligible(g, mb)) {
    m400::Epi400 epi{};
    epi.multiplier = multiplier; epi.amin = amin; epi.db_sub = multiplier * db_multiplier;
    epi.group_max = group_max; epi.rows_per_group = rows_per_group < 1 ? 1 : rows_per_group;
    return launch_fft400<m400::EPI400_MEL_DB>(g, mb, wav, window, twiddle, out, epi, (hipStream_t)stream);
  }
  rc = launch_generic<EPI_MEL>(g, mb, wav, window, twiddle, out, (hipStream_t)stream);
  if (rc != AAMD_OK) return rc;
  return aamd_amplitude_to_db_f32(out, out, g.rows * g.n_frames * (int64_t)mb.n_mels, multiplier, amin,
                                  db_multiplier, group_max,
                                  (rows_per_group < 1 ? 1 : rows_per_group) * g.n_frames * (int64_t)mb.n_mels,
                                  stream);
}

int aamd_melspectrogram_lognorm_f32(const float* wav, const float* window, const float* twiddle,
                                    const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc, float gain,
                                    const float* mean, const float* invstddev, int64_t out_frames, void* stream) {
  DeviceScope dev_scope_(wav);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(wav && window && twiddle && out && mean && invstddev, "null buffer");
  AAMD_CHECK_ARG(desc->power > 0.0f, "mel spectrogram needs power > 0");
  AAMD_CHECK_ARG(desc->onesided, "mel spectrogram needs a onesided spectrum");
  AAMD_CHECK_ARG(out_frames >= desc->n_frames, "out_frames must be >= n_frames");
  MelBandsDev mb;
  rc = validate_bands(bands, g.n_freq, mb);
  if (rc != AAMD_OK) return rc;
  if (mel400_eligible(g, mb)) {
    m400::Epi400 epi{};
    epi.gain = gain; epi.mean = mean; epi.invstd = invstddev; epi.out_frames = out_frames;
    return launch_fft400<m400::EPI400_MEL_NORM>(g, mb, wav, window, twiddle, out, epi, (hipStream_t)stream);
  }
  if (out_frames != g.n_frames)
    return fail(AAMD_EUNSUPPORTED, "audio_amd: padded feature rows need the n_fft = 400 fast path");
  rc = launch_generic<EPI_MEL>(g, mb, wav, window, twiddle, out, (hipStream_t)stream);
  if (rc != AAMD_OK) return rc;
  const int64_t n = g.rows * g.n_frames * (int64_t)mb.n_mels;
  if (n == 0) return AAMD_OK;
  hipLaunchKernelGGL(lognorm_kernel, dim3(grid_for(n, 256, dev_props().cu_count * 16)), dim3(256), 0,
                     (hipStream_t)stream, out, n, mb.n_mels, gain, mean, invstddev);
  return launch_check();
}

int aamd_melspectrogram_pcm16_f32(const int16_t* wav, const float* window, const float* twiddle,
                                  const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc, float gain,
                                  const float* mean, const float* invstddev, int64_t out_frames, void* stream) {
  DeviceScope dev_scope_(wav);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(wav && window && twiddle && out, "null buffer");
  AAMD_CHECK_ARG((mean == nullptr) == (invstddev == nullptr), "mean and invstddev come together");
  AAMD_CHECK_ARG(desc->power > 0.0f && desc->onesided, "mel spectrogram needs power > 0 and a onesided spectrum");
  MelBandsDev mb;
  rc = validate_bands(bands, g.n_freq, mb);
  if (rc != AAMD_OK) return rc;
  if (!mel400_eligible(g, mb) || (g.hop != 160 && g.hop != 200))
    return fail(AAMD_EUNSUPPORTED, "audio_amd: int16 PCM input is served by the n_fft = 400, hop 160 / 200 kernel only");
  hipStream_t s = (hipStream_t)stream;
  if (mean != nullptr) {
    AAMD_CHECK_ARG(out_frames >= desc->n_frames, "out_frames must be >= n_frames");
    m400::Epi400 epi{};
    epi.gain = gain; epi.mean = mean; epi.invstd = invstddev; epi.out_frames = out_frames;
    return g.hop == 160 ? launch_fft400_h<m400::EPI400_MEL_NORM, 8, int16_t>(g, mb, wav, window, twiddle, out, epi, s)
                        : launch_fft400_h<m400::EPI400_MEL_NORM, 10, int16_t>(g, mb, wav, window, twiddle, out, epi, s);
  }
  return g.hop == 160 ? launch_fft400_h<m400::EPI400_MEL, 8, int16_t>(g, mb, wav, window, twiddle, out, m400::Epi400{}, s)
                      : launch_fft400_h<m400::EPI400_MEL, 10, int16_t>(g, mb, wav, window, twiddle, out, m400::Epi400{}, s);
}

// Reduced-precision waveforms (the reference takes any floating dtype, functional/functional.py:1413-1414 and every
// transform's forward; a half / bfloat16 pipeline hands such tensors over): read as they are, converted to float in the gather
// of the radix-20x20 kernel -- half the input bytes, no separate cast pass.  Arithmetic and output stay float32 (the host
// casts the result to the input dtype, which is what the reference returns).  n_fft = 400, hop 160 / 200 only; every other
// shape takes the host's cast + the float kernels.
int aamd_melspectrogram_lowp_f32(const void* wav, int32_t wav_dtype, const float* window, const float* twiddle,
                                 const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc, void* stream) {
  DeviceScope dev_scope_(wav);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(wav && window && twiddle && out, "null buffer");
  AAMD_CHECK_ARG(wav_dtype == AAMD_DTYPE_F16 || wav_dtype == AAMD_DTYPE_BF16, "wav_dtype: AAMD_DTYPE_F16 or AAMD_DTYPE_BF16");
  AAMD_CHECK_ARG(desc->power > 0.0f && desc->onesided, "mel spectrogram needs power > 0 and a onesided spectrum");
  MelBandsDev mb;
  rc = validate_bands(bands, g.n_freq, mb);
  if (rc != AAMD_OK) return rc;
  if (!mel400_eligible(g, mb) || (g.hop != 160 && g.hop != 200))
    return fail(AAMD_EUNSUPPORTED, "audio_amd: half / bfloat16 input is read directly by the n_fft = 400, hop 160 / 200 kernel only");
  hipStream_t s = (hipStream_t)stream;
  const m400::Epi400 epi{};
  if (wav_dtype == AAMD_DTYPE_F16) {
    const _Float16* w = static_cast<const _Float16*>(wav);
    return g.hop == 160 ? launch_fft400_h<m400::EPI400_MEL, 8, _Float16>(g, mb, w, window, twiddle, out, epi, s)
                        : launch_fft400_h<m400::EPI400_MEL, 10, _Float16>(g, mb, w, window, twiddle, out, epi, s);
  }
  const __bf16* w = static_cast<const __bf16*>(wav);
  return g.hop == 160 ? launch_fft400_h<m400::EPI400_MEL, 8, __bf16>(g, mb, w, window, twiddle, out, epi, s)
                      : launch_fft400_h<m400::EPI400_MEL, 10, __bf16>(g, mb, w, window, twiddle, out, epi, s);
}

int aamd_melspectrogram_pcm16_interleaved_f32(const int16_t* pcm, int32_t channels, const float* window, const float* twiddle,
                                              const aamd_mel_bands* bands, float* out, const aamd_stft_desc* desc,
                                              float gain, const float* mean, const float* invstddev, int64_t out_frames,
                                              void* stream) {
  if (channels == 1)
    return aamd_melspectrogram_pcm16_f32(pcm, window, twiddle, bands, out, desc, gain, mean, invstddev, out_frames, stream);
  if (channels != 2)
    return fail(AAMD_EUNSUPPORTED, "audio_amd: interleaved PCM is read directly for 1 or 2 channels only (transpose first)");
  DeviceScope dev_scope_(pcm);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(pcm && window && twiddle && out, "null buffer");
  AAMD_CHECK_ARG(g.rows % 2 == 0, "rows must be clips * channels");
  AAMD_CHECK_ARG((mean == nullptr) == (invstddev == nullptr), "mean and invstddev come together");
  AAMD_CHECK_ARG(desc->power > 0.0f && desc->onesided, "mel spectrogram needs power > 0 and a onesided spectrum");
  AAMD_CHECK_ARG(reinterpret_cast<uintptr_t>(pcm) % 4 == 0, "interleaved stereo PCM must be 4-byte aligned");
  MelBandsDev mb;
  rc = validate_bands(bands, g.n_freq, mb);
  if (rc != AAMD_OK) return rc;
  if (!mel400_eligible(g, mb) || (g.hop != 160 && g.hop != 200))
    return fail(AAMD_EUNSUPPORTED, "audio_amd: int16 PCM input is served by the n_fft = 400, hop 160 / 200 kernel only");
  hipStream_t s = (hipStream_t)stream;
  const m400::PcmStereo* w2 = reinterpret_cast<const m400::PcmStereo*>(pcm);      // one (L, R) word per sample time
  if (mean != nullptr) {
    AAMD_CHECK_ARG(out_frames >= desc->n_frames, "out_frames must be >= n_frames");
    m400::Epi400 epi{};
    epi.gain = gain; epi.mean = mean; epi.invstd = invstddev; epi.out_frames = out_frames;
    return g.hop == 160 ? launch_fft400_h<m400::EPI400_MEL_NORM, 8, m400::PcmStereo>(g, mb, w2, window, twiddle, out, epi, s)
                        : launch_fft400_h<m400::EPI400_MEL_NORM, 10, m400::PcmStereo>(g, mb, w2, window, twiddle, out, epi, s);
  }
  return g.hop == 160 ? launch_fft400_h<m400::EPI400_MEL, 8, m400::PcmStereo>(g, mb, w2, window, twiddle, out, m400::Epi400{}, s)
                      : launch_fft400_h<m400::EPI400_MEL, 10, m400::PcmStereo>(g, mb, w2, window, twiddle, out, m400::Epi400{}, s);
}

int aamd_spectrogram_grad_f32(const float* spec, const float* dpower, float* out, int64_t n, float power, void* stream) {
  DeviceScope dev_scope_(spec);
  AAMD_CHECK_ARG(spec && dpower && out, "null buffer");
  AAMD_CHECK_ARG(n >= 0 && power > 0.0f, "bad sizes / power");
  if (n == 0) return AAMD_OK;
  hipLaunchKernelGGL(spec_grad_kernel, dim3(grid_for(n, 256, dev_props().cu_count * 16)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float2*>(spec), dpower, reinterpret_cast<float2*>(out), n, power);
  return launch_check();
}

int aamd_melspectrogram_grad_f32(float* spec_inout, const float* dmel, const aamd_mel_bands* bands_t, int64_t n_vec,
                                 int32_t n_freq, int32_t n_mels, float power, void* stream) {
  DeviceScope dev_scope_(spec_inout);
  AAMD_CHECK_ARG(spec_inout && dmel, "null buffer");
  AAMD_CHECK_ARG(n_vec >= 0 && n_freq >= 1 && n_mels >= 1 && power > 0.0f, "bad sizes / power");
  MelBandsDev bt;
  int rc = validate_bands(bands_t, n_mels, bt);          // table of fb^T: one band of mels per bin
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(bt.n_mels == n_freq, "the transposed band table must have one band per bin");
  const int64_t n = n_vec * n_freq;
  if (n == 0) return AAMD_OK;
  hipLaunchKernelGGL(mel_grad_kernel, dim3(grid_for(n, 256, dev_props().cu_count * 16)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<float2*>(spec_inout), dmel, bt, n_vec, n_mels, power);
  return launch_check();
}

int aamd_kaldi_features_f32(const float* wav, const float* window, const float* twiddle, const aamd_mel_bands* bands,
                            float* out, const aamd_kaldi_desc* d, void* stream) {
  DeviceScope dev_scope_(wav);
  AAMD_CHECK_ARG(d != nullptr && wav && window && twiddle && out, "null buffer");
  AAMD_CHECK_ARG(d->n_samples >= 0 && d->n_frames >= 0, "negative sizes");
  AAMD_CHECK_ARG(d->shift >= 1 && d->win >= 2 && d->win <= d->n_fft, "need shift >= 1 and 2 <= win <= n_fft");
  AAMD_CHECK_ARG(d->preemphasis >= 0.0f && d->preemphasis <= 1.0f, "preemphasis must be in [0, 1]");
  AAMD_CHECK_ARG(d->n_fft % 2 == 0, "the padded window must be even (compliance/kaldi.py:139-141)");
  AAMD_CHECK_ARG(d->dither == 0.0f || d->noise != nullptr, "dither needs the noise buffer");
  if (d->n_frames == 0) return AAMD_OK;
  MelBandsDev mb{};
  if (bands != nullptr) {
    int rc = validate_bands(bands, d->n_fft / 2 + 1, mb);
    if (rc != AAMD_OK) return rc;
    AAMD_CHECK_ARG(d->n_cols >= mb.n_mels && d->first_col >= 0 && d->first_col + mb.n_mels <= d->n_cols &&
                   d->energy_col < d->n_cols, "bad output columns");
  }
  p2::KaldiGeom kg{};
  kg.n_samples = d->n_samples; kg.n_frames = d->n_frames; kg.shift = d->shift; kg.win = d->win;
  kg.snip_edges = d->snip_edges; kg.pad_left = d->win / 2 - d->shift / 2;
  kg.preemph = d->preemphasis; kg.remove_dc = d->remove_dc_offset; kg.raw_energy = d->raw_energy;
  kg.log_energy_floor = d->energy_floor > 0.0f ? std::log(d->energy_floor) : -INFINITY;
  kg.eps = 1.1920928955078125e-07f;
  kg.use_power = d->use_power; kg.use_log = d->use_log;
  kg.energy_col = d->energy_col; kg.first_col = d->first_col; kg.n_cols = d->n_cols;
  kg.noise = d->dither != 0.0f ? d->noise : nullptr; kg.dither = d->dither;
  kg.n_utt = d->n_utt > 1 ? d->n_utt : 1;
  kg.utt_stride = d->n_utt > 1 ? d->utt_stride : d->n_samples;
  AAMD_CHECK_ARG(kg.utt_stride >= d->n_samples, "utt_stride < n_samples");
  const bool pow2 = d->n_fft == 256 || d->n_fft == 512 || d->n_fft == 1024 || d->n_fft == 2048;
  if (!pow2 || force_generic()) {
    // any even padded window: mixed-radix Stockham stages in LDS (csrc/kaldi_generic.h)
    kgen::Plan plan{};
    plan.n_fft = d->n_fft;
    plan.n_stages = plan_radices(d->n_fft, plan.radix);
    if (plan.n_stages < 0 || d->n_fft > 8192)
      return fail(AAMD_EUNSUPPORTED, "audio_amd: padded window too long / too many prime factors for the Kaldi front-end");
    const int pb = kgen::pairs_per_block(d->n_fft);
    size_t lds = kgen::lds_floats(d->n_fft, pb) * sizeof(float);
    const bool long_win = lds > dev_props().lds_per_block_optin;     // ~5 750 .. 8 192: the layout without the LDS twiddle table
    if (long_win) lds = kgen::lds_floats_long(d->n_fft, pb) * sizeof(float);
    if (lds > dev_props().lds_per_block_optin)
      return fail(AAMD_EUNSUPPORTED, "audio_amd: padded window too long for the LDS");
    const int64_t bpu = (d->n_frames + 2 * pb - 1) / (2 * pb);
    const int64_t nblk = bpu * kg.n_utt;
    AAMD_CHECK_ARG(nblk < (1ll << 31), "too many frames for one launch");
    const auto* twg = reinterpret_cast<const cplx<float>*>(twiddle);
    if (bands == nullptr) {
      auto kk = long_win ? kgen::kaldi_generic_kernel<0, 1> : kgen::kaldi_generic_kernel<0>;
      if (lds > 48 * 1024)
        AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kk, dim3((unsigned)nblk), dim3(kgen::kThreads), lds, (hipStream_t)stream, kg, plan, pb, (int)bpu, wav,
                         window, twg, mb, out);
    } else {
      auto kk = long_win ? kgen::kaldi_generic_kernel<1, 1> : kgen::kaldi_generic_kernel<1>;
      if (lds > 48 * 1024)
        AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(kk, dim3((unsigned)nblk), dim3(kgen::kThreads), lds, (hipStream_t)stream, kg, plan, pb, (int)bpu, wav,
                         window, twg, mb, out);
    }
    return launch_check();
  }
  const int64_t n_pairs = (d->n_frames + 1) / 2 * kg.n_utt;
  int64_t blocks = (int64_t)dev_props().cu_count * 4;
  const int64_t need = (n_pairs + p2::kWaves - 1) / p2::kWaves;
  if (blocks > need) blocks = need;
  const auto* twc = reinterpret_cast<const p2::C32*>(twiddle);
#define AAMD_KALDI(EE, MODE)                                                                                       \
  {                                                                                                                \
    const size_t lds2 = (size_t)p2::kWaves * p2::Cfg<EE>::lds_complex * sizeof(p2::C32);                           \
    auto k2 = p2::kaldi_pow2_kernel<EE, MODE>;                                                                     \
    if (lds2 > 48 * 1024)                                                                                          \
      AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)lds2));                                                                    \
    hipLaunchKernelGGL(k2, dim3((unsigned)blocks), dim3(64 * p2::kWaves), lds2, (hipStream_t)stream, kg, wav,      \
                       window, twc, mb, out);                                                                      \
  }
  if (bands == nullptr) {
    if (d->n_fft == 256) AAMD_KALDI(4, 0) else if (d->n_fft == 512) AAMD_KALDI(8, 0) else if (d->n_fft == 1024) AAMD_KALDI(16, 0) else AAMD_KALDI(32, 0)
  } else {
    if (d->n_fft == 256) AAMD_KALDI(4, 1) else if (d->n_fft == 512) AAMD_KALDI(8, 1) else if (d->n_fft == 1024) AAMD_KALDI(16, 1) else AAMD_KALDI(32, 1)
  }
#undef AAMD_KALDI
  return launch_check();
}

int aamd_istft_f32(const float* spec, const float* window, const float* twiddle, const float* inv_envelope,
                   float* out, const aamd_stft_desc* desc, int32_t adjoint, void* stream) {
  DeviceScope dev_scope_(spec);
  AAMD_CHECK_ARG(desc != nullptr && spec && window && twiddle && out, "null buffer");
  AAMD_CHECK_ARG(desc->rows >= 0 && desc->length >= 0 && desc->n_frames >= 0, "negative sizes");
  AAMD_CHECK_ARG(desc->n_fft >= 2 && desc->hop >= 1 && desc->pad >= 0, "n_fft must be >= 2, hop >= 1, pad >= 0");
  AAMD_CHECK_ARG(desc->pad_mode >= 0 && desc->pad_mode <= 3, "bad pad_mode");
  if (!desc->onesided) return fail(AAMD_EUNSUPPORTED, "audio_amd: inverse STFT needs a onesided spectrum");
  if (desc->rows == 0 || desc->length == 0 || desc->n_frames == 0) return AAMD_OK;
  OlaGeom og{};
  StftGeom& g = og.g;
  g.rows = desc->rows; g.length = desc->length; g.row_stride = desc->length;
  g.n_fft = desc->n_fft; g.hop = desc->hop; g.pad = desc->pad; g.center = desc->center;
  g.pad_mode = desc->pad_mode; g.onesided = 1; g.n_frames = desc->n_frames;
  g.n_freq = desc->n_fft / 2 + 1;
  g.scale = 1.0f; g.power = 0.0f;
  g.n_stages = plan_radices(desc->n_fft, g.radix);
  if (g.n_stages < 0) return fail(AAMD_EUNSUPPORTED, "audio_amd: n_fft has too many prime factors");
  og.interior = adjoint ? 0.5f : 1.0f;
  og.scale = desc->scale * (adjoint ? 1.0f : 1.0f / (float)desc->n_fft);
  og.scale_d = (double)desc->scale * (adjoint ? 1.0 : 1.0 / (double)desc->n_fft);
  if (g.n_fft == 400 && (g.hop == 100 || g.hop == 160 || g.hop == 200) && g.center && g.pad == 0 &&
      !force_generic()) {
    // radix-20x20 register FFT run backwards (istft400.h)
    m400::Inv400Geom ig{g, og.interior};
    const int tiles_per_row = (g.n_frames + m400::kFramesPerWave - 1) / m400::kFramesPerWave;
    const int64_t n_tiles = g.rows * tiles_per_row;
    int64_t blocks = dev_props().cu_count;
    const int64_t need = (n_tiles + m400::kInvWaves - 1) / m400::kInvWaves;
    if (blocks > need) blocks = need;
    const auto* sp = reinterpret_cast<const cplx<float>*>(spec);
#define AAMD_I400(HH)                                                                                             \
    {                                                                                                             \
      const size_t lds4 = ((size_t)m400::kInvWaves * m400::Hop<HH>::lds_dwords + m400::kConstDwords) * sizeof(float); \
      auto k4 = m400::istft400_kernel<HH>;                                                                        \
      AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds4));                                                                   \
      hipLaunchKernelGGL(k4, dim3((unsigned)blocks), dim3(64 * m400::kInvWaves), lds4, (hipStream_t)stream, ig, sp, \
                         window, twiddle, inv_envelope, out, og.scale, tiles_per_row, n_tiles);                   \
    }
    if (g.hop == 100) AAMD_I400(5) else if (g.hop == 200) AAMD_I400(10) else AAMD_I400(8)
#undef AAMD_I400
    return launch_check();
  }
  if ((g.n_fft == 256 || g.n_fft == 512 || g.n_fft == 1024 || g.n_fft == 2048) && !force_generic()) {
    // register-resident wave FFT run as the inverse (stft_pow2.h)
    p2::InvGeom ig{g, og.interior};
    const int64_t ppr = (g.n_frames + 1) / 2, n_pairs = g.rows * ppr;
    const auto* sp = reinterpret_cast<const p2::C32*>(spec);
    const auto* twc = reinterpret_cast<const p2::C32*>(twiddle);
    int64_t blocks = (int64_t)dev_props().cu_count * (g.n_fft <= 512 ? 8 : g.n_fft == 1024 ? 4 : 2);
    const int64_t need = (n_pairs + p2::kWaves - 1) / p2::kWaves;
    if (blocks > need) blocks = need;
    // runs of consecutive pairs per wave (overlap-add in an LDS ring, plain stores); hop > n_fft leaves gaps the ring
    // logic does not model: pair-at-a-time atomics there
    const bool use_runs = g.hop <= g.n_fft && (policy() & AAMD_POLICY_ISTFT_ATOMIC) == 0;
    const int run_len = 16;
    const int64_t rpr = (ppr + run_len - 1) / run_len, n_runs = g.rows * rpr;
    if (use_runs) {
      const int64_t need_r = (n_runs + p2::kWaves - 1) / p2::kWaves;
      if (blocks > need_r) blocks = need_r;
    }
#define AAMD_IP2(EE)                                                                                              \
    {                                                                                                             \
      if (use_runs) {                                                                                             \
        const size_t lds2 = (size_t)p2::kWaves * (2 * p2::Cfg<EE>::lds_complex + 2 * p2::Cfg<EE>::N) * sizeof(float); \
        auto k2 = p2::istft_pow2_run_kernel<EE>;                                                                  \
        if (lds2 > 48 * 1024)                                                                                     \
          AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds2));                                                               \
        hipLaunchKernelGGL(k2, dim3((unsigned)blocks), dim3(64 * p2::kWaves), lds2, (hipStream_t)stream, ig, sp,  \
                           window, twc, inv_envelope, out, og.scale, ppr, rpr, n_runs, run_len);                  \
      } else {                                                                                                    \
        const size_t lds2 = (size_t)p2::kWaves * p2::Cfg<EE>::lds_complex * sizeof(p2::C32);                      \
        auto k2 = p2::istft_pow2_kernel<EE>;                                                                      \
        if (lds2 > 48 * 1024)                                                                                     \
          AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds2));                                                               \
        hipLaunchKernelGGL(k2, dim3((unsigned)blocks), dim3(64 * p2::kWaves), lds2, (hipStream_t)stream, ig, sp,  \
                           window, twc, inv_envelope, out, og.scale, ppr, n_pairs);                               \
      }                                                                                                           \
    }
    if (g.n_fft == 256) AAMD_IP2(4) else if (g.n_fft == 512) AAMD_IP2(8) else if (g.n_fft == 1024) AAMD_IP2(16) else AAMD_IP2(32)
#undef AAMD_IP2
    return launch_check();
  }
  int pb = gen_pairs_per_block(g.n_fft);
  const int pairs_per_row = (g.n_frames + 1) / 2;
  if (pb > pairs_per_row) pb = pairs_per_row;
  const int bpr = (pairs_per_row + pb - 1) / pb;
  const int64_t blocks = g.rows * bpr;
  AAMD_CHECK_ARG(blocks < (1ll << 31), "too many frames for one launch");
  size_t lds = ((size_t)2 * g.n_fft + (size_t)4 * pb * gen_seq_len(g.n_fft)) * sizeof(float);
  auto kern = ola_kernel<float>;
  if (lds > dev_props().lds_per_block_optin) {     // long windows: twiddles from memory (stft_generic.h, gen_lds_floats_long)
    lds = gen_lds_floats_long(g.n_fft, pb) * sizeof(float);
    kern = ola_kernel<float, 1>;
    if (lds > dev_props().lds_per_block_optin) return fail(AAMD_EUNSUPPORTED, "audio_amd: n_fft too large for the LDS");
  }
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kGenThreads), lds, (hipStream_t)stream, og, spec, window,
                     reinterpret_cast<const cplx<float>*>(twiddle), inv_envelope, out, pb, bpr);
  return launch_check();
}

// ---- float64 entry points (csrc/f64_paths.h): autograd / gradcheck precision, not throughput ------------------------

int aamd_spectrogram_f64(const double* wav, const double* window, const double* twiddle, double* out,
                         const aamd_stft_desc* desc, void* stream) {
  DeviceScope dev_scope_(wav);
  StftGeom g;
  int rc = validate_desc(desc, g);
  if (rc != AAMD_OK) return rc;
  AAMD_CHECK_ARG(wav && window && twiddle && out, "null buffer");
  if (g.rows == 0) return AAMD_OK;
  int pb = gen_pairs_per_block(g.n_fft);
  const int pairs_per_row = (g.n_frames + 1) / 2;
  if (pb > pairs_per_row) pb = pairs_per_row;
  size_t lds = gen_lds_floats(g.n_fft, g.n_freq, pb) * sizeof(double);
  while (lds > dev_props().lds_per_block_optin && pb > 1) {
    pb /= 2;
    lds = gen_lds_floats(g.n_fft, g.n_freq, pb) * sizeof(double);
  }
  if (lds > dev_props().lds_per_block_optin) return fail(AAMD_EUNSUPPORTED, "audio_amd: n_fft too large for the LDS (float64)");
  const int bpr = (pairs_per_row + pb - 1) / pb;
  const int64_t blocks = g.rows * bpr;
  AAMD_CHECK_ARG(blocks < (1ll << 31), "too many frames for one launch");
  auto kern = stft_generic_kernel<double, EPI_SPEC>;
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  MelBandsDev mb{};
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kGenThreads), lds, (hipStream_t)stream, g, wav, window,
                     reinterpret_cast<const cplx<double>*>(twiddle), mb, out, pb, bpr);
  return launch_check();
}

int aamd_istft_f64(const double* spec, const double* window, const double* twiddle, const double* inv_envelope,
                   double* out, const aamd_stft_desc* desc, int32_t adjoint, void* stream) {
  DeviceScope dev_scope_(spec);
  AAMD_CHECK_ARG(desc != nullptr && spec && window && twiddle && out, "null buffer");
  AAMD_CHECK_ARG(desc->rows >= 0 && desc->length >= 0 && desc->n_frames >= 0, "negative sizes");
  AAMD_CHECK_ARG(desc->n_fft >= 2 && desc->hop >= 1 && desc->pad >= 0, "n_fft must be >= 2, hop >= 1, pad >= 0");
  AAMD_CHECK_ARG(desc->pad_mode >= 0 && desc->pad_mode <= 3, "bad pad_mode");
  if (!desc->onesided) return fail(AAMD_EUNSUPPORTED, "audio_amd: inverse STFT needs a onesided spectrum");
  if (desc->rows == 0 || desc->length == 0 || desc->n_frames == 0) return AAMD_OK;
  OlaGeom og{};
  StftGeom& g = og.g;
  g.rows = desc->rows; g.length = desc->length; g.row_stride = desc->length;
  g.n_fft = desc->n_fft; g.hop = desc->hop; g.pad = desc->pad; g.center = desc->center;
  g.pad_mode = desc->pad_mode; g.onesided = 1; g.n_frames = desc->n_frames;
  g.n_freq = desc->n_fft / 2 + 1;
  g.scale = 1.0f; g.power = 0.0f;
  g.n_stages = plan_radices(desc->n_fft, g.radix);
  if (g.n_stages < 0) return fail(AAMD_EUNSUPPORTED, "audio_amd: n_fft has too many prime factors");
  og.interior = adjoint ? 0.5f : 1.0f;
  og.scale = desc->scale * (adjoint ? 1.0f : 1.0f / (float)desc->n_fft);
  og.scale_d = (double)desc->scale * (adjoint ? 1.0 : 1.0 / (double)desc->n_fft);
  int pb = gen_pairs_per_block(g.n_fft);
  const int pairs_per_row = (g.n_frames + 1) / 2;
  if (pb > pairs_per_row) pb = pairs_per_row;
  size_t lds = ((size_t)2 * g.n_fft + (size_t)4 * pb * gen_seq_len(g.n_fft)) * sizeof(double);
  while (lds > dev_props().lds_per_block_optin && pb > 1) {
    pb /= 2;
    lds = ((size_t)2 * g.n_fft + (size_t)4 * pb * gen_seq_len(g.n_fft)) * sizeof(double);
  }
  if (lds > dev_props().lds_per_block_optin) return fail(AAMD_EUNSUPPORTED, "audio_amd: n_fft too large for the LDS (float64)");
  const int bpr = (pairs_per_row + pb - 1) / pb;
  const int64_t blocks = g.rows * bpr;
  AAMD_CHECK_ARG(blocks < (1ll << 31), "too many frames for one launch");
  auto kern = ola_kernel<double>;
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kGenThreads), lds, (hipStream_t)stream, og, spec, window,
                     reinterpret_cast<const cplx<double>*>(twiddle), inv_envelope, out, pb, bpr);
  return launch_check();
}

int aamd_lfilter_f64(const double* x, const double* a, const double* b, double* y, int64_t batch, int32_t channels,
                     int64_t length, int32_t n_order, int32_t n_coeff_rows, int32_t n_stages, int32_t clamp, void* stream) {
  DeviceScope dev_scope_(x);
  AAMD_CHECK_ARG(x && a && b && y, "null buffer");
  AAMD_CHECK_ARG(batch >= 0 && channels >= 1 && length >= 0 && n_order >= 1, "bad sizes");
  AAMD_CHECK_ARG(n_coeff_rows == 1 || n_coeff_rows == channels, "coefficient rows must be 1 or channels");
  if (n_stages != 1) return fail(AAMD_EUNSUPPORTED, "audio_amd: float64 lfilter runs one stage per call");
  const int64_t n_seq = batch * channels;
  if (n_seq == 0 || length == 0) return AAMD_OK;
  AAMD_CHECK_ARG((n_seq + 63) / 64 < (1ll << 31), "too many sequences for one launch");
  hipLaunchKernelGGL(f64::lfilter_kernel, dim3((unsigned)((n_seq + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x, a, b, y,
                     n_seq, channels, length, n_order, n_coeff_rows, clamp);
  return launch_check();
}

int aamd_resample_f64(const double* wav, const double* kernel, double* out, int64_t rows, int64_t length,
                      int64_t row_stride, int32_t orig, int32_t new_, int32_t width, int64_t out_len, void* stream) {
  DeviceScope dev_scope_(wav);
  AAMD_CHECK_ARG(wav && kernel && out, "null buffer");
  AAMD_CHECK_ARG(rows >= 0 && length >= 0 && orig >= 1 && new_ >= 1 && width >= 0 && out_len >= 0 && row_stride >= length,
                 "bad sizes");
  const int64_t n = rows * out_len;
  if (n == 0) return AAMD_OK;
  AAMD_CHECK_ARG((n + 255) / 256 < (1ll << 31), "too many samples for one launch");
  hipLaunchKernelGGL(f64::resample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wav, kernel,
                     out, rows, length, row_stride, orig, new_, width, out_len);
  return launch_check();
}

int aamd_fftconvolve_f64(const double* x, const double* y, double* out, int64_t rows, int64_t n_x_rows, int64_t n_y_rows,
                         int64_t nx, int64_t ny, const int64_t* x_row_of, const int64_t* y_row_of, int64_t start,
                         int64_t out_len, void* stream) {
  DeviceScope dev_scope_(x);
  AAMD_CHECK_ARG(x && y && out, "null buffer");
  AAMD_CHECK_ARG(rows >= 0 && nx >= 1 && ny >= 1 && start >= 0 && out_len >= 0 && start + out_len <= nx + ny - 1, "bad sizes");
  AAMD_CHECK_ARG((x_row_of != nullptr || n_x_rows == rows) && (y_row_of != nullptr || n_y_rows == rows), "row maps missing");
  const int64_t n = rows * out_len;
  if (n == 0) return AAMD_OK;
  AAMD_CHECK_ARG((n + 255) / 256 < (1ll << 31), "too many samples for one launch");
  hipLaunchKernelGGL(f64::conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, out, rows,
                     nx, ny, x_row_of, y_row_of, start, out_len);
  return launch_check();
}

int aamd_phase_vocoder_f32(const float* spec, const float* phase_advance, float* out, const aamd_vocoder_desc* d,
                           void* stream) {
  DeviceScope dev_scope_(spec);
  AAMD_CHECK_ARG(d != nullptr && spec && phase_advance && out, "null buffer");
  AAMD_CHECK_ARG(d->rows >= 0 && d->n_freq >= 1 && d->n_frames_in >= 0 && d->n_frames_out >= 0, "bad sizes");
  AAMD_CHECK_ARG(d->rate > 0.0, "rate must be positive");
  if (d->rows == 0 || d->n_frames_out == 0) return AAMD_OK;
  VocoderGeom g{d->rows, d->n_freq, d->n_frames_in, d->n_frames_out, d->in_stride_row, d->in_stride_freq,
                d->in_stride_frame, d->out_stride_row, d->out_stride_freq, d->out_stride_frame, d->rate};
  const int64_t chains = d->rows * d->n_freq;
  AAMD_CHECK_ARG((chains + 255) / 256 < (1ll << 31), "too many chains for one launch");
  hipLaunchKernelGGL(phase_vocoder_kernel, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g,
                     reinterpret_cast<const cplx<float>*>(spec), phase_advance, reinterpret_cast<cplx<float>*>(out));
  return launch_check();
}

int aamd_griffinlim_update_f32(const float* rebuilt, float* tprev, const float* magnitude, float* next, int64_t n,
                               float momentum, void* stream) {
  DeviceScope dev_scope_(rebuilt);
  AAMD_CHECK_ARG(rebuilt && tprev && magnitude && next, "null buffer");
  AAMD_CHECK_ARG(n >= 0, "bad size");
  if (n == 0) return AAMD_OK;
  const int blocks = grid_for(n, 256, dev_props().cu_count * 16);
  hipLaunchKernelGGL(griffinlim_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const cplx<float>*>(rebuilt), reinterpret_cast<cplx<float>*>(tprev), magnitude,
                     reinterpret_cast<cplx<float>*>(next), n, momentum);
  return launch_check();
}

int aamd_mel_scale_f32(const float* spec, const aamd_mel_bands* bands, float* out, int64_t rows,
                       int32_t n_frames, int32_t n_freq, void* stream) {
  DeviceScope dev_scope_(spec);
  AAMD_CHECK_ARG(spec && out, "null buffer");
  AAMD_CHECK_ARG(rows >= 0 && n_frames >= 0 && n_freq >= 1, "bad sizes");
  MelBandsDev mb;
  int rc = validate_bands(bands, n_freq, mb);
  if (rc != AAMD_OK) return rc;
  const int64_t n_vec = rows * n_frames;
  if (n_vec == 0) return AAMD_OK;
  const size_t lds = ms_lds_floats(mb.n_mels, mb.max_width, n_freq) * sizeof(float);
  if (lds <= 96 * 1024) {                                  // band table + 16 spectrum rows in LDS, persistent workgroups
    if (lds > 48 * 1024)
      AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mel_scale_lds_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = grid_for(n_vec, kMsVec, dev_props().cu_count * 8);
    hipLaunchKernelGGL(mel_scale_lds_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, spec, mb, out, n_vec, n_freq);
    return launch_check();
  }
  const int blocks = grid_for(n_vec * mb.n_mels, 256, dev_props().cu_count * 16);
  hipLaunchKernelGGL(mel_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, spec, mb, out,
                     n_vec, n_freq);
  return launch_check();
}

// launch geometry of db_group_kernel: one workgroup per kDbChunk elements of one group
static int db_grid(int64_t n, int64_t group_size, int64_t* chunks_per_group, int64_t* blocks) {
  const int64_t n_groups = (n + group_size - 1) / group_size;
  *chunks_per_group = (group_size + kDbChunk - 1) / kDbChunk;
  *blocks = n_groups * *chunks_per_group;
  return *blocks < (1ll << 31) ? AAMD_OK : AAMD_EINVAL;
}

int aamd_amplitude_to_db_f32(const float* x, float* out, int64_t n, float multiplier, float amin,
                             float db_multiplier, float* group_max, int64_t group_size, void* stream) {
  DeviceScope dev_scope_(x);
  AAMD_CHECK_ARG(x && (out || group_max), "null buffer");
  AAMD_CHECK_ARG(n >= 0, "negative size");
  AAMD_CHECK_ARG(group_max == nullptr || group_size >= 1, "group_size must be >= 1");
  if (n == 0) return AAMD_OK;
  const int64_t gs = group_max ? group_size : n;
  int64_t cpg, blocks;
  AAMD_CHECK_ARG(db_grid(n, gs, &cpg, &blocks) == AAMD_OK, "too many chunks for one launch");
  hipStream_t s = (hipStream_t)stream;
  if (group_max == nullptr)
    hipLaunchKernelGGL((db_group_kernel<true, false, false>), dim3((unsigned)blocks), dim3(256), 0, s, x, out, n, multiplier,
                       amin, db_multiplier, group_max, gs, cpg, 0.0f);
  else if (out != nullptr)
    hipLaunchKernelGGL((db_group_kernel<true, true, false>), dim3((unsigned)blocks), dim3(256), 0, s, x, out, n, multiplier,
                       amin, db_multiplier, group_max, gs, cpg, 0.0f);
  else                                                     // maximum only: first pass of a top_db conversion
    hipLaunchKernelGGL((db_group_kernel<false, true, false>), dim3((unsigned)blocks), dim3(256), 0, s, x, out, n, multiplier,
                       amin, db_multiplier, group_max, gs, cpg, 0.0f);
  return launch_check();
}

int aamd_amplitude_to_db_clamped_f32(const float* x, float* out, int64_t n, float multiplier, float amin,
                                     float db_multiplier, const float* group_max, int64_t group_size, float top_db,
                                     void* stream) {
  DeviceScope dev_scope_(x);
  AAMD_CHECK_ARG(x && out && group_max, "null buffer");
  AAMD_CHECK_ARG(n >= 0 && group_size >= 1, "bad sizes");
  if (n == 0) return AAMD_OK;
  int64_t cpg, blocks;
  AAMD_CHECK_ARG(db_grid(n, group_size, &cpg, &blocks) == AAMD_OK, "too many chunks for one launch");
  hipLaunchKernelGGL((db_group_kernel<true, false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, n,
                     multiplier, amin, db_multiplier, const_cast<float*>(group_max), group_size, cpg, top_db);
  return launch_check();
}

int aamd_db_clamp_f32(const float* x, float* out, int64_t n, const float* group_max,
                      int64_t group_size, float top_db, void* stream) {
  DeviceScope dev_scope_(x);
  AAMD_CHECK_ARG(x && out && group_max, "null buffer");
  AAMD_CHECK_ARG(n >= 0 && group_size >= 1, "bad sizes");
  if (n == 0) return AAMD_OK;
  const int blocks = grid_for(n, 256, dev_props().cu_count * 16);
  hipLaunchKernelGGL(db_clamp_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, n,
                     group_max, group_size, top_db);
  return launch_check();
}

int aamd_mfcc_dct_f32(const float* mel, const float* dct, float* out, int64_t n_vec, int32_t n_mels,
                      int32_t n_mfcc, int32_t log_mode, const float* group_max,
                      int64_t vec_per_group, float top_db, void* stream) {
  DeviceScope dev_scope_(mel);
  AAMD_CHECK_ARG(mel && dct && out, "null buffer");
  AAMD_CHECK_ARG(n_vec >= 0 && n_mels >= 1 && n_mfcc >= 1, "bad sizes");
  AAMD_CHECK_ARG(log_mode >= 0 && log_mode <= 2, "bad log_mode");
  AAMD_CHECK_ARG(vec_per_group >= 1 || group_max == nullptr, "vec_per_group must be >= 1");
  if (n_vec == 0) return AAMD_OK;
  const int nt = (n_mfcc + 15) / 16;
  if (n_mels % 4 == 0 && n_mels <= 16 * kDctMaxChunks && nt <= 4 && reinterpret_cast<uintptr_t>(mel) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(out) % 16 == 0 && !force_generic()) {
    const size_t flds = (size_t)dct_frag_floats(n_mels, n_mfcc) * sizeof(float);
    if (flds <= 64 * 1024) {
      const int64_t tiles = (n_vec + kDctFramesPerTile - 1) / kDctFramesPerTile;
      const int blocks = grid_for(tiles, 4, dev_props().cu_count * 8);
      const int64_t vpg = vec_per_group < 1 ? 1 : vec_per_group;
#define AAMD_DCT(NT)                                                                                   \
  do {                                                                                                 \
    if (flds > 48 * 1024)                                                                              \
      AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfcc_dct_mfma_kernel<NT>),            \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));            \
    hipLaunchKernelGGL(mfcc_dct_mfma_kernel<NT>, dim3(blocks), dim3(256), flds, (hipStream_t)stream,   \
                       mel, dct, out, n_vec, n_mels, n_mfcc, log_mode, group_max, vpg, top_db);        \
    return launch_check();                                                                             \
  } while (0)
      switch (nt) {
        case 1: AAMD_DCT(1);
        case 2: AAMD_DCT(2);
        case 3: AAMD_DCT(3);
        default: AAMD_DCT(4);
      }
#undef AAMD_DCT
    }
  }
  const size_t lds = ((size_t)n_mels * n_mfcc + (size_t)kMfccVecPerBlock * n_mels) * sizeof(float);
  if (lds > 160 * 1024) return fail(AAMD_EUNSUPPORTED, "audio_amd: dct matrix too large for LDS");
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfcc_dct_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int blocks = grid_for(n_vec, kMfccVecPerBlock, dev_props().cu_count * 8);
  hipLaunchKernelGGL(mfcc_dct_kernel, dim3(blocks), dim3(256), lds, (hipStream_t)stream, mel, dct, out,
                     n_vec, n_mels, n_mfcc, log_mode, group_max, vec_per_group < 1 ? 1 : vec_per_group,
                     top_db);
  return launch_check();
}

int aamd_resample_f32(const float* wav, const float* kernel, float* out, int64_t rows, int64_t length,
                      int64_t row_stride, int32_t orig, int32_t new_, int32_t width, int64_t out_len,
                      void* stream) {
  DeviceScope dev_scope_(wav);
  AAMD_CHECK_ARG(wav && kernel && out, "null buffer");
  AAMD_CHECK_ARG(rows >= 0 && length >= 0 && orig >= 1 && new_ >= 1 && width >= 0, "bad sizes");
  AAMD_CHECK_ARG(row_stride >= length, "row_stride < length");
  const int64_t expect = (new_ * length + orig - 1) / orig;
  AAMD_CHECK_ARG(out_len == expect, "out_len must be ceil(new*length/orig)");
  if (rows == 0 || out_len == 0) return AAMD_OK;
  ResampleGeom g;
  g.rows = rows; g.length = length; g.row_stride = row_stride; g.out_len = out_len;
  g.orig = orig; g.new_ = new_; g.width = width; g.taps = 2 * width + orig;
  const int64_t nq = (out_len + new_ - 1) / new_;
  // aim for ~2048 outputs per workgroup, halo within 96 KiB of LDS
  int qt = (2048 + new_ - 1) / new_;
  if (qt < 1) qt = 1;
  if (qt > nq) qt = (int)nq;
  const int64_t lds_budget = 96 * 1024 / sizeof(float);
  while (qt > 1 && (int64_t)(qt - 1) * orig + g.taps > lds_budget) --qt;
  g.qt = qt;
  g.use_lds = ((int64_t)(qt - 1) * orig + g.taps <= lds_budget) ? 1 : 0;
  g.nq_tiles = (int)((nq + qt - 1) / qt);
  const int64_t blocks = rows * g.nq_tiles;
  AAMD_CHECK_ARG(blocks < (1ll << 31), "too many tiles for one launch");
  const size_t lds = g.use_lds ? ((size_t)(qt - 1) * orig + g.taps) * sizeof(float) : 0;
  if (lds > 48 * 1024)
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(resample_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, g, wav,
                     kernel, out);
  return launch_check();
}

// tools only (tools/rsm_census.py): the time stamps the f16 resampler records under AAMD_RSM_LAB=64
extern "C" __attribute__((visibility("default"))) int aamd_debug_rsm_census(long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(rsm::g_rsm_census), sizeof(long long) * (size_t)n) == hipSuccess ? 0 : -1;
}

int aamd_resample_banded_f32(const float* wav, const float* kernel, float* out, int64_t rows,
                             int64_t length, int64_t row_stride, int32_t orig, int32_t new_, int32_t width,
                             int64_t out_len, const aamd_resample_bands* bands, void* stream) {
  return aamd_resample_prepared_f32(wav, kernel, out, rows, length, row_stride, orig, new_, width, out_len, bands, nullptr, stream);
}

int64_t aamd_resample_frag_bytes(int32_t orig, int32_t new_, const aamd_resample_bands* bands) {
  if (bands == nullptr || new_ < 1 || bands->n_tiles != (new_ + 15) / 16) return 0;
  const int ks = rsm::pick_ks(bands->tap_span, orig);
  return ks == 0 ? 0 : rsm::frag_bytes(bands->n_tiles, ks);
}

int aamd_resample_frag_build_f32(const float* kernel, int32_t orig, int32_t new_, int32_t width,
                                 const aamd_resample_bands* bands, void* frag, void* stream) {
  DeviceScope dev_scope_(kernel);
  AAMD_CHECK_ARG(kernel && frag && bands, "null buffer");
  AAMD_CHECK_ARG(orig >= 1 && new_ >= 1 && width >= 0, "bad sizes");
  const int n_tiles = (new_ + 15) / 16;
  AAMD_CHECK_ARG(bands->n_tiles == n_tiles && bands->tap_lo != nullptr && bands->tap_span >= 1, "band table must have ceil(new/16) tiles");
  const int ks = rsm::pick_ks(bands->tap_span, orig);
  if (ks == 0) return fail(AAMD_EUNSUPPORTED, "audio_amd: band wider than 448 taps: no matrix-core kernel, no prepared fragments");
  AAMD_CHECK_ARG(reinterpret_cast<uintptr_t>(frag) % 16 == 0, "fragment table must be 16-byte aligned");
  const int taps = 2 * width + orig;
  for (int t = 0; t < n_tiles; ++t)
    AAMD_CHECK_ARG(bands->tap_lo[t] >= 0 && bands->tap_lo[t] < taps, "tap_lo outside the tap table");
  rsm::Geom g{};
  g.orig = orig; g.new_ = new_; g.width = width; g.taps = taps;
  for (int pt0 = 0; pt0 < n_tiles; pt0 += rsm::kMaxPhaseTiles) {          // (the band starts ride in the kernel arguments, 14 tiles a launch)
    g.pt0 = pt0;
    g.n_pt = n_tiles - pt0 < rsm::kMaxPhaseTiles ? n_tiles - pt0 : rsm::kMaxPhaseTiles;
    for (int t = 0; t < g.n_pt; ++t) g.tap_lo[t] = bands->tap_lo[pt0 + t];
    const int n = g.n_pt * (ks / 8) * 64;
    hipLaunchKernelGGL(rsm::frag_build_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, ks, kernel,
                       static_cast<uint32_t*>(frag));
  }
  return launch_check();
}

int aamd_resample_prepared_f32(const float* wav, const float* kernel, float* out, int64_t rows,
                               int64_t length, int64_t row_stride, int32_t orig, int32_t new_, int32_t width,
                               int64_t out_len, const aamd_resample_bands* bands, const void* frag, void* stream) {
  DeviceScope dev_scope_(wav);
  const int n_tiles = (new_ + 15) / 16;
  const int ks = bands ? rsm::pick_ks(bands->tap_span, orig) : 0;
  if (bands == nullptr || ks == 0 || force_generic())
    return aamd_resample_f32(wav, kernel, out, rows, length, row_stride, orig, new_, width, out_len, stream);
  AAMD_CHECK_ARG(wav && kernel && out, "null buffer");
  AAMD_CHECK_ARG(rows >= 0 && length >= 0 && orig >= 1 && new_ >= 1 && width >= 0, "bad sizes");
  AAMD_CHECK_ARG(row_stride >= length, "row_stride < length");
  AAMD_CHECK_ARG(out_len == (new_ * length + orig - 1) / orig, "out_len must be ceil(new*length/orig)");
  AAMD_CHECK_ARG(bands->n_tiles == n_tiles && bands->tap_lo != nullptr && bands->tap_span >= 1,
                 "band table must have ceil(new/16) tiles");
  if (rows == 0 || out_len == 0) return AAMD_OK;
  const int taps = 2 * width + orig;
  for (int t = 0; t < n_tiles; ++t)
    AAMD_CHECK_ARG(bands->tap_lo[t] >= 0 && bands->tap_lo[t] < taps, "tap_lo outside the tap table");
  rsm::Geom g{};
#ifdef AAMD_LAB
  static const int rsm_lab = [] { const char* e = std::getenv("AAMD_RSM_LAB"); return e ? std::atoi(e) : 0; }();   // tools only
  g.lab = rsm_lab;
#endif
  g.frag = static_cast<const uint32_t*>(frag);       // (read by the f16 kernels only)
  AAMD_CHECK_ARG(reinterpret_cast<uintptr_t>(frag) % 16 == 0, "fragment table must be 16-byte aligned");
  g.rows = rows; g.length = length; g.row_stride = row_stride; g.out_len = out_len;
  g.orig = orig; g.new_ = new_; g.width = width; g.taps = taps;
  g.vec_in = (reinterpret_cast<uintptr_t>(wav) % 16 == 0) && (row_stride % 4 == 0);
  g.vec_out = (reinterpret_cast<uintptr_t>(out) % 16 == 0) && (out_len % 4 == 0) && (new_ % 4 == 0);
  const int64_t nq = (out_len + new_ - 1) / new_;
  const int max_cw = rsm::max_compute_waves(ks);
  const size_t lds_cap = dev_props().lds_per_block_optin ? dev_props().lds_per_block_optin : 64 * 1024;
  for (int pt0 = 0; pt0 < n_tiles; pt0 += max_cw) {
    g.pt0 = pt0;
    g.n_pt = n_tiles - pt0 < max_cw ? n_tiles - pt0 : max_cw;
    int max_lo = 0;
    for (int t = 0; t < g.n_pt; ++t) {
      g.tap_lo[t] = bands->tap_lo[pt0 + t];
      if (g.tap_lo[t] > max_lo) max_lo = g.tap_lo[t];
    }
    const bool f16 = (policy() & AAMD_POLICY_RESAMPLE_FP32) == 0;
    const bool rd64 = f16 && rsm::b64_ok(ks, orig) && (policy() & AAMD_POLICY_RESAMPLE_B32) == 0;
    if (!rsm::plan_chunk(g, ks, f16, nq, max_lo, lds_cap))   // a single q-group does not fit (huge orig): scalar kernel
      return aamd_resample_f32(wav, kernel, out, rows, length, row_stride, orig, new_, width, out_len, stream);
    const int qg = g.qg;
    const int qc = rsm::chunk_q(g);
    const size_t lds = 2 * (size_t)g.buf_floats * sizeof(float) + (f16 ? 48 : 0);       // + the chunk-maximum slots and arrival counters
    g.chunks_per_row = (int)((nq + qc - 1) / qc);
    g.n_chunks = rows * g.chunks_per_row;
    AAMD_CHECK_ARG(g.n_chunks < (1ll << 31), "too many chunks for one launch");
    // persistent workgroups: as many per CU as the 16 wave slots (128 registers) and the LDS hold -- a one-tile rate pair
    // has only a handful of compute waves per workgroup
    const int wg_waves = g.n_pt * qg + g.n_loaders;
    int per_cu = ks >= 80 ? 1 : 16 / wg_waves;
    if (per_cu > (int)(lds_cap / lds)) per_cu = (int)(lds_cap / lds);
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)dev_props().cu_count * per_cu;
    if (blocks > g.n_chunks) blocks = g.n_chunks;
    g.chunks_per_block = (int)((g.n_chunks + blocks - 1) / blocks);
    blocks = (g.n_chunks + g.chunks_per_block - 1) / g.chunks_per_block;
    const int threads = 64 * wg_waves;
#ifdef AAMD_LAB
#define AAMD_RSM_F16(KS) (g.lab == 0 ? rsm::resample_f16_kernel<KS, 0> : g.lab == 64 ? rsm::resample_f16_kernel<KS, 1> : rsm::resample_f16_kernel<KS, 2>)
#define AAMD_RSM_RD64(KS) (g.lab == 64 ? (full ? rsm::kernel_rd64<KS, 1, 1>() : rsm::kernel_rd64<KS, 1>()) : (full ? rsm::kernel_rd64<KS, 0, 1>() : rsm::kernel_rd64<KS, 0>()))
#else
#define AAMD_RSM_F16(KS) (rsm::resample_f16_kernel<KS, 0>)
#define AAMD_RSM_RD64(KS) (full ? rsm::kernel_rd64<KS, 0, 1>() : rsm::kernel_rd64<KS, 0>())
#endif
#define AAMD_RSM(KS)                                                                                  \
  do {                                                                                                \
    auto kern = !f16 ? rsm::resample_mfma_kernel<KS> : AAMD_RSM_F16(KS);                              \
    /* 8-byte operand reads: odd orig, KS = 80 / 104 / 112 (resample_mfma.h, b64_rot) */              \
    const bool full = rd64 && rsm::chunk_is_full(g, KS);   /* padded chunk: the branch-free loader instantiation */ \
    if (f16 && rd64) kern = AAMD_RSM_RD64(KS);                                                        \
    if (lds > 48 * 1024)                                                                              \
      AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));            \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), lds, (hipStream_t)stream, g, wav, \
                       kernel, out);                                                                  \
  } while (0)
    switch (ks) {
      case 16: AAMD_RSM(16); break;
      case 48: AAMD_RSM(48); break;
      case 80: AAMD_RSM(80); break;
      case 104: AAMD_RSM(104); break;      // (odd orig only: pick_ks)
      default: AAMD_RSM(112); break;
    }
#undef AAMD_RSM
#undef AAMD_RSM_F16
#undef AAMD_RSM_RD64
    int rc = launch_check();
    if (rc != AAMD_OK) return rc;
  }
  return AAMD_OK;
}

int aamd_resample_sparse_f32(const float* wav, const float* taps_compact, const int32_t* tap_lo, float* out, int64_t rows,
                             int64_t length, int64_t row_stride, int32_t orig, int32_t new_, int32_t width, int32_t span,
                             int64_t out_len, void* stream) {
  DeviceScope dev_scope_(wav);
  AAMD_CHECK_ARG(wav && taps_compact && tap_lo && out, "null buffer");
  AAMD_CHECK_ARG(rows >= 0 && length >= 0 && orig >= 1 && new_ >= 1 && width >= 0 && span >= 1, "bad sizes");
  AAMD_CHECK_ARG(row_stride >= length, "row_stride < length");
  AAMD_CHECK_ARG(out_len == ((int64_t)new_ * length + orig - 1) / orig, "out_len must be ceil(new*length/orig)");
  const int64_t n = rows * out_len;
  if (n == 0) return AAMD_OK;
  AAMD_CHECK_ARG((n + 255) / 256 < (1ll << 31), "too many samples for one launch");
  hipLaunchKernelGGL(resample_sparse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wav,
                     taps_compact, tap_lo, out, rows, length, row_stride, orig, new_, width, span, out_len);
  return launch_check();
}

int aamd_lfilter_f32(const float* x, const float* a, const float* b, float* y, int64_t batch,
                     int32_t channels, int64_t length, int32_t n_order, int32_t n_coeff_rows,
                     int32_t n_stages, int32_t clamp, void* stream) {
  DeviceScope dev_scope_(x);
  AAMD_CHECK_ARG(x && a && b && y, "null buffer");
  AAMD_CHECK_ARG(batch >= 0 && channels >= 1 && length >= 0, "bad sizes");
  AAMD_CHECK_ARG(n_order >= 1 && n_stages >= 1, "n_order and n_stages must be >= 1");
  AAMD_CHECK_ARG(clamp >= 0 && clamp <= 2, "clamp must be 0, 1 (after every stage) or 2 (after the last stage only)");
  AAMD_CHECK_ARG(n_coeff_rows == 1 || n_coeff_rows == channels, "n_coeff_rows must be 1 or channels");
  const int64_t n_seq = batch * channels;
  if (n_seq == 0 || length == 0) return AAMD_OK;
  hipStream_t s = (hipStream_t)stream;
  const int d = n_order - 1;
  // biquad-class filters: W waves per sequence with shuffle scans (lfilter_wave.h); W fills the chip
  if (n_order <= 3 && n_stages <= lfw::kMaxCascade && length < (1ll << 30) && !force_generic()) {
    int W = 1;
    while (W < lfw::kMaxWaves && n_seq * (2 * W) <= 8192 && (int64_t)W * lfw::kWaveBlock < length) W *= 2;
    const size_t cap = dev_props().lds_per_block_optin ? dev_props().lds_per_block_optin : 64 * 1024;
#ifdef AAMD_LAB
    static const int lab_w = [] { const char* e = std::getenv("AAMD_LFW_W"); return e ? std::atoi(e) : 0; }();    // tools only
    if (lab_w >= 1 && lab_w <= lfw::kMaxWaves && (lab_w & (lab_w - 1)) == 0) W = lab_w;
#endif
    while (W > 1 && lfw::lds_bytes(W, n_stages) > cap) W /= 2;
    const size_t lds = lfw::lds_bytes(W, n_stages);
    if (lds <= cap) {
      const int blocks = grid_for(n_seq, 1, dev_props().cu_count * 8);
#ifdef AAMD_LAB
      static const int lab = [] { const char* e = std::getenv("AAMD_LFW_LAB"); return e ? std::atoi(e) : 0; }();   // tools only
#else
      constexpr int lab = 0;
#endif
      const int vec_ok = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0) &&
                         (length % 4 == 0) && lab != 64;      // lab 64: the dword copies also for aligned rows
#define AAMD_LFW(LL, MW, VV)                                                                                       \
      {                                                                                                            \
        AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lfw::lfilter_wave_kernel<LL, MW, VV>),          \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                       \
        hipLaunchKernelGGL((lfw::lfilter_wave_kernel<LL, MW, VV>), dim3(blocks), dim3(64 * W), lds, s, x, a, b, y, \
                           n_seq, channels, length, n_order, n_coeff_rows, n_stages, clamp);                       \
      }
#ifndef AAMD_LAB
#define AAMD_LFW_LABS(MW)                                                                                          \
      { if (vec_ok) AAMD_LFW(0, MW, true) else AAMD_LFW(0, MW, false) }
#else
#define AAMD_LFW_LABS(MW)                                                                                          \
      switch (vec_ok ? lab : 0) {                                                                                  \
        case 1: AAMD_LFW(1, MW, true) break;                                                                       \
        case 2: AAMD_LFW(2, MW, true) break;                                                                       \
        case 4: AAMD_LFW(4, MW, true) break;                                                                       \
        case 8: AAMD_LFW(8, MW, true) break;                                                                       \
        case 15: AAMD_LFW(15, MW, true) break;                                                                     \
        case 16: AAMD_LFW(16, MW, true) break;                                                                     \
        case 32: AAMD_LFW(32, MW, true) break;                                                                     \
        case 63: AAMD_LFW(63, MW, true) break;                                                                     \
        case 128: AAMD_LFW(128, MW, true) break;                                                                   \
        case 256: AAMD_LFW(256, MW, true) break;                                                                   \
        case 512: AAMD_LFW(512, MW, true) break;                                                                   \
        case 896: AAMD_LFW(896, MW, true) break;                                                                   \
        default:                                                                                                   \
          if (vec_ok) AAMD_LFW(0, MW, true) else AAMD_LFW(0, MW, false)                                            \
      }
#endif
#ifdef AAMD_LAB
      static const int lab_pipe = [] { const char* e = std::getenv("AAMD_LFW_PIPE"); return e ? std::atoi(e) : -1; }();   // tools only
#else
      constexpr int lab_pipe = -1;
#endif
      const bool pipe = vec_ok && W >= 8 && lab_pipe != 0 && lfw::pipe_lds_bytes(8, n_stages) <= cap;
      if (pipe && n_stages >= 3 && lab_pipe != 1) {   // + 4 mover waves that own the copies and the stores
        const size_t plds = lfw::pipe_lds_bytes(8, n_stages);
#define AAMD_LFWM(LL)                                                                                              \
        {                                                                                                          \
          AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lfw::lfilter_wave_mover_kernel<LL>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));                    \
          hipLaunchKernelGGL((lfw::lfilter_wave_mover_kernel<LL>), dim3(blocks), dim3(64 * (8 + lfw::kMovers)),    \
                             plds, s, x, a, b, y, n_seq, channels, length, n_order, n_coeff_rows, n_stages, clamp);\
        }
#ifdef AAMD_LAB
        switch (lab) {
          case 1: AAMD_LFWM(1) break;
          case 16: AAMD_LFWM(16) break;
          case 32: AAMD_LFWM(32) break;
          case 48: AAMD_LFWM(48) break;
          default: AAMD_LFWM(0)
        }
#else
        AAMD_LFWM(0)
#endif
#undef AAMD_LFWM
      } else if (pipe) {        // two tiles per wave, copies and stores spread over the stages (8 waves)
        W = 8;
        const size_t plds = lfw::pipe_lds_bytes(8, n_stages);
#define AAMD_LFWP(LL)                                                                                              \
        {                                                                                                          \
          AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lfw::lfilter_wave_pipe_kernel<LL>),           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));                    \
          hipLaunchKernelGGL((lfw::lfilter_wave_pipe_kernel<LL>), dim3(blocks), dim3(512), plds, s, x, a, b, y,    \
                             n_seq, channels, length, n_order, n_coeff_rows, n_stages, clamp);                     \
        }
#ifdef AAMD_LAB
        switch (lab) {
          case 1: AAMD_LFWP(1) break;
          case 15: AAMD_LFWP(15) break;
          case 16: AAMD_LFWP(16) break;
          case 32: AAMD_LFWP(32) break;
          case 48: AAMD_LFWP(48) break;
          case 50: AAMD_LFWP(50) break;
          case 52: AAMD_LFWP(52) break;
          case 56: AAMD_LFWP(56) break;
          case 63: AAMD_LFWP(63) break;
          default: AAMD_LFWP(0)
        }
#else
        AAMD_LFWP(0)
#endif
#undef AAMD_LFWP
      } else if (W <= 8) {      // 512 threads: the 256-register instantiation
        AAMD_LFW_LABS(8)
      } else {
        AAMD_LFW_LABS(16)
      }
#undef AAMD_LFW_LABS
#undef AAMD_LFW
      return launch_check();
    }
  }
#define AAMD_LF(D) return launch_lfilter<D>(x, a, b, y, n_seq, channels, length, n_order, n_coeff_rows, n_stages, clamp, s)
  if (d <= 1) AAMD_LF(1);
  if (d <= 2) AAMD_LF(2);
  if (d <= 3) AAMD_LF(3);
  if (d <= 4) AAMD_LF(4);
  if (d <= 6) AAMD_LF(6);
  if (d <= 8) AAMD_LF(8);
  if (d <= 12) AAMD_LF(12);
  if (d <= 16) AAMD_LF(16);
#undef AAMD_LF
  return fail(AAMD_EUNSUPPORTED, "audio_amd: lfilter order > 16 not supported");
}

// taps above this use overlap-save on the LDS FFT; below, the tiled time-domain kernel is cheaper
static const int64_t kFftConvMinTaps = 192;

static bool fftconv_use_fft(int64_t n_taps) {
  return n_taps > kFftConvMinTaps && !force_generic();
}

// partitions of the delay-line plan, 0 when the tap count (or the policy) rules it out
static int64_t fftconv_fdl_parts(int64_t n_taps) {
  const int64_t np = (n_taps + fco::kHop - 1) / fco::kHop;
  return ((policy() & AAMD_POLICY_FFTCONV_NO_FDL) || np < 2 || np > fco::kMaxFdlParts) ? 0 : np;
}
// the plan of one call: the cost model of fco::plan_fdl, or (policy, tests) the delay line whenever it is possible at all
static bool fftconv_pick_fdl(int64_t rows, int64_t taps, int64_t out_len, fco::FdlGeom& f) {
  if (!fftconv_fdl_parts(taps)) return false;
  const bool cheaper = fco::plan_fdl(rows, taps, out_len, dev_props().cu_count, f);
  return cheaper || ((policy() & AAMD_POLICY_FFTCONV_FDL) && f.n_blocks >= 2);
}

// the real-block kernel (fftconv_fdr.h, plan 3): EVERY FFT-eligible tap count up to 32768 (round 5; 24576 in round 4) -- plain
// overlap-save on real blocks up to 8192 taps (one partition, no delay line), the register delay line of up to three delayed
// spectra for 8193 .. 32768 -- unless ANY of the three FFTCONV
// policy bits is set: NO_FDL / FDL / COMPLEX all select the complex-block kernels (plans 1 / 2) for all tap counts, also for
// <= 8192 taps where plan 3 is not a delay line at all (the bits exist for A/B runs against the round-1..3 kernels)
static bool fftconv_pick_fdr(int64_t rows, int64_t taps, int64_t out_len, fdr::Geom& g) {
  if (policy() & (AAMD_POLICY_FFTCONV_NO_FDL | AAMD_POLICY_FFTCONV_FDL | AAMD_POLICY_FFTCONV_COMPLEX)) return false;
  return fdr::plan(rows, taps, out_len, dev_props().cu_count, g);
}

int64_t aamd_fftconvolve_workspace(int64_t rows, int64_t n_x_rows, int64_t n_y_rows, int64_t nx, int64_t ny) {
  (void)rows;
  const bool swap = ny > nx;
  const int64_t taps = swap ? nx : ny;
  const int64_t tap_rows = swap ? n_x_rows : n_y_rows;
  if (!fftconv_use_fft(taps)) return 0;
  fco::Geom g{};
  fco::plan(taps, 1, g);
  // twiddles | tap spectra (of whichever plan runs) | delay-line rings, one per workgroup (the slice is not known here)
  const int64_t np_fdl = fftconv_fdl_parts(taps);
  const int64_t n_spec = tap_rows * (np_fdl > g.n_part ? np_fdl : g.n_part);
  return (int64_t)sizeof(fco::C32) * fco::kN * (1 + n_spec + (np_fdl > 2 ? np_fdl - 2 : 0) * dev_props().cu_count);
}

int aamd_fftconvolve_plan(int64_t rows, int64_t nx, int64_t ny, int64_t out_len) {
  const int64_t taps = ny > nx ? nx : ny;
  if (!fftconv_use_fft(taps)) return 0;
  fdr::Geom fg{};
  if (fftconv_pick_fdr(rows, taps, out_len, fg)) return 3;
  fco::FdlGeom f{};
  return fftconv_pick_fdl(rows, taps, out_len, f) ? 2 : 1;
}

int aamd_fftconvolve_f32(const float* x, const float* y, float* out, int64_t rows, int64_t n_x_rows,
                         int64_t n_y_rows, int64_t nx, int64_t ny, const int64_t* x_row_of,
                         const int64_t* y_row_of, int64_t start, int64_t out_len, void* workspace,
                         void* stream) {
  return aamd_fftconvolve_staged_f32(x, y, out, rows, n_x_rows, n_y_rows, nx, ny, x_row_of, y_row_of, start, out_len,
                                     workspace, AAMD_FFTCONV_PREPARE | AAMD_FFTCONV_RUN, stream);
}

// stages: PREPARE lays the twiddles and the tap spectra of the plan down in the workspace (two small launches), RUN walks the
// rows.  A caller that convolves many batches with ONE impulse response prepares once and runs with the same workspace
// afterwards (the plan must be the same: aamd_fftconvolve_plan, same policy, same tap rows).
int aamd_fftconvolve_staged_f32(const float* x, const float* y, float* out, int64_t rows, int64_t n_x_rows,
                                int64_t n_y_rows, int64_t nx, int64_t ny, const int64_t* x_row_of,
                                const int64_t* y_row_of, int64_t start, int64_t out_len, void* workspace,
                                int32_t stages, void* stream) {
  const bool prep = (stages & AAMD_FFTCONV_PREPARE) != 0, run = (stages & AAMD_FFTCONV_RUN) != 0;
  AAMD_CHECK_ARG((prep || run) && !(stages & ~(AAMD_FFTCONV_PREPARE | AAMD_FFTCONV_RUN)), "stages: PREPARE, RUN or both");
  DeviceScope dev_scope_(run ? (const void*)x : (const void*)workspace);
  AAMD_CHECK_ARG(!run || (x && y && out), "null buffer");
  AAMD_CHECK_ARG(!prep || y, "null tap buffer");
  AAMD_CHECK_ARG(rows >= 0 && nx >= 1 && ny >= 1 && n_x_rows >= 1 && n_y_rows >= 1, "bad sizes");
  AAMD_CHECK_ARG(start >= 0 && out_len >= 0 && start + out_len <= nx + ny - 1, "slice outside the full convolution");
  if (rows == 0 || out_len == 0) return AAMD_OK;
  // stream the SHORTER operand as taps (convolution commutes)
  const bool swap = ny > nx;
  const float* xa = swap ? y : x;
  const float* ya = swap ? x : y;
  const int64_t nxa = swap ? ny : nx, nya = swap ? nx : ny;
  const int64_t tap_rows = swap ? n_x_rows : n_y_rows;
  const int64_t* xmap = swap ? y_row_of : x_row_of;
  const int64_t* ymap = swap ? x_row_of : y_row_of;
  hipStream_t s = (hipStream_t)stream;
  AAMD_CHECK_ARG(!prep || ya, "null tap buffer");
  if (fftconv_use_fft(nya)) {
    AAMD_CHECK_ARG(workspace != nullptr, "fftconvolve needs the workspace of aamd_fftconvolve_workspace()");
    AAMD_CHECK_ARG(reinterpret_cast<uintptr_t>(workspace) % 8 == 0, "workspace must be 8-byte aligned");
    fco::Geom g{};
    g.rows = rows; g.nx = nxa; g.ny = nya; g.start = start; g.out_len = out_len;
    fco::plan(nya, out_len, g);
    fco::C32* tw = reinterpret_cast<fco::C32*>(workspace);
    fco::C32* H = tw + fco::kN;
    const size_t lds = (size_t)fco::kLdsComplex * sizeof(fco::C32);
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fco::spectrum_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fco::overlap_save_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (prep) hipLaunchKernelGGL(fco::twiddle_kernel, dim3(fco::kN / 256), dim3(256), 0, s, tw);
    fdr::Geom fg{};
    fg.rows = rows; fg.nx = nxa; fg.ny = nya; fg.start = start; fg.out_len = out_len;
    if (fftconv_pick_fdr(rows, nya, out_len, fg)) {
      // real blocks of 16384 samples as 8192-point complex FFTs, the delay line in registers (fftconv_fdr.h).  The tap spectra
      // (8192 complex per partition) fit the space the workspace reserves for the complex-block plans (16384 per partition).
      const size_t lds_r = (size_t)fdr::kLdsComplex * sizeof(fco::C32);
      AAMD_CHECK_ARG(tap_rows * fg.n_part < (1ll << 31) && rows * fg.segs < (1ll << 31), "too many tap rows / work items");
      if (prep) {
        AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fdr::spectrum_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
        hipLaunchKernelGGL(fdr::spectrum_kernel, dim3((unsigned)(tap_rows * fg.n_part)), dim3(fdr::kThreads), lds_r, s, nya,
                           fg.n_part, ya, tw, H);
      }
      if (!run) return launch_check();
      int64_t blocks = dev_props().cu_count;
      if (blocks > rows * fg.segs) blocks = rows * fg.segs;
#define AAMD_FDR(NP)                                                                                          \
      do {                                                                                                    \
        AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fdr::delay_line_kernel<NP>),               \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));                \
        hipLaunchKernelGGL(fdr::delay_line_kernel<NP>, dim3((unsigned)blocks), dim3(fdr::kThreads), lds_r, s, \
                           fg, xa, tw, H, xmap, ymap, out);                                                   \
      } while (0)
      if (fg.n_part == 1) AAMD_FDR(1); else if (fg.n_part == 2) AAMD_FDR(2); else if (fg.n_part == 3) AAMD_FDR(3); else AAMD_FDR(4);
#undef AAMD_FDR
      return launch_check();
    }
    fco::FdlGeom f{};
    f.rows = rows; f.nx = nxa; f.ny = nya; f.start = start; f.out_len = out_len;
    if (fftconv_pick_fdl(rows, nya, out_len, f)) {
      // frequency-domain delay line: one forward + one inverse FFT per block step
      fco::Geom gs = g;
      gs.n_part = f.n_part; gs.part_taps = fco::kHop;
      AAMD_CHECK_ARG(tap_rows * f.n_part < (1ll << 31), "too many tap rows");
      if (prep)
        hipLaunchKernelGGL(fco::spectrum_kernel, dim3((unsigned)(tap_rows * f.n_part)), dim3(fco::kThreads), lds, s, gs,
                           ya, tw, H);
      if (!run) return launch_check();
      const int64_t n_spec = tap_rows * (f.n_part > g.n_part ? f.n_part : g.n_part);
      fco::C32* ring = H + n_spec * fco::kN;
      int64_t blocks = dev_props().cu_count;
      if (blocks > rows * f.segs) blocks = rows * f.segs;
#define AAMD_FDL(NP)                                                                                          \
      do {                                                                                                    \
        AAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fco::overlap_save_fdl_kernel<NP>),         \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
        hipLaunchKernelGGL(fco::overlap_save_fdl_kernel<NP>, dim3((unsigned)blocks), dim3(fco::kPhys), lds, s, \
                           f, xa, tw, H, ring, xmap, ymap, out);                                              \
      } while (0)
      switch (f.n_part) {
        case 2: AAMD_FDL(2); break;
        case 3: AAMD_FDL(3); break;
        default: AAMD_FDL(4); break;
      }
#undef AAMD_FDL
      return launch_check();
    }
    AAMD_CHECK_ARG(tap_rows * g.n_part < (1ll << 31), "too many tap rows");
    if (prep)
      hipLaunchKernelGGL(fco::spectrum_kernel, dim3((unsigned)(tap_rows * g.n_part)), dim3(fco::kThreads), lds, s, g,
                         ya, tw, H);
    if (!run) return launch_check();
    const int64_t items = rows * g.n_pairs;
    int64_t blocks = dev_props().cu_count;
    if (blocks > items) blocks = items;
    hipLaunchKernelGGL(fco::overlap_save_kernel, dim3((unsigned)blocks), dim3(fco::kPhys), lds, s, g, xa, tw, H,
                       xmap, ymap, out);
    return launch_check();
  }
  if (!run) return AAMD_OK;             // the time-domain kernel reads the taps as they are: nothing to prepare
  FcGeom g;
  g.rows = rows; g.start = start; g.out_len = out_len;
  g.nx = nxa;
  g.ny = nya;
  g.n_tiles = (int)((out_len + kFcTN - 1) / kFcTN);
  const int64_t blocks = rows * g.n_tiles;
  AAMD_CHECK_ARG(blocks < (1ll << 31), "too many tiles for one launch");
  hipLaunchKernelGGL(fftconv_direct_kernel, dim3((unsigned)blocks), dim3(kFcThreads), 0, s,
                     g, xa, ya, xmap, ymap, out);
  return launch_check();
}

}  // extern "C"

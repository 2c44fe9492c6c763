// CPU replay of the HIP kernels' per-thread phase functions (audio_amd/csrc/*.h compiled with
// g++, no GPU).  Each driver below mirrors the corresponding __global__ kernel's control flow,
// replacing "all threads run phase X, then __syncthreads()" by a loop over thread ids.
// TEST INFRASTRUCTURE ONLY: lets the build container check index math / algorithms of the
// exact device source before a GPU is available.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <array>
#include <vector>

#include "../../include/audio_amd.h"
#include "../../audio_amd/csrc/db_mfcc.h"
#include "../../audio_amd/csrc/fftconv.h"
#include "../../audio_amd/csrc/fftconv_os.h"
#include "../../audio_amd/csrc/fftconv_fdr.h"
#include "../../audio_amd/csrc/istft.h"
#include "../../audio_amd/csrc/vocoder.h"
#include "../../audio_amd/csrc/stft_pow2.h"
#include "../../audio_amd/csrc/istft400.h"
#include "../../audio_amd/csrc/kaldi_generic.h"
#include "../../audio_amd/csrc/lfilter.h"
#include "../../audio_amd/csrc/lfilter_wave.h"
#include "../../audio_amd/csrc/melspec400.h"
#include "../../audio_amd/csrc/resample.h"
#include "../../audio_amd/csrc/resample_mfma.h"
#include "../../audio_amd/csrc/stft_generic.h"

using namespace aamd;

static void fill_geom(const aamd_stft_desc* d, StftGeom& g) {
  g.rows = d->rows; g.length = d->length; g.row_stride = d->row_stride;
  g.n_fft = d->n_fft; g.hop = d->hop; g.pad = d->pad; g.center = d->center;
  g.pad_mode = d->pad_mode; g.onesided = d->onesided; g.n_frames = d->n_frames;
  g.n_freq = d->onesided ? d->n_fft / 2 + 1 : d->n_fft;
  g.scale = d->scale; g.power = d->power;
  g.n_stages = plan_radices(d->n_fft, g.radix);
}

// stft_pow2_kernel: one "wave" of 64 lanes per frame pair, phases separated where the kernel has wave_lds_sync()
template <int E>
static int sim_pow2_e(const float* wav, const float* window, const float* tw, const MelBandsDev& mb, float* out,
                      const StftGeom& g, int epi_mel) {
  using namespace p2;
  constexpr int N = Cfg<E>::N;
  const C32* twc = reinterpret_cast<const C32*>(tw);
  std::vector<LaneTab<E>> lt(64);
  for (int l = 0; l < 64; ++l) lane_tab<E>(l, window, twc, g.scale, lt[l]);
  std::vector<C32> lds(Cfg<E>::lds_complex);
  std::vector<std::array<C32, p2::Cfg<E>::VR>> v(64), z(64);
  std::vector<std::array<C32, E / 2 + 1>> A(64), B(64);
  const int64_t ppr = (g.n_frames + 1) / 2;
  const int opf = epi_mel ? mb.n_mels : (g.power <= 0.0f ? N + 2 : N / 2 + 1);
  std::vector<float> tab;
  const float* mel_tab = nullptr;
  if (epi_mel && mel_in_lds(mb.n_mels, mb.max_width)) {
    tab.resize(mel_lds_floats(mb.n_mels, mb.max_width));
    for (int t = 0; t < 256; ++t) mel_stage(t, 256, mb, tab.data());
    mel_tab = tab.data();
  }
  for (int64_t pair = 0; pair < g.rows * ppr; ++pair) {
    const int64_t row = pair / ppr, ta = 2 * (pair - row * ppr);
    for (int l = 0; l < 64; ++l) { load_pair<E>(l, g, wav + row * g.row_stride, ta, lt[l], v[l].data()); stage_a<E>(lt[l], v[l].data()); }
    for (int l = 0; l < 64; ++l) xch1_write<E>(l, v[l].data(), lds.data());
    for (int l = 0; l < 64; ++l) { xch1_read<E>(l, lds.data(), v[l].data()); stage_b<E>(lt[l], v[l].data()); }
    for (int l = 0; l < 64; ++l) xch2_write<E>(l, v[l].data(), lds.data());
    for (int l = 0; l < 64; ++l) { xch2_read<E>(l, lds.data(), v[l].data()); stage_c<E>(v[l].data(), z[l].data()); }
    if (E < 8) {
      for (int l = 0; l < 64; ++l) redist_write<E>(l, z[l].data(), lds.data());
      for (int l = 0; l < 64; ++l) redist_read<E>(l, lds.data(), z[l].data());
    }
    for (int l = 0; l < 64; ++l) xch3_write<E>(l, z[l].data(), lds.data());
    for (int l = 0; l < 64; ++l) finish_bins<E>(l, z[l].data(), lds.data(), A[l].data(), B[l].data());
    float* out_row = out + row * g.n_frames * (int64_t)opf;
    if (!epi_mel) {
      for (int l = 0; l < 64; ++l) store_spec<E>(l, g, A[l].data(), B[l].data(), ta, out_row);
    } else {
      F2* P = reinterpret_cast<F2*>(lds.data());
      for (int l = 0; l < 64; ++l) power_rows<E>(l, g, A[l].data(), B[l].data(), P);
      const int tail_G = mel_tab ? mel_tail_lanes(mb.n_mels) : 1;
      for (int l = 0; l < 64; ++l) mel_rows<E>(l, g, mb, mel_tab, P, ta, out_row, tail_G > 1 ? mel_tail_first(mb.n_mels) : mb.n_mels);
      if (tail_G > 1) {
        float pa[64], pb[64], ta_[64], tb_[64];
        for (int l = 0; l < 64; ++l) mel_tail_partial(l, mb, mel_tab, P, tail_G, pa[l], pb[l]);
        for (int step = 1; step < tail_G; step *= 2) {
          for (int l = 0; l < 64; ++l) { const int sl = mel_tail_src(step, l); ta_[l] = sl >= 0 ? pa[sl] : 0.0f; tb_[l] = sl >= 0 ? pb[sl] : 0.0f; }
          for (int l = 0; l < 64; ++l) { pa[l] += ta_[l]; pb[l] += tb_[l]; }
        }
        for (int l = 0; l < 64; ++l) mel_tail_store(l, g, mb, tail_G, pa[l], pb[l], ta, out_row);
      }
    }
  }
  return 0;
}

// Write tracing for the inverse kernels (set by sim_trace_writes): per output sample, how many PLAIN stores and how many
// atomic adds hit it.  A sequential replay cannot show a store/atomic race by its value; the counts can.
static int32_t* g_trace_store = nullptr;
static int32_t* g_trace_add = nullptr;
static int g_force_generic = 0;   // sim_kaldi_features: replay the generic (any even size) kernel also for 256 ... 2048
static const float* g_trace_base = nullptr;

template <int E>
static int sim_istft_pow2_e(const float* spec, const float* window, const float* tw, const float* inv_env, float* out,
                            const StftGeom& g, float interior, float out_scale, int runs) {
  using namespace p2;
  constexpr int N = Cfg<E>::N, F = N / 2 + 1;
  const C32* twc = reinterpret_cast<const C32*>(tw);
  const C32* sp = reinterpret_cast<const C32*>(spec);
  InvGeom ig{g, interior};
  std::vector<LaneTab<E>> lt(64);
  for (int l = 0; l < 64; ++l) lane_tab<E>(l, window, twc, 2.0f * out_scale, lt[l]);
  std::vector<C32> lds(Cfg<E>::lds_complex);
  std::vector<float> ring(2 * N, 0.0f);
  std::vector<std::array<C32, p2::Cfg<E>::VR>> v(64), z(64);
  const int64_t ppr = (g.n_frames + 1) / 2;
  const int64_t c = (g.center ? N / 2 : 0) + g.pad;
  auto add = [](float* p, float x) { *p += x; if (g_trace_add) ++g_trace_add[p - g_trace_base]; };
  auto store = [](float* p, float x) { *p = x; if (g_trace_store) ++g_trace_store[p - g_trace_base]; };
  auto fft_pair = [&](int64_t row, int64_t ta, bool vb) {
    const C32* Sa = sp + (row * g.n_frames + ta) * (int64_t)F;
    for (int l = 0; l < 64; ++l) { inv_load<E>(l, ig, Sa, vb ? Sa + F : nullptr, v[l].data()); stage_a<E>(lt[l], v[l].data()); }
    for (int l = 0; l < 64; ++l) xch1_write<E>(l, v[l].data(), lds.data());
    for (int l = 0; l < 64; ++l) { xch1_read<E>(l, lds.data(), v[l].data()); stage_b<E>(lt[l], v[l].data()); }
    for (int l = 0; l < 64; ++l) xch2_write<E>(l, v[l].data(), lds.data());
    for (int l = 0; l < 64; ++l) { xch2_read<E>(l, lds.data(), v[l].data()); stage_c<E>(v[l].data(), z[l].data()); }
    if (E < 8) {
      for (int l = 0; l < 64; ++l) redist_write<E>(l, z[l].data(), lds.data());
      for (int l = 0; l < 64; ++l) redist_read<E>(l, lds.data(), z[l].data());
    }
  };
  if (!runs) {
    for (int64_t pair = 0; pair < g.rows * ppr; ++pair) {
      const int64_t row = pair / ppr, ta = 2 * (pair - row * ppr);
      const bool vb = ta + 1 < g.n_frames;
      fft_pair(row, ta, vb);
      for (int l = 0; l < 64; ++l) inv_store<E>(l, ig, lt[l], z[l].data(), ta, vb, inv_env, out + row * g.length, add);
    }
    return 0;
  }
  // istft_pow2_run_kernel; plain stores are modelled as stores (a double write would show up as a wrong value)
  const int run_len = runs;
  const int64_t rpr = (ppr + run_len - 1) / run_len;
  for (int64_t run = 0; run < g.rows * rpr; ++run) {
    const int64_t row = run / rpr, p_lo = (run - row * rpr) * run_len;
    const int64_t p_hi = p_lo + run_len < ppr ? p_lo + run_len : ppr;
    const RunPlan rp = run_plan<E>(g, p_lo, p_hi);
    float* out_row = out + row * g.length;
    int64_t flushed = 0;
    for (int64_t p = p_lo; p < p_hi; ++p) {
      const int64_t ta = 2 * p;
      const bool vb = ta + 1 < g.n_frames;
      fft_pair(row, ta, vb);
      if (p >= rp.pi_lo && p <= rp.pi_hi) {
        const int64_t sa = ta * (int64_t)g.hop - c;
        if (p == rp.pi_lo) flushed = sa;
        for (int l = 0; l < 64; ++l) ring_add<E>(l, lt[l], z[l].data(), sa, g.hop, true, ring.data());
        const int64_t s1 = p == rp.pi_hi ? sa + g.hop + N : sa + 2 * (int64_t)g.hop;
        for (int l = 0; l < 64; ++l) ring_flush<E>(l, rp, flushed, s1, inv_env, ring.data(), out_row, add, store);
        flushed = s1;
      } else {
        for (int l = 0; l < 64; ++l) inv_store<E>(l, ig, lt[l], z[l].data(), ta, vb, inv_env, out_row, add);
      }
    }
  }
  for (float r : ring) if (r != 0.0f) return -3;      // every sample must have been flushed
  return 0;
}

template <int E>
static int sim_kaldi_e(const float* wav, const float* window, const float* tw, const MelBandsDev& mb, float* out,
                       const p2::KaldiGeom& kg, int mode) {
  using namespace p2;
  const C32* twc = reinterpret_cast<const C32*>(tw);
  std::vector<LaneTab<E>> lt(64);
  for (int l = 0; l < 64; ++l) lane_tab<E>(l, window, twc, 2.0f, lt[l]);
  std::vector<C32> lds(Cfg<E>::lds_complex);
  std::vector<std::array<C32, p2::Cfg<E>::VR>> v(64), z(64);
  std::vector<std::array<C32, E / 2 + 1>> A(64), B(64);
  std::vector<std::array<float, E>> ra(64), rpa(64), rb(64), rpb(64), ya(64), yb(64);
  auto wave_sum = [](float* s) {           // the kernel's __shfl_xor butterfly, bit for bit
    for (int off = 32; off > 0; off >>= 1) { float t[64]; for (int l = 0; l < 64; ++l) t[l] = s[l] + s[l ^ off]; std::memcpy(s, t, sizeof(t)); }
    return s[0];
  };
  const float inv_win = 1.0f / (float)kg.win;
  for (int64_t pair = 0; pair < (kg.n_frames + 1) / 2; ++pair) {
    const int64_t ta = 2 * pair;
    float s[64];
    for (int l = 0; l < 64; ++l) { kaldi_load<E>(l, kg, wav, ta, ra[l].data(), rpa[l].data()); kaldi_load<E>(l, kg, wav, ta + 1, rb[l].data(), rpb[l].data()); }
    float mean_a = 0, mean_b = 0, ea = 0, eb = 0;
    if (kg.remove_dc) {
      for (int l = 0; l < 64; ++l) s[l] = kaldi_partial_sum<E>(ra[l].data());
      mean_a = wave_sum(s) * inv_win;
      for (int l = 0; l < 64; ++l) s[l] = kaldi_partial_sum<E>(rb[l].data());
      mean_b = wave_sum(s) * inv_win;
    }
    if (kg.raw_energy) {
      for (int l = 0; l < 64; ++l) s[l] = kaldi_partial_sumsq<E>(l, kg, ra[l].data(), mean_a);
      ea = kaldi_log_energy(kg, wave_sum(s));
      for (int l = 0; l < 64; ++l) s[l] = kaldi_partial_sumsq<E>(l, kg, rb[l].data(), mean_b);
      eb = kaldi_log_energy(kg, wave_sum(s));
    }
    for (int l = 0; l < 64; ++l) {
      kaldi_shape<E>(l, kg, lt[l].win, ra[l].data(), rpa[l].data(), mean_a, ya[l].data());
      kaldi_shape<E>(l, kg, lt[l].win, rb[l].data(), rpb[l].data(), mean_b, yb[l].data());
    }
    if (!kg.raw_energy) {
      for (int l = 0; l < 64; ++l) s[l] = kaldi_partial_sumsq<E>(l, kg, ya[l].data(), 0.0f);
      ea = kaldi_log_energy(kg, wave_sum(s));
      for (int l = 0; l < 64; ++l) s[l] = kaldi_partial_sumsq<E>(l, kg, yb[l].data(), 0.0f);
      eb = kaldi_log_energy(kg, wave_sum(s));
    }
    for (int l = 0; l < 64; ++l) {
      for (int e = 0; e < E; ++e) v[l][e] = C32{0.5f * ya[l][e], 0.5f * yb[l][e]};
      stage_a<E>(lt[l], v[l].data());
    }
    for (int l = 0; l < 64; ++l) xch1_write<E>(l, v[l].data(), lds.data());
    for (int l = 0; l < 64; ++l) { xch1_read<E>(l, lds.data(), v[l].data()); stage_b<E>(lt[l], v[l].data()); }
    for (int l = 0; l < 64; ++l) xch2_write<E>(l, v[l].data(), lds.data());
    for (int l = 0; l < 64; ++l) { xch2_read<E>(l, lds.data(), v[l].data()); stage_c<E>(v[l].data(), z[l].data()); }
    if (E < 8) {
      for (int l = 0; l < 64; ++l) redist_write<E>(l, z[l].data(), lds.data());
      for (int l = 0; l < 64; ++l) redist_read<E>(l, lds.data(), z[l].data());
    }
    for (int l = 0; l < 64; ++l) xch3_write<E>(l, z[l].data(), lds.data());
    for (int l = 0; l < 64; ++l) finish_bins<E>(l, z[l].data(), lds.data(), A[l].data(), B[l].data());
    if (mode == 0) {
      for (int l = 0; l < 64; ++l) kaldi_store_spec<E>(l, kg, A[l].data(), B[l].data(), ta, ea, eb, out);
    } else {
      F2* P = reinterpret_cast<F2*>(lds.data());
      for (int l = 0; l < 64; ++l) kaldi_power_rows<E>(l, kg, A[l].data(), B[l].data(), P);
      for (int l = 0; l < 64; ++l) kaldi_fbank_rows<E>(l, kg, mb, P, ta, ea, eb, out);
    }
  }
  return 0;
}

template <int H>
static int sim_istft400_h(const float* spec, const float* window, const float* tw400, const float* inv_env, float* out,
                          const StftGeom& g, float interior, float out_scale) {
  using namespace m400;
  using HC = Hop<H>;
  alignas(16) static float lds[HC::lds_dwords];
  alignas(16) static float ctab[kConstDwords];
  for (int tid = 0; tid < 256; ++tid) const_tab_build(tid, 256, window, tw400, 2.0f * out_scale, ctab);
  LaneConst c[64];
  for (int l = 0; l < 64; ++l) lane_init(l, ctab, c[l]);
  Inv400Geom ig{g, interior};
  const cplx<float>* sp = reinterpret_cast<const cplx<float>*>(spec);
  const int tiles_per_row = (g.n_frames + kFramesPerWave - 1) / kFramesPerWave;
  static float xr[64][20], xi[64][20], vr[64][20], vi[64][20], zr[64][20], zi[64][20];
  auto add = [](float* p, float v) { *p += v; if (g_trace_add) ++g_trace_add[p - g_trace_base]; };
  auto store = [](float* p, float v) { *p = v; if (g_trace_store) ++g_trace_store[p - g_trace_base]; };
  for (int64_t tile = 0; tile < g.rows * tiles_per_row; ++tile) {
    const int64_t row = tile / tiles_per_row, t0 = (tile - row * tiles_per_row) * kFramesPerWave;
    const int64_t left = g.n_frames - t0;
    const int n_valid = left < kFramesPerWave ? (int)left : kFramesPerWave;
    for (int l = 0; l < 64; ++l) inv400_load(c[l], ig, sp + row * g.n_frames * (int64_t)kSpecBins, t0, xr[l], xi[l]);
    for (int l = 0; l < 64; ++l) phase_a_core<false>(c[l], xr[l], xi[l], lds);
    for (int l = 0; l < 64; ++l) phase_b1_load(c[l], lds, vr[l], vi[l]);
    for (int l = 0; l < 64; ++l) inv400_zero<H>(l, lds);
    for (int l = 0; l < 64; ++l) dft20(vr[l], vi[l], zr[l], zi[l]);
    for (int ph = 0; ph < 4; ++ph)
      for (int l = 0; l < 64; ++l)
        inv400_add<H>(c[l], ctab + 20 * kTwRow + 20 * c[l].col, zr[l], zi[l], ph & 1, ph >> 1, n_valid, lds);
    for (int l = 0; l < 64; ++l) inv400_flush<H>(l, ig, t0, n_valid, lds, inv_env, out + row * g.length, add, store);
  }
  return 0;
}

extern "C" {

int sim_stft_generic(const float* wav, const float* window, const float* tw, const aamd_mel_bands* bands,
                     float* out, const aamd_stft_desc* d, int epi_mel) {
  StftGeom g; fill_geom(d, g);
  if (g.n_stages < 0) return -1;
  MelBandsDev mb{};
  if (epi_mel) { mb.n_mels = bands->n_mels; mb.max_width = bands->max_width; mb.lo = bands->lo; mb.width = bands->width; mb.weights = bands->weights; }
  const int nthr = kGenThreads, N = g.n_fft, SL = gen_seq_len(N);
  int pb = gen_pairs_per_block(N);
  const int pairs_per_row = (g.n_frames + 1) / 2;
  if (pb > pairs_per_row) pb = pairs_per_row;
  const int bpr = (pairs_per_row + pb - 1) / pb;
  std::vector<cplx<float>> A((size_t)pb * SL), B((size_t)pb * SL);
  std::vector<float> P((size_t)2 * pb * g.n_freq);
  const cplx<float>* twc = reinterpret_cast<const cplx<float>*>(tw);
  const int opf = epi_mel ? mb.n_mels : (g.power <= 0.f ? 2 * g.n_freq : g.n_freq);
  for (int64_t row = 0; row < g.rows; ++row)
    for (int chunk = 0; chunk < bpr; ++chunk) {
      const int64_t t0 = (int64_t)chunk * 2 * pb;
      const float* wr = wav + row * g.row_stride;
      for (int tid = 0; tid < nthr; ++tid) gen_load<float>(tid, nthr, g, wr, window, t0, pb, A.data());
      cplx<float>* x = A.data(); cplx<float>* y = B.data();
      int s = 1;
      for (int st = 0; st < g.n_stages; ++st) {
        for (int tid = 0; tid < nthr; ++tid) gen_stage<float>(tid, nthr, N, g.radix[st], s, pb, x, y, twc);
        s *= g.radix[st];
        std::swap(x, y);
      }
      float* out_row = out + row * g.n_frames * (int64_t)opf;
      if (!epi_mel) {
        for (int tid = 0; tid < nthr; ++tid) gen_store_spec<float>(tid, nthr, g, x, pb, t0, out_row);
      } else {
        for (int tid = 0; tid < nthr; ++tid) gen_power_rows<float>(tid, nthr, g, x, pb, P.data());
        for (int tid = 0; tid < nthr; ++tid) gen_mel<float>(tid, nthr, g, mb, P.data(), pb, t0, out_row);
      }
    }
  return 0;
}

// Replay of ola_kernel (inverse STFT / STFT adjoint): same launcher logic as aamd_istft_f32.
int sim_stft_pow2(const float* wav, const float* window, const float* tw, const aamd_mel_bands* bands, float* out,
                  const aamd_stft_desc* d) {
  StftGeom g{};
  fill_geom(d, g);
  MelBandsDev mb{};
  if (bands) { mb.n_mels = bands->n_mels; mb.max_width = bands->max_width; mb.lo = bands->lo; mb.width = bands->width; mb.weights = bands->weights; }
  if (!g.onesided) return -1;
  if (g.n_fft == 256) return sim_pow2_e<4>(wav, window, tw, mb, out, g, bands != nullptr);
  if (g.n_fft == 512) return sim_pow2_e<8>(wav, window, tw, mb, out, g, bands != nullptr);
  if (g.n_fft == 1024) return sim_pow2_e<16>(wav, window, tw, mb, out, g, bands != nullptr);
  if (g.n_fft == 2048) return sim_pow2_e<32>(wav, window, tw, mb, out, g, bands != nullptr);
  return -2;
}

// phase_vocoder_kernel / griffinlim_update_kernel: one chain / element per "thread"
int sim_phase_vocoder(const float* spec, const float* phase_advance, float* out, const aamd_vocoder_desc* d) {
  VocoderGeom g{d->rows, d->n_freq, d->n_frames_in, d->n_frames_out, d->in_stride_row, d->in_stride_freq,
                d->in_stride_frame, d->out_stride_row, d->out_stride_freq, d->out_stride_frame, d->rate};
  const cplx<float>* in = reinterpret_cast<const cplx<float>*>(spec);
  cplx<float>* o = reinterpret_cast<cplx<float>*>(out);
  for (int64_t chain = 0; chain < g.rows * g.n_freq; ++chain) {
    const int64_t row = chain / g.n_freq;
    const int f = (int)(chain - row * g.n_freq);
    vocoder_chain(g, in + row * g.in_row + f * g.in_f, phase_advance[f], o + row * g.out_row + f * g.out_f);
  }
  return 0;
}

int sim_griffinlim_update(const float* rebuilt, float* tprev, const float* mag, float* next, int64_t n, float momentum) {
  const cplx<float>* r = reinterpret_cast<const cplx<float>*>(rebuilt);
  cplx<float>* tp = reinterpret_cast<cplx<float>*>(tprev);
  cplx<float>* nx = reinterpret_cast<cplx<float>*>(next);
  for (int64_t i = 0; i < n; ++i) griffinlim_update_elem(r[i], tp[i], mag[i], momentum, nx[i]);
  return 0;
}

void sim_trace_writes(const float* base, int32_t* stores, int32_t* adds) {
  g_trace_base = base; g_trace_store = stores; g_trace_add = adds;
}

int sim_istft400(const float* spec, const float* window, const float* tw, const float* inv_env, float* out,
                 const aamd_stft_desc* d, int adjoint) {
  StftGeom g{};
  fill_geom(d, g);
  g.onesided = 1; g.n_freq = 201; g.row_stride = g.length;
  if (g.n_fft != 400 || !g.center || g.pad != 0) return -2;
  const float interior = adjoint ? 0.5f : 1.0f;
  const float scale = d->scale * (adjoint ? 1.0f : 1.0f / 400.0f);
  if (g.hop == 100) return sim_istft400_h<5>(spec, window, tw, inv_env, out, g, interior, scale);
  if (g.hop == 160) return sim_istft400_h<8>(spec, window, tw, inv_env, out, g, interior, scale);
  if (g.hop == 200) return sim_istft400_h<10>(spec, window, tw, inv_env, out, g, interior, scale);
  return -2;
}

void sim_set_force_generic(int v) { g_force_generic = v; }

int sim_kaldi_features(const float* wav, const float* window, const float* tw, const aamd_mel_bands* bands, float* out,
                       const aamd_kaldi_desc* d) {
  p2::KaldiGeom kg{};
  kg.n_samples = d->n_samples; kg.n_frames = d->n_frames; kg.shift = d->shift; kg.win = d->win;
  kg.snip_edges = d->snip_edges; kg.pad_left = d->win / 2 - d->shift / 2;
  kg.preemph = d->preemphasis; kg.remove_dc = d->remove_dc_offset; kg.raw_energy = d->raw_energy;
  kg.log_energy_floor = d->energy_floor > 0.0f ? std::log(d->energy_floor) : -INFINITY;
  kg.eps = 1.1920928955078125e-07f;
  kg.use_power = d->use_power; kg.use_log = d->use_log;
  kg.energy_col = d->energy_col; kg.first_col = d->first_col; kg.n_cols = d->n_cols;
  MelBandsDev mb{};
  if (bands) { mb.n_mels = bands->n_mels; mb.max_width = bands->max_width; mb.lo = bands->lo; mb.width = bands->width; mb.weights = bands->weights; }
  kg.noise = d->dither != 0.0f ? d->noise : nullptr; kg.dither = d->dither;
  kg.n_utt = 1; kg.utt_stride = d->n_samples;
  const int mode = bands ? 1 : 0;
  const bool pow2 = d->n_fft == 256 || d->n_fft == 512 || d->n_fft == 1024 || d->n_fft == 2048;
  if (!pow2 || g_force_generic) {   // replay of kgen::kaldi_generic_kernel, phase by phase
    using namespace kgen;
    Plan plan{};
    plan.n_fft = d->n_fft;
    plan.n_stages = plan_radices(d->n_fft, plan.radix);
    if (plan.n_stages < 0) return -1;
    const int N = d->n_fft, pb = pairs_per_block(N), nf = 2 * pb, nthr = kThreads;
    std::vector<float> mem(lds_floats(N, pb) + 16);
    const Lds l = carve(mem.data(), N, pb);
    const cplx<float>* twc = reinterpret_cast<const cplx<float>*>(tw);
    for (int i = 0; i < N; ++i) l.twl[i] = twc[i];
    for (int64_t t0 = 0; t0 < kg.n_frames; t0 += nf) {
      for (int tid = 0; tid < nthr; ++tid) pass_sum(tid, nthr, nf, kg, wav, t0, l.red);
      for (int tid = 0; tid < nthr; ++tid) fold_mean(tid, nthr, nf, kg, l.red, l.stat);
      if (kg.raw_energy) {
        for (int tid = 0; tid < nthr; ++tid) pass_sumsq(tid, nthr, nf, kg, wav, t0, l.stat, l.red);
        for (int tid = 0; tid < nthr; ++tid) fold_energy(tid, nthr, nf, kg, l.red, l.stat);
      }
      for (int tid = 0; tid < nthr; ++tid) pass_shape(tid, nthr, nf, N, kg, wav, window, t0, l.stat, l.bufA, l.red);
      if (!kg.raw_energy)
        for (int tid = 0; tid < nthr; ++tid) fold_energy(tid, nthr, nf, kg, l.red, l.stat);
      cplx<float>* x = l.bufA; cplx<float>* y = l.bufB;
      int sstride = 1;
      for (int st = 0; st < plan.n_stages; ++st) {
        for (int tid = 0; tid < nthr; ++tid) gen_stage<float>(tid, nthr, N, plan.radix[st], sstride, pb, x, y, l.twl);
        sstride *= plan.radix[st];
        std::swap(x, y);
      }
      if (mode == 0) {
        for (int tid = 0; tid < nthr; ++tid) store_spec(tid, nthr, nf, N, kg, x, t0, l.stat, out);
      } else {
        for (int tid = 0; tid < nthr; ++tid) power_rows(tid, nthr, nf, N, kg, x, l.P);
        for (int tid = 0; tid < nthr; ++tid) fbank_rows(tid, nthr, nf, N, kg, mb, l.P, t0, l.stat, out);
      }
    }
    return 0;
  }
  if (d->n_fft == 256) return sim_kaldi_e<4>(wav, window, tw, mb, out, kg, mode);
  if (d->n_fft == 512) return sim_kaldi_e<8>(wav, window, tw, mb, out, kg, mode);
  if (d->n_fft == 1024) return sim_kaldi_e<16>(wav, window, tw, mb, out, kg, mode);
  if (d->n_fft == 2048) return sim_kaldi_e<32>(wav, window, tw, mb, out, kg, mode);
  return -2;
}

int sim_istft_pow2(const float* spec, const float* window, const float* tw, const float* inv_env, float* out,
                   const aamd_stft_desc* d, int adjoint, int runs) {
  StftGeom g{};
  fill_geom(d, g);
  g.onesided = 1; g.n_freq = g.n_fft / 2 + 1; g.row_stride = g.length;
  const float interior = adjoint ? 0.5f : 1.0f;
  const float scale = d->scale * (adjoint ? 1.0f : 1.0f / (float)d->n_fft);
  if (g.n_fft == 256) return sim_istft_pow2_e<4>(spec, window, tw, inv_env, out, g, interior, scale, runs);
  if (g.n_fft == 512) return sim_istft_pow2_e<8>(spec, window, tw, inv_env, out, g, interior, scale, runs);
  if (g.n_fft == 1024) return sim_istft_pow2_e<16>(spec, window, tw, inv_env, out, g, interior, scale, runs);
  if (g.n_fft == 2048) return sim_istft_pow2_e<32>(spec, window, tw, inv_env, out, g, interior, scale, runs);
  return -2;
}

int sim_istft(const float* spec, const float* window, const float* tw, const float* inv_env, float* out,
              const aamd_stft_desc* d, int adjoint) {
  OlaGeom og{};
  fill_geom(d, og.g);
  StftGeom& g = og.g;
  g.onesided = 1; g.n_freq = g.n_fft / 2 + 1; g.row_stride = g.length;
  if (g.n_stages < 0) return -1;
  og.interior = adjoint ? 0.5f : 1.0f;
  og.scale = d->scale * (adjoint ? 1.0f : 1.0f / (float)d->n_fft);
  const int nthr = kGenThreads, N = g.n_fft, SL = gen_seq_len(N);
  int pb = gen_pairs_per_block(N);
  const int pairs_per_row = (g.n_frames + 1) / 2;
  if (pb > pairs_per_row) pb = pairs_per_row;
  const int bpr = (pairs_per_row + pb - 1) / pb;
  std::vector<cplx<float>> A((size_t)pb * SL), B((size_t)pb * SL);
  const cplx<float>* twc = reinterpret_cast<const cplx<float>*>(tw);
  for (int64_t row = 0; row < g.rows; ++row)
    for (int chunk = 0; chunk < bpr; ++chunk) {
      const int64_t t0 = (int64_t)chunk * 2 * pb;
      for (int tid = 0; tid < nthr; ++tid)
        ola_load<float>(tid, nthr, og, spec + row * g.n_frames * 2 * (int64_t)g.n_freq, t0, pb, A.data());
      cplx<float>* x = A.data(); cplx<float>* y = B.data();
      int s = 1;
      for (int st = 0; st < g.n_stages; ++st) {
        for (int tid = 0; tid < nthr; ++tid) gen_stage<float>(tid, nthr, N, g.radix[st], s, pb, x, y, twc);
        s *= g.radix[st];
        std::swap(x, y);
      }
      int64_t nf = g.n_frames - t0; if (nf > 2 * pb) nf = 2 * pb;
      const int span = (int)(nf - 1) * g.hop + N;
      for (int j = 0; j < span; ++j) {
        const int64_t i = ola_target(g, t0 * g.hop + j);
        if (i < 0) continue;
        float v = ola_gather<float>(og, x, window, (int)nf, j);
        if (inv_env) v *= inv_env[i];
        out[row * g.length + i] += v;
      }
    }
  return 0;
}

// epi: 0 mel, 1 mel + dB (db = {multiplier, amin, db_sub}; gmax[rows / rows_per_group] max-reduced),
//      2 spectrogram |X|^power (bands unused)
// fused MFCC (EPI400_MFCC): the extra operands of the epilogue, set before sim_melspec400(epi_mode = 4)
static struct { std::vector<float> frag; int n_mfcc = 0; float top_db = 0.f; float* tile_min = nullptr; int fixup = 0; int fix_count = 0; } g_mfcc;
int sim_mfcc_fused_setup(const float* dct, int n_mels, int n_mfcc, float top_db, float* tile_min, int fixup) {
  // what mfcc_frag_build_kernel writes: binary16 hi / lo planes of the scaled DCT weights, in 16-byte pieces
  g_mfcc.frag.assign(m400::kMfccFragFloats, 0.f);
  uint16_t* fh = reinterpret_cast<uint16_t*>(g_mfcc.frag.data());
  for (int t = 0; t < m400::kMfccMT; ++t)
    for (int sidx = 0; sidx < m400::kMfccSteps; ++sidx)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const float v = m400::mfcc_frag_value(dct, n_mels, n_mfcc, t, sidx, lane, j);
          const uint16_t hi = rsm::f16_bits(v);
          fh[m400::mfcc_frag_piece(t, sidx, 0, lane) * 8 + j] = hi;
          fh[m400::mfcc_frag_piece(t, sidx, 1, lane) * 8 + j] = rsm::f16_bits(v - rsm::f16_value(hi));
        }
  g_mfcc.n_mfcc = n_mfcc; g_mfcc.top_db = top_db; g_mfcc.tile_min = tile_min; g_mfcc.fixup = fixup; g_mfcc.fix_count = 0;
  return 0;
}
int sim_mfcc_fused_fix_count() { return g_mfcc.fix_count; }
}  // extern "C" (pause: template)
template <int H, typename TIn>
static int sim_melspec400_h(const TIn* wav, const float* window, const float* tw400, const aamd_mel_bands* bands,
                            float* out, int64_t rows, int64_t length, int64_t row_stride, int n_frames, float scale,
                            int epi_mode, const float* db, float* gmax, int64_t rows_per_group, float power,
                            int want_wide) {
  using namespace m400;
  using HC = Hop<H>;
  constexpr int kHop = HC::hop;
  MelBandsDev mb{};
  if (epi_mode != EPI400_SPEC) {
    mb = MelBandsDev{bands->n_mels, bands->max_width, bands->lo, bands->width, bands->weights, bands->lane_order};
    if (mel_ws(mb.max_width) > kMelMaxTaps + 4 || mel_rounds(mb.n_mels) > kMelMaxRounds) return -2;
  }
  Epi400 epi{};
  if (epi_mode == EPI400_MEL_DB || epi_mode == EPI400_MFCC) { epi.multiplier = db[0]; epi.amin = db[1]; epi.db_sub = db[2]; }
  if (epi_mode == EPI400_MFCC && (mb.n_mels != kMfccMels || g_mfcc.n_mfcc <= 0 || g_mfcc.n_mfcc % 4 || g_mfcc.n_mfcc > 48)) return -5;
  // MEL_NORM: db = {gain, out_frames}; gmax = [mean(n_mels) | invstddev(n_mels)]
  if (epi_mode == EPI400_MEL_NORM) { epi.gain = db[0]; epi.out_frames = (int64_t)db[1]; epi.mean = gmax; epi.invstd = gmax + bands->n_mels; }
  epi.power = power;
  alignas(16) static float lds[HC::lds_dwords];
  alignas(16) static float tab[kMelMaxRounds * kMelSlots * (kMelMaxTaps + 4 + 2) + 256];
  alignas(16) static float ctab[kConstDwords];
  for (int tid = 0; tid < 256; ++tid) const_tab_build(tid, 256, window, tw400, scale, ctab);
  MelTab mt{};
  if (epi_mode != EPI400_SPEC && bands->table400 != nullptr) {   // the prebuilt image (host or device builder), as the kernel copies it
    std::memcpy(tab, bands->table400, (size_t)mel_tab_dwords(mb.n_mels, mb.max_width) * 4);
    mel_tab_layout(mb, tab, mt);
  } else if (epi_mode != EPI400_SPEC) {
    for (int tid = 0; tid < 256; ++tid) mel_tab_rounds(tid, 256, mb, tab, mt);
    for (int tid = 0; tid < 256; ++tid) mel_tab_fill(tid, 256, mb, tab, mt);
  }
  LaneConst c[64];
  for (int l = 0; l < 64; ++l) lane_init(l, ctab, c[l]);
  const int tiles_per_row = (n_frames + kFramesPerWave - 1) / kFramesPerWave;
  // same launch-time switches as launch_mel400() in c_api.hip
  const bool in_aligned = (row_stride % (16 / (int)sizeof(TIn)) == 0);
  const bool out_wide = (epi_mode == EPI400_SPEC) || (want_wide && mb.n_mels % 4 == 0);
  static float X[64][HC::nx], vr[64][20], vi[64][20], zr[64][20], zi[64][20], qr[64][10], qi[64][10];
  static float acc_a[64][kMelMaxRounds], acc_b[64][kMelMaxRounds];
  auto staged = [&](int64_t t0) {
    return in_aligned && (t0 * kHop - kPad >= 0) && ((t0 + kFramesPerWave - 1) * kHop + (kN - kPad) <= length) &&
           (t0 + kFramesPerWave <= n_frames);
  };
  // what the 5 LDS-DMA instructions of stage_issue() do: staging piece u <- tile piece stage_src_piece(u)
  auto stage = [&](int64_t row, int64_t t0) {
    const TIn* src = wav + (row / InTraits<TIn>::chans) * row_stride + (t0 * kHop - kPad);
    using SG = Stage<H, TIn>;
    for (int u = 0; u < 64 * SG::ndma; ++u)      // 16-B pieces, byte for byte
      std::memcpy(reinterpret_cast<char*>(lds + kSOff) + 16 * u,
                  reinterpret_cast<const char*>(src) + 16 * stage_src_piece<H, TIn>(u), 16);
  };
  // one wave walks all tiles in order, exactly like the kernel's tile loop
  const int64_t n_tiles = rows * tiles_per_row;
  bool cur_staged = n_tiles > 0 && staged(0);
  if (cur_staged) stage(0, 0);
  const bool fix = epi_mode == EPI400_MFCC && g_mfcc.fixup;
  for (int64_t tile = 0; tile < n_tiles; ++tile) {
    const int64_t row = tile / tiles_per_row, t0 = (tile % tiles_per_row) * kFramesPerWave;
    const int64_t nrow = (tile + 1) / tiles_per_row, nt0 = ((tile + 1) % tiles_per_row) * kFramesPerWave;
    const bool nxt_staged = (tile + 1 < n_tiles) && staged(nt0);
    float fix_cut = -INFINITY;
    if (fix) {   // the fix-up pass redoes flagged tiles only, staging their samples on the spot
      fix_cut = gmax[row / rows_per_group] - g_mfcc.top_db;
      cur_staged = staged(t0);
      if (!(g_mfcc.tile_min[tile] < fix_cut)) continue;
      ++g_mfcc.fix_count;
      if (cur_staged) stage(row, t0);
    }
    const TIn* wr = wav + (row / InTraits<TIn>::chans) * row_stride;
    const int chan = (int)(row % InTraits<TIn>::chans);
    for (int l = 0; l < 64; ++l) {
      if (cur_staged) gather_lds<H, TIn>(c[l], reinterpret_cast<const TIn*>(lds + kSOff), X[l], chan);
      else gather_global<H, TIn>(c[l], wr, length, t0, n_frames, X[l], chan);
    }
    for (int l = 0; l < 64; ++l) phase_a<H>(c[l], X[l], lds);
    for (int l = 0; l < 64; ++l) phase_b1_load(c[l], lds, vr[l], vi[l]);
    if (nxt_staged && !fix) stage(nrow, nt0);
    for (int l = 0; l < 64; ++l) dft20(vr[l], vi[l], zr[l], zi[l]);
    for (int l = 0; l < 64; ++l) phase_b2_send(c[l], zr[l], zi[l], qr[l], qi[l]);
    // exchange_partner(): in-place DPP swap with lane ^ 1, self-paired columns (0 and 10) keep their own
    static float xr[64][10], xi[64][10];
    for (int l = 0; l < 64; ++l) {
      const bool self = (c[l].col == 0) || (c[l].col == 10);
      for (int i = 0; i < 10; ++i) { xr[l][i] = self ? qr[l][i] : qr[l ^ 1][i]; xi[l][i] = self ? qi[l][i] : qi[l ^ 1][i]; }
    }
    // the DPP quad_perm [1,0,3,2] swap: lane l receives lane l ^ 1's q
    if (epi_mode == EPI400_SPEC) {
      const int64_t left = n_frames - t0;
      const int n_valid = left < kFramesPerWave ? (int)left : kFramesPerWave;
      if (epi.power > 0.0f) {
        const int64_t a0 = (row * n_frames + t0) * (int64_t)kSpecBins;
        for (int l = 0; l < 64; ++l)
          phase_b2_spec(c[l], zr[l], zi[l], xr[l], xi[l], epi.power, (int)(a0 & 3), lds);
        for (int l = 0; l < 64; ++l) {
          if (n_valid == kFramesPerWave) store_spec_full(l, lds, out, a0);     // (the kernel's choice for full tiles)
          else store_spec(l, lds, out, a0, n_valid * kSpecBins);
        }
      } else {
        for (int half = 0; half < 2; ++half) {
          const int64_t a0 = (row * n_frames + t0 + 3 * half) * (int64_t)(2 * kSpecBins);
          const int nv = n_valid - 3 * half < 3 ? n_valid - 3 * half : 3;
          for (int l = 0; l < 64; ++l) phase_b2_spec_complex(c[l], zr[l], zi[l], xr[l], xi[l], half, (int)(a0 & 3), lds);
          if (nv > 0) for (int l = 0; l < 64; ++l) store_spec(l, lds, out, a0, nv * 2 * kSpecBins);
        }
      }
      cur_staged = nxt_staged;
      continue;
    }
    for (int l = 0; l < 64; ++l) phase_b2(c[l], zr[l], zi[l], xr[l], xi[l], lds);
    for (int l = 0; l < 64; ++l) phase_b2_pad(l, lds);
    for (int l = 0; l < 64; ++l) phase_c(c[l], mt, lds, acc_a[l], acc_b[l]);
    if (epi_mode == EPI400_MEL_DB || epi_mode == EPI400_MFCC) {
      float m = -INFINITY, tmin = INFINITY;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < mt.n_rounds; ++r) {
          acc_a[l][r] = epi_db(acc_a[l][r], epi);
          acc_b[l][r] = epi_db(acc_b[l][r], epi);
          const bool real = mt.row_mel[r * kMelSlots + c[l].pi] >= 0;
          if (t0 + 2 * c[l].p < n_frames) { m = std::fmax(m, acc_a[l][r]); if (real) tmin = std::fmin(tmin, acc_a[l][r]); }
          if (t0 + 2 * c[l].p + 1 < n_frames) { m = std::fmax(m, acc_b[l][r]); if (real) tmin = std::fmin(tmin, acc_b[l][r]); }
          if (epi_mode == EPI400_MFCC) { acc_a[l][r] = std::fmax(acc_a[l][r], fix_cut); acc_b[l][r] = std::fmax(acc_b[l][r], fix_cut); }
        }
      if (gmax && !fix) { float& g = gmax[row / rows_per_group]; g = std::fmax(g, m); }
      if (epi_mode == EPI400_MFCC && !fix) g_mfcc.tile_min[tile] = tmin;
    }
    if (epi_mode == EPI400_MFCC) {
      // the staged binary16 planes + the fragment maps of v_mfma_f32_16x16x32_f16 / 16x16x16_f16 (A[l & 15][8 (l >> 4) + j],
      // B[8 (l >> 4) + j][l & 15], C row 4 (l >> 4) + r).  Column n of B: n = 0 .. 5 the hi plane of frame n, n = 8 .. 13 the lo
      // plane of frame n - 8 (mfcc_b_index); A_lo B, then A_hi B; lane j += lane j + 8 of its row of 16 (the DPP row_ror:8 add)
      uint16_t* hs = reinterpret_cast<uint16_t*>(lds);
      uint16_t* ls = hs + kMfccPlaneHalves;
      for (int l = 0; l < 64; ++l) {
        for (int r = 0; r < mt.n_rounds; ++r) {
          const int m = mt.row_mel[r * kMelSlots + c[l].pi];
          if (m < 0) return -7;                 // (the kernel traps: 80 mels = 4 full rounds)
          const float va = acc_a[l][r] * kMfccYScale, vb = acc_b[l][r] * kMfccYScale;
          const uint16_t ha = rsm::f16_bits(va), hb = rsm::f16_bits(vb);
          hs[2 * c[l].p * kMfccMels + m] = ha;
          ls[2 * c[l].p * kMfccMels + m] = rsm::f16_bits(va - rsm::f16_value(ha));
          hs[(2 * c[l].p + 1) * kMfccMels + m] = hb;
          ls[(2 * c[l].p + 1) * kMfccMels + m] = rsm::f16_bits(vb - rsm::f16_value(hb));
        }
      }
      static float C3[kMfccMT][16][16];
      std::memset(C3, 0, sizeof(C3));
      const uint16_t* fh = reinterpret_cast<const uint16_t*>(g_mfcc.frag.data());
      for (int sidx = 0; sidx < kMfccSteps; ++sidx)
        for (int hl = 1; hl >= 0; --hl)
          for (int t = 0; t < kMfccMT; ++t)
            for (int i = 0; i < 16; ++i)
              for (int j = 0; j < 16; ++j)
                for (int g4 = 0; g4 < 4; ++g4)
                  for (int e = 0; e < (sidx < 2 ? 8 : 4); ++e) {
                    const int la = i + 16 * g4, lb = j + 16 * g4;
                    const float av = rsm::f16_value(fh[mfcc_frag_piece(t, sidx, hl, la) * 8 + e]);
                    const float bv = rsm::f16_value(hs[mfcc_b_index(lb, sidx) + e]);
                    C3[t][i][j] += av * bv;
                  }
      for (int l = 0; l < 64; ++l) {
        const int j = l & 15;
        if (j >= kFramesPerWave || t0 + j >= n_frames) continue;
        float* orow = out + (row * (int64_t)n_frames + t0 + j) * (int64_t)g_mfcc.n_mfcc;
        for (int t = 0; t < kMfccMT; ++t)
          for (int r = 0; r < 4; ++r) {
            const int k0 = 16 * t + 4 * (l >> 4);
            if (k0 < g_mfcc.n_mfcc) orow[k0 + r] = C3[t][4 * (l >> 4) + r][j] + C3[t][4 * (l >> 4) + r][(j + 8) & 15];
          }
      }
      cur_staged = nxt_staged;
      continue;
    }
    if (epi_mode == EPI400_MEL_NORM) {
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < mt.n_rounds; ++r) {
          const int m = mt.row_mel[r * kMelSlots + c[l].pi];
          const float mu = m >= 0 ? epi.mean[m] : 0.0f, is = m >= 0 ? epi.invstd[m] : 0.0f;
          acc_a[l][r] = (epi_plog(acc_a[l][r] * epi.gain) - mu) * is;
          acc_b[l][r] = (epi_plog(acc_b[l][r] * epi.gain) - mu) * is;
        }
    }
    float* out_row = out + row * (epi_mode == EPI400_MEL_NORM ? epi.out_frames : (int64_t)n_frames) * (int64_t)mb.n_mels;
    if (out_wide) {
      for (int l = 0; l < 64; ++l) store_stage(c[l], mt, acc_a[l], acc_b[l], lds);
      for (int l = 0; l < 64; ++l) store_wide(l, mt, lds, out_row, t0, n_frames);
    } else {
      for (int l = 0; l < 64; ++l) store_direct(c[l], mt, acc_a[l], acc_b[l], out_row, t0, n_frames);
    }
    cur_staged = nxt_staged;
  }
  return 0;
}

extern "C" {
int sim_melspec400(const void* wav, const float* window, const float* tw400, const aamd_mel_bands* bands,
                   float* out, int64_t rows, int64_t length, int64_t row_stride, int n_frames, float scale,
                   int epi_mode, const float* db, float* gmax, int64_t rows_per_group, float power, int want_wide,
                   int hop, int in_i16) {
#define SIM_M400(H) return sim_melspec400_h<H, float>(static_cast<const float*>(wav), window, tw400, bands, out, rows, length, \
                                               row_stride, n_frames, scale, epi_mode, db, gmax, rows_per_group, power, want_wide)
#define SIM_M400_I16(H) return sim_melspec400_h<H, int16_t>(static_cast<const int16_t*>(wav), window, tw400, bands, out, rows, \
                                               length, row_stride, n_frames, scale, epi_mode, db, gmax, rows_per_group, power, want_wide)
#define SIM_M400_ST(H) return sim_melspec400_h<H, aamd::m400::PcmStereo>(static_cast<const aamd::m400::PcmStereo*>(wav), window, tw400, bands, out, rows, \
                                               length, row_stride, n_frames, scale, epi_mode, db, gmax, rows_per_group, power, want_wide)
  if (in_i16 == 2) {            // interleaved 16-bit stereo: rows = 2 x clips, row_stride in sample times
    if (hop == 160) SIM_M400_ST(8);
    if (hop == 200) SIM_M400_ST(10);
    return -4;
  }
  if (in_i16) {
    if (hop == 160) SIM_M400_I16(8);
    if (hop == 200) SIM_M400_I16(10);
    return -4;
  }
  if (hop == 100) SIM_M400(5);
  if (hop == 200) SIM_M400(10);
  if (hop == 160) SIM_M400(8);
#undef SIM_M400
  return -3;
}


// Replay of mfcc_dct_mfma_kernel: the A/B/C fragment maps of v_mfma_f32_16x16x4_f32
// (A[l&15][l>>4], B[l>>4][l&15], C row = 4*(l>>4)+i, col = l&15) applied to the same index math.
int sim_mfcc_dct_mfma(const float* mel, const float* dct, float* out, int64_t n_vec, int n_mels, int n_mfcc,
                      int log_mode, const float* group_max, int64_t vec_per_group, float top_db) {
  const int kc = (n_mels + 15) / 16, NT = (n_mfcc + 15) / 16;
  std::vector<float> frag((size_t)NT * kc * 4 * 64);
  for (size_t i = 0; i < frag.size(); ++i) {
    const int lane = (int)(i & 63), slot = (int)(i >> 6);
    const int j = slot & 3, c = (slot >> 2) % kc, nt = (slot >> 2) / kc;
    frag[i] = dct_frag_value(dct, n_mels, n_mfcc, nt, c, j, lane);
  }
  const bool clampy = log_mode != 1 && top_db >= 0.0f && group_max != nullptr;
  const int64_t n_tiles = (n_vec + kDctFramesPerTile - 1) / kDctFramesPerTile;
  for (int64_t tile = 0; tile < n_tiles; ++tile) {
    std::vector<float> C(NT * 16 * 16, 0.f);   // [nt][row][col]
    for (int c = 0; c < kc; ++c) {
      float y[64][4];
      for (int lane = 0; lane < 64; ++lane) {
        const int f = lane & 15, g = lane >> 4;
        const int64_t v = tile * kDctFramesPerTile + f;
        for (int j = 0; j < 4; ++j) y[lane][j] = 0.f;
        if (v < n_vec && 16 * c + 4 * g < n_mels) {
          float cut = -INFINITY;
          if (clampy) cut = group_max[v / vec_per_group] - top_db;
          for (int j = 0; j < 4; ++j) y[lane][j] = mfcc_log(mel[v * n_mels + 16 * c + 4 * g + j], log_mode, cut);
        }
      }
      for (int nt = 0; nt < NT; ++nt)
        for (int j = 0; j < 4; ++j)
          for (int row = 0; row < 16; ++row)
            for (int col = 0; col < 16; ++col)
              for (int k = 0; k < 4; ++k) {
                const float a = frag[(size_t)(((nt * kc + c) * 4 + j) * 64) + (row + 16 * k)];   // lane = row + 16 k
                const float b = y[col + 16 * k][j];                                               // lane = col + 16 k
                C[(nt * 16 + row) * 16 + col] += a * b;
              }
    }
    for (int lane = 0; lane < 64; ++lane) {
      const int f = lane & 15, g = lane >> 4;
      const int64_t v = tile * kDctFramesPerTile + f;
      if (v >= n_vec) continue;
      for (int nt = 0; nt < NT; ++nt)
        for (int i = 0; i < 4; ++i) {
          const int k0 = 16 * nt + 4 * g;
          if (k0 + i < n_mfcc) out[v * n_mfcc + k0 + i] = C[(nt * 16 + 4 * g + i) * 16 + f];
        }
    }
  }
  return 0;
}

int sim_resample(const float* wav, const float* kern, float* out, int64_t rows, int64_t length, int64_t row_stride,
                 int orig, int new_, int width, int64_t out_len, int qt, int use_lds) {
  ResampleGeom g;
  g.rows = rows; g.length = length; g.row_stride = row_stride; g.out_len = out_len;
  g.orig = orig; g.new_ = new_; g.width = width; g.taps = 2 * width + orig; g.qt = qt; g.use_lds = use_lds;
  const int64_t nq = (out_len + new_ - 1) / new_;
  g.nq_tiles = (int)((nq + qt - 1) / qt);
  std::vector<float> xs((size_t)(qt - 1) * orig + g.taps + 1);
  for (int64_t row = 0; row < rows; ++row)
    for (int tile = 0; tile < g.nq_tiles; ++tile) {
      const int64_t q0 = (int64_t)tile * qt;
      const float* wr = wav + row * row_stride;
      if (use_lds) for (int tid = 0; tid < 256; ++tid) resample_stage(tid, 256, g, wr, q0, xs.data());
      for (int tid = 0; tid < 256; ++tid) resample_compute(tid, 256, g, kern, wr, xs.data(), q0, out + row * out_len);
    }
  return 0;
}

// Replay of resample_sparse_kernel (one thread per output sample over the compacted tap table)
int sim_resample_sparse(const float* wav, const float* hb, const int32_t* lo, float* out, int64_t rows, int64_t length,
                        int64_t row_stride, int orig, int new_, int width, int span, int64_t out_len) {
  for (int64_t row = 0; row < rows; ++row)
    for (int64_t i = 0; i < out_len; ++i)
      out[row * out_len + i] = resample_sparse_one(wav + row * row_stride, length, hb, lo, orig, new_, width, span, i);
  return 0;
}

// Replay of rsm::resample_mfma_kernel: same Geom set-up as aamd_resample_banded_f32, the loader's
// piece copies into a chunk buffer, and the MFMA fragment maps (A: lane m + 16 k, B: lane n + 16 k,
// C: lane n + 16 (m / 4), element m % 4) applied to a_frag / b_base / store_c.
// f16 = 1: replay of rsm::resample_f16_kernel (chunk maximum -> power-of-two scale, samples and taps split into two
// binary16 numbers, hi*hi + hi*lo + lo*hi with the 16x16x32 fragment maps); f16 = 0: rsm::resample_mfma_kernel
// f16 = 2: the same kernel with the 8-byte operand reads (template argument RD = 1: tiles dealt by the parity of q, odd lane
// groups walking the contraction steps in the order b64_sigma -- b64_step / b_base64 / store_c_at), including the alignment and
// buffer-bound claims the pair reads rest on (-4 / -5 when one does not hold); -6 when the geometry is not served by it
int sim_resample_mfma(const float* wav, const float* kern, float* out, int64_t rows, int64_t length,
                      int64_t row_stride, int orig, int new_, int width, int64_t out_len,
                      const int32_t* tap_lo, int tap_span, int vec_ok, int f16) {
  const bool rd64 = f16 == 2;
  using namespace rsm;
  const int n_tiles = (new_ + 15) / 16;
  const int ks = pick_ks(tap_span, orig);
  if (ks == 0) return -2;
  if (rd64 && !b64_ok(ks, orig)) return -6;
  Geom g{};
  g.rows = rows; g.length = length; g.row_stride = row_stride; g.out_len = out_len;
  g.orig = orig; g.new_ = new_; g.width = width; g.taps = 2 * width + orig;
  g.vec_in = vec_ok && (row_stride % 4 == 0);
  g.vec_out = vec_ok && (out_len % 4 == 0) && (new_ % 4 == 0);
  const int64_t nq = (out_len + new_ - 1) / new_;
  const int max_cw = max_compute_waves(ks);
  for (int pt0 = 0; pt0 < n_tiles; pt0 += max_cw) {
    g.pt0 = pt0;
    g.n_pt = n_tiles - pt0 < max_cw ? n_tiles - pt0 : max_cw;
    int max_lo = 0;
    for (int t = 0; t < g.n_pt; ++t) { g.tap_lo[t] = tap_lo[pt0 + t]; if (g.tap_lo[t] > max_lo) max_lo = g.tap_lo[t]; }
    if (!plan_chunk(g, ks, f16 != 0, nq, max_lo, 160 * 1024)) return -3;   // the launcher's own chunk geometry
    const int qg = g.qg;
    const int qc = chunk_q(g);
    g.chunks_per_row = (int)((nq + qc - 1) / qc);
    g.n_chunks = rows * g.chunks_per_row;
    std::vector<float> buf(g.buf_floats);
    for (int64_t cid = 0; cid < g.n_chunks; ++cid) {
      const int64_t row = cid / g.chunks_per_row;
      const int64_t qc0 = (cid - row * g.chunks_per_row) * qc;
      const float* wrow = wav + row * row_stride;
      const int64_t a0 = chunk_a0(g, qc0);
      for (int j = 0; j < g.buf_floats / 4; ++j) {
        const F4 v = load_piece(g, wrow, a0, j);
        buf[4 * j] = v.x; buf[4 * j + 1] = v.y; buf[4 * j + 2] = v.z; buf[4 * j + 3] = v.w;
      }
      const int shift = (int)((qc0 * orig - width) - a0);
      float sc = 1.0f, inv = 1.0f;
      std::vector<uint32_t> pk;
      if (f16) {
        uint32_t mxb = 0;
        for (float v : buf) { uint32_t u; std::memcpy(&u, &v, 4); u &= 0x7fffffffu; if (u > mxb) mxb = u; }
        chunk_scale(mxb, sc, inv);
        pk.resize(buf.size());
        for (size_t i = 0; i < buf.size(); ++i) pk[i] = pack_hl_cut(buf[i] * sc);      // (the samples' 4-operation split; the taps: pack_hl)
      }
      for (int w = 0; w < g.n_pt * qg * g.rounds; ++w) {      // (rounds: the same wave, its next q-group)
        const int pt_l = w % g.n_pt, qgi = w / g.n_pt, pt = pt0 + pt_l, lo = g.tap_lo[pt_l];
        for (int half = 0; half < 2; ++half) {
          const int qt = 2 * qgi + half;
          float Cm[16][16] = {};
          if (rd64) {
            // tile `half` holds q = 32 qgi + 2 n + half; the kernel calls the 8-byte aligned one tile A (half = (lo + shift) & 1)
            const bool aligned = half == ((lo + shift) & 1);
            for (int sidx = 0; sidx < ks / 8; ++sidx)
              for (int k4 = 0; k4 < 4; ++k4) {
                const int sig = b64_step(ks, sidx, k4);
                for (int n = 0; n < 16; ++n) {
                  const int b0 = b_base64(g, qgi, half, lo, ks, shift, n + 16 * k4) + 8 * sig;
                  if (((b0 & 1) == 0) != aligned) return -4;                 // every lane of a tile has the tile's parity
                  // the pair reads: [b0, b0 + 8) for the aligned tile, [b0 - 1, b0 + 9) for the other one
                  if (b0 - (aligned ? 0 : 1) < 0 || b0 + (aligned ? 7 : 8) >= g.buf_floats) return -5;
                }
              }
            for (int sidx = 0; sidx < ks / 8; ++sidx)
              for (int term = 0; term < 3; ++term)
                for (int m = 0; m < 16; ++m)
                  for (int n = 0; n < 16; ++n)
                    for (int k4 = 0; k4 < 4; ++k4)
                      for (int e = 0; e < 8; ++e) {
                        const int sig = b64_step(ks, sidx, k4);
                        uint32_t ahi, alo;
                        a_pack16_at(g, kern, pt, lo + ks * k4 + 8 * sig + 2 * (e >> 1), m + 16 * k4, ahi, alo);
                        const uint16_t a_h = (uint16_t)(ahi >> (16 * (e & 1))), a_l = (uint16_t)(alo >> (16 * (e & 1)));
                        const int bidx = b_base64(g, qgi, half, lo, ks, shift, n + 16 * k4) + 8 * sig + e;
                        const uint16_t b_h = (uint16_t)(pk[bidx] & 0xffffu), b_l = (uint16_t)(pk[bidx] >> 16);
                        const float a = f16_value(term == 0 ? a_l : a_h), b = f16_value(term == 1 ? b_l : b_h);
                        Cm[m][n] += a * b;
                      }
            for (int lane = 0; lane < 64; ++lane) {
              const int n = lane & 15, gq = lane >> 4;
              store_c_at(g, out + row * out_len, qc0 + 32 * qgi + 2 * n + half, pt, lane, Cm[4 * gq][n] * inv,
                         Cm[4 * gq + 1][n] * inv, Cm[4 * gq + 2][n] * inv, Cm[4 * gq + 3][n] * inv);
            }
            continue;
          }
          if (f16) {
            // lane (row m / column n, group k4): A elements 8 s + e of its group, B the 8 consecutive samples
            for (int sidx = 0; sidx < ks / 8; ++sidx)
              for (int term = 0; term < 3; ++term)          // lo*hi, hi*lo, hi*hi
                for (int m = 0; m < 16; ++m)
                  for (int n = 0; n < 16; ++n)
                    for (int k4 = 0; k4 < 4; ++k4)
                      for (int e = 0; e < 8; ++e) {
                        uint32_t ahi, alo;
                        a_pack16(g, kern, pt, lo, ks, sidx, e >> 1, m + 16 * k4, ahi, alo);
                        const uint16_t a_h = (uint16_t)(ahi >> (16 * (e & 1))), a_l = (uint16_t)(alo >> (16 * (e & 1)));
                        const int bidx = b_base(g, qt, lo, ks, shift, n + 16 * k4) + 8 * sidx + e;
                        if (bidx < 0 || bidx >= g.buf_floats) return -3;
                        const uint16_t b_h = (uint16_t)(pk[bidx] & 0xffffu), b_l = (uint16_t)(pk[bidx] >> 16);
                        const float a = f16_value(term == 0 ? a_l : a_h), b = f16_value(term == 1 ? b_l : b_h);
                        Cm[m][n] += a * b;
                      }
            for (int lane = 0; lane < 64; ++lane) {
              const int n = lane & 15, gq = lane >> 4;
              store_c(g, out + row * out_len, qc0, qt, pt, lane, Cm[4 * gq][n] * inv, Cm[4 * gq + 1][n] * inv,
                      Cm[4 * gq + 2][n] * inv, Cm[4 * gq + 3][n] * inv);
            }
            continue;
          }
          for (int kk = 0; kk < ks; ++kk)
            for (int m = 0; m < 16; ++m)
              for (int n = 0; n < 16; ++n)
                for (int k = 0; k < 4; ++k) {
                  const int bidx = b_base(g, qt, lo, ks, shift, n + 16 * k) + kk;
                  if (bidx < 0 || bidx >= g.buf_floats) return -3;   // read outside the chunk buffer
                  Cm[m][n] += a_frag(g, kern, pt, lo, ks, kk, m + 16 * k) * buf[bidx];
                }
          for (int lane = 0; lane < 64; ++lane) {
            const int n = lane & 15, gq = lane >> 4;
            store_c(g, out + row * out_len, qc0, qt, pt, lane, Cm[4 * gq][n], Cm[4 * gq + 1][n], Cm[4 * gq + 2][n],
                    Cm[4 * gq + 3][n]);
          }
        }
      }
    }
  }
  return 0;
}

}  // extern C (pause)
template <int D>
static int sim_lfilter_d(const float* x, const float* a, const float* b, float* y, int64_t n_seq, int channels,
                         int64_t length, int n_order, int n_rows, int n_stages, int clamp) {
  using L = LfLds<D>;
  std::vector<float> blkv((size_t)L::blk_floats, 0.f);
  std::vector<double> tabv((size_t)L::total + (size_t)n_stages * L::stage_doubles, 0.0);
  float* blk = blkv.data();
  double* tab = tabv.data();
  double* stage_store = tab + L::total;
  std::vector<LfThread<D>> th(kLfThreads);
  for (int64_t seq = 0; seq < n_seq; ++seq) {
    const int ch = (int)(seq % channels);
    const int crow = (n_rows == 1) ? 0 : ch;
    for (int st = 0; st < n_stages; ++st) {
      const int64_t coff = ((int64_t)st * n_rows + crow) * n_order;
      for (int tid = 0; tid < kLfThreads; ++tid) lf_tables_coeffs<D>(tid, a + coff, b + coff, n_order, tab);
      for (int e = 0; e < 2 * D; ++e) tab[L::cx + e] = 0.0;
      for (int tid = 0; tid < kLfThreads; ++tid) lf_tables_response<D>(tid, tab);
      for (int k = 1; k < kLfScanSteps; ++k)
        for (int tid = 0; tid < kLfThreads; ++tid) lf_tables_square<D>(tid, k, tab);
      for (int tid = 0; tid < kLfThreads; ++tid) lf_tables_small<D>(tid, tab);
      lf_tables_neff<D>(tab);
      for (int i = 0; i < L::stage_doubles; ++i) stage_store[st * L::stage_doubles + i] = tab[L::H + i];
    }
    const float* xs = x + seq * length;
    float* ys = y + seq * length;
    for (int64_t n0 = 0; n0 < length; n0 += kLfBlock) {
      for (int i = 0; i < kLfBlock; ++i) {
        const int64_t n = n0 + i;
        blk[(i / kLfChunk) * kLfChunkStride + (i % kLfChunk)] = (n < length) ? xs[n] : 0.f;
      }
      for (int st = 0; st < n_stages; ++st) {
        for (int i = 0; i < L::stage_doubles; ++i) tab[L::H + i] = stage_store[st * L::stage_doubles + i];
        for (int tid = 0; tid < kLfThreads; ++tid) lf_chunk_pass<D>(tid, blk, tab, th[tid]);
        lf_save_input_carry<D>(blk, tab);
        bool src_is_a = true;
        const int n_eff = (int)tab[L::neff];
        for (int k = 0; k < n_eff; ++k) {
          for (int tid = 0; tid < kLfThreads; ++tid) lf_scan_step<D>(tid, k, tab, th[tid], src_is_a);
          src_is_a = !src_is_a;
        }
        for (int tid = 0; tid < kLfThreads; ++tid) lf_correct_store<D>(tid, blk, tab, th[tid], src_is_a, clamp == 1 || (clamp == 2 && st == n_stages - 1));
        lf_save_output_carry<D>(tab, src_is_a);
        for (int i = 0; i < 2 * D; ++i) stage_store[st * L::stage_doubles + (L::cx - L::H) + i] = tab[L::cx + i];
      }
      for (int i = 0; i < kLfBlock; ++i) {
        const int64_t n = n0 + i;
        if (n < length) ys[n] = blk[(i / kLfChunk) * kLfChunkStride + (i % kLfChunk)];
      }
    }
  }
  return 0;
}

// Replay of lfw::lfilter_wave_kernel: W waves x 64 lanes per sequence, DPP moves as array reads (lfw::scan_src is the
// lane map of both), phases cut where the kernel has its workgroup barrier.
extern "C" int sim_lfilter_wave(const float* x, const float* a, const float* b, float* y, int64_t batch, int channels,
                                int64_t length, int n_order, int n_rows, int n_stages, int clamp, int W) {
  using namespace lfw;
  if (n_order > 3 || n_stages > kMaxCascade || W > kMaxWaves) return -1;
  const int64_t n_seq = batch * channels;
  const int64_t block_len = (int64_t)W * kWaveBlock;
  std::vector<float> tiles((size_t)W * kTile), tabs((size_t)n_stages * kTabFloats), xch(xch_floats(W, n_stages));
  std::vector<std::vector<float>> v(W * 64, std::vector<float>(kCh)), z(W * 64, std::vector<float>(kCh));
  std::vector<float> s0(W * 64), s1(W * 64), t0(64), t1(64), hin0(W), hin1(W), e0(W), e1(W);
  auto arr = [](std::vector<float>& r) -> float(&)[kCh] { return *reinterpret_cast<float(*)[kCh]>(r.data()); };
  // one DPP step over 64 lanes: lane l adds M_l . (value of lane src(l)), zeros without a source
  auto dpp_step = [&](float* p0, float* p1, int step, const float* tab, bool fold) {
    for (int lane = 0; lane < 64; ++lane) {
      const int src = scan_src(step, lane);
      t0[lane] = src >= 0 ? p0[src] : 0.0f;
      t1[lane] = src >= 0 ? p1[src] : 0.0f;
    }
    for (int lane = 0; lane < 64; ++lane)
      mat_acc(scan_mat(tab, step, lane), t0[lane], t1[lane], p0[lane], p1[lane]);
  };
  for (int64_t seq = 0; seq < n_seq; ++seq) {
    const int crow = (n_rows == 1) ? 0 : (int)(seq % channels);
    for (int st = 0; st < n_stages; ++st) {
      const int64_t coff = ((int64_t)st * n_rows + crow) * n_order;
      build_stage(a + coff, b + coff, n_order, tabs.data() + st * kTabFloats);
    }
    std::fill(xch.begin(), xch.end(), 0.0f);
    const float* xs = x + seq * length;
    float* ys = y + seq * length;
    int parity = 0, sbuf = 0;
    for (int64_t n0 = 0; n0 < length; n0 += block_len, parity ^= 1) {
      for (int w = 0; w < W; ++w) {   // the block's samples and the 64 before them, through the tile
        float* tile = tiles.data() + (size_t)w * kTile;
        const int64_t nw = n0 + (int64_t)w * kWaveBlock;
        for (int sidx = -64; sidx < kWaveBlock; ++sidx)
          tile[tile_idx(sidx)] = (nw + sidx >= 0 && nw + sidx < length) ? xs[nw + sidx] : 0.0f;
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < kCh; ++j) v[w * 64 + lane][j] = tile[tile_idx(kCh * lane + j)];
        hin0[w] = nw > 0 ? tile[tile_idx(-1)] : 0.0f;
        hin1[w] = nw > 0 ? tile[tile_idx(-2)] : 0.0f;
      }
      for (int st = 0; st < n_stages; ++st, sbuf ^= 1) {
        const float* tab = tabs.data() + st * kTabFloats;
        for (int w = 0; w < W; ++w) {
          for (int lane = 0; lane < 64; ++lane) {
            float hu0 = hin0[w], hu1 = hin1[w];
            if (lane > 0) { hu0 = v[w * 64 + lane - 1][kCh - 1]; hu1 = v[w * 64 + lane - 1][kCh - 2]; }
            z[w * 64 + lane] = v[w * 64 + lane];   // the kernel filters in place; keep v for the neighbour's history
            chunk_pass(tab, arr(z[w * 64 + lane]), hu0, hu1, s0[w * 64 + lane], s1[w * 64 + lane]);
          }
          for (int k = 0; k < 6; ++k) dpp_step(&s0[w * 64], &s1[w * 64], k, tab, false);
          xch[xch_S(W, sbuf, w)] = s0[w * 64 + 63];
          xch[xch_S(W, sbuf, w) + 1] = s1[w * 64 + 63];
        }
        // the barrier
        const int cin = xch_carry(W, n_stages, parity, st);
        for (int w = 0; w < W; ++w) {
          float f0[64], f1[64];
          for (int lane = 0; lane < 64; ++lane) {
            const bool on = (lane & 15) < W;
            f0[lane] = on ? xch[xch_S(W, sbuf, lane & 15)] : 0.0f;
            f1[lane] = on ? xch[xch_S(W, sbuf, lane & 15) + 1] : 0.0f;
          }
          for (int lane = 0; lane < 64; ++lane) {   // contribution of S_(lane) to wave w, then a plain prefix sum over the row
            float g0, g1;
            mat_apply(tab + kTabPowW + 4 * fold_entry(w, lane), f0[lane], f1[lane], g0, g1);
            f0[lane] = g0;
            f1[lane] = g1;
          }
          for (int k = 0; k < 4; ++k) {
            for (int lane = 0; lane < 64; ++lane) {
              const int src = scan_src(k, lane);
              t0[lane] = src >= 0 ? f0[src] : 0.0f;
              t1[lane] = src >= 0 ? f1[src] : 0.0f;
            }
            for (int lane = 0; lane < 64; ++lane) { f0[lane] += t0[lane]; f1[lane] += t1[lane]; }
          }
          const int from = w > 0 ? w - 1 : 0;
          fold_finish(tab, w, f0[from], f1[from], xch[cin], xch[cin + 1], e0[w], e1[w]);
          for (int lane = 0; lane < 64; ++lane)
            mat_acc(tab + kTabPow + 4 * lane, e0[w], e1[w], s0[w * 64 + lane], s1[w * 64 + lane]);
          for (int lane = 0; lane < 64; ++lane) {
            const float tt0 = lane ? s0[w * 64 + lane - 1] : e0[w], tt1 = lane ? s1[w * 64 + lane - 1] : e1[w];
            correct_clamp(tab, tt0, tt1, stage_clamp(clamp, st, n_stages), arr(z[w * 64 + lane]));
          }
        }
        for (int w = 0; w < W; ++w) {
          if (w == W - 1) {
            const int cout = xch_carry(W, n_stages, parity ^ 1, st);
            xch[cout] = s0[w * 64 + 63]; xch[cout + 1] = s1[w * 64 + 63];
          }
          hin0[w] = clamp1(e0[w], stage_clamp(clamp, st, n_stages));
          hin1[w] = clamp1(e1[w], stage_clamp(clamp, st, n_stages));
          for (int lane = 0; lane < 64; ++lane) v[w * 64 + lane] = z[w * 64 + lane];
        }
      }
      for (int w = 0; w < W; ++w) {
        float* tile = tiles.data() + (size_t)w * kTile;
        const int64_t nw = n0 + (int64_t)w * kWaveBlock;
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < kCh; ++j) tile[tile_idx(kCh * lane + j)] = v[w * 64 + lane][j];
        for (int sidx = 0; sidx < kWaveBlock; ++sidx)
          if (nw + sidx < length) ys[nw + sidx] = tile[tile_idx(sidx)];
      }
    }
  }
  return 0;
}

extern "C" {
int sim_lfilter(const float* x, const float* a, const float* b, float* y, int64_t batch, int channels, int64_t length,
                int n_order, int n_rows, int n_stages, int clamp) {
  const int d = n_order - 1;
  const int64_t n_seq = batch * channels;
#define SIM_LF(D) return sim_lfilter_d<D>(x, a, b, y, n_seq, channels, length, n_order, n_rows, n_stages, clamp)
  if (d <= 1) SIM_LF(1);
  if (d <= 2) SIM_LF(2);
  if (d <= 3) SIM_LF(3);
  if (d <= 4) SIM_LF(4);
  if (d <= 6) SIM_LF(6);
  if (d <= 8) SIM_LF(8);
  if (d <= 12) SIM_LF(12);
  if (d <= 16) SIM_LF(16);
  return -1;
}

// The tile hand-out of the n_fft = 400 kernel with tail pools (m400::pool_tile and the claim logic of melspec400_kernel):
// nb workgroups of `waves` waves, each wave a state machine that is advanced one atomic operation at a time in a seeded random
// order -- LDS queue claim, pool ticket request, pool ticket read (with the reset by the launch's last ticket).  `launches`
// launches in a row on the same counters.  visits[t] counts how often tile t was run.  Returns 0, or: -1 a counter was not
// back at zero after a launch, -2 a wave drew a ticket after the reset, -3 the launch did not terminate.
// P < 0: the launcher's own share (pool_share); P = 0: no pools.  xcd_remap as in the kernel (lb from blockIdx).
int sim_mel400_pool(int nb, int tiles_per_block, int64_t n_tiles, int P, int waves, int launches, uint32_t seed, int32_t* visits) {
  using namespace m400;
  if (P < 0) P = pool_share(tiles_per_block);
  if (P > tiles_per_block) P = tiles_per_block;
  const bool pooled = P > 0;
  const int np = pool_count(nb);
  std::vector<uint32_t> ctr((size_t)np, 0u);
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  struct Wave { int wg, state; unsigned cur_idx, nxt_idx, ticket; bool cur_ok, nxt_ok, nxt_pool, first; };
  for (int l = 0; l < launches; ++l) {
    std::vector<unsigned> queue((size_t)nb, (unsigned)waves);
    std::vector<Wave> ws;
    for (int b = 0; b < nb; ++b)
      for (int w = 0; w < waves; ++w) ws.push_back(Wave{b, 0, (unsigned)w, 0u, 0u, false, false, false, true});
    std::vector<char> reset_done((size_t)np, 0);
    size_t alive = ws.size();
    for (int64_t step = 0; alive > 0; ++step) {
      if (step > (int64_t)40 * (n_tiles + (int64_t)nb * waves) + 1000) return -3;
      Wave& w = ws[rnd() % ws.size()];
      if (w.state == 9) continue;
      const int lb = ((nb & 7) == 0) ? (w.wg & 7) * (nb >> 3) + (w.wg >> 3) : w.wg;
      const unsigned blk_first = (unsigned)lb * (unsigned)tiles_per_block;
      unsigned blk_count = 0;
      if (blk_first < (unsigned)n_tiles) {
        blk_count = (unsigned)n_tiles - blk_first;
        if (blk_count > (unsigned)tiles_per_block) blk_count = (unsigned)tiles_per_block;
      }
      unsigned n_static = blk_count;
      if (pooled && n_static > (unsigned)(tiles_per_block - P)) n_static = (unsigned)(tiles_per_block - P);
      const int pid = lb % np, mem = pool_members(nb, np, pid);
      auto request = [&]() {
        if (reset_done[pid]) return false;                      // a ticket drawn behind the reset: the protocol is broken
        w.ticket = ctr[pid]++;
        return true;
      };
      auto take = [&](unsigned& idx, bool& ok, bool& again) {   // pool_take of the kernel, one draw
        again = false;
        if (w.ticket + 1u == (unsigned)(mem * (P + waves))) { ctr[pid] = 0u; reset_done[pid] = 1; }
        bool ex;
        const unsigned t = pool_tile(np, P, tiles_per_block, (unsigned)n_tiles, pid, mem, w.ticket, ex);
        ok = t != kNoTile;
        idx = t - blk_first;
        if (!ok && !ex) again = true;
      };
      switch (w.state) {
        case 0:      // before the tile loop
          w.cur_ok = w.cur_idx < (pooled ? n_static : blk_count);
          if (pooled && !w.cur_ok) { if (!request()) return -2; w.state = 1; } else w.state = 2;
          break;
        case 1: {    // pool_take for the first tile
          bool again;
          take(w.cur_idx, w.cur_ok, again);
          if (again) { if (!request()) return -2; } else w.state = 2;
          break;
        }
        case 2:      // loop head: claim the next tile from the LDS queue
          if (!w.cur_ok) { w.state = 9; --alive; break; }
          w.nxt_idx = queue[w.wg]++;
          w.nxt_ok = w.nxt_idx < (pooled ? n_static : blk_count);
          w.nxt_pool = pooled && !w.nxt_ok;
          w.state = w.nxt_pool ? 3 : 5;
          break;
        case 3:      // pool_request behind the gather
          if (!request()) return -2;
          w.state = 4;
          break;
        case 4: {    // pool_take in front of the next tile's DMA
          bool again;
          take(w.nxt_idx, w.nxt_ok, again);
          if (again) { if (!request()) return -2; } else w.state = 5;
          break;
        }
        case 5:      // the rest of the tile: it is computed and stored
          ++visits[blk_first + w.cur_idx];
          w.cur_idx = w.nxt_idx; w.cur_ok = w.nxt_ok;
          w.state = 2;
          break;
      }
    }
    for (int p_ = 0; p_ < np; ++p_)
      if (ctr[p_] != 0u) return -1;
  }
  return 0;
}

// LDS bytes of the generic STFT / Kaldi kernels: layout 0 = full (twiddle table in LDS), 1 = long windows (stft_generic.h,
// gen_lds_floats_long), 2 = the Kaldi front-end's long layout
int64_t sim_gen_lds_bytes(int n_fft, int pb, int layout) {
  if (layout == 0) return (int64_t)(gen_lds_floats(n_fft, n_fft / 2 + 1, pb) * sizeof(float));
  if (layout == 1) return (int64_t)(gen_lds_floats_long(n_fft, pb) * sizeof(float));
  return (int64_t)(kgen::lds_floats_long(n_fft, pb) * sizeof(float));
}

// The launcher's chunk geometry of the banded resampler for one set of phase tiles (rsm::plan_chunk), for property tests:
// out = {qg, rounds, n_loaders, buf_floats, waves, chunk_q}; returns 0 when even one q-group does not fit.
int sim_rsm_plan(int orig, int new_, int width, int tap_span, int max_lo, int64_t nq, int f16, int64_t lds_cap, int* out) {
  using namespace rsm;
  const int ks = pick_ks(tap_span, orig);
  if (ks == 0) return -2;
  Geom g{};
  g.orig = orig; g.new_ = new_; g.width = width; g.taps = 2 * width + orig;
  const int n_tiles = (new_ + 15) / 16, max_cw = max_compute_waves(ks);
  g.pt0 = 0;
  g.n_pt = n_tiles < max_cw ? n_tiles : max_cw;
  const bool ok = plan_chunk(g, ks, f16 != 0, nq, max_lo, (size_t)lds_cap);
  out[0] = g.qg; out[1] = g.rounds; out[2] = g.n_loaders; out[3] = g.buf_floats;
  out[4] = g.n_pt * g.qg + g.n_loaders; out[5] = chunk_q(g); out[6] = ks; out[7] = loader_pieces_per_lane(ks);
  return ok ? 1 : 0;
}

// Bank model of the 8-byte operand reads of rsm::resample_f16_kernel<KS, ., 1> (MI355X_MICROARCH.md, LDS table: a ds_read_b64 is
// served in two groups of 32 lanes over 64 dword banks): the number of loop steps in which a read of tile A or tile B is NOT
// conflict-free, for one phase tile (tap_lo) and chunk phase (shift).  The addresses are the kernel's own (b_base64 + 8 b64_step).
int sim_rsm_b64_conflicted_steps(int ks, int orig, int tap_lo, int shift) {
  using namespace rsm;
  if (!b64_ok(ks, orig)) return -1;
  Geom g{};
  g.orig = orig;
  const int hA = (tap_lo + shift) & 1;
  int bad_steps = 0;
  for (int s = 0; s < ks / 8; ++s) {
    bool bad = false;
    for (int tile = 0; tile < 2; ++tile)                  // 0: the aligned tile, 1: the other one, read from one dword earlier
      for (int half = 0; half < 2; ++half)                // lanes 0 .. 31 / 32 .. 63
        for (int pair = 0; pair < 4; ++pair) {
          int64_t first[64];
          for (int b = 0; b < 64; ++b) first[b] = -1;
          for (int lane = 32 * half; lane < 32 * half + 32; ++lane) {
            const int h = tile == 0 ? hA : hA ^ 1;
            const int64_t a = (int64_t)b_base64(g, 0, h, tap_lo, ks, shift, lane) + 8 * b64_step(ks, s, lane >> 4) + (tile ? 1 : 0) + 2 * pair;
            if (a & 1) return -2;                         // not 8-byte aligned: the layout's own claim
            for (int w = 0; w < 2; ++w) {
              const int bank = (int)(((a + w) % 64 + 64) % 64);
              if (first[bank] >= 0 && first[bank] != a + w) bad = true;
              first[bank] = a + w;
            }
          }
        }
    bad_steps += bad;
  }
  return bad_steps;
}

// Replay of the overlap-save path (fco::spectrum_kernel + fco::overlap_save_kernel): the same
// launcher logic as aamd_fftconvolve_f32, 1024 "threads" per phase, phases separated where the
// kernel has barriers.  Twiddles as the device twiddle_kernel computes them (fp64 -> fp32).
int sim_fftconv_os(const float* x, const float* y, float* out, int64_t rows, int64_t n_x_rows, int64_t n_y_rows,
                   int64_t nx, int64_t ny, const int64_t* x_row_of, const int64_t* y_row_of, int64_t start,
                   int64_t out_len, int cu_count) {   // cu_count <= 0: never the delay-line plan; returns 1 when it ran
  using namespace fco;
  const bool swap = ny > nx;
  const float* xa = swap ? y : x; const float* ya = swap ? x : y;
  const int64_t nxa = swap ? ny : nx, nya = swap ? nx : ny;
  const int64_t tap_rows = swap ? n_x_rows : n_y_rows;
  const int64_t* xmap = swap ? y_row_of : x_row_of; const int64_t* ymap = swap ? x_row_of : y_row_of;
  Geom g{};
  g.rows = rows; g.nx = nxa; g.ny = nya; g.start = start; g.out_len = out_len;
  plan(nya, out_len, g);
  std::vector<C32> tw(kN), lds(kLdsComplex), H((size_t)tap_rows * g.n_part * kN);
  for (int m = 0; m < kN; ++m) {
    const double a = -2.0 * M_PI * (double)m / (double)kN;
    tw[m] = C32{(float)std::cos(a), (float)std::sin(a)};
  }
  C32* tl = lds.data() + kLdsData;
  for (int t = 0; t < kThreads; ++t) twiddle_tables(t, tw.data(), tl);
  auto fwd = [&]() {
    for (int t = 0; t < kThreads; ++t) pass16<16384, false>(t, lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) pass16<1024, false>(t, lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) pass16<64, false>(t, lds.data(), tl);
  };
  for (int64_t b = 0; b < tap_rows * g.n_part; ++b) {
    const int64_t yrow = b / g.n_part; const int p = (int)(b - yrow * g.n_part);
    for (int t = 0; t < kThreads; ++t) load_taps(t, g, ya + yrow * g.ny, p, lds.data());
    fwd();
    for (int t = 0; t < kThreads; ++t) middle_spectrum(t, lds.data(), H.data() + b * kN, 1.0f / (float)kN);
  }
  std::vector<std::array<C32, 16>> v(kThreads), acc(kThreads);
  auto arr = [](std::array<C32, 16>& a) -> C32 (&)[16] { return *reinterpret_cast<C32 (*)[16]>(a.data()); };
  FdlGeom f{};
  f.rows = rows; f.nx = nxa; f.ny = nya; f.start = start; f.out_len = out_len;
  if (cu_count > 0 && plan_fdl(rows, nya, out_len, cu_count, f)) {
    // the delay-line plan (fco::overlap_save_fdl_kernel): spectra of kHop-tap partitions, a ring of NP spectra per item
    const int NP = f.n_part;
    Geom gs = g;
    gs.n_part = NP; gs.part_taps = kHop;
    std::vector<C32> Hf((size_t)tap_rows * NP * kN), ring((size_t)(NP > 2 ? NP - 2 : 1) * kN);
    std::vector<std::array<C32, 16>> zprev(kThreads);
    for (int64_t b = 0; b < tap_rows * NP; ++b) {
      const int64_t yrow = b / NP; const int p = (int)(b - yrow * NP);
      for (int t = 0; t < kThreads; ++t) load_taps(t, gs, ya + yrow * gs.ny, p, lds.data());
      fwd();
      for (int t = 0; t < kThreads; ++t) middle_spectrum(t, lds.data(), Hf.data() + b * kN, 1.0f / (float)kN);
    }
    for (int64_t item = 0; item < rows * f.segs; ++item) {
      const int64_t row = item / f.segs, j_lo = (item - row * f.segs) * f.seg_blocks;
      const int64_t j_hi = j_lo + f.seg_blocks < f.n_blocks ? j_lo + f.seg_blocks : f.n_blocks;
      const int64_t hn = (j_hi - j_lo + 1) / 2;
      const int64_t rx = xmap ? xmap[row] : row, ry = ymap ? ymap[row] : row;
      const float* xr = xa + rx * f.nx;
      const C32* Hr = Hf.data() + ry * NP * (int64_t)kN;
      int wslot = 0;
      for (int64_t s = -(NP - 1); s < hn; ++s) {
        const bool produce = s >= 0;
        for (int t = 0; t < kThreads; ++t) {
          fdl_load(t, f, xr, j_lo + s, j_lo + hn + s, arr(v[t]));
          first_pass_from_regs(t, arr(v[t]), lds.data(), tl);
        }
        for (int t = 0; t < kThreads; ++t) pass16<1024, false>(t, lds.data(), tl);
        for (int t = 0; t < kThreads; ++t) pass16<64, false>(t, lds.data(), tl);
        for (int t = 0; t < kThreads; ++t) {
          if (NP == 2) middle_fdl<2>(t, lds.data(), Hr, ring.data(), wslot, produce, arr(zprev[t]));
          else if (NP == 3) middle_fdl<3>(t, lds.data(), Hr, ring.data(), wslot, produce, arr(zprev[t]));
          else middle_fdl<4>(t, lds.data(), Hr, ring.data(), wslot, produce, arr(zprev[t]));
        }
        if (produce) {
          for (int t = 0; t < kThreads; ++t) pass16<64, true>(t, lds.data(), tl);
          for (int t = 0; t < kThreads; ++t) pass16<1024, true>(t, lds.data(), tl);
          for (int t = 0; t < kThreads; ++t) {
            last_pass_to_regs(t, lds.data(), tl, arr(v[t]));
            fdl_store(t, f, arr(v[t]), j_lo + s, j_lo + hn + s, j_hi, out + row * out_len);
          }
        }
        if (NP > 2) wslot = wslot + 1 == NP - 2 ? 0 : wslot + 1;
      }
    }
    return 1;
  }
  for (int64_t item = 0; item < rows * g.n_pairs; ++item) {
    const int64_t row = item / g.n_pairs, j0 = 2 * (item - row * g.n_pairs);
    const int64_t rx = xmap ? xmap[row] : row, ry = ymap ? ymap[row] : row;
    const float* xr = xa + rx * g.nx;
    for (int t = 0; t < kThreads; ++t) {
      for (int k = 0; k < 16; ++k) acc[t][k] = C32{0.0f, 0.0f};
      load_pair_regs(t, g, xr, 0, j0, arr(v[t]));
    }
    for (int p = 0; p < g.n_part; ++p) {
      for (int t = 0; t < kThreads; ++t) first_pass_from_regs(t, arr(v[t]), lds.data(), tl);
      for (int t = 0; t < kThreads; ++t) pass16<1024, false>(t, lds.data(), tl);
      for (int t = 0; t < kThreads; ++t) pass16<64, false>(t, lds.data(), tl);
      for (int t = 0; t < kThreads; ++t) {
        if (p + 1 < g.n_part) load_pair_regs(t, g, xr, p + 1, j0, arr(v[t]));
        middle_accumulate(t, lds.data(), H.data() + (ry * g.n_part + p) * (int64_t)kN, arr(acc[t]));
      }
    }
    for (int t = 0; t < kThreads; ++t) middle_finish(t, arr(acc[t]), lds.data());
    for (int t = 0; t < kThreads; ++t) pass16<64, true>(t, lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) pass16<1024, true>(t, lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) {
      last_pass_to_regs(t, lds.data(), tl, arr(v[t]));
      store_pair_regs(t, g, arr(v[t]), j0, out + row * out_len);
    }
  }
  return 0;
}

// Replay of the real-block delay line (fdr::spectrum_kernel + fdr::delay_line_kernel<NP>): the launcher's logic of
// aamd_fftconvolve_f32 (plan 3), 1024 "threads" per phase, phases separated where the kernel has barriers or wave hand-offs; the
// neighbour exchange of the radix-2 stage is the DPP swap of the device.  Returns 1 when the plan serves the shape, else 0.
int sim_fftconv_fdr(const float* x, const float* y, float* out, int64_t rows, int64_t n_x_rows, int64_t n_y_rows,
                    int64_t nx, int64_t ny, const int64_t* x_row_of, const int64_t* y_row_of, int64_t start,
                    int64_t out_len, int cu_count) {
  using namespace fdr;
  const bool swap = ny > nx;
  const float* xa = swap ? y : x; const float* ya = swap ? x : y;
  const int64_t nxa = swap ? ny : nx, nya = swap ? nx : ny;
  const int64_t tap_rows = swap ? n_x_rows : n_y_rows;
  const int64_t* xmap = swap ? y_row_of : x_row_of; const int64_t* ymap = swap ? x_row_of : y_row_of;
  Geom g{};
  g.rows = rows; g.nx = nxa; g.ny = nya; g.start = start; g.out_len = out_len;
  if (!plan(rows, nya, out_len, cu_count, g)) return 0;
  const int NP = g.n_part;
  std::vector<C32> tw(fco::kN), lds(kLdsComplex), H((size_t)tap_rows * NP * kHPerPart);
  for (int m = 0; m < fco::kN; ++m) {
    const double a = -2.0 * M_PI * (double)m / (double)fco::kN;
    tw[m] = C32{(float)std::cos(a), (float)std::sin(a)};
  }
  C32* tl = lds.data() + kLdsData;
  for (int t = 0; t < kThreads; ++t) twiddle_tables(t, tw.data(), tl);
  std::vector<MidConst> mc(kThreads);
  for (int t = 0; t < kThreads; ++t) mid_init(t, tw.data(), mc[t]);
  using A8 = std::array<C32, 8>;
  auto arr = [](A8& a) -> C32 (&)[8] { return *reinterpret_cast<C32 (*)[8]>(a.data()); };
  std::vector<A8> v(kThreads), o(kThreads), z0(kThreads), z1(kThreads), z2(kThreads), z3(kThreads), acc(kThreads);
  auto forward = [&]() {
    for (int t = 0; t < kThreads; ++t) first_pass_from_regs(t, arr(v[t]), lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) pass_m128<false>(t, lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) pass_m16<false>(t, lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) pass_m2_fwd_a(t, lds.data(), tl, arr(o[t]));
    for (int t = 0; t < kThreads; ++t) pass_m2_fwd_b(t, arr(o[t]), arr(o[t ^ 1]), lds.data());
  };
  auto inverse = [&]() {
    for (int t = 0; t < kThreads; ++t) pass_m2_inv_a(t, lds.data(), arr(o[t]));
    for (int t = 0; t < kThreads; ++t) pass_m2_inv_b(t, arr(o[t]), arr(o[t ^ 1]), lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) pass_m16<true>(t, lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) pass_m128<true>(t, lds.data(), tl);
    for (int t = 0; t < kThreads; ++t) last_pass_to_regs(t, lds.data(), tl, arr(v[t]));
  };
  for (int64_t b = 0; b < tap_rows * NP; ++b) {
    const int64_t yrow = b / NP; const int p = (int)(b - yrow * NP);
    for (int t = 0; t < kThreads; ++t) load_taps(t, nya, ya + yrow * nya, p, arr(v[t]));
    forward();
    for (int t = 0; t < kThreads; ++t) {
      mid_split(t, lds.data(), mc[t], arr(z0[t]));
      for (int i = 0; i < 8; ++i)
        H[(size_t)b * kHPerPart + h_index(0, i, t)] = C32{z0[t][i].x * kSpectrumScale, z0[t][i].y * kSpectrumScale};
    }
  }
  for (int64_t item = 0; item < rows * g.segs; ++item) {
    const int64_t row = item / g.segs, j_lo = (item - row * g.segs) * g.seg_blocks;
    const int64_t j_hi = j_lo + g.seg_blocks < g.n_blocks ? j_lo + g.seg_blocks : g.n_blocks;
    const int64_t rx = xmap ? xmap[row] : row, ry = ymap ? ymap[row] : row;
    const float* xr = xa + rx * g.nx;
    const C32* Hr = H.data() + ry * NP * kHPerPart;
    const bool vin = ((rx * g.nx) & 1) == 0, vout = ((row * out_len) & 1) == 0;   // what 8-byte alignment of a row means here
    for (int t = 0; t < kThreads; ++t)
      for (int i = 0; i < 8; ++i) z1[t][i] = z2[t][i] = z3[t][i] = C32{0.0f, 0.0f};
    for (int64_t j = j_lo - (NP - 1); j < j_hi; ++j) {
      const bool produce = j >= j_lo;
      for (int t = 0; t < kThreads; ++t) {       // the delayed partitions first, as the kernel does (same summation order)
        C32 h[8];
        for (int i = 0; i < 8; ++i) acc[t][i] = C32{0.0f, 0.0f};
        if (produce) {
          if (NP > 1) {
            for (int i = 0; i < 8; ++i) h[i] = Hr[h_index(1, i, t)];
            mid_mac(t, h, arr(z1[t]), arr(acc[t]));
          }
          if (NP > 2) {
            for (int i = 0; i < 8; ++i) h[i] = Hr[h_index(2, i, t)];
            mid_mac(t, h, arr(z2[t]), arr(acc[t]));
          }
          if (NP > 3) {
            for (int i = 0; i < 8; ++i) h[i] = Hr[h_index(3, i, t)];
            mid_mac(t, h, arr(z3[t]), arr(acc[t]));
          }
        }
      }
      for (int t = 0; t < kThreads; ++t) load_block(t, g, xr, j, vin, arr(v[t]));
      forward();
      for (int t = 0; t < kThreads; ++t) {
        mid_split(t, lds.data(), mc[t], arr(z0[t]));
        if (produce) {
          C32 h[8];
          for (int i = 0; i < 8; ++i) h[i] = Hr[h_index(0, i, t)];
          mid_mac(t, h, arr(z0[t]), arr(acc[t]));
        }
      }
      // (every thread has read its quads before any thread writes: on the device a thread rewrites only the cells it read)
      for (int t = 0; t < kThreads; ++t) {
        if (produce) mid_merge(t, arr(acc[t]), mc[t], lds.data());
        z3[t] = z2[t]; z2[t] = z1[t]; z1[t] = z0[t];
      }
      if (produce) {
        inverse();
        for (int t = 0; t < kThreads; ++t) store_block(t, g, arr(v[t]), j, j_hi, vout, out + row * out_len);
      }
    }
  }
  return 1;
}

int sim_fftconv(const float* x, const float* y, float* out, int64_t rows, int64_t nx, int64_t ny,
                const int64_t* x_row_of, const int64_t* y_row_of, int64_t start, int64_t out_len) {
  FcGeom g; g.rows = rows; g.start = start; g.out_len = out_len;
  const bool swap = ny > nx;
  const float* xa = swap ? y : x; const float* ya = swap ? x : y;
  g.nx = swap ? ny : nx; g.ny = swap ? nx : ny;
  const int64_t* xmap = swap ? y_row_of : x_row_of; const int64_t* ymap = swap ? x_row_of : y_row_of;
  g.n_tiles = (int)((out_len + kFcTN - 1) / kFcTN);
  std::vector<float> xs(kFcTN + kFcTY), ys(kFcTY);
  for (int64_t row = 0; row < rows; ++row)
    for (int tile = 0; tile < g.n_tiles; ++tile) {
      const int64_t rx = xmap ? xmap[row] : row, ry = ymap ? ymap[row] : row;
      const float* xr = xa + rx * g.nx; const float* yr = ya + ry * g.ny;
      const int64_t n0 = start + (int64_t)tile * kFcTN;
      static float acc[kFcThreads][kFcOutPerThread];
      std::memset(acc, 0, sizeof(acc));
      for (int64_t j0 = 0; j0 < g.ny; j0 += kFcTY) {
        const int64_t xbase = n0 - j0 - (kFcTY - 1);
        if (xbase >= g.nx || xbase + kFcTN + kFcTY - 1 <= 0) continue;
        for (int tid = 0; tid < kFcThreads; ++tid) fc_stage(tid, kFcThreads, g, xr, yr, j0, xbase, xs.data(), ys.data());
        for (int tid = 0; tid < kFcThreads; ++tid) fc_accumulate(tid, xs.data(), ys.data(), acc[tid]);
      }
      for (int tid = 0; tid < kFcThreads; ++tid)
        for (int i = 0; i < kFcOutPerThread; ++i) {
          const int64_t n = (int64_t)tile * kFcTN + tid + i * kFcThreads;
          if (n < out_len) out[row * out_len + n] = acc[tid][i];
        }
    }
  return 0;
}

}  // extern "C"

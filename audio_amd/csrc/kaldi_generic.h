// Kaldi-compatible front-end for ANY even padded window size (compliance/kaldi.py:125-151: `round_to_power_of_two =
// False` keeps the padded window at the frame length -- 400 samples at 16 kHz / 25 ms, 200 at 8 kHz, 1102 at 44.1 kHz),
// the sizes the register FFT of stft_pow2.h (256 / 512 / 1024 / 2048) does not serve.
//
// One 256-thread workgroup conditions and transforms `pb` frame PAIRS (pb = 2, or 1 for very long windows):
//   frames -> [+ dither noise] -> DC removal -> raw log-energy -> pre-emphasis -> window -> zero padding to N
//   (kaldi.py:154-217 _get_window), z = a + i b, mixed-radix Stockham stages of stft_generic.h in LDS,
//   separation of the two real spectra, then log-power rows (kaldi.py:306-315) or mel-bank rows (:625-643).
// The per-frame sums (mean, energy) are reduced through LDS in a fixed order (thread partials, then one serial pass per
// frame): deterministic, and the CPU replay (tests/cpu_sim/sim.cpp) reproduces it exactly.
#pragma once
#include "stft_generic.h"
#include "stft_pow2.h"

namespace aamd {
namespace kgen {

using p2::KaldiGeom;
using p2::kaldi_log_energy;
using p2::kaldi_sample;

constexpr int kThreads = 256;

AAMD_HD int pairs_per_block(int n_fft) { return n_fft <= 2400 ? 2 : 1; }
AAMD_HD size_t lds_floats(int n_fft, int pb) {
  return (size_t)2 * n_fft + (size_t)4 * pb * gen_seq_len(n_fft) + (size_t)2 * pb * (n_fft / 2 + 2) + kThreads + 4 * 2 * pb;
}

struct Lds {
  cplx<float>* twl;    // [N]
  cplx<float>* bufA;   // [pb][SL]
  cplx<float>* bufB;   // [pb][SL]
  float* P;            // [2 pb][N/2 + 2]
  float* red;          // [kThreads]
  float* stat;         // [2 pb][4]: mean, log energy, -, -
};
AAMD_HD Lds carve(float* base, int N, int pb) {
  Lds l;
  const int SL = gen_seq_len(N);
  l.twl = reinterpret_cast<cplx<float>*>(base);
  l.bufA = l.twl + N;
  l.bufB = l.bufA + pb * SL;
  l.P = reinterpret_cast<float*>(l.bufB + pb * SL);
  l.red = l.P + 2 * pb * (N / 2 + 2);
  l.stat = l.red + kThreads;
  return l;
}
// the long-window layout (round 5, stft_generic.h gen_lds_floats_long): no twiddle table (read from memory), the power rows take the
// ping-pong buffer the last stage left free (`P` is set by the kernel)
AAMD_HD size_t lds_floats_long(int n_fft, int pb) { return (size_t)4 * pb * gen_seq_len(n_fft) + kThreads + 4 * 2 * pb; }
AAMD_HD Lds carve_long(float* base, int N, int pb) {
  Lds l;
  const int SL = gen_seq_len(N);
  l.twl = nullptr;
  l.bufA = reinterpret_cast<cplx<float>*>(base);
  l.bufB = l.bufA + pb * SL;
  l.P = nullptr;
  l.red = reinterpret_cast<float*>(l.bufB + pb * SL);
  l.stat = l.red + kThreads;
  return l;
}

// thread tid works on frame f = tid / tpf with its tpf-strided share of the samples
struct Who {
  int f, sub, tpf;
};
AAMD_HD Who who(int tid, int nthr, int nf) {
  Who w;
  w.tpf = nthr / nf;
  w.f = tid / w.tpf;
  w.sub = tid - w.f * w.tpf;
  return w;
}

// pass 1: partial sums of the raw frame (for the mean)
AAMD_HD void pass_sum(int tid, int nthr, int nf, const KaldiGeom& kg, const float* x, int64_t t0, float* red) {
  const Who w = who(tid, nthr, nf);
  const int64_t t = t0 + w.f;
  float s = 0.0f;
  if (t < kg.n_frames)
    for (int j = w.sub; j < kg.win; j += w.tpf) s += kaldi_sample(kg, x, t, j);
  red[tid] = s;
}
// one thread per frame folds the partials in a fixed order
AAMD_HD void fold_mean(int tid, int nthr, int nf, const KaldiGeom& kg, const float* red, float* stat) {
  const Who w = who(tid, nthr, nf);
  if (w.sub != 0) return;
  float s = 0.0f;
  for (int i = 0; i < w.tpf; ++i) s += red[w.f * w.tpf + i];
  stat[4 * w.f] = kg.remove_dc ? s / (float)kg.win : 0.0f;
}
// pass 2: partial sums of (x - mean)^2 (raw log-energy)
AAMD_HD void pass_sumsq(int tid, int nthr, int nf, const KaldiGeom& kg, const float* x, int64_t t0, const float* stat,
                        float* red) {
  const Who w = who(tid, nthr, nf);
  const int64_t t = t0 + w.f;
  const float mean = stat[4 * w.f];
  float s = 0.0f;
  if (t < kg.n_frames)
    for (int j = w.sub; j < kg.win; j += w.tpf) {
      const float d = kaldi_sample(kg, x, t, j) - mean;
      s += d * d;
    }
  red[tid] = s;
}
AAMD_HD void fold_energy(int tid, int nthr, int nf, const KaldiGeom& kg, const float* red, float* stat) {
  const Who w = who(tid, nthr, nf);
  if (w.sub != 0) return;
  float s = 0.0f;
  for (int i = 0; i < w.tpf; ++i) s += red[w.f * w.tpf + i];
  stat[4 * w.f + 1] = kaldi_log_energy(kg, s);
}
// pass 3: y[j] = ((x[j] - mean) - c (x[max(j - 1, 0)] - mean)) w[j] into the real / imaginary part of the pair's sequence,
// zeros for j >= win and for frames past the end; partial sums of y^2 (log-energy of the windowed frame)
AAMD_HD void pass_shape(int tid, int nthr, int nf, int N, const KaldiGeom& kg, const float* x, const float* window,
                        int64_t t0, const float* stat, cplx<float>* buf, float* red) {
  const Who w = who(tid, nthr, nf);
  const int64_t t = t0 + w.f;
  const float mean = stat[4 * w.f];
  const int SL = gen_seq_len(N);
  float* dst = reinterpret_cast<float*>(buf + (w.f >> 1) * SL) + (w.f & 1);
  const bool live = t < kg.n_frames;
  float s = 0.0f;
  for (int j = w.sub; j < N; j += w.tpf) {
    float y = 0.0f;
    if (live && j < kg.win) {
      const float d = kaldi_sample(kg, x, t, j) - mean;
      const float dp = kaldi_sample(kg, x, t, j > 0 ? j - 1 : 0) - mean;
      y = (d - kg.preemph * dp) * window[j];
    }
    dst[2 * gen_pad(j)] = y;
    s += y * y;
  }
  red[tid] = s;
}

AAMD_HD void separate(const cplx<float>* X, int N, int k, cplx<float>& A, cplx<float>& B) {
  const cplx<float> zk = X[gen_pad(k)];
  const cplx<float> zm = X[gen_pad(k == 0 ? 0 : N - k)];
  A = {0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
  B = {0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x)};
}

// spectrogram rows: log(max(|X|^2, eps)), column 0 = the log energy (kaldi.py:306-315)
AAMD_HD void store_spec(int tid, int nthr, int nf, int N, const KaldiGeom& kg, const cplx<float>* X, int64_t t0,
                        const float* stat, float* out) {
  const int F = N / 2 + 1, SL = gen_seq_len(N);
  for (int idx = tid; idx < nf * F; idx += nthr) {
    const int f = idx / F, k = idx - f * F;
    const int64_t t = t0 + f;
    if (t >= kg.n_frames) continue;
    cplx<float> A, B;
    separate(X + (f >> 1) * SL, N, k, A, B);
    const cplx<float> v = (f & 1) ? B : A;
    out[t * (int64_t)F + k] = k == 0 ? stat[4 * f + 1] : log(fmax(v.x * v.x + v.y * v.y, kg.eps));
  }
}
// |X|^2 (or |X|) rows of every frame of the block, one zero behind the Nyquist bin (band reads may touch it)
AAMD_HD void power_rows(int tid, int nthr, int nf, int N, const KaldiGeom& kg, const cplx<float>* X, float* P) {
  const int F = N / 2 + 1, SL = gen_seq_len(N), PS = F + 1;
  for (int idx = tid; idx < nf * PS; idx += nthr) {
    const int f = idx / PS, k = idx - f * PS;
    float p = 0.0f;
    if (k < F) {
      cplx<float> A, B;
      separate(X + (f >> 1) * SL, N, k, A, B);
      const cplx<float> v = (f & 1) ? B : A;
      p = v.x * v.x + v.y * v.y;
      if (!kg.use_power) p = sqrt(p);
    }
    P[f * PS + k] = p;
  }
}
// mel-bank rows (kaldi.py:625-643) + the energy column
AAMD_HD void fbank_rows(int tid, int nthr, int nf, int N, const KaldiGeom& kg, const MelBandsDev& mb, const float* P,
                        int64_t t0, const float* stat, float* out) {
  const int PS = N / 2 + 2;
  for (int idx = tid; idx < nf * mb.n_mels; idx += nthr) {
    const int f = idx / mb.n_mels, m = idx - f * mb.n_mels;
    const int64_t t = t0 + f;
    if (t >= kg.n_frames) continue;
    const int lo = mb.lo[m], w = mb.width[m];
    const float* wt = mb.weights + (int64_t)m * mb.max_width;
    const float* Pf = P + f * PS + lo;
    float acc = 0.0f;
    for (int i = 0; i < w; ++i) acc += wt[i] * Pf[i];
    if (kg.use_log) acc = log(fmax(acc, kg.eps));
    out[t * (int64_t)kg.n_cols + kg.first_col + m] = acc;
  }
  if (kg.energy_col >= 0)
    for (int f = tid; f < nf; f += nthr)
      if (t0 + f < kg.n_frames) out[(t0 + f) * (int64_t)kg.n_cols + kg.energy_col] = stat[4 * f + 1];
}

struct Plan {
  int n_fft, n_stages;
  int radix[kMaxStages];
};

#if defined(__HIPCC__)
// MODE 0: kaldi.spectrogram rows [N/2 + 1]; MODE 1: kaldi.fbank rows [n_cols]
template <int MODE, int LONG = 0>
__global__ void __launch_bounds__(kThreads)
kaldi_generic_kernel(KaldiGeom kg, Plan plan, int pb, int blocks_per_utt, const float* wav,
                     const float* __restrict__ window /* [N], 0 past win */, const cplx<float>* __restrict__ tw,
                     MelBandsDev mb, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_kg[];
  const int N = plan.n_fft, nf = 2 * pb;
  Lds l = LONG ? carve_long(reinterpret_cast<float*>(smem_kg), N, pb) : carve(reinterpret_cast<float*>(smem_kg), N, pb);
  const cplx<float>* twl = LONG ? tw : l.twl;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t utt = blockIdx.x / blocks_per_utt;            // batch extension: one utterance per group of workgroups
  const int64_t t0 = (int64_t)(blockIdx.x - utt * blocks_per_utt) * nf;
  wav += utt * kg.utt_stride;
  out += utt * kg.n_frames * (MODE == 0 ? (int64_t)(N / 2 + 1) : (int64_t)kg.n_cols);
  if (kg.noise != nullptr) kg.noise += utt * kg.n_frames * kg.win;
  if (!LONG)
    for (int i = tid; i < N; i += nthr) l.twl[i] = tw[i];
  pass_sum(tid, nthr, nf, kg, wav, t0, l.red);
  __syncthreads();
  fold_mean(tid, nthr, nf, kg, l.red, l.stat);
  __syncthreads();
  if (kg.raw_energy) {
    pass_sumsq(tid, nthr, nf, kg, wav, t0, l.stat, l.red);
    __syncthreads();
    fold_energy(tid, nthr, nf, kg, l.red, l.stat);
    __syncthreads();
  }
  pass_shape(tid, nthr, nf, N, kg, wav, window, t0, l.stat, l.bufA, l.red);
  __syncthreads();
  if (!kg.raw_energy) {
    fold_energy(tid, nthr, nf, kg, l.red, l.stat);
    __syncthreads();
  }
  cplx<float>* x = l.bufA;
  cplx<float>* y = l.bufB;
  int s = 1;
  for (int st = 0; st < plan.n_stages; ++st) {
    const int r = plan.radix[st];
    gen_stage<float>(tid, nthr, N, r, s, pb, x, y, twl);
    __syncthreads();
    s *= r;
    cplx<float>* tmp = x; x = y; y = tmp;
  }
  if (MODE == 0) {
    store_spec(tid, nthr, nf, N, kg, x, t0, l.stat, out);
  } else {
    if (LONG) l.P = reinterpret_cast<float*>(y);      // 2 pb (N / 2 + 2) floats into a buffer of 2 pb SL
    power_rows(tid, nthr, nf, N, kg, x, l.P);
    __syncthreads();
    fbank_rows(tid, nthr, nf, N, kg, mb, l.P, t0, l.stat, out);
  }
}
#endif  // __HIPCC__

}  // namespace kgen
}  // namespace aamd

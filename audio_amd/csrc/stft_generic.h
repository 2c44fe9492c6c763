// Generic STFT path: any n_fft (mixed radix, arbitrary prime factors), any hop / padding /
// window / normalisation / power, with optional fused banded-mel epilogue.
//
// One workgroup owns a run of consecutive frames of one waveform.  Per frame:
//   load+window (reflect/replicate/circular/constant index math, exact integers)
//   -> Stockham autosort FFT, ping-pong between two LDS buffers, one barrier per stage
//   -> epilogue straight from LDS (|X|^p or complex, or banded mel reduction).
// This is the correctness workhorse for every shape that is not the (400,160) headline
// kernel in melspec400.h; both are checked against the same oracle.
//
// Semantics restated from the reference: functional/functional.py:112-145 (F.spectrogram),
// torch/functional.py:675-681 (centre padding), transforms/_transforms.py:403-415 (MelScale).
#pragma once
#include "hd.h"

namespace aamd {

constexpr int kMaxStages = 16;

struct StftGeom {
  int64_t rows, length, row_stride;
  int n_fft, hop, pad, center, pad_mode, onesided, n_frames, n_freq;
  float scale, power;
  int n_stages;
  int radix[kMaxStages];
};

struct MelBandsDev {
  int n_mels, max_width;
  const int32_t* lo;
  const int32_t* width;
  const float* weights;
  const int32_t* order;   // mel400 only: table row -> mel (lane assignment), -1 = unused row; null = identity
};

// ---- phase 1: gather one frame, multiply by the window, write complex (v, 0) ------------
template <typename T>
AAMD_HD void stft_load_frame(int tid, int nthr, const StftGeom& g, const T* wav_row,
                             const T* window, int64_t t, cplx<T>* buf) {
  const int64_t L1 = g.length + 2 * (int64_t)g.pad;  // after F.spectrogram's zero pad
  const int64_t base = t * (int64_t)g.hop - (g.center ? g.n_fft / 2 : 0);
  for (int n = tid; n < g.n_fft; n += nthr) {
    int64_t i1 = base + n;
    int64_t s1 = g.center ? pad_source_index(i1, L1, g.pad_mode) : i1;
    T v = 0;
    if (s1 >= 0) {
      int64_t s0 = s1 - g.pad;
      if (s0 >= 0 && s0 < g.length) v = wav_row[s0];
    }
    buf[n] = {v * window[n], (T)0};
  }
}

// ---- phase 2: one Stockham (DIF, autosort) stage of radix r, sub-length n = N/s ---------
//   y[q + s(r p + k)] = W_n^{pk} * sum_j x[q + s(p + m j)] W_r^{jk},  m = n/r,
//   p in [0,m), q in [0,s), k in [0,r).   One work item per output element.
template <typename T>
AAMD_HD void stockham_stage(int tid, int nthr, int N, int r, int s, const cplx<T>* x,
                            cplx<T>* y, const cplx<T>* tw) {
  const int n = N / s;
  const int m = n / r;
  const int nb = N / r;  // butterflies
  for (int u = tid; u < N; u += nthr) {
    const int k = u / nb;
    const int i = u - k * nb;
    const int p = i / s;
    const int q = i - p * s;
    cplx<T> acc = {0, 0};
    const int wstep = (int)(((int64_t)nb * k) % N);  // W_r^{k} as a power of W_N
    int widx = 0;
    for (int j = 0; j < r; ++j) {
      cplx<T> a = x[q + s * (p + m * j)];
      acc = cadd(acc, cmul(a, tw[widx]));
      widx += wstep;
      if (widx >= N) widx -= N;
    }
    const int tidx = (int)(((int64_t)s * p * k) % N);  // W_n^{pk}
    acc = cmul(acc, tw[tidx]);
    y[q + s * (r * p + k)] = acc;
  }
}

template <typename T>
AAMD_HD T mag_pow(T re, T im, float power) {
  T m2 = re * re + im * im;
  if (power == 2.0f) return m2;
  T m = sqrt(m2);
  if (power == 1.0f) return m;
  return pow(m, (T)power);
}

// ---- phase 3a: spectrogram epilogue ----------------------------------------------------
template <typename T>
AAMD_HD void stft_store_spec(int tid, int nthr, const StftGeom& g, const cplx<T>* X,
                             T* out_frame) {
  for (int k = tid; k < g.n_freq; k += nthr) {
    T re = X[k].x * (T)g.scale, im = X[k].y * (T)g.scale;
    if (g.power <= 0.0f) {
      out_frame[2 * k] = re;
      out_frame[2 * k + 1] = im;
    } else {
      out_frame[k] = mag_pow(re, im, g.power);
    }
  }
}

// ---- phase 3b/3c: power spectrum to LDS, then banded mel reduction ------------------------
template <typename T>
AAMD_HD void stft_power_to_lds(int tid, int nthr, const StftGeom& g, const cplx<T>* X, T* P) {
  for (int k = tid; k < g.n_freq; k += nthr) {
    T re = X[k].x * (T)g.scale, im = X[k].y * (T)g.scale;
    P[k] = mag_pow(re, im, g.power);
  }
}

template <typename T>
AAMD_HD void mel_from_lds(int tid, int nthr, const MelBandsDev& mb, const T* P, T* out_frame) {
  for (int m = tid; m < mb.n_mels; m += nthr) {
    const int lo = mb.lo[m], w = mb.width[m];
    const float* wt = mb.weights + (int64_t)m * mb.max_width;
    T acc = 0;
    for (int i = 0; i < w; ++i) acc += (T)wt[i] * P[lo + i];
    out_frame[m] = acc;
  }
}

#if defined(__HIPCC__)
enum { EPI_SPEC = 0, EPI_MEL = 1 };

template <typename T, int EPI>
__global__ void __launch_bounds__(256)
stft_generic_kernel(StftGeom g, const T* __restrict__ wav, const T* __restrict__ window,
                    const cplx<T>* __restrict__ tw, MelBandsDev mb, T* __restrict__ out,
                    int frames_per_block, int blocks_per_row) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cplx<T>* bufA = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* bufB = bufA + g.n_fft;
  T* P = reinterpret_cast<T*>(bufB + g.n_fft);

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t row = blockIdx.x / blocks_per_row;
  const int chunk = blockIdx.x - (int)row * blocks_per_row;
  const T* wav_row = wav + row * g.row_stride;
  const int out_per_frame =
      EPI == EPI_MEL ? mb.n_mels : (g.power <= 0.0f ? 2 * g.n_freq : g.n_freq);

  for (int f = 0; f < frames_per_block; ++f) {
    const int64_t t = (int64_t)chunk * frames_per_block + f;
    if (t >= g.n_frames) break;
    stft_load_frame<T>(tid, nthr, g, wav_row, window, t, bufA);
    __syncthreads();
    cplx<T>* x = bufA;
    cplx<T>* y = bufB;
    int s = 1;
    for (int st = 0; st < g.n_stages; ++st) {
      const int r = g.radix[st];
      stockham_stage<T>(tid, nthr, g.n_fft, r, s, x, y, tw);
      __syncthreads();
      s *= r;
      cplx<T>* tmp = x; x = y; y = tmp;
    }
    T* out_frame = out + (row * g.n_frames + t) * (int64_t)out_per_frame;
    if (EPI == EPI_SPEC) {
      stft_store_spec<T>(tid, nthr, g, x, out_frame);
    } else {
      stft_power_to_lds<T>(tid, nthr, g, x, P);
      __syncthreads();
      mel_from_lds<T>(tid, nthr, mb, P, out_frame);
    }
    __syncthreads();
  }
}
#endif  // __HIPCC__

// Host helper: factor n into radices (4s first, then 2, 3, 5, remaining primes).
inline int plan_radices(int n, int* radix) {
  int ns = 0;
  while (n % 4 == 0 && ns < kMaxStages) { radix[ns++] = 4; n /= 4; }
  for (int p = 2; n > 1 && ns < kMaxStages; ) {
    if (n % p == 0) { radix[ns++] = p; n /= p; }
    else { p += (p == 2) ? 1 : 2; if ((int64_t)p * p > n && n > 1) { radix[ns++] = n; n = 1; } }
  }
  return n == 1 ? ns : -1;
}

}  // namespace aamd

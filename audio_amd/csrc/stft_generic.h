// Generic STFT path: any n_fft (mixed radix, arbitrary prime factors), any hop / padding /
// window / normalisation / power, with optional fused banded-mel epilogue.
//
// One workgroup owns 2 PB consecutive frames of one waveform (PB frame PAIRS, each pair one complex FFT):
//   load+window (reflect/replicate/circular/constant index math, exact integers)
//   -> Stockham autosort FFT, ping-pong between two LDS buffers, one barrier per stage, ONE radix-r
//      butterfly per work item (radix 2/3/4/5 in registers, any other prime by its r-point DFT),
//      twiddles from an LDS copy of the W_N table
//   -> epilogue straight from LDS: separate the two real spectra, |X|^p or complex, or banded mel.
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
  const float* table400;  // mel400 only: prebuilt LDS image of the band table (m400::mel_tab_dwords dwords), or null
  int table_sig;          // mel400 only: shape of that image (chunk counts per round as nibbles), 0 = unknown
};

// ---- geometry of one workgroup: PB frame PAIRS (2 PB consecutive frames of one waveform) ------------
// Two real frames a, b = a + 1 are transformed as ONE complex sequence z = a + i b and separated in
// the epilogue with X_a[k] = (Z[k] + conj Z[N-k]) / 2,  X_b[k] = (Z[k] - conj Z[N-k]) / (2i).
constexpr int kGenThreads = 256;
AAMD_HD int gen_pairs_per_block(int n_fft) {
  // Small workgroups win here: the kernel is latency-bound (one barrier per stage), so more resident
  // workgroups per CU beat fuller workgroups (measured: 2 pairs per block at n_fft = 512 is 1.45x faster
  // than 4, and one wave per pair without barriers is slower still).
  int p = 1024 / n_fft;
  return p < 1 ? 1 : (p > 8 ? 8 : p);
}
// Sequence buffers are index-padded (one complex every 32) so that the power-of-two strides of the
// Stockham output pattern (r p + k) do not pile onto a few LDS banks.
AAMD_HD int gen_pad(int i) { return i + (i >> 5); }
AAMD_HD int gen_seq_len(int n_fft) { return gen_pad(n_fft - 1) + 1; }
// LDS floats: twiddle table N complex | two ping-pong buffers of PB padded sequences | PB x 2 power rows
AAMD_HD size_t gen_lds_floats(int n_fft, int n_freq, int pb) {
  return (size_t)2 * n_fft + (size_t)4 * pb * gen_seq_len(n_fft) + (size_t)2 * pb * n_freq;
}
// The long-window layout (round 5; n_fft of about 5 750 .. 8 192 in float32: 0.36 .. 0.51 s windows at 16 kHz, n_fft = 8192 at 44.1 / 48 kHz):
// only the two ping-pong buffers -- the twiddles are read from the table in memory (64 KB: L2-resident) and the power rows of the
// mel epilogue take the ping-pong buffer that the last stage left free.  Slower per stage than the LDS table; used only when the
// full layout does not fit.
AAMD_HD size_t gen_lds_floats_long(int n_fft, int pb) { return (size_t)4 * pb * gen_seq_len(n_fft); }

template <typename T>
AAMD_HD T stft_sample(const StftGeom& g, const T* wav_row, int64_t t, int n) {
  const int64_t L1 = g.length + 2 * (int64_t)g.pad;  // after F.spectrogram's zero pad
  const int64_t i1 = t * (int64_t)g.hop - (g.center ? g.n_fft / 2 : 0) + n;
  const int64_t s1 = g.center ? pad_source_index(i1, L1, g.pad_mode) : i1;
  if (s1 < 0) return (T)0;
  const int64_t s0 = s1 - g.pad;
  return (s0 >= 0 && s0 < g.length) ? wav_row[s0] : (T)0;
}

// ---- phase 1: gather PB frame pairs, multiply by the window, write z = a + i b ----------------------
//   Frames that lie entirely inside the waveform (all but the first / last few) take plain coalesced
//   loads; only the edge frames pay for the padding index math.
template <typename T>
AAMD_HD void gen_load(int tid, int nthr, const StftGeom& g, const T* wav_row, const T* window, int64_t t0,
                      int pb, cplx<T>* buf) {
  const int N = g.n_fft, SL = gen_seq_len(N);
  const int64_t cpad = g.center ? g.n_fft / 2 : 0;
  for (int pair = 0; pair < pb; ++pair) {        // per-pair invariants hoisted out of the sample loop
    const int64_t ta = t0 + 2 * pair, tb = ta + 1;
    const int64_t ba = ta * (int64_t)g.hop - cpad - g.pad, bb = ba + g.hop;   // source index of sample 0
    const bool va = ta < g.n_frames, vb = tb < g.n_frames;
    const bool ia = va && ba >= 0 && ba + N <= g.length, ib = vb && bb >= 0 && bb + N <= g.length;
    cplx<T>* dst = buf + pair * SL;
    for (int n = tid; n < N; n += nthr) {
      const T w = window[n];
      const T a = ia ? wav_row[ba + n] : (va ? stft_sample<T>(g, wav_row, ta, n) : (T)0);
      const T b = ib ? wav_row[bb + n] : (vb ? stft_sample<T>(g, wav_row, tb, n) : (T)0);
      dst[gen_pad(n)] = {a * w, b * w};
    }
  }
}

// ---- radix butterflies (forward, e^{-2 pi i jk/r}), in place on v[0..r) ------------------------------
template <typename T>
AAMD_HD void bfly2(cplx<T>* v) {
  const cplx<T> a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}
template <typename T>
AAMD_HD void bfly4(cplx<T>* v) {
  const cplx<T> s0 = cadd(v[0], v[2]), s1 = csub(v[0], v[2]), s2 = cadd(v[1], v[3]), s3 = csub(v[1], v[3]);
  const cplx<T> m = {s3.y, -s3.x};   // -i * s3
  v[0] = cadd(s0, s2);
  v[1] = cadd(s1, m);
  v[2] = csub(s0, s2);
  v[3] = csub(s1, m);
}
template <typename T>
AAMD_HD void bfly8(cplx<T>* v) {   // 2 x radix-4 on even / odd inputs, W8 twiddles, radix-2 combine
  const T h = (T)0.70710678118654752440;
  cplx<T> e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
  bfly4<T>(e);
  bfly4<T>(o);
  const cplx<T> o1 = {h * (o[1].x + o[1].y), h * (o[1].y - o[1].x)};    // o1 * W8^1 = o1 * (1 - i)/sqrt2
  const cplx<T> o2 = {o[2].y, -o[2].x};                                 // o2 * W8^2 = -i o2
  const cplx<T> o3 = {h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y)};   // o3 * W8^3 = o3 * (-1 - i)/sqrt2
  v[0] = cadd(e[0], o[0]); v[4] = csub(e[0], o[0]);
  v[1] = cadd(e[1], o1);   v[5] = csub(e[1], o1);
  v[2] = cadd(e[2], o2);   v[6] = csub(e[2], o2);
  v[3] = cadd(e[3], o3);   v[7] = csub(e[3], o3);
}
template <typename T>
AAMD_HD void bfly3(cplx<T>* v) {
  const T h = (T)0.86602540378443864676;   // sin(2 pi / 3)
  const cplx<T> t = cadd(v[1], v[2]), d = csub(v[1], v[2]);
  const cplx<T> c = {v[0].x - (T)0.5 * t.x, v[0].y - (T)0.5 * t.y};
  const cplx<T> e = {h * d.y, -h * d.x};   // -i * h * d
  v[0] = cadd(v[0], t);
  v[1] = cadd(c, e);
  v[2] = csub(c, e);
}
template <typename T>
AAMD_HD void bfly5(cplx<T>* v) {
  const T C1 = (T)0.30901699437494742, C2 = (T)-0.80901699437494742;
  const T S1 = (T)0.95105651629515357, S2 = (T)0.58778525229247313;
  const cplx<T> t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
  const cplx<T> m1 = {v[0].x + C1 * t1.x + C2 * t2.x, v[0].y + C1 * t1.y + C2 * t2.y};
  const cplx<T> m2 = {v[0].x + C2 * t1.x + C1 * t2.x, v[0].y + C2 * t1.y + C1 * t2.y};
  const cplx<T> u1 = {S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y};
  const cplx<T> u2 = {S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y};
  v[0] = {v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y};
  v[1] = {m1.x + u1.y, m1.y - u1.x};   // m1 - i u1
  v[4] = {m1.x - u1.y, m1.y + u1.x};
  v[2] = {m2.x + u2.y, m2.y - u2.x};   // m2 - i u2
  v[3] = {m2.x - u2.y, m2.y + u2.x};
}

// load R inputs (stride `in_stride`), butterfly in registers, twiddle, store R outputs (stride `out_stride`)
template <typename T, int R>
AAMD_HD void gen_bfly_static(const cplx<T>* x, cplx<T>* y, int xi, int yi, int in_stride, int out_stride,
                             int step, const cplx<T>* tw) {
  cplx<T> v[R];
#pragma unroll
  for (int j = 0; j < R; ++j) v[j] = x[gen_pad(xi + in_stride * j)];
  if (R == 8) bfly8<T>(v);
  else if (R == 4) bfly4<T>(v);
  else if (R == 5) bfly5<T>(v);
  else if (R == 2) bfly2<T>(v);
  else bfly3<T>(v);
  y[gen_pad(yi)] = v[0];
#pragma unroll
  for (int k = 1; k < R; ++k) y[gen_pad(yi + out_stride * k)] = cmul(v[k], tw[step * k]);
}

// ---- phase 2: one Stockham (autosort) stage of radix r on PB sequences: one BUTTERFLY per work item --
//   y[q + s(r p + k)] = W_N^{s p k} * sum_j x[q + s(p + m j)] W_r^{jk},  m = N / (s r),
//   p in [0,m), q in [0,s), k in [0,r);  s p k < N, so the twiddle index needs no reduction.
template <typename T>
AAMD_HD void gen_stage(int tid, int nthr, int N, int r, int s, int pb, const cplx<T>* x, cplx<T>* y,
                       const cplx<T>* tw) {
  const int nb = N / r;          // butterflies per sequence
  const int m = nb / s;
  const int SL = gen_seq_len(N);
  for (int it = tid; it < pb * nb; it += nthr) {
    const int pair = it / nb, i = it - pair * nb;
    const int p = i / s, q = i - p * s;
    const cplx<T>* xq = x + pair * SL;
    cplx<T>* yq = y + pair * SL;
    const int xi = q + s * p, yi = q + s * r * p;      // un-padded indices inside the sequence
    const int step = s * p;      // W_N^{s p k}
    if (r == 8) {
      gen_bfly_static<T, 8>(xq, yq, xi, yi, s * m, s, step, tw);
    } else if (r == 4) {
      gen_bfly_static<T, 4>(xq, yq, xi, yi, s * m, s, step, tw);
    } else if (r == 5) {
      gen_bfly_static<T, 5>(xq, yq, xi, yi, s * m, s, step, tw);
    } else if (r == 2) {
      gen_bfly_static<T, 2>(xq, yq, xi, yi, s * m, s, step, tw);
    } else if (r == 3) {
      gen_bfly_static<T, 3>(xq, yq, xi, yi, s * m, s, step, tw);
    } else {   // any other prime radix: direct r-point DFT, W_r^{jk} = W_N^{(N/r) (jk mod r)}
      for (int k = 0; k < r; ++k) {
        cplx<T> acc = xq[gen_pad(xi)];
        int e = 0;
        for (int j = 1; j < r; ++j) {
          e += k;
          if (e >= r) e -= r;
          acc = cadd(acc, cmul(xq[gen_pad(xi + s * m * j)], tw[nb * e]));
        }
        yq[gen_pad(yi + s * k)] = (k == 0) ? acc : cmul(acc, tw[step * k]);
      }
    }
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

// the two real spectra of pair `pair` at bin k (scaled)
template <typename T>
AAMD_HD void gen_separate(const StftGeom& g, const cplx<T>* X, int pair, int k, cplx<T>& A, cplx<T>& B) {
  const int N = g.n_fft, SL = gen_seq_len(N);
  const cplx<T> zk = X[pair * SL + gen_pad(k)];
  const cplx<T> zm = X[pair * SL + gen_pad(k == 0 ? 0 : N - k)];
  const T h = (T)0.5 * (T)g.scale;
  A = {h * (zk.x + zm.x), h * (zk.y - zm.y)};       // (Z[k] + conj Z[N-k]) / 2
  B = {h * (zk.y + zm.y), h * (zm.x - zk.x)};       // (Z[k] - conj Z[N-k]) / (2i)
}

// ---- phase 3a: spectrogram epilogue (2 PB frames x n_freq bins, frame-major output) --------------------
template <typename T>
AAMD_HD void gen_store_spec(int tid, int nthr, const StftGeom& g, const cplx<T>* X, int pb, int64_t t0,
                            T* out_row /* frame 0 of this waveform */) {
  const int opf = g.power <= 0.0f ? 2 * g.n_freq : g.n_freq;
  for (int idx = tid; idx < 2 * pb * g.n_freq; idx += nthr) {
    const int f = idx / g.n_freq, k = idx - f * g.n_freq;
    const int64_t t = t0 + f;
    if (t >= g.n_frames) continue;
    cplx<T> A, B;
    gen_separate<T>(g, X, f >> 1, k, A, B);
    const cplx<T> v = (f & 1) ? B : A;
    T* o = out_row + t * (int64_t)opf;
    if (g.power <= 0.0f) {
      o[2 * k] = v.x;
      o[2 * k + 1] = v.y;
    } else {
      o[k] = mag_pow(v.x, v.y, g.power);
    }
  }
}

// ---- phase 3b/3c: power rows to LDS, then banded mel reduction ---------------------------------------
template <typename T>
AAMD_HD void gen_power_rows(int tid, int nthr, const StftGeom& g, const cplx<T>* X, int pb, T* P) {
  for (int idx = tid; idx < pb * g.n_freq; idx += nthr) {
    const int pair = idx / g.n_freq, k = idx - pair * g.n_freq;
    cplx<T> A, B;
    gen_separate<T>(g, X, pair, k, A, B);
    P[(2 * pair) * g.n_freq + k] = mag_pow(A.x, A.y, g.power);
    P[(2 * pair + 1) * g.n_freq + k] = mag_pow(B.x, B.y, g.power);
  }
}

template <typename T>
AAMD_HD void gen_mel(int tid, int nthr, const StftGeom& g, const MelBandsDev& mb, const T* P, int pb, int64_t t0,
                     T* out_row) {
  for (int idx = tid; idx < 2 * pb * mb.n_mels; idx += nthr) {
    const int f = idx / mb.n_mels, m = idx - f * mb.n_mels;
    const int64_t t = t0 + f;
    if (t >= g.n_frames) continue;
    const int lo = mb.lo[m], w = mb.width[m];
    const float* wt = mb.weights + (int64_t)m * mb.max_width;
    const T* Pf = P + f * g.n_freq + lo;
    T acc = 0;
    for (int i = 0; i < w; ++i) acc += (T)wt[i] * Pf[i];
    out_row[t * (int64_t)mb.n_mels + m] = acc;
  }
}

#if defined(__HIPCC__)
enum { EPI_SPEC = 0, EPI_MEL = 1 };

// LONG = 1: the long-window layout (gen_lds_floats_long): twiddles from memory, power rows in the free ping-pong buffer
template <typename T, int EPI, int LONG = 0>
__global__ void __launch_bounds__(kGenThreads)
stft_generic_kernel(StftGeom g, const T* __restrict__ wav, const T* __restrict__ window,
                    const cplx<T>* __restrict__ tw, MelBandsDev mb, T* __restrict__ out,
                    int pairs_per_block, int blocks_per_row) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = g.n_fft, pb = pairs_per_block, SL = gen_seq_len(N);
  cplx<T>* twl_lds = reinterpret_cast<cplx<T>*>(smem);
  cplx<T>* bufA = LONG ? twl_lds : twl_lds + N;
  cplx<T>* bufB = bufA + pb * SL;
  T* P = reinterpret_cast<T*>(bufB + pb * SL);
  const cplx<T>* twl = LONG ? tw : twl_lds;

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t row = blockIdx.x / blocks_per_row;
  const int chunk = blockIdx.x - (int)row * blocks_per_row;
  const int64_t t0 = (int64_t)chunk * 2 * pb;
  const T* wav_row = wav + row * g.row_stride;
  if (!LONG)
    for (int i = tid; i < N; i += nthr) twl_lds[i] = tw[i];
  gen_load<T>(tid, nthr, g, wav_row, window, t0, pb, bufA);
  __syncthreads();
  cplx<T>* x = bufA;
  cplx<T>* y = bufB;
  int s = 1;
  for (int st = 0; st < g.n_stages; ++st) {
    const int r = g.radix[st];
    gen_stage<T>(tid, nthr, N, r, s, pb, x, y, twl);
    __syncthreads();
    s *= r;
    cplx<T>* tmp = x; x = y; y = tmp;
  }
  if (EPI == EPI_SPEC) {
    const int opf = g.power <= 0.0f ? 2 * g.n_freq : g.n_freq;
    gen_store_spec<T>(tid, nthr, g, x, pb, t0, out + row * g.n_frames * (int64_t)opf);
  } else {
    if (LONG) P = reinterpret_cast<T*>(y);            // (2 pb n_freq floats into a buffer of 2 pb SL: the other one holds the spectra)
    gen_power_rows<T>(tid, nthr, g, x, pb, P);
    __syncthreads();
    gen_mel<T>(tid, nthr, g, mb, P, pb, t0, out + row * g.n_frames * (int64_t)mb.n_mels);
  }
}
#endif  // __HIPCC__

// Host helper: factor n into radices.  Powers of two become 8s with the remainder as 4s (never a lone
// radix-2 stage when 4 x 4 is possible: 2^4 -> 4,4; 2^10 -> 8,8,4,4), then 3, 5, remaining primes.
inline int plan_radices(int n, int* radix) {
  int ns = 0, e = 0;
  while (n % 2 == 0) { n /= 2; ++e; }
  int n4 = (e % 3 == 1 && e >= 4) ? 2 : (e % 3 == 2 ? 1 : 0);
  int n2 = (e == 1) ? 1 : 0;
  int n8 = (e - 2 * n4 - n2) / 3;
  for (int i = 0; i < n8 && ns < kMaxStages; ++i) radix[ns++] = 8;
  for (int i = 0; i < n4 && ns < kMaxStages; ++i) radix[ns++] = 4;
  for (int i = 0; i < n2 && ns < kMaxStages; ++i) radix[ns++] = 2;
  for (int p = 3; n > 1 && ns < kMaxStages; ) {
    if (n % p == 0) { radix[ns++] = p; n /= p; }
    else { p += 2; if ((int64_t)p * p > n && n > 1) { radix[ns++] = n; n = 1; } }
  }
  return n == 1 ? ns : -1;
}

}  // namespace aamd

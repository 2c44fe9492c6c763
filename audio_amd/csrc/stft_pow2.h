// Spectrogram / MelSpectrogram for power-of-two n_fft = 64 E (E = 8, 16, 32: n_fft = 512, 1024, 2048), any hop, window
// and padding mode (functional/functional.py:112-145, transforms/_transforms.py:403-415, 612-622): the frames
// the radix-20x20 kernel (melspec400.h) does not cover but that TTS / vocoder / librosa-style front-ends use.
//
// One WAVE transforms one frame pair (two real frames a, b = a + 1 packed as z = a + i b) with the whole
// sequence in registers, E complex values per lane, and no workgroup barrier anywhere:
//   n = l + 64 e,  k = k1 + E (k2a + 8 k2b),  l = l1 + 8 l2   (l = lane)
//   stage A  lane l:            DFT-E over e of w[n] z[n]            -> Y[l][k1], times W_N^(l k1)
//   exchange 1 (LDS, [k1][l], rows padded to 72): lane (k1lo, l1) gathers l2 = 0..7 for k1 = k1lo + 8 k1hi
//   stage B  lane (k1lo, l1):   DFT-8 over l2                         -> [k2a], times W_64^(l1 k2a)
//   exchange 2 (LDS, [g = k1 + E k2a][l1], rows padded to 9): lane g mod 64 gathers l1 = 0..7 for g = lane + 64 rhi
//   stage C  lane:              DFT-8 over l1                         -> Z[g + 8 E k2b]
// so the lane ends up with Z[lane + 64 j], j = 0..E-1: natural order, 64 consecutive bins per register -- the
// output rows leave as fully coalesced stores straight from registers.  The partner bin Z[N - k] of the real-pair
// separation comes through one more LDS pass of the upper half.  All three exchanges are conflict-free b64
// accesses (strides 72 and 9 complex) and are ordered by the wave's own program order (LDS is in-order per wave).
// Constants per lane (window taps, W_N^(l k1), W_64^(l1 k2a)) live in registers for the whole launch; waves are
// persistent and stride over the frame pairs.
#pragma once
#include "hd.h"
#include "stft_generic.h"

namespace aamd {
namespace p2 {

constexpr int kWaves = 4;                    // waves per workgroup (independent of each other)
constexpr int kRow1 = 72;                    // exchange-1 row stride (complex): 72 * 2 = 16 (mod 128) dwords
constexpr int kRow2 = 9;                     // exchange-2 row stride (complex)

template <int E>
struct Cfg {
  static constexpr int N = 64 * E;
  static constexpr int G = E >= 8 ? E / 8 : 1;   // DFT-8 groups per lane in stages B and C
  static constexpr int VR = E >= 8 ? E : 8;      // registers a lane needs in stages B / C (E = 4: 32 lanes x 8)
  static constexpr int active = 8 * E < 64 ? 8 * E : 64;   // lanes that take part in stages B / C
  static constexpr int lds_complex = E * kRow1;          // = 8 E * kRow2; >= N (exchange 3), >= (N/2+1) floats x 2
  static_assert(E == 4 || E == 8 || E == 16 || E == 32, "n_fft = 256, 512, 1024 or 2048");
  static_assert(E * kRow1 == 8 * E * kRow2, "one region serves both exchanges");
};

using C32 = cplx<float>;

// cos / sin of 2 pi q / 32
AAMD_HD constexpr float cos32(int q) {
  constexpr float c[32] = {1.0f, 0.98078528f, 0.923879533f, 0.831469612f, 0.707106781f, 0.555570233f, 0.382683432f,
                           0.195090322f, 0.0f, -0.195090322f, -0.382683432f, -0.555570233f, -0.707106781f,
                           -0.831469612f, -0.923879533f, -0.98078528f, -1.0f, -0.98078528f, -0.923879533f,
                           -0.831469612f, -0.707106781f, -0.555570233f, -0.382683432f, -0.195090322f, 0.0f,
                           0.195090322f, 0.382683432f, 0.555570233f, 0.707106781f, 0.831469612f, 0.923879533f,
                           0.98078528f};
  return c[q & 31];
}
AAMD_HD constexpr float sin32(int q) { return cos32(q - 8); }

// in-register forward DFT of size M (natural order in and out): decimation in time on top of bfly8
template <int M>
struct Dft {
  static AAMD_HD void run(C32* v) {
    constexpr int H = M / 2;
    C32 e[H], o[H];
#pragma unroll
    for (int i = 0; i < H; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    Dft<H>::run(e);
    Dft<H>::run(o);
#pragma unroll
    for (int k = 0; k < H; ++k) {
      C32 t;
      if (k == 0) {
        t = o[0];
      } else if (2 * k == H) {
        t = C32{o[k].y, -o[k].x};                       // W_M^(M/4) = -i
      } else {
        const float c = cos32(k * (32 / M)), s = sin32(k * (32 / M));   // W_M^k = c - i s
        t = C32{o[k].x * c + o[k].y * s, o[k].y * c - o[k].x * s};
      }
      v[k] = cadd(e[k], t);
      v[k + H] = csub(e[k], t);
    }
  }
};
template <>
struct Dft<8> {
  static AAMD_HD void run(C32* v) { bfly8<float>(v); }
};
template <>
struct Dft<4> {
  static AAMD_HD void run(C32* v) { bfly4<float>(v); }
};

// per-lane constants
template <int E>
struct LaneTab {
  float win[E];        // 0.5 * scale * window[l + 64 e]
  C32 twA[E];          // W_N^(l k1)              (twA[0] unused)
  C32 twB[8];          // W_64^(l1 k2a), l1 = l & 7 (twB[0] unused)
};

template <int E>
AAMD_HD void lane_tab(int lane, const float* window, const C32* tw /* W_N^m, m < N */, float scale, LaneTab<E>& lt) {
  constexpr int N = Cfg<E>::N;
#pragma unroll
  for (int e = 0; e < E; ++e) lt.win[e] = window[lane + 64 * e] * (0.5f * scale);
#pragma unroll
  for (int k1 = 0; k1 < E; ++k1) lt.twA[k1] = tw[(lane * k1) % N];
#pragma unroll
  for (int k = 0; k < 8; ++k) lt.twB[k] = tw[(E * (lane & 7) * k) % N];
}

// ---- stage A input: z[l + 64 e] = w (a + i b) ---------------------------------------------------------
// raw samples of the pair (no window yet): the kernel fetches them one pair ahead
template <int E>
AAMD_HD void load_raw(int lane, const StftGeom& g, const float* wav_row, int64_t ta, float* ra, float* rb) {
  constexpr int N = Cfg<E>::N;
  const int64_t tb = ta + 1;
  const int64_t cpad = g.center ? N / 2 : 0;
  const int64_t ba = ta * (int64_t)g.hop - cpad - g.pad, bb = ba + g.hop;
  const bool vb = tb < g.n_frames;
  if (vb && ba >= 0 && bb + N <= g.length) {                     // interior pair (wave-uniform): coalesced loads
    const float* pa = wav_row + ba;
    const float* pb = wav_row + bb;
#pragma unroll
    for (int e = 0; e < E; ++e) { ra[e] = (pa + 64 * e)[lane]; rb[e] = (pb + 64 * e)[lane]; }
    return;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int n = lane + 64 * e;
    ra[e] = stft_sample<float>(g, wav_row, ta, n);
    rb[e] = vb ? stft_sample<float>(g, wav_row, tb, n) : 0.0f;
  }
}
template <int E>
AAMD_HD void apply_window(const LaneTab<E>& lt, const float* ra, const float* rb, C32* v) {
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = C32{ra[e] * lt.win[e], rb[e] * lt.win[e]};
}
template <int E>
AAMD_HD void load_pair(int lane, const StftGeom& g, const float* wav_row, int64_t ta, const LaneTab<E>& lt, C32* v) {
  float ra[E], rb[E];
  load_raw<E>(lane, g, wav_row, ta, ra, rb);
  apply_window<E>(lt, ra, rb, v);
}

template <int E>
AAMD_HD void stage_a(const LaneTab<E>& lt, C32* v) {
  Dft<E>::run(v);
#pragma unroll
  for (int k1 = 1; k1 < E; ++k1) v[k1] = cmul(v[k1], lt.twA[k1]);
}

template <int E>
AAMD_HD void xch1_write(int lane, const C32* v, C32* lds) {
#pragma unroll
  for (int k1 = 0; k1 < E; ++k1) lds[k1 * kRow1 + lane] = v[k1];
}
template <int E>
AAMD_HD void xch1_read(int lane, const C32* lds, C32* v) {
  const int l1 = lane & 7, k1lo = lane >> 3;
  if (lane >= Cfg<E>::active) return;
#pragma unroll
  for (int h = 0; h < Cfg<E>::G; ++h)
#pragma unroll
    for (int l2 = 0; l2 < 8; ++l2) v[8 * h + l2] = lds[(k1lo + 8 * h) * kRow1 + l1 + 8 * l2];
}

template <int E>
AAMD_HD void stage_b(const LaneTab<E>& lt, C32* v) {
#pragma unroll
  for (int h = 0; h < Cfg<E>::G; ++h) {
    bfly8<float>(v + 8 * h);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[8 * h + k] = cmul(v[8 * h + k], lt.twB[k]);
  }
}

template <int E>
AAMD_HD void xch2_write(int lane, const C32* v, C32* lds) {
  const int l1 = lane & 7, k1lo = lane >> 3;
  if (lane >= Cfg<E>::active) return;
#pragma unroll
  for (int h = 0; h < Cfg<E>::G; ++h)
#pragma unroll
    for (int k2a = 0; k2a < 8; ++k2a) lds[(k1lo + 8 * h + E * k2a) * kRow2 + l1] = v[8 * h + k2a];
}
template <int E>
AAMD_HD void xch2_read(int lane, const C32* lds, C32* v) {
  if (lane >= Cfg<E>::active) return;
#pragma unroll
  for (int r = 0; r < Cfg<E>::G; ++r)
#pragma unroll
    for (int l1 = 0; l1 < 8; ++l1) v[8 * r + l1] = lds[(lane + 64 * r) * kRow2 + l1];
}

// stage C: z[j] = Z[lane + 64 j]
template <int E>
AAMD_HD void stage_c(C32* v, C32* z) {
  constexpr int G = Cfg<E>::G;
#pragma unroll
  for (int r = 0; r < G; ++r) {
    bfly8<float>(v + 8 * r);
#pragma unroll
    for (int k2b = 0; k2b < 8; ++k2b) z[r + G * k2b] = v[8 * r + k2b];
  }
}

// E = 4 (n_fft = 256) only: stage C leaves Z[lane + 32 j], j < 8, on 32 lanes; one more LDS pass spreads it to the
// layout every epilogue expects, Z[lane + 64 j], j < 4, on 64 lanes
template <int E>
AAMD_HD void redist_write(int lane, const C32* z, C32* lds) {
  if (lane >= Cfg<E>::active) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) lds[lane + Cfg<E>::active * j] = z[j];
}
template <int E>
AAMD_HD void redist_read(int lane, const C32* lds, C32* z) {
#pragma unroll
  for (int j = 0; j < E; ++j) z[j] = lds[lane + 64 * j];
}

// exchange 3: upper half in natural order, then the partner Z[N - k] of every bin this lane finishes
template <int E>
AAMD_HD void xch3_write(int lane, const C32* z, C32* lds) {
#pragma unroll
  for (int j = E / 2; j < E; ++j) lds[lane + 64 * j] = z[j];
}
// A, B = the two real spectra at bin k (the 0.5 * scale factor is already in the window)
AAMD_HD void separate(C32 zk, C32 zm, C32& A, C32& B) {
  A = C32{zk.x + zm.x, zk.y - zm.y};        // (Z[k] + conj Z[N-k])
  B = C32{zk.y + zm.y, zm.x - zk.x};        // (Z[k] - conj Z[N-k]) / i
}
// bins of this lane: k = lane + 64 j, j < E/2, plus k = N/2 on lane 0 (slot E/2)
template <int E>
AAMD_HD void finish_bins(int lane, const C32* z, const C32* lds, C32* A, C32* B) {
  constexpr int N = Cfg<E>::N;
#pragma unroll
  for (int j = 0; j < E / 2; ++j) {
    const int k = lane + 64 * j;
    const C32 zm = (k == 0) ? z[0] : lds[N - k];
    separate(z[j], zm, A[j], B[j]);
  }
  separate(z[E / 2], z[E / 2], A[E / 2], B[E / 2]);     // k = N/2 (meaningful on lane 0 only)
}

// ---- epilogues ---------------------------------------------------------------------------------------
template <int E>
AAMD_HD void store_spec(int lane, const StftGeom& g, const C32* A, const C32* B, int64_t ta, float* out_row) {
  constexpr int N = Cfg<E>::N;
  const bool vb = ta + 1 < g.n_frames;
  if (g.power <= 0.0f) {                                   // complex: interleaved (re, im)
    C32* oa = reinterpret_cast<C32*>(out_row + ta * (int64_t)(N + 2));
    C32* ob = oa + (N / 2 + 1);
#pragma unroll
    for (int j = 0; j < E / 2; ++j) {
      (oa + 64 * j)[lane] = A[j];
      if (vb) (ob + 64 * j)[lane] = B[j];
    }
    if (lane == 0) {
      oa[N / 2] = A[E / 2];
      if (vb) ob[N / 2] = B[E / 2];
    }
    return;
  }
  float* oa = out_row + ta * (int64_t)(N / 2 + 1);
  float* ob = oa + (N / 2 + 1);
#pragma unroll
  for (int j = 0; j < E / 2; ++j) {
    (oa + 64 * j)[lane] = mag_pow(A[j].x, A[j].y, g.power);
    if (vb) (ob + 64 * j)[lane] = mag_pow(B[j].x, B[j].y, g.power);
  }
  if (lane == 0) {
    oa[N / 2] = mag_pow(A[E / 2].x, A[E / 2].y, g.power);
    if (vb) ob[N / 2] = mag_pow(B[E / 2].x, B[E / 2].y, g.power);
  }
}

// power rows of the pair in LDS, interleaved per bin: P[k] = (|A[k]|^p, |B[k]|^p), one zero entry after the last bin
template <int E>
AAMD_HD void power_rows(int lane, const StftGeom& g, const C32* A, const C32* B, F2* P /* [N/2 + 2] */) {
  constexpr int F = Cfg<E>::N / 2 + 1;
#pragma unroll
  for (int j = 0; j < E / 2; ++j) {
    F2 p;
    p.x = mag_pow(A[j].x, A[j].y, g.power);
    p.y = mag_pow(B[j].x, B[j].y, g.power);
    P[lane + 64 * j] = p;
  }
  if (lane == 0) {
    F2 p;
    p.x = mag_pow(A[E / 2].x, A[E / 2].y, g.power);
    p.y = mag_pow(B[E / 2].x, B[E / 2].y, g.power);
    P[F - 1] = p;
    p.x = 0.0f; p.y = 0.0f;
    P[F] = p;
  }
}

// Banded filterbank staged in LDS once per workgroup (dense rows, odd stride, zero padded so that a band can be
// walked two taps at a time): [n_mels][mel_stride] weights, then lo[n_mels], width[n_mels].  Used when it fits
// kMelLdsBytes; otherwise the rows are read from global memory, tap by tap.
constexpr int kMelLdsBytes = 32 * 1024;
AAMD_HD int mel_stride(int max_width) { return (max_width + 2) | 1; }
AAMD_HD int mel_lds_floats(int n_mels, int max_width) { return n_mels * mel_stride(max_width) + 2 * n_mels; }
AAMD_HD bool mel_in_lds(int n_mels, int max_width) {
  return (size_t)mel_lds_floats(n_mels, max_width) * sizeof(float) <= (size_t)kMelLdsBytes;
}
AAMD_HD void mel_stage(int tid, int nthr, const MelBandsDev& mb, float* tab) {
  const int ms = mel_stride(mb.max_width);
  for (int i = tid; i < mb.n_mels * ms; i += nthr) {
    const int m = i / ms, j = i - m * ms;
    tab[i] = j < mb.max_width ? mb.weights[(int64_t)m * mb.max_width + j] : 0.0f;
  }
  int* lo = reinterpret_cast<int*>(tab + mb.n_mels * ms);
  for (int m = tid; m < mb.n_mels; m += nthr) { lo[m] = mb.lo[m]; lo[mb.n_mels + m] = mb.width[m]; }
}

// out[t][m] = sum_i w[m][i] P[t][lo[m] + i] for both frames of the pair: lane -> mel (neighbouring lanes hold
// neighbouring mels: similar band widths, and the two output rows leave as coalesced stores)
template <int E>
AAMD_HD void mel_rows(int lane, const StftGeom& g, const MelBandsDev& mb, const float* tab /* LDS table or null */,
                      const F2* P, int64_t ta, float* out_row, int m_end) {
  const bool vb = ta + 1 < g.n_frames;
  for (int m = lane; m < m_end; m += 64) {
    float acc_a = 0.0f, acc_b = 0.0f;
    if (tab) {
      const int ms = mel_stride(mb.max_width);
      const int* lo_tab = reinterpret_cast<const int*>(tab + mb.n_mels * ms);
      const int lo = lo_tab[m], w = lo_tab[mb.n_mels + m];
      const float* wt = tab + m * ms;
      const F2* Pf = P + lo;
      for (int i = 0; i < w; i += 2) {               // the table and P are zero padded past the band
        const F2 p0 = Pf[i], p1 = Pf[i + 1];
        const float w0 = wt[i], w1 = wt[i + 1];
        acc_a += w0 * p0.x; acc_b += w0 * p0.y;
        acc_a += w1 * p1.x; acc_b += w1 * p1.y;
      }
    } else {
      const int lo = mb.lo[m], w = mb.width[m];
      const float* wt = mb.weights + (int64_t)m * mb.max_width;
      for (int i = 0; i < w; ++i) {
        const F2 p = P[lo + i];
        acc_a += wt[i] * p.x; acc_b += wt[i] * p.y;
      }
    }
    out_row[ta * (int64_t)mb.n_mels + m] = acc_a;
    if (vb) out_row[(ta + 1) * (int64_t)mb.n_mels + m] = acc_b;
  }
}

// Round 6: the LAST round of the walk holds the widest bands on the fewest lanes (80 mels: 16 lanes walk 13 two-tap steps for
// n_fft = 512, 26 for 1024, while 48 lanes are masked off -- and an LDS instruction costs its passes whatever the mask).  Its
// R = n_mels - 64 floor((n_mels - 1) / 64) mels get G = 2 / 4 / 8 lanes each (the largest power of two with G R <= 64): lane ->
// (mel m0 + lane / G, part lane % G), part p walks the taps [p tl, (p + 1) tl) of the band (tl = ceil(w / G), even), the G partial
// sums meet in the part-0 lane through row_shl adds.  mel_tail_src names the source lane of a reduction step (-1: none) -- the
// kernel's DPP controls and the CPU replay's array reads are the same map.
AAMD_HD int mel_tail_lanes(int n_mels) {
  const int r = n_mels - 64 * ((n_mels - 1) / 64);
  int G = 1;
  while (G < 8 && 2 * G * r <= 64) G *= 2;
  return G;
}
AAMD_HD int mel_tail_first(int n_mels) { return 64 * ((n_mels - 1) / 64); }
AAMD_HD int mel_tail_src(int step /* 1, 2, 4 */, int lane) { return (lane & 15) + step < 16 ? lane + step : -1; }
AAMD_HD void mel_tail_partial(int lane, const MelBandsDev& mb, const float* tab, const F2* P, int G, float& pa, float& pb) {
  pa = 0.0f;
  pb = 0.0f;
  const int m = mel_tail_first(mb.n_mels) + lane / G, part = lane % G;
  if (m >= mb.n_mels) return;
  const int ms = mel_stride(mb.max_width);
  const int* lo_tab = reinterpret_cast<const int*>(tab + mb.n_mels * ms);
  const int lo = lo_tab[m], w = lo_tab[mb.n_mels + m];
  const int tl = ((w + G - 1) / G + 1) & ~1;
  const int i0 = part * tl, i1 = i0 + tl < w ? i0 + tl : w;
  const float* wt = tab + m * ms;
  const F2* Pf = P + lo;
  for (int i = i0; i < i1; i += 2) {                 // (an odd end reads one zero-padded tap, as the one-lane walk does)
    const F2 p0 = Pf[i], p1 = Pf[i + 1];
    const float w0 = wt[i], w1 = wt[i + 1];
    pa += w0 * p0.x; pb += w0 * p0.y;
    pa += w1 * p1.x; pb += w1 * p1.y;
  }
}
AAMD_HD void mel_tail_store(int lane, const StftGeom& g, const MelBandsDev& mb, int G, float pa, float pb, int64_t ta,
                            float* out_row) {
  const int m = mel_tail_first(mb.n_mels) + lane / G;
  if (lane % G != 0 || m >= mb.n_mels) return;
  out_row[ta * (int64_t)mb.n_mels + m] = pa;
  if (ta + 1 < g.n_frames) out_row[(ta + 1) * (int64_t)mb.n_mels + m] = pb;
}

// ---- inverse: frames -> waveform (torch.istft numerator / envelope, or the adjoint of the onesided STFT) ------
// IFFT(Z) = conj(FFT(conj Z)) on the SAME stages: the lane builds conj Z[l + 64 e] from the two onesided spectra
// (Hermitian extension; `interior` = 1 for irfft, 0.5 for the adjoint), and ends with time samples n = lane + 64 j.
struct InvGeom {
  StftGeom g;            // length = output samples per row; n_frames, hop, pad, center, pad_mode as for the forward
  float interior;
};

template <int E>
AAMD_HD void inv_load(int lane, const InvGeom& ig, const C32* Sa, const C32* Sb /* null: no frame b */, C32* v) {
  constexpr int N = Cfg<E>::N;
  // all loads first, branch-free (a missing frame b re-reads frame a and is zeroed by `mb`): a branch between two loads
  // makes the compiler drain the memory counter, i.e. one round trip per load
  const float mb = Sb ? 1.0f : 0.0f;
  const C32* Sb2 = Sb ? Sb : Sa;
  C32 a[E], b[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = lane + 64 * e;
    const int kk = k < N - k ? k : N - k;                    // onesided bin that defines Z[k]
    a[e] = Sa[kk];
    b[e] = Sb2[kk];
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = lane + 64 * e;
    const int kk = k < N - k ? k : N - k;
    const bool edge = (kk == 0) | (2 * kk == N);
    const float wgt = edge ? 1.0f : ig.interior;
    const float sgn = edge ? 0.0f : (2 * k > N ? -1.0f : 1.0f);   // Im: ignored at DC / Nyquist, negated in the mirror half
    const float ar = a[e].x * wgt, ai = a[e].y * (wgt * sgn);
    const float br = b[e].x * (wgt * mb), bi = b[e].y * (wgt * mb * sgn);
    v[e] = C32{ar - bi, -(ai + br)};                         // conj(A + i B)
  }
}

// overlap-add of the pair's two frames: x_a[n] = Re, x_b[n] = -Im of the stage-C output, times the window
// (lt.win carries the output scale), through the forward's padding map run backwards (ola_target)
AAMD_HD int64_t inv_target(const StftGeom& g, int64_t u) {
  const int64_t L1 = g.length + 2 * (int64_t)g.pad;
  const int64_t i1 = u - (g.center ? g.n_fft / 2 : 0);
  const int64_t s1 = g.center ? pad_source_index(i1, L1, g.pad_mode) : i1;
  if (s1 < 0 || s1 >= L1) return -1;
  const int64_t s0 = s1 - g.pad;
  return (s0 >= 0 && s0 < g.length) ? s0 : -1;
}

template <int E, typename AddFn>
AAMD_HD void inv_store(int lane, const InvGeom& ig, const LaneTab<E>& lt, const C32* z, int64_t ta, bool vb,
                       const float* inv_env, float* out_row, AddFn add) {
  constexpr int N = Cfg<E>::N;
  const StftGeom& g = ig.g;
  const int64_t cpad = g.center ? N / 2 : 0;
  const int64_t ba = ta * (int64_t)g.hop - cpad - g.pad, bb = ba + g.hop;
  const bool interior = ba >= 0 && bb + N <= g.length;        // wave-uniform: no index folding anywhere in the pair
#pragma unroll
  for (int j = 0; j < E; ++j) {
    const int n = lane + 64 * j;
    float xa = z[j].x * lt.win[j], xb = -z[j].y * lt.win[j];
    int64_t ia, ib;
    if (interior) { ia = ba + n; ib = bb + n; }
    else { ia = inv_target(g, ta * (int64_t)g.hop + n); ib = vb ? inv_target(g, (ta + 1) * (int64_t)g.hop + n) : -1; }
    if (ia >= 0) add(out_row + ia, inv_env ? xa * inv_env[ia] : xa);
    if (vb && ib >= 0) add(out_row + ib, inv_env ? xb * inv_env[ib] : xb);
  }
}

// Run-based overlap-add: a wave walks a RUN of consecutive pairs of one row and sums their overlapping frames in a
// wave-private LDS ring (2 N floats, LDS float adds), writing every finished sample once.  Samples that only this run's
// interior frames touch are PLAIN stores; the halo a neighbouring run, an edge pair or a folded (reflected / wrapped)
// contribution can also reach is added atomically.  Cuts the global atomics of the pair-at-a-time scheme (every sample
// hit N / hop times) to the ~N / (run_len hop) halo share.
struct RunPlan {
  int64_t excl_lo, excl_hi;      // samples s with excl_lo <= s < excl_hi belong to this run alone
  int64_t pi_lo, pi_hi;          // interior pairs of the run (inclusive; pi_lo > pi_hi: none)
};
template <int E>
AAMD_HD RunPlan run_plan(const StftGeom& g, int64_t p_lo, int64_t p_hi /* exclusive */) {
  constexpr int N = Cfg<E>::N;
  const int64_t c = (g.center ? N / 2 : 0) + g.pad, hop = g.hop;
  // pair p is interior iff 2 p hop - c >= 0 and (2 p + 1) hop - c + N <= length (and frame 2 p + 1 exists)
  int64_t first = (c + 2 * hop - 1) / (2 * hop);
  int64_t last_t = (g.length + c - N) / hop;                       // last frame that ends inside the row
  if (g.length + c - N < 0) last_t = -1;
  if (last_t > g.n_frames - 1) last_t = g.n_frames - 1;
  int64_t last = (last_t - 1) / 2;                                   // 2 p + 1 <= last_t
  if (last_t < 1) last = -1;
  RunPlan rp;
  rp.pi_lo = p_lo > first ? p_lo : first;
  rp.pi_hi = (p_hi - 1) < last ? (p_hi - 1) : last;
  const int64_t tA = 2 * rp.pi_lo, tB = 2 * rp.pi_hi + 1;
  rp.excl_lo = (tA - 1) * hop + N - c;
  if (rp.excl_lo < c + 1) rp.excl_lo = c + 1;                        // folded / wrapped edge contributions land below c + 1 ...
  rp.excl_hi = (tB + 1) * hop - c;
  if (rp.excl_hi > g.length - c - 1) rp.excl_hi = g.length - c - 1;  // ... and above length - c - 1
  return rp;
}
// add the pair's two frames into the ring (positions are row-local output samples, ring index = s mod 2 N): plain
// read-add-write -- lanes of one frame touch distinct words, frame b starts after frame a's writes (in-order LDS);
// LDS float atomics cost ~64 cycles per instruction here and made the kernel slower than global atomics
template <int E>
AAMD_HD void ring_add(int lane, const LaneTab<E>& lt, const C32* z, int64_t sa, int hop, bool vb, float* ring) {
  constexpr int M = 2 * Cfg<E>::N - 1;
  float cur[E];
#pragma unroll
  for (int j = 0; j < E; ++j) cur[j] = ring[(sa + lane + 64 * j) & M];
#pragma unroll
  for (int j = 0; j < E; ++j) ring[(sa + lane + 64 * j) & M] = cur[j] + z[j].x * lt.win[j];
  if (!vb) return;
#pragma unroll
  for (int j = 0; j < E; ++j) cur[j] = ring[(sa + hop + lane + 64 * j) & M];
#pragma unroll
  for (int j = 0; j < E; ++j) ring[(sa + hop + lane + 64 * j) & M] = cur[j] - z[j].y * lt.win[j];
}
// write out (and clear) the finished samples [s0, s1)
template <int E, typename AddFn, typename StoreFn>
AAMD_HD void ring_flush(int lane, const RunPlan& rp, int64_t s0, int64_t s1, const float* inv_env, float* ring,
                        float* out_row, AddFn add, StoreFn store) {
  constexpr int M = 2 * Cfg<E>::N - 1;
  for (int64_t s = s0 + lane; s < s1; s += 64) {
    float v = ring[s & M];
    ring[s & M] = 0.0f;
    if (inv_env) v *= inv_env[s];
    if (s >= rp.excl_lo && s < rp.excl_hi) store(out_row + s, v);
    else add(out_row + s, v);
  }
}

// ---- Kaldi-compatible front-end (compliance/kaldi.py:154-217 _get_window, :229-315 spectrogram, :514-645 fbank) ------
// Frames of `win` samples every `shift` samples (snip_edges, or Kaldi's edge-repeating reflection), per frame:
// DC removal, raw log-energy, pre-emphasis, window, zero padding to N = 64 E, power spectrum, then either
// log-power rows or mel-bank energies -- the same register FFT, a different way in and out.
struct KaldiGeom {
  int64_t n_samples, n_frames;
  int32_t shift, win;                 // frame shift / length in samples (win <= N)
  int32_t snip_edges, pad_left;       // !snip_edges: frame t starts at t * shift - pad_left of the reflected signal
  float preemph;                      // 0 = off
  int32_t remove_dc, raw_energy;
  float log_energy_floor;             // log(energy_floor), or -inf when energy_floor == 0
  float eps;                          // torch.finfo(float32).eps
  int32_t use_power, use_log;
  int32_t energy_col, first_col, n_cols;   // output row: [n_cols]; energy_col < 0: no energy column
  const float* noise;                 // dither: unit Gaussian noise [n_frames][win] (kaldi.py:180-183), or null
  float dither;
  int64_t n_utt, utt_stride;          // batch extension: n_utt waveforms of n_samples each, utt_stride apart; outputs / noise per utterance
};

// sample j of frame t (kaldi.py:44-83 _get_strided): snip_edges reads the signal as is; otherwise the signal is
// extended by its mirror image WITH the edge sample repeated ([2, 1, 0, 0, 1, 2])
AAMD_HD float kaldi_sample(const KaldiGeom& kg, const float* x, int64_t t, int j) {
  int64_t i = t * (int64_t)kg.shift + j - (kg.snip_edges ? 0 : kg.pad_left);
  if (i < 0) i = -1 - i;
  if (i >= kg.n_samples) i = 2 * kg.n_samples - 1 - i;
  const float v = (i >= 0 && i < kg.n_samples) ? x[i] : 0.0f;
  // dither is added to the FRAMED signal (one independent draw per frame and tap, also where frames overlap)
  return kg.noise != nullptr ? v + kg.dither * kg.noise[t * (int64_t)kg.win + j] : v;
}

// raw samples of the frame and their predecessors (for the pre-emphasis), this lane's E taps; taps >= win are 0
template <int E>
AAMD_HD void kaldi_load(int lane, const KaldiGeom& kg, const float* x, int64_t t, float* r, float* rp) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = lane + 64 * e;
    const bool in = j < kg.win && t < kg.n_frames;
    r[e] = in ? kaldi_sample(kg, x, t, j) : 0.0f;
    rp[e] = in ? kaldi_sample(kg, x, t, j > 0 ? j - 1 : 0) : 0.0f;
  }
}
template <int E>
AAMD_HD float kaldi_partial_sum(const float* r) {
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < E; ++e) s += r[e];
  return s;
}
template <int E>
AAMD_HD float kaldi_partial_sumsq(int lane, const KaldiGeom& kg, const float* r, float mean) {
  float s = 0.0f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const float d = (lane + 64 * e < kg.win) ? r[e] - mean : 0.0f;
    s += d * d;
  }
  return s;
}
// DC removal + pre-emphasis + window for one frame: y[j] = ((x[j] - mean) - c (x[max(j-1, 0)] - mean)) w[j]
template <int E>
AAMD_HD void kaldi_shape(int lane, const KaldiGeom& kg, const float* win, const float* r, const float* rp, float mean,
                         float* y) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const bool in = lane + 64 * e < kg.win;
    const float d = r[e] - mean, dp = rp[e] - mean;
    y[e] = in ? (d - kg.preemph * dp) * win[e] : 0.0f;
  }
}
AAMD_HD float kaldi_log_energy(const KaldiGeom& kg, float sumsq) {
  const float le = log(fmax(sumsq, kg.eps));
  return fmax(le, kg.log_energy_floor);
}
// spectrogram rows (kaldi.py:306-315): log(max(|X|^2, eps)), column 0 replaced by the log energy
template <int E>
AAMD_HD void kaldi_store_spec(int lane, const KaldiGeom& kg, const C32* A, const C32* B, int64_t ta, float ea, float eb,
                              float* out) {
  constexpr int N = Cfg<E>::N;
  const bool vb = ta + 1 < kg.n_frames;
  float* oa = out + ta * (int64_t)(N / 2 + 1);
  float* ob = oa + (N / 2 + 1);
#pragma unroll
  for (int j = 0; j < E / 2; ++j) {
    const int k = lane + 64 * j;
    const float pa = log(fmax(A[j].x * A[j].x + A[j].y * A[j].y, kg.eps));
    const float pb = log(fmax(B[j].x * B[j].x + B[j].y * B[j].y, kg.eps));
    oa[k] = k == 0 ? ea : pa;
    if (vb) ob[k] = k == 0 ? eb : pb;
  }
  if (lane == 0) {
    oa[N / 2] = log(fmax(A[E / 2].x * A[E / 2].x + A[E / 2].y * A[E / 2].y, kg.eps));
    if (vb) ob[N / 2] = log(fmax(B[E / 2].x * B[E / 2].x + B[E / 2].y * B[E / 2].y, kg.eps));
  }
}
// |X| or |X|^2 rows of the pair in LDS, interleaved per bin (as power_rows)
template <int E>
AAMD_HD void kaldi_power_rows(int lane, const KaldiGeom& kg, const C32* A, const C32* B, F2* P) {
  constexpr int F = Cfg<E>::N / 2 + 1;
#pragma unroll
  for (int j = 0; j <= E / 2; ++j) {
    if (j == E / 2 && lane != 0) continue;
    F2 p;
    p.x = A[j].x * A[j].x + A[j].y * A[j].y;
    p.y = B[j].x * B[j].x + B[j].y * B[j].y;
    if (!kg.use_power) { p.x = sqrt(p.x); p.y = sqrt(p.y); }
    P[j == E / 2 ? F - 1 : lane + 64 * j] = p;
  }
  if (lane == 0) { F2 z; z.x = 0.0f; z.y = 0.0f; P[F] = z; }
}
// mel-bank rows (kaldi.py:625-643): sum over the band, log(max(., eps)), plus the energy column
template <int E>
AAMD_HD void kaldi_fbank_rows(int lane, const KaldiGeom& kg, const MelBandsDev& mb, const F2* P, int64_t ta, float ea,
                              float eb, float* out) {
  const bool vb = ta + 1 < kg.n_frames;
  float* oa = out + ta * (int64_t)kg.n_cols;
  float* ob = oa + kg.n_cols;
  for (int m = lane; m < mb.n_mels; m += 64) {
    const int lo = mb.lo[m], w = mb.width[m];
    const float* wt = mb.weights + (int64_t)m * mb.max_width;
    float acc_a = 0.0f, acc_b = 0.0f;
    for (int i = 0; i < w; ++i) {
      const F2 p = P[lo + i];
      acc_a += wt[i] * p.x; acc_b += wt[i] * p.y;
    }
    if (kg.use_log) { acc_a = log(fmax(acc_a, kg.eps)); acc_b = log(fmax(acc_b, kg.eps)); }
    oa[kg.first_col + m] = acc_a;
    if (vb) ob[kg.first_col + m] = acc_b;
  }
  if (lane == 0 && kg.energy_col >= 0) {
    oa[kg.energy_col] = ea;
    if (vb) ob[kg.energy_col] = eb;
  }
}

#if defined(__HIPCC__)
AAMD_D void wave_lds_sync() {     // program order within the wave is the only ordering these exchanges need
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lane l <- lane l + N of its row of 16 (zeros past the row's end)
template <int N>
AAMD_D float row_shl0(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}

template <int E, int EPI>
__global__ void __launch_bounds__(64 * kWaves)
stft_pow2_kernel(StftGeom g, const float* __restrict__ wav, const float* __restrict__ window,
                 const C32* __restrict__ tw, MelBandsDev mb, float* __restrict__ out, int64_t pairs_per_row,
                 int64_t n_pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p2[];
  constexpr int N = Cfg<E>::N;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  C32* lds = reinterpret_cast<C32*>(smem_p2) + wave * Cfg<E>::lds_complex;
  const float* mel_tab = nullptr;
  if (EPI == EPI_MEL && mel_in_lds(mb.n_mels, mb.max_width)) {
    float* tab = reinterpret_cast<float*>(reinterpret_cast<C32*>(smem_p2) + kWaves * Cfg<E>::lds_complex);
    mel_stage(threadIdx.x, blockDim.x, mb, tab);
    __syncthreads();                       // the only workgroup barrier: before the persistent loop
    mel_tab = tab;
  }
  const int tail_G = (EPI == EPI_MEL && mel_tab != nullptr) ? mel_tail_lanes(mb.n_mels) : 1;      // lanes per mel in the last round
  LaneTab<E> lt;
  lane_tab<E>(lane, window, tw, g.scale, lt);
  const int64_t n_waves = (int64_t)gridDim.x * kWaves;
  const int opf = (EPI == EPI_MEL) ? mb.n_mels : (g.power <= 0.0f ? N + 2 : N / 2 + 1);
  constexpr bool kPrefetch = E <= 16;        // E = 32 has no registers to spare
  float ra[E], rb[E];
  const int64_t pair0 = (int64_t)blockIdx.x * kWaves + wave;
  if (kPrefetch && pair0 < n_pairs) {
    const int64_t row = pair0 / pairs_per_row;
    load_raw<E>(lane, g, wav + row * g.row_stride, 2 * (pair0 - row * pairs_per_row), ra, rb);
  }
#pragma unroll 1
  for (int64_t pair = pair0; pair < n_pairs; pair += n_waves) {
    const int64_t row = pair / pairs_per_row;
    const int64_t ta = 2 * (pair - row * pairs_per_row);
    C32 v[Cfg<E>::VR], z[Cfg<E>::VR];
    if (kPrefetch) {
      apply_window<E>(lt, ra, rb, v);
      const int64_t nxt = pair + n_waves;                       // in flight during this pair's FFT
      if (nxt < n_pairs) {
        const int64_t nrow = nxt / pairs_per_row;
        load_raw<E>(lane, g, wav + nrow * g.row_stride, 2 * (nxt - nrow * pairs_per_row), ra, rb);
      }
    } else {
      load_pair<E>(lane, g, wav + row * g.row_stride, ta, lt, v);
    }
    stage_a<E>(lt, v);
    wave_lds_sync();                       // the previous pair's epilogue reads are done
    xch1_write<E>(lane, v, lds);
    wave_lds_sync();
    xch1_read<E>(lane, lds, v);
    stage_b<E>(lt, v);
    wave_lds_sync();
    xch2_write<E>(lane, v, lds);
    wave_lds_sync();
    xch2_read<E>(lane, lds, v);
    stage_c<E>(v, z);
    if (E < 8) { wave_lds_sync(); redist_write<E>(lane, z, lds); wave_lds_sync(); redist_read<E>(lane, lds, z); }
    wave_lds_sync();
    xch3_write<E>(lane, z, lds);
    wave_lds_sync();
    C32 A[E / 2 + 1], B[E / 2 + 1];
    finish_bins<E>(lane, z, lds, A, B);
    float* out_row = out + row * g.n_frames * (int64_t)opf;
    if (EPI == EPI_SPEC) {
      store_spec<E>(lane, g, A, B, ta, out_row);
    } else {
      wave_lds_sync();
      F2* P = reinterpret_cast<F2*>(lds);
      power_rows<E>(lane, g, A, B, P);
      wave_lds_sync();
      mel_rows<E>(lane, g, mb, mel_tab, P, ta, out_row, tail_G > 1 ? mel_tail_first(mb.n_mels) : mb.n_mels);
      if (tail_G > 1) {
        // (an opaque copy of the lane number: the tail's lane-derived addresses hoisted out of the persistent loop cost E = 8 its
        // fourth wave per SIMD -- 134 registers)
        int tl_lane = lane;
        asm volatile("" : "+v"(tl_lane));
        float pa, pb;
        mel_tail_partial(tl_lane, mb, mel_tab, P, tail_G, pa, pb);
        // (all lanes take part: the row_shl moves read zeros past the row's end; a group never crosses a row of 16)
        pa += row_shl0<1>(pa); pb += row_shl0<1>(pb);
        if (tail_G > 2) { pa += row_shl0<2>(pa); pb += row_shl0<2>(pb); }
        if (tail_G > 4) { pa += row_shl0<4>(pa); pb += row_shl0<4>(pb); }
        mel_tail_store(tl_lane, g, mb, tail_G, pa, pb, ta, out_row);
      }
    }
  }
}
template <int E>
__global__ void __launch_bounds__(64 * kWaves)
istft_pow2_kernel(InvGeom ig, const C32* __restrict__ spec, const float* __restrict__ window,
                  const C32* __restrict__ tw, const float* __restrict__ inv_env, float* __restrict__ out,
                  float out_scale, int64_t pairs_per_row, int64_t n_pairs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p2[];
  constexpr int F = Cfg<E>::N / 2 + 1;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  C32* lds = reinterpret_cast<C32*>(smem_p2) + wave * Cfg<E>::lds_complex;
  LaneTab<E> lt;
  lane_tab<E>(lane, window, tw, 2.0f * out_scale, lt);        // lane_tab folds 0.5 * scale into the window
  const int64_t n_waves = (int64_t)gridDim.x * kWaves;
  auto add = [](float* p, float v) { atomicAdd(p, v); };
#pragma unroll 1
  for (int64_t pair = (int64_t)blockIdx.x * kWaves + wave; pair < n_pairs; pair += n_waves) {
    const int64_t row = pair / pairs_per_row;
    const int64_t ta = 2 * (pair - row * pairs_per_row);
    const bool vb = ta + 1 < ig.g.n_frames;
    const C32* Sa = spec + (row * ig.g.n_frames + ta) * (int64_t)F;
    C32 v[Cfg<E>::VR], z[Cfg<E>::VR];
    inv_load<E>(lane, ig, Sa, vb ? Sa + F : nullptr, v);
    stage_a<E>(lt, v);
    wave_lds_sync();
    xch1_write<E>(lane, v, lds);
    wave_lds_sync();
    xch1_read<E>(lane, lds, v);
    stage_b<E>(lt, v);
    wave_lds_sync();
    xch2_write<E>(lane, v, lds);
    wave_lds_sync();
    xch2_read<E>(lane, lds, v);
    stage_c<E>(v, z);
    if (E < 8) { wave_lds_sync(); redist_write<E>(lane, z, lds); wave_lds_sync(); redist_read<E>(lane, lds, z); }
    inv_store<E>(lane, ig, lt, z, ta, vb, inv_env, out + row * ig.g.length, add);
  }
}
template <int E>
__global__ void __launch_bounds__(64 * kWaves)
istft_pow2_run_kernel(InvGeom ig, const C32* __restrict__ spec, const float* __restrict__ window,
                      const C32* __restrict__ tw, const float* __restrict__ inv_env, float* __restrict__ out,
                      float out_scale, int64_t pairs_per_row, int64_t runs_per_row, int64_t n_runs, int run_len) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p2[];
  constexpr int N = Cfg<E>::N, F = N / 2 + 1;
  constexpr int kWaveFloats = 2 * Cfg<E>::lds_complex + 2 * N;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* wbase = reinterpret_cast<float*>(smem_p2) + wave * kWaveFloats;
  C32* lds = reinterpret_cast<C32*>(wbase);
  float* ring = wbase + 2 * Cfg<E>::lds_complex;
  for (int i = lane; i < 2 * N; i += 64) ring[i] = 0.0f;
  LaneTab<E> lt;
  lane_tab<E>(lane, window, tw, 2.0f * out_scale, lt);
  const StftGeom& g = ig.g;
  const int64_t c = (g.center ? N / 2 : 0) + g.pad;
  const int64_t n_waves = (int64_t)gridDim.x * kWaves;
  auto add = [](float* p, float v) { atomicAdd(p, v); };
  auto store = [](float* p, float v) { *p = v; };
  wave_lds_sync();
#pragma unroll 1
  for (int64_t run = (int64_t)blockIdx.x * kWaves + wave; run < n_runs; run += n_waves) {
    const int64_t row = run / runs_per_row;
    const int64_t p_lo = (run - row * runs_per_row) * run_len;
    const int64_t p_hi = p_lo + run_len < pairs_per_row ? p_lo + run_len : pairs_per_row;
    const RunPlan rp = run_plan<E>(g, p_lo, p_hi);
    float* out_row = out + row * g.length;
    const float* env = inv_env;
    int64_t flushed = 0;
#pragma unroll 1
    for (int64_t p = p_lo; p < p_hi; ++p) {
      const int64_t ta = 2 * p;
      const bool vb = ta + 1 < g.n_frames;
      const C32* Sa = spec + (row * g.n_frames + ta) * (int64_t)F;
      C32 v[Cfg<E>::VR], z[Cfg<E>::VR];
      inv_load<E>(lane, ig, Sa, vb ? Sa + F : nullptr, v);
      stage_a<E>(lt, v);
      wave_lds_sync();
      xch1_write<E>(lane, v, lds);
      wave_lds_sync();
      xch1_read<E>(lane, lds, v);
      stage_b<E>(lt, v);
      wave_lds_sync();
      xch2_write<E>(lane, v, lds);
      wave_lds_sync();
      xch2_read<E>(lane, lds, v);
      stage_c<E>(v, z);
    if (E < 8) { wave_lds_sync(); redist_write<E>(lane, z, lds); wave_lds_sync(); redist_read<E>(lane, lds, z); }
      if (p >= rp.pi_lo && p <= rp.pi_hi) {                   // interior pair: through the ring
        const int64_t sa = ta * (int64_t)g.hop - c;
        if (p == rp.pi_lo) flushed = sa;
        ring_add<E>(lane, lt, z, sa, g.hop, true, ring);
        wave_lds_sync();
        const int64_t s1 = p == rp.pi_hi ? sa + g.hop + N : sa + 2 * (int64_t)g.hop;   // run end: everything left
        ring_flush<E>(lane, rp, flushed, s1, env, ring, out_row, add, store);
        flushed = s1;
        wave_lds_sync();
      } else {                                                // edge pair: index map per sample, atomics
        inv_store<E>(lane, ig, lt, z, ta, vb, env, out_row, add);
      }
    }
  }
}

AAMD_D float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// MODE 0: kaldi.spectrogram rows [N/2 + 1]; MODE 1: kaldi.fbank rows [n_cols]
template <int E, int MODE>
__global__ void __launch_bounds__(64 * kWaves)
kaldi_pow2_kernel(KaldiGeom kg, const float* wav, const float* __restrict__ window /* [N], 0 past win */,
                  const C32* __restrict__ tw, MelBandsDev mb, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_p2[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  C32* lds = reinterpret_cast<C32*>(smem_p2) + wave * Cfg<E>::lds_complex;
  LaneTab<E> lt;
  lane_tab<E>(lane, window, tw, 2.0f, lt);                   // win = window (lane_tab folds 0.5 * scale)
  const int64_t ppu = (kg.n_frames + 1) / 2;                  // frame pairs per utterance
  const int64_t n_pairs = ppu * kg.n_utt;
  const int64_t n_waves = (int64_t)gridDim.x * kWaves;
  const float inv_win = 1.0f / (float)kg.win;
  const float* const wav0 = wav;
  float* const out0 = out;
  const float* const noise0 = kg.noise;
  const int64_t out_row = MODE == 0 ? (Cfg<E>::N / 2 + 1) : kg.n_cols;
#pragma unroll 1
  for (int64_t gp = (int64_t)blockIdx.x * kWaves + wave; gp < n_pairs; gp += n_waves) {
    const int64_t utt = gp / ppu, pair = gp - utt * ppu;
    wav = wav0 + utt * kg.utt_stride;                         // this utterance's samples, rows and dither draws
    out = out0 + utt * kg.n_frames * out_row;
    kg.noise = noise0 ? noise0 + utt * kg.n_frames * kg.win : nullptr;
    const int64_t ta = 2 * pair;
    float ra[E], rpa[E], rb[E], rpb[E], ya[E], yb[E];
    kaldi_load<E>(lane, kg, wav, ta, ra, rpa);
    kaldi_load<E>(lane, kg, wav, ta + 1, rb, rpb);
    const float mean_a = kg.remove_dc ? wave_sum(kaldi_partial_sum<E>(ra)) * inv_win : 0.0f;
    const float mean_b = kg.remove_dc ? wave_sum(kaldi_partial_sum<E>(rb)) * inv_win : 0.0f;
    float ea = 0.0f, eb = 0.0f;
    if (kg.raw_energy) {
      ea = kaldi_log_energy(kg, wave_sum(kaldi_partial_sumsq<E>(lane, kg, ra, mean_a)));
      eb = kaldi_log_energy(kg, wave_sum(kaldi_partial_sumsq<E>(lane, kg, rb, mean_b)));
    }
    kaldi_shape<E>(lane, kg, lt.win, ra, rpa, mean_a, ya);
    kaldi_shape<E>(lane, kg, lt.win, rb, rpb, mean_b, yb);
    if (!kg.raw_energy) {
      ea = kaldi_log_energy(kg, wave_sum(kaldi_partial_sumsq<E>(lane, kg, ya, 0.0f)));
      eb = kaldi_log_energy(kg, wave_sum(kaldi_partial_sumsq<E>(lane, kg, yb, 0.0f)));
    }
    C32 v[Cfg<E>::VR], z[Cfg<E>::VR];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = C32{0.5f * ya[e], 0.5f * yb[e]};   // separate() returns 2 X
    stage_a<E>(lt, v);
    wave_lds_sync();
    xch1_write<E>(lane, v, lds);
    wave_lds_sync();
    xch1_read<E>(lane, lds, v);
    stage_b<E>(lt, v);
    wave_lds_sync();
    xch2_write<E>(lane, v, lds);
    wave_lds_sync();
    xch2_read<E>(lane, lds, v);
    stage_c<E>(v, z);
    if (E < 8) { wave_lds_sync(); redist_write<E>(lane, z, lds); wave_lds_sync(); redist_read<E>(lane, lds, z); }
    wave_lds_sync();
    xch3_write<E>(lane, z, lds);
    wave_lds_sync();
    C32 A[E / 2 + 1], B[E / 2 + 1];
    finish_bins<E>(lane, z, lds, A, B);
    if (MODE == 0) {
      kaldi_store_spec<E>(lane, kg, A, B, ta, ea, eb, out);
    } else {
      wave_lds_sync();
      F2* P = reinterpret_cast<F2*>(lds);
      kaldi_power_rows<E>(lane, kg, A, B, P);
      wave_lds_sync();
      kaldi_fbank_rows<E>(lane, kg, mb, P, ta, ea, eb, out);
    }
  }
}
#endif  // __HIPCC__

}  // namespace p2
}  // namespace aamd

// Headline kernel: fused MelSpectrogram for n_fft = 400, hop = 160 (the RNN-T / Whisper
// style front-end, BASELINE.json config 2), power = 2, centre + reflect padding.
//
// Design (MI355X, wave64; no workgroup barriers in the tile loop -- every wave owns private LDS):
//   * two REAL frames a, b = a + 1 are packed as one COMPLEX 400-point FFT  z = a + i b
//     (no redundant half-spectrum work, no post-twiddle multiply);
//   * 400 = 20 x 20 Cooley-Tukey.  A 20-lane group owns one frame pair; each lane runs a
//     20-point DFT entirely in registers (Good-Thomas 4x5: five radix-4 + four radix-5
//     butterflies, NO internal twiddles), multiplies by its W400^(b c) twiddles, and the
//     20x20 transposition goes once through LDS (row stride 22 complex: conflict free for
//     both the column write and the row read); a second in-register DFT-20 finishes the FFT;
//   * 3 pairs (60 lanes) = 6 frames per wave per tile; frame b starts 160 = 8*20 samples after
//     frame a, so a lane fetches 28 (not 40) strided samples per pair;
//   * the a/b spectra are separated with the conj-symmetry rule.  Lanes are laid out so that the
//     lane holding column 20-c sits next to the lane holding column c: the partner bin
//     Z[400-k] arrives through a DPP quad_perm [1,0,3,2] swap (wavefront shuffle on the VALU,
//     no LDS traffic); the two self-paired columns (0 and 10) are patched with selects;
//   * |A|^2, |B|^2 for k = 0..200 go to LDS interleaved as float2 (a, b) per bin;
//   * mel = banded reduction over that LDS power spectrum: lane (pair, slot) owns mels
//     slot + 20 r for BOTH frames of the pair; the band table (even-aligned band starts, zero
//     padded weights) lives in LDS, read as b64 (2 weights) + b128 (2 bins x 2 frames).
//
// Reference semantics: transforms/_transforms.py:612-622 (MelSpectrogram.forward),
// functional/functional.py:112-145, torch/functional.py:675-681; framing is bit-exact
// (index reflection i<0 -> -i, i>=L -> 2(L-1)-i).
#pragma once
#include "hd.h"
#include "stft_generic.h"

namespace aamd {
namespace m400 {

constexpr int kN = 400;
constexpr int kHop = 160;
constexpr int kPad = 200;
constexpr int kFramesPerWave = 6;
// LDS strides picked with tools/lds_conflicts.py (bank model of MI355X_MICROARCH.md):
constexpr int kTRow = 44;                      // dwords per transposition row: 20 complex + pad, 16-B aligned
                                               // rows -> conflict-free ds_read_b128; column writes 6 array
                                               // cycles = the ds_write_b64 issue cost
constexpr int kTPair = 20 * kTRow;             // 880 dwords per pair
constexpr int kLdsDwordsPerWave = 3 * kTPair;  // 2640 dwords = 10560 B (the P rows alias it)
constexpr int kPK = 212;                       // readable bins per P row: 201 + zeroed tail, even
constexpr int kPPair = 448;                    // dwords per pair of P rows (>= 2 * kPK)
constexpr int kMelSlots = 20;                  // mels per round
constexpr int kMelMaxRounds = 8;               // n_mels <= 160
constexpr int kMelMaxTaps = 64;                // widest padded band

static_assert(3 * kPPair <= kLdsDwordsPerWave && 2 * kPK <= kPPair, "P rows must fit in the transposition buffer");
static_assert(3 * (kPK - 201) <= 64, "one lane per tail bin");

// column held by the lane at position pi of a 20-lane group (pass 2), and its inverse:
// positions (0,1) = columns (0,10), then (2j, 2j+1) = (j, 20-j): lane ^ 1 holds column 20 - c.
AAMD_HD constexpr int col_of_pos(int pi) {
  return pi == 0 ? 0 : pi == 1 ? 10 : (pi & 1) ? 20 - (pi >> 1) : (pi >> 1);
}
AAMD_HD constexpr int pos_of_col(int c) {
  return c == 0 ? 0 : c == 10 ? 1 : c < 10 ? 2 * c : 2 * (20 - c) + 1;
}

// ---- banded filterbank in LDS (built once per launch by every workgroup) ------------------
struct MelTab {
  const float* w;    // [n_mels][ws]: weight of bin lo2[m] + j, zero outside the band
  const int* lo2;    // [n_mels]: even band start (<= first non-zero bin)
  const int* rw2;    // [n_rounds]: even tap count of the round (wave-uniform trip count)
  int n_mels, ws, n_rounds;
};

AAMD_HD int mel_rounds(int n_mels) { return (n_mels + kMelSlots - 1) / kMelSlots; }
// row stride: covers the widest even-aligned band, = 2 mod 4 (b64 rows of 20 mels hit distinct banks)
AAMD_HD int mel_ws(int max_width) {
  const int w2 = (max_width + 2) & ~1;
  return (w2 & 3) == 0 ? w2 + 2 : w2;
}
AAMD_HD int mel_tab_dwords(int n_mels, int max_width) {
  return n_mels * mel_ws(max_width) + n_mels + kMelMaxRounds;
}

AAMD_HD void mel_tab_build(int tid, int nthr, const MelBandsDev& mb, float* base, MelTab& mt) {
  mt.n_mels = mb.n_mels;
  mt.ws = mel_ws(mb.max_width);
  mt.n_rounds = mel_rounds(mb.n_mels);
  float* w = base;
  int* lo2 = reinterpret_cast<int*>(base + mb.n_mels * mt.ws);
  int* rw2 = lo2 + mb.n_mels;
  // every thread derives the per-round tap counts it needs (n_mels is small)
  for (int i = tid; i < mb.n_mels * mt.ws; i += nthr) {
    const int m = i / mt.ws, j = i - m * mt.ws;
    const int r = m / kMelSlots;
    int rw = 0;
    for (int q = r * kMelSlots; q < (r + 1) * kMelSlots && q < mb.n_mels; ++q) {
      const int e = (mb.width[q] + (mb.lo[q] & 1) + 1) & ~1;
      rw = e > rw ? e : rw;
    }
    int l2 = mb.lo[m] & ~1;
    if (l2 + rw > kPK) l2 = kPK - rw;
    const int off = mb.lo[m] - l2;
    w[i] = (j >= off && j - off < mb.width[m]) ? mb.weights[m * mb.max_width + (j - off)] : 0.0f;
    if (j == 0) lo2[m] = l2;
    if (j == 0 && m == r * kMelSlots) rw2[r] = rw;
  }
  mt.w = w; mt.lo2 = lo2; mt.rw2 = rw2;
}

struct LaneConst {
  float twr[19], twi[19];  // W400^(b*c), c = 1..19 (pass-1 role b = pi)
  float win[20];           // 0.5 * scale * window[b + 20 q]
  int p, pi, col;          // pair 0..2, position in the 20-lane group, pass-2 column
  int active;              // lanes 60..63 shadow lanes 40..43 but never store
};

AAMD_HD void lane_init(int lane, const float* window, const float* tw400, float scale,
                       LaneConst& c) {
  c.active = lane < 60;
  const int l = c.active ? lane : lane - 20;
  c.p = l / 20;
  c.pi = l - 20 * c.p;
  c.col = col_of_pos(c.pi);
#pragma unroll
  for (int s = 1; s < 20; ++s) {
    const int idx = (c.pi * s) % kN;
    c.twr[s - 1] = tw400[2 * idx];
    c.twi[s - 1] = tw400[2 * idx + 1];
  }
#pragma unroll
  for (int q = 0; q < 20; ++q) c.win[q] = window[c.pi + 20 * q] * (0.5f * scale);
}

// ---- in-register DFT-20, forward (e^{-2 pi i nk/20}), natural order in and out -----------
AAMD_HD void dft4(float ar, float ai, float br, float bi, float cr, float ci, float dr, float di,
                  float* o_r, float* o_i) {
  const float s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;
  const float s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
  o_r[0] = s0r + s2r; o_i[0] = s0i + s2i;
  o_r[1] = s1r + s3i; o_i[1] = s1i - s3r;   // s1 - i*s3
  o_r[2] = s0r - s2r; o_i[2] = s0i - s2i;
  o_r[3] = s1r - s3i; o_i[3] = s1i + s3r;   // s1 + i*s3
}

AAMD_HD void dft5(const float* xr, const float* xi, float* o_r, float* o_i) {
  constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f;
  constexpr float S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;
  const float t1r = xr[1] + xr[4], t1i = xi[1] + xi[4];
  const float t2r = xr[2] + xr[3], t2i = xi[2] + xi[3];
  const float t3r = xr[1] - xr[4], t3i = xi[1] - xi[4];
  const float t4r = xr[2] - xr[3], t4i = xi[2] - xi[3];
  const float m1r = xr[0] + C1 * t1r + C2 * t2r, m1i = xi[0] + C1 * t1i + C2 * t2i;
  const float m2r = xr[0] + C2 * t1r + C1 * t2r, m2i = xi[0] + C2 * t1i + C1 * t2i;
  const float u1r = S1 * t3r + S2 * t4r, u1i = S1 * t3i + S2 * t4i;
  const float u2r = S2 * t3r - S1 * t4r, u2i = S2 * t3i - S1 * t4i;
  o_r[0] = xr[0] + t1r + t2r; o_i[0] = xi[0] + t1i + t2i;
  o_r[1] = m1r + u1i; o_i[1] = m1i - u1r;   // m1 - i*u1
  o_r[4] = m1r - u1i; o_i[4] = m1i + u1r;
  o_r[2] = m2r + u2i; o_i[2] = m2i - u2r;   // m2 - i*u2
  o_r[3] = m2r - u2i; o_i[3] = m2i + u2r;
}

// Good-Thomas: n = (5 n1 + 4 n2) mod 20, k = (5 k1 + 16 k2) mod 20.
AAMD_HD void dft20(const float (&xr)[20], const float (&xi)[20], float (&yr)[20], float (&yi)[20]) {
  float tr[4][5], ti[4][5];
#pragma unroll
  for (int n2 = 0; n2 < 5; ++n2) {
    float o_r[4], o_i[4];
    const int i0 = (4 * n2) % 20, i1 = (5 + 4 * n2) % 20, i2 = (10 + 4 * n2) % 20,
              i3 = (15 + 4 * n2) % 20;
    dft4(xr[i0], xi[i0], xr[i1], xi[i1], xr[i2], xi[i2], xr[i3], xi[i3], o_r, o_i);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) { tr[k1][n2] = o_r[k1]; ti[k1][n2] = o_i[k1]; }
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    float o_r[5], o_i[5];
    dft5(tr[k1], ti[k1], o_r, o_i);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) {
      const int k = (5 * k1 + 16 * k2) % 20;
      yr[k] = o_r[k2];
      yi[k] = o_i[k2];
    }
  }
}

AAMD_HD int64_t reflect_idx(int64_t i, int64_t len) {
  if (i < 0) i = -i;
  if (i >= len) i = 2 * (len - 1) - i;
  return i;
}

// ---- phase A: gather + window + DFT-20 over q + twiddle + transposed LDS write ----------
//   lane (p, b) owns samples n = b + 20 q of frames a = t0 + 2p and a + 1; the 28 fetched
//   samples X[q] sit at signal index a*160 - 200 + b + 20 q: frame a uses X[0..19], frame
//   a + 1 uses X[8..27].
template <bool EDGE>
AAMD_HD void phase_a(const LaneConst& c, const float* wav_row, int64_t length, int64_t t0,
                     int n_frames, float* lds) {
  float X[28];
  const int64_t ta = t0 + 2 * c.p;
  const int64_t i0 = ta * kHop - kPad + c.pi;
  if (!EDGE) {
    const float* src = wav_row + i0;
#pragma unroll
    for (int q = 0; q < 28; ++q) X[q] = src[20 * q];
  } else {
#pragma unroll
    for (int q = 0; q < 28; ++q) {
      const int64_t i = i0 + 20 * q;
      // q < 20 belongs to frame a (and to a + 1 when q >= 8); q >= 20 only to frame a + 1
      const bool need = (q < 20) ? (ta < n_frames) : (ta + 1 < n_frames);
      X[q] = need ? wav_row[reflect_idx(i, length)] : 0.0f;
    }
  }
  float xr[20], xi[20], yr[20], yi[20];
  const bool vb = !EDGE || (ta + 1 < n_frames);
#pragma unroll
  for (int q = 0; q < 20; ++q) {
    xr[q] = X[q] * c.win[q];
    xi[q] = vb ? X[q + 8] * c.win[q] : 0.0f;
  }
  dft20(xr, xi, yr, yi);
  if (c.active) {
    float* colp = lds + kTPair * c.p + 2 * c.pi;
#pragma unroll
    for (int s = 0; s < 20; ++s) {
      float vr = yr[s], vi = yi[s];
      if (s != 0) {
        const float wr = c.twr[s - 1], wi = c.twi[s - 1];
        vr = yr[s] * wr - yi[s] * wi;
        vi = yr[s] * wi + yi[s] * wr;
      }
      *reinterpret_cast<F2*>(colp + kTRow * pos_of_col(s)) = F2{vr, vi};
    }
  }
}

// ---- phase B1: read own row, DFT-20 over b  ->  Z[col + 20 d] in registers ----------------
AAMD_HD void phase_b1(const LaneConst& c, const float* lds, float (&zr)[20], float (&zi)[20]) {
  float vr[20], vi[20];
  const float* row = lds + kTPair * c.p + kTRow * c.pi;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const F4 v = *reinterpret_cast<const F4*>(row + 4 * j);
    vr[2 * j] = v.x;
    vi[2 * j] = v.y;
    vr[2 * j + 1] = v.z;
    vi[2 * j + 1] = v.w;
  }
  dft20(vr, vi, zr, zi);
}

// ---- phase B2a: the values the neighbour lane (lane ^ 1) needs: q[i] = Z-register 10 + i,
//   except on the column-0 lane, whose conjugate partner of bin 20 u is its OWN register
//   (20 - u) % 20 = (19 - u) + 1: rotate by one so every lane can use index 19 - u.
AAMD_HD void phase_b2_send(const LaneConst& c, const float (&zr)[20], const float (&zi)[20],
                           float (&qr)[10], float (&qi)[10]) {
  const bool c0 = (c.col == 0);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int j = 10 + i;
    // load both candidates first: a select between array ELEMENTS, never between indices
    const float nr = zr[j], ni = zi[j], rr = zr[(j + 1) % 20], ri = zi[(j + 1) % 20];
    qr[i] = c0 ? rr : nr;
    qi[i] = c0 ? ri : ni;
  }
}

// ---- phase B2b: separate the two real spectra, |.|^2, write interleaved P rows ------------
//   g = q of lane ^ 1 (DPP swap).  conj(Z[400-k]) for k = col + 20u is g[9 - u]; the self-paired
//   columns 0 and 10 take their own q instead.
AAMD_HD void phase_b2(const LaneConst& c, const float (&zr)[20], const float (&zi)[20],
                      const float (&qr)[10], const float (&qi)[10], const float (&gr)[10],
                      const float (&gi)[10], float* lds) {
  if (!c.active) return;
  float* P = lds + kPPair * c.p + 2 * c.col;
  const bool self = (c.col == 0) || (c.col == 10);
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    const float sr = qr[9 - u], si = qi[9 - u], pr = gr[9 - u], pi_ = gi[9 - u];
    const float cr = self ? sr : pr;
    const float ci = self ? si : pi_;
    const float ar = zr[u] + cr, ai = zi[u] - ci;   // 2*A = Z[k] + conj(Z[N-k])
    const float br = zr[u] - cr, bi = zi[u] + ci;   // |2*B|: Z[k] - conj(Z[N-k])
    *reinterpret_cast<F2*>(P + 40 * u) = F2{ar * ar + ai * ai, br * br + bi * bi};
  }
  if (c.col == 0) {  // k = 200 (Nyquist) is its own conjugate partner: A = Re, B = Im
    const float ar = zr[10] + zr[10], bi = zi[10] + zi[10];
    *reinterpret_cast<F2*>(P + 400) = F2{ar * ar, bi * bi};
  }
}

// zero bins 201..211 of each P row so that band reads past bin 200 (weight 0) never touch
// stale transposition data
AAMD_HD void phase_b2_pad(int lane, float* lds) {
  constexpr int kTail = kPK - 201;
  if (lane < 3 * kTail) {
    const int p = lane / kTail, j = lane - p * kTail;
    *reinterpret_cast<F2*>(lds + kPPair * p + 2 * (201 + j)) = F2{0.0f, 0.0f};
  }
}

// ---- phase C: banded mel reduction from the LDS power rows --------------------------------
//   lane (p, slot) computes mel m = slot + 20 r of frames 2p and 2p + 1 in round r; the tap
//   count rw2[r] is wave-uniform.
AAMD_HD void phase_c(const LaneConst& c, const MelTab& mt, const float* lds, float* out_row,
                     int64_t t0, int n_frames) {
  const int64_t ta = t0 + 2 * c.p;
  const bool va = c.active && ta < n_frames, vb = c.active && ta + 1 < n_frames;
  const float* Pp = lds + kPPair * c.p;
  float* oa = out_row + ta * (int64_t)mt.n_mels;
  for (int r = 0; r < mt.n_rounds; ++r) {
    const int m = r * kMelSlots + c.pi;
    const bool ok = m < mt.n_mels;
    const int mm = ok ? m : 0;
    const float* wt = mt.w + mm * mt.ws;
    const float* P = Pp + 2 * mt.lo2[mm];
    const int n2 = mt.rw2[r];
    float acc_a = 0.0f, acc_b = 0.0f;
    for (int j = 0; j < n2; j += 2) {
      const F2 w = *reinterpret_cast<const F2*>(wt + j);
      const F4 q = *reinterpret_cast<const F4*>(P + 2 * j);   // (a, b) of bins j, j + 1
      acc_a += w.x * q.x;
      acc_b += w.x * q.y;
      acc_a += w.y * q.z;
      acc_b += w.y * q.w;
    }
    if (ok && va) oa[m] = acc_a;
    if (ok && vb) oa[mt.n_mels + m] = acc_b;
  }
}

#if defined(__HIPCC__)
__device__ __forceinline__ void wave_lds_fence() {
  // LDS ops of one wave execute in order; this only stops the compiler from moving
  // LDS accesses across the hand-off between lanes.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// value of lane ^ 1 (DPP quad_perm [1,0,3,2]): a VALU move, no LDS traffic
__device__ __forceinline__ float swap_adjacent(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

__global__ void __launch_bounds__(256, 2)
melspec400_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                  const float* __restrict__ tw400, MelBandsDev mb, float* __restrict__ out,
                  int64_t rows, int64_t length, int64_t row_stride, int n_frames, float scale,
                  int tiles_per_row, int64_t n_tiles, int tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float smem400[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  float* lds = smem400 + wave * kLdsDwordsPerWave;

  MelTab mt;
  mel_tab_build(threadIdx.x, blockDim.x, mb, smem400 + 4 * kLdsDwordsPerWave, mt);
  __syncthreads();

  LaneConst c;
  lane_init(lane, window, tw400, scale, c);

  // XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous
  // range of tiles so the frame-overlap re-reads stay inside one L2.
  const int nb = gridDim.x;
  int lb = blockIdx.x;
  if ((nb & 7) == 0) lb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  const int64_t first = ((int64_t)lb * 4 + wave) * tiles_per_wave;
  int64_t last = first + tiles_per_wave;
  if (last > n_tiles) last = n_tiles;

  // (row, tile-in-row) advance incrementally: one division per wave, none per tile
  int64_t row = first / tiles_per_row;
  int tir = (int)(first - row * tiles_per_row);
  for (int64_t tile = first; tile < last; ++tile, ++tir) {
    if (tir == tiles_per_row) { tir = 0; ++row; }
    const int64_t t0 = (int64_t)tir * kFramesPerWave;
    const float* wav_row = wav + row * row_stride;
    const bool interior = (t0 * kHop - kPad >= 0) &&
                          ((t0 + kFramesPerWave - 1) * kHop + (kN - kPad) <= length) &&
                          (t0 + kFramesPerWave <= n_frames);
    if (interior) phase_a<false>(c, wav_row, length, t0, n_frames, lds);
    else          phase_a<true>(c, wav_row, length, t0, n_frames, lds);
    wave_lds_fence();
    float zr[20], zi[20], qr[10], qi[10], gr[10], gi[10];
    phase_b1(c, lds, zr, zi);
    phase_b2_send(c, zr, zi, qr, qi);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      gr[i] = swap_adjacent(qr[i]);
      gi[i] = swap_adjacent(qi[i]);
    }
    wave_lds_fence();
    phase_b2(c, zr, zi, qr, qi, gr, gi, lds);
    phase_b2_pad(lane, lds);
    wave_lds_fence();
    phase_c(c, mt, lds, out + row * n_frames * (int64_t)mb.n_mels, t0, n_frames);
    wave_lds_fence();
  }
}
#endif  // __HIPCC__

}  // namespace m400
}  // namespace aamd

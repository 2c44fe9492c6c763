// Headline kernel: fused MelSpectrogram for n_fft = 400, hop = 160 (the RNN-T / Whisper
// style front-end, BASELINE.json config 2), power = 2, centre + reflect padding.
//
// Design (MI355X, wave64; no barriers -- every wave owns private LDS):
//   * two REAL frames a, b are packed as one COMPLEX 400-point FFT  z = a + i b
//     (no redundant half-spectrum work, no post-twiddle multiply);
//   * 400 = 20 x 20 Cooley-Tukey.  A 20-lane group owns one frame pair; each lane runs a
//     20-point DFT entirely in registers (Good-Thomas 4x5: five radix-4 + four radix-5
//     butterflies, NO internal twiddles), multiplies by its W400^(r s) twiddles, and the
//     20x20 transposition goes once through LDS (row stride 21 complex = conflict free for
//     both the column write and the row read); a second in-register DFT-20 finishes the FFT;
//   * 3 pairs (60 lanes) = 6 frames per wave per iteration;
//   * the a/b spectra are separated with the conj-symmetry rule; the partner bin Z[400-k]
//     lives in lane (20 - s) mod 20 of the same group and is fetched with ds_bpermute
//     (wavefront shuffle), so only |A|^2, |B|^2 for k = 0..200 are ever written to LDS;
//   * mel = banded reduction over the LDS power spectrum: the filterbank's band table lives
//     in LDS; rounds of 10 mels x 6 frames with a wave-uniform tap count per round.
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
constexpr int kTRow = 42;                 // dwords per transposition row (20 complex + 1 pad)
constexpr int kPStride = 218;             // dwords per power-spectrum row: 201 bins + 17 zeros, = 10 mod 16
constexpr int kLdsDwordsPerWave = 60 * kTRow;  // 2520 dwords = 10080 B (the 6 P rows alias it)
constexpr int kMelsPerRound = 10;         // phase C: 6 frames x 10 mels = 60 lanes per round
constexpr int kMelChunk = 8;              // taps fetched per batch of LDS reads

// Per-workgroup LDS copy of the banded filterbank (built once per launch by every workgroup).
struct MelTab {
  const float* w;    // [n_mels][wpad], zero padded
  const int* lo;     // [n_mels]
  const int* rw;     // [n_rounds]: widest band among the round's mels (wave-uniform trip count)
  int n_mels, wpad, n_rounds;
};

AAMD_HD int mel_wpad(int max_width) { return max_width | 1; }   // odd stride: conflict-free rows
AAMD_HD int mel_rounds(int n_mels) { return (n_mels + kMelsPerRound - 1) / kMelsPerRound; }
AAMD_HD int mel_tab_dwords(int n_mels, int max_width) {
  return n_mels * mel_wpad(max_width) + n_mels + mel_rounds(n_mels);
}

AAMD_HD void mel_tab_build(int tid, int nthr, const MelBandsDev& mb, float* base, MelTab& mt) {
  mt.n_mels = mb.n_mels;
  mt.wpad = mel_wpad(mb.max_width);
  mt.n_rounds = mel_rounds(mb.n_mels);
  float* w = base;
  int* lo = reinterpret_cast<int*>(base + mb.n_mels * mt.wpad);
  int* rw = lo + mb.n_mels;
  for (int i = tid; i < mb.n_mels * mt.wpad; i += nthr) {
    const int m = i / mt.wpad, j = i - m * mt.wpad;
    w[i] = (j < mb.width[m]) ? mb.weights[m * mb.max_width + j] : 0.0f;
  }
  for (int m = tid; m < mb.n_mels; m += nthr) lo[m] = mb.lo[m];
  for (int r = tid; r < mt.n_rounds; r += nthr) {
    int mx = 0;
    for (int m = r * kMelsPerRound; m < (r + 1) * kMelsPerRound && m < mb.n_mels; ++m)
      mx = mb.width[m] > mx ? mb.width[m] : mx;
    rw[r] = mx;
  }
  mt.w = w; mt.lo = lo; mt.rw = rw;
}

struct LaneConst {
  // W400^(r*s) = ta[s & 3] * tb[s >> 2] with ta[b] = w^b (b = 1..3), tb[a] = w^(4a) (a = 1..4),
  // w = W400^r: 14 registers instead of 40; 12 of the 20 twiddles cost one extra complex multiply.
  float tar[3], tai[3], tbr[4], tbi[4];
  float win[20];           // 0.5 * scale * window[r + 20 q]
  int p, r;                // pair index (0..2), row/column index inside the pair (0..19)
  int active;              // lanes 60..63 shadow pair 2 but never store
};

AAMD_HD void lane_init(int lane, const float* window, const float* tw400, float scale,
                       LaneConst& c) {
  c.active = lane < 60;
  const int l = c.active ? lane : lane - 20;
  c.p = l / 20;
  c.r = l - 20 * c.p;
#pragma unroll
  for (int b = 1; b < 4; ++b) {
    const int idx = (c.r * b) % kN;
    c.tar[b - 1] = tw400[2 * idx];
    c.tai[b - 1] = tw400[2 * idx + 1];
  }
#pragma unroll
  for (int a = 1; a < 5; ++a) {
    const int idx = (c.r * 4 * a) % kN;
    c.tbr[a - 1] = tw400[2 * idx];
    c.tbi[a - 1] = tw400[2 * idx + 1];
  }
#pragma unroll
  for (int q = 0; q < 20; ++q) c.win[q] = window[c.r + 20 * q] * (0.5f * scale);
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
//   lane (p, r) owns samples n = r + 20 q of frames a = t0 + 2p, b = a + 1.
template <bool EDGE>
AAMD_HD void phase_a(const LaneConst& c, const float* wav_row, int64_t length, int64_t t0,
                     int n_frames, float* lds) {
  float xr[20], xi[20], yr[20], yi[20];
  const int64_t ta = t0 + 2 * c.p;
  const int64_t ia0 = ta * kHop - kPad + c.r;
#pragma unroll
  for (int q = 0; q < 20; ++q) {
    const int64_t ia = ia0 + 20 * q, ib = ia + kHop;
    float xa, xb;
    if (!EDGE) {
      xa = wav_row[ia];
      xb = wav_row[ib];
    } else {
      xa = (ta < n_frames) ? wav_row[reflect_idx(ia, length)] : 0.0f;
      xb = (ta + 1 < n_frames) ? wav_row[reflect_idx(ib, length)] : 0.0f;
    }
    xr[q] = xa * c.win[q];
    xi[q] = xb * c.win[q];
  }
  dft20(xr, xi, yr, yi);
  if (c.active) {
    float* col = lds + kTRow * (20 * c.p) + 2 * c.r;
#pragma unroll
    for (int s = 0; s < 20; ++s) {
      float vr = yr[s], vi = yi[s];
      const int b = s & 3, a = s >> 2;
      if (b != 0) {
        const float tr_ = vr * c.tar[b - 1] - vi * c.tai[b - 1];
        vi = vr * c.tai[b - 1] + vi * c.tar[b - 1];
        vr = tr_;
      }
      if (a != 0) {
        const float tr_ = vr * c.tbr[a - 1] - vi * c.tbi[a - 1];
        vi = vr * c.tbi[a - 1] + vi * c.tbr[a - 1];
        vr = tr_;
      }
      col[kTRow * s] = vr;
      col[kTRow * s + 1] = vi;
    }
  }
}

// ---- phase B1: read own row, DFT-20 over r  ->  Z[s + 20 u] in registers ------------------
AAMD_HD void phase_b1(const LaneConst& c, const float* lds, float (&zr)[20], float (&zi)[20]) {
  float vr[20], vi[20];
  const float* row = lds + kTRow * (20 * c.p + c.r);
#pragma unroll
  for (int j = 0; j < 20; ++j) { vr[j] = row[2 * j]; vi[j] = row[2 * j + 1]; }
  dft20(vr, vi, zr, zi);
}

AAMD_HD int partner_lane(const LaneConst& c) { return 20 * c.p + (20 - c.r) % 20; }

// ---- phase B2: separate the two real spectra, |.|^2, write P rows ------------------------
//   g[i] = partner's Z[10 + i].  conj(Z[400-k]) for k = s + 20u is partner idx 19-u (s != 0),
//   or own-lane idx (20-u) mod 20 when s == 0 (partner == self).
AAMD_HD void phase_b2(const LaneConst& c, const float (&zr)[20], const float (&zi)[20],
                      const float (&gr)[10], const float (&gi)[10], float* lds) {
  if (!c.active) return;
  float* pa = lds + kPStride * (2 * c.p) + c.r;
  float* pb = pa + kPStride;
  const bool s0 = (c.r == 0);
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    float cr, ci;
    if (u == 0) {
      cr = s0 ? zr[0] : gr[9];
      ci = s0 ? zi[0] : gi[9];
    } else {
      cr = s0 ? gr[10 - u] : gr[9 - u];
      ci = s0 ? gi[10 - u] : gi[9 - u];
    }
    const float ar = zr[u] + cr, ai = zi[u] - ci;   // 2*A = Z[k] + conj(Z[N-k])
    const float br = zr[u] - cr, bi = zi[u] + ci;   // |2*B|: Z[k] - conj(Z[N-k])
    pa[20 * u] = ar * ar + ai * ai;
    pb[20 * u] = br * br + bi * bi;
  }
  if (s0) {  // k = 200 (Nyquist): partner idx 10 of this same lane
    const float ar = zr[10] + gr[0], ai = zi[10] - gi[0];
    const float br = zr[10] - gr[0], bi = zi[10] + gi[0];
    pa[200] = ar * ar + ai * ai;
    pb[200] = br * br + bi * bi;
  }
}

// zero the 17-float tail of each P row so that band reads past bin 200 (weight 0) never touch
// stale transposition data of another frame
AAMD_HD void phase_b2_pad(int lane, float* lds) {
  constexpr int kTail = kPStride - 201;
  for (int i = lane; i < kFramesPerWave * kTail; i += 64) {
    const int f = i / kTail, j = i - f * kTail;
    lds[kPStride * f + 201 + j] = 0.0f;
  }
}

// ---- phase C: banded mel reduction from the 6 LDS power rows ----------------------------------
//   round r: lane (f, mi) -> frame f, mel m = 10 r + mi; mel widths grow with m, so the
//   wave-uniform tap count rw[r] tracks each lane's own band width closely (46 tap slots per
//   lane for the 80-mel bank vs 37 ideal).  Each round is a straight-line body selected by the
//   uniform tap count: all 2*W LDS reads are issued back to back, then W FMAs.
constexpr int kMelMaxTaps = 16;   // bands wider than this use the chunked loop
constexpr int kMelMaxRounds = 16;

template <int W>
AAMD_HD float mel_dot(const float* wt, const float* P) {
  float wv[W], pv[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { wv[j] = wt[j]; pv[j] = P[j]; }
  float acc = 0.0f;
#pragma unroll
  for (int j = 0; j < W; ++j) acc += wv[j] * pv[j];
  return acc;
}

AAMD_HD float mel_dot_n(int rw, const float* wt, const float* P) {
  switch (rw) {
    case 0: return 0.0f;
    case 1: return mel_dot<1>(wt, P);
    case 2: return mel_dot<2>(wt, P);
    case 3: return mel_dot<3>(wt, P);
    case 4: return mel_dot<4>(wt, P);
    case 5: return mel_dot<5>(wt, P);
    case 6: return mel_dot<6>(wt, P);
    case 7: return mel_dot<7>(wt, P);
    case 8: return mel_dot<8>(wt, P);
    case 9: return mel_dot<9>(wt, P);
    case 10: return mel_dot<10>(wt, P);
    case 11: return mel_dot<11>(wt, P);
    case 12: return mel_dot<12>(wt, P);
    case 13: return mel_dot<13>(wt, P);
    case 14: return mel_dot<14>(wt, P);
    case 15: return mel_dot<15>(wt, P);
    case 16: return mel_dot<16>(wt, P);
    default: break;
  }
  float acc = 0.0f;
  for (int i0 = 0; i0 < rw; i0 += kMelMaxTaps) {
    float part = 0.0f;
    for (int j = 0; j < kMelMaxTaps && i0 + j < rw; ++j) part += wt[i0 + j] * P[i0 + j];
    acc += part;
  }
  return acc;
}

AAMD_HD void phase_c(int lane, const MelTab& mt, const float* lds, float* out_row, int64_t t0,
                     int n_frames) {
  const bool lane_ok = lane < 60;
  const int f = lane_ok ? lane / kMelsPerRound : 0;
  const int mi = lane_ok ? lane - kMelsPerRound * f : 0;
  const float* Prow = lds + kPStride * f;
  const bool frame_ok = lane_ok && (t0 + f < n_frames);
  float* orow = out_row + (t0 + f) * (int64_t)mt.n_mels;
  // band starts of this lane's mel in every round, fetched together
  int lo[kMelMaxRounds];
#pragma unroll
  for (int r = 0; r < kMelMaxRounds; ++r) {
    const int m = r * kMelsPerRound + mi;
    lo[r] = (r < mt.n_rounds && m < mt.n_mels) ? mt.lo[m] : 0;
  }
#pragma unroll
  for (int r = 0; r < kMelMaxRounds; ++r) {
    if (r < mt.n_rounds) {
      const int m = r * kMelsPerRound + mi;
      const bool ok = m < mt.n_mels;
      const float* wt = mt.w + (ok ? m : 0) * mt.wpad;
      const float acc = mel_dot_n(mt.rw[r], wt, Prow + lo[r]);
      if (ok && frame_ok) orow[m] = acc;
    }
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

__global__ void __launch_bounds__(256)
melspec400_kernel(const float* __restrict__ wav, const float* __restrict__ window,
                  const float* __restrict__ tw400, MelBandsDev mb, float* __restrict__ out,
                  int64_t rows, int64_t length, int64_t row_stride, int n_frames, float scale,
                  int tiles_per_row, int64_t n_tiles, int ablate) {
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
  const int per_xcd = nb >> 3;
  int lb = blockIdx.x;
  if ((nb & 7) == 0) lb = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t waves_total = (int64_t)nb * 4;
  const int partner = partner_lane(c);

  for (int64_t tile = (int64_t)lb * 4 + wave; tile < n_tiles; tile += waves_total) {
    const int64_t row = tile / tiles_per_row;
    const int64_t t0 = (tile - row * tiles_per_row) * kFramesPerWave;
    const float* wav_row = wav + row * row_stride;
    const bool interior = (t0 * kHop - kPad >= 0) &&
                          ((t0 + kFramesPerWave - 1) * kHop + (kN - kPad) <= length) &&
                          (t0 + kFramesPerWave <= n_frames);
    // `ablate` bits are a profiling aid only (tools/gpu_microbench.py); 0 in production
    if (ablate & 1) wav_row = wav;           // every tile re-reads row 0 (cache-resident input)
    if (interior) phase_a<false>(c, wav_row, length, t0, n_frames, lds);
    else          phase_a<true>(c, wav_row, length, t0, n_frames, lds);
    wave_lds_fence();
    float zr[20], zi[20], gr[10], gi[10];
    phase_b1(c, lds, zr, zi);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      gr[i] = __shfl(zr[10 + i], partner, 64);
      gi[i] = __shfl(zi[10 + i], partner, 64);
    }
    wave_lds_fence();
    phase_b2(c, zr, zi, gr, gi, lds);
    phase_b2_pad(lane, lds);
    wave_lds_fence();
    if (!(ablate & 2)) phase_c(lane, mt, lds, out + row * n_frames * (int64_t)mb.n_mels, t0, n_frames);
    else if (lane == 0) out[tile] = lds[lane];
    wave_lds_fence();
  }
}
#endif  // __HIPCC__

}  // namespace m400
}  // namespace aamd

// Biquad-class IIR filtering (order <= 2, any cascade depth <= 8): W waves cooperate on ONE sequence,
// with wave-local shuffle scans and two workgroup barriers per stage and block (BASELINE config 5a:
// 256 sequences x 480 000 samples per GPU, fused 4-biquad cascade).
//
// Reference semantics (functional/filtering.py:1027-1099, libtorchaudio/lfilter.cpp:17-48): per stage
//   w[n] = sum_k b^[k] x[n-k];  y[n] = w[n] - sum_{k>=1} a^[k] y[n-k];  clamp(y) AFTER the recursion.
// A workgroup of W waves walks its sequence in blocks of W x 2048 samples; wave w owns samples
// [2048 w, 2048 (w+1)) of the block, lane l the 32 consecutive samples of chunk l:
//   * 16-B coalesced global loads (next block prefetched into registers), transposed through a
//     wave-private LDS tile (chunk stride 36 floats: conflict-free b128 both ways);
//   * per stage: chunk pass from ZERO recursion state (true input history), chunk-end states combined
//     inside the wave by a 6-step shuffle scan with the powers M^(2^k) of the 32-step transition matrix;
//     the wave-end states are exchanged through LDS (barrier 1) and folded sequentially (W <= 16 steps of
//     a 2x2 product) into the true state E_w entering each wave; lane l adds M^(l+1) E_w, takes the true
//     state entering its chunk from lane l-1 and applies the homogeneous correction H[j] . T; clamp;
//     the clamped chunk stays in registers as the next stage's input (tails exchanged at barrier 2);
//   * a cascade reads x once and writes y once.
// Tables (a^, b^, H[32][2], M^(2^k), M^(l+1)) are built per (sequence, stage) in fp64 by wave 0.
#pragma once
#include "hd.h"

namespace aamd {
namespace lfw {

constexpr int kCh = 32;                    // samples per lane and block
constexpr int kWaveBlock = 64 * kCh;       // 2048 samples per wave and block
constexpr int kRow = 36;                   // LDS floats per chunk row (32 + pad, 16-B aligned)
constexpr int kTile = 64 * kRow;           // 2304 floats per wave
constexpr int kScanSteps = 6;
constexpr int kMaxCascade = 8;
constexpr int kMaxWaves = 16;

// per-stage table (floats): ah[3] bh[3] pad[2] | H[32][2] | M^(2^k), k = 0..6 [7][4] | M^(l+1), l = 0..63 [64][4]
constexpr int kTabAB = 0, kTabH = 8, kTabM = kTabH + 2 * kCh, kTabPow = kTabM + 4 * (kScanSteps + 1);
constexpr int kTabFloats = kTabPow + 4 * 64;                   // 356
// exchange area (floats): S[W][2] | tails[2 parity][stages + 1][W][2] | carry_y[2 parity][stages][2]
AAMD_HD int xch_S(int w) { return 2 * w; }
AAMD_HD int xch_tail(int W, int n_stages, int parity, int st, int w) {
  return 2 * W + ((parity * (n_stages + 1) + st) * W + w) * 2;
}
AAMD_HD int xch_carry(int W, int n_stages, int parity, int st) {
  return 2 * W + 2 * (n_stages + 1) * W * 2 + (parity * n_stages + st) * 2;
}
AAMD_HD int xch_floats(int W, int n_stages) { return 2 * W + 4 * (n_stages + 1) * W + 4 * n_stages; }
AAMD_HD size_t lds_bytes(int W, int n_stages) {
  return ((size_t)W * kTile + (size_t)n_stages * kTabFloats + xch_floats(W, n_stages) + 4) * sizeof(float);
}

// ---- tables of one stage (fp64) ---------------------------------------------------------------
AAMD_HD void build_stage(const float* a_row, const float* b_row, int n_order, float* tab) {
  const float a0 = a_row[0];
  float ah[3], bh[3];
  for (int k = 0; k < 3; ++k) {
    ah[k] = (k < n_order) ? a_row[k] / a0 : 0.0f;     // same fp32 division as the reference
    bh[k] = (k < n_order) ? b_row[k] / a0 : 0.0f;
    tab[kTabAB + k] = ah[k];
    tab[kTabAB + 3 + k] = bh[k];
  }
  double M[2][2];
  for (int d = 0; d < 2; ++d) {
    double h0 = (d == 0) ? 1.0 : 0.0, h1 = (d == 1) ? 1.0 : 0.0;   // y[-1], y[-2]
    double prev = 0.0, last = 0.0;
    for (int j = 0; j < kCh; ++j) {
      const double y = -(double)ah[1] * h0 - (double)ah[2] * h1;
      h1 = h0;
      h0 = y;
      tab[kTabH + 2 * j + d] = (float)y;
      prev = last;
      last = y;
    }
    M[0][d] = last;   // y[31]
    M[1][d] = prev;   // y[30]
  }
  double P[2][2] = {{M[0][0], M[0][1]}, {M[1][0], M[1][1]}};       // M^(l+1)
  for (int l = 0; l < 64; ++l) {
    for (int e = 0; e < 2; ++e)
      for (int d = 0; d < 2; ++d) tab[kTabPow + 4 * l + 2 * e + d] = (float)P[e][d];
    double Q[2][2];
    for (int e = 0; e < 2; ++e)
      for (int d = 0; d < 2; ++d) Q[e][d] = M[e][0] * P[0][d] + M[e][1] * P[1][d];
    for (int e = 0; e < 2; ++e)
      for (int d = 0; d < 2; ++d) P[e][d] = Q[e][d];
  }
  for (int k = 0; k <= kScanSteps; ++k) {
    for (int e = 0; e < 2; ++e)
      for (int d = 0; d < 2; ++d) tab[kTabM + 4 * k + 2 * e + d] = (float)M[e][d];
    double M2[2][2];
    for (int e = 0; e < 2; ++e)
      for (int d = 0; d < 2; ++d) M2[e][d] = M[e][0] * M[0][d] + M[e][1] * M[1][d];
    for (int e = 0; e < 2; ++e)
      for (int d = 0; d < 2; ++d) M[e][d] = M2[e][d];
  }
}

// ---- phase 1: filter the lane's chunk from zero recursion state; (hu0, hu1) = x[-1], x[-2] -------
//      x is overwritten by the zero-state response z (in place: one register array per lane)
AAMD_HD void chunk_pass(const float* tab, float (&x)[kCh], float hu0, float hu1, float& s0, float& s1) {
  const float a1 = tab[kTabAB + 1], a2 = tab[kTabAB + 2];
  const float b0 = tab[kTabAB + 3], b1 = tab[kTabAB + 4], b2 = tab[kTabAB + 5];
  float hz0 = 0.0f, hz1 = 0.0f;
#pragma unroll
  for (int j = 0; j < kCh; ++j) {
    const float u = x[j];
    float w = b2 * hu1;          // oldest tap first (lfilter.cpp:40-43)
    w += b1 * hu0;
    w += b0 * u;
    float y = w;
    y -= a2 * hz1;
    y -= a1 * hz0;
    hu1 = hu0; hu0 = u;
    hz1 = hz0; hz0 = y;
    x[j] = y;
  }
  s0 = hz0;
  s1 = hz1;
}

// (r0, r1) = M . (n0, n1) for a 2x2 M stored row-major
AAMD_HD void mat_apply(const float* M, float n0, float n1, float& r0, float& r1) {
  r0 = M[0] * n0 + M[1] * n1;
  r1 = M[2] * n0 + M[3] * n1;
}

// ---- phase 2: one scan step: s += M^(2^k) . s_from_lane(l - 2^k)  (only lanes l >= 2^k) ----------
AAMD_HD void scan_step(const float* tab, int k, bool active, float n0, float n1, float& s0, float& s1) {
  if (!active) return;
  float a0, a1;
  mat_apply(tab + kTabM + 4 * k, n0, n1, a0, a1);
  s0 += a0;
  s1 += a1;
}

// ---- phase 3: true state E_w entering wave w: fold the published wave-end states ----------------
//   E_0 = carry (true state entering the block);  E_{i+1} = S_i + M^(64 chunks) E_i
//   The published states are fetched in two batches of 8 BEFORE the serial chain (round 2: with the LDS read inside the
//   loop the last wave paid 15 dependent LDS round trips per stage and block while 15 others waited at barrier 2 for it).
AAMD_HD void fold_entering(const float* tab, const float* S, int w, float c0, float c1, float& e0, float& e1) {
  e0 = c0;
  e1 = c1;
  const float m00 = tab[kTabM + 4 * kScanSteps], m01 = tab[kTabM + 4 * kScanSteps + 1];
  const float m10 = tab[kTabM + 4 * kScanSteps + 2], m11 = tab[kTabM + 4 * kScanSteps + 3];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (8 * half >= w) break;
    F2 sr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sr[i] = *reinterpret_cast<const F2*>(S + xch_S(8 * half + i));   // S[i] for i >= W is unused
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (8 * half + i < w) {
        const float p0 = m00 * e0 + m01 * e1, p1 = m10 * e0 + m11 * e1;
        e0 = sr[i].x + p0;
        e1 = sr[i].y + p1;
      }
    }
  }
}

// ---- phase 4: homogeneous correction with the true state (t0, t1) entering the chunk, clamp ------
AAMD_HD void correct_clamp(const float* tab, float t0, float t1, int clamp, float (&z)[kCh]) {
  const float* H = tab + kTabH;
#pragma unroll
  for (int j = 0; j < kCh; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
    // keep at most 8 samples' table reads in flight: hoisting all 64 H values spills at 128 VGPRs
    if ((j & 7) == 0) __builtin_amdgcn_sched_barrier(0);
#endif
    float y = z[j];
    y += H[2 * j] * t0;
    y += H[2 * j + 1] * t1;
    if (clamp) y = fmin(fmax(y, -1.0f), 1.0f);
    z[j] = y;
  }
}

// LDS tile index of sample s of the wave's 2048 samples (chunk s / 32, padded rows)
AAMD_HD int tile_idx(int s) { return (s >> 5) * kRow + (s & 31); }

#if defined(__HIPCC__)
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// out of line: the fp64 table builder runs once per sequence and must not raise the register
// pressure of the block loop
__device__ __attribute__((noinline)) void build_stage_call(const float* a_row, const float* b_row, int n_order,
                                                            float* tab) {
  build_stage(a_row, b_row, n_order, tab);
}

// LAB != 0: profiling variants (wrong results by design) -- bit 0: no barrier 2, bit 1: no barrier 1, bit 2: no scan,
// bit 3: no fold / correction.  Selected by the hidden AAMD_LFW_LAB environment value (tools only).
template <int LAB = 0>
__global__ void __launch_bounds__(1024)
lfilter_wave_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                    float* __restrict__ y, int64_t n_seq, int channels, int64_t length, int n_order,
                    int n_coeff_rows, int n_stages, int clamp, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) float smem_lfw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  float* tile = smem_lfw + wave * kTile;
  float* tabs = smem_lfw + W * kTile;
  float* xch = tabs + n_stages * kTabFloats;
  const int64_t block_len = (int64_t)W * kWaveBlock;

  for (int64_t seq = blockIdx.x; seq < n_seq; seq += gridDim.x) {
    const int ch = (int)(seq % channels);
    const int crow = (n_coeff_rows == 1) ? 0 : ch;
    __syncthreads();
    if (wave == 0) {
      if (lane < n_stages) {   // lane st builds the tables of stage st
        const int64_t coff = ((int64_t)lane * n_coeff_rows + crow) * n_order;
        build_stage_call(a + coff, b + coff, n_order, tabs + lane * kTabFloats);
      }
      for (int i = lane; i < xch_floats(W, n_stages); i += 64) xch[i] = 0.0f;   // zero initial conditions
    }
    __syncthreads();
    const float* xs = x + seq * length;
    float* ys = y + seq * length;

    // whole 2048-sample wave block inside the sequence (all but the last block): one base address +
    // immediate offsets; the ragged tail takes the element-wise path
    auto fetch = [&](int64_t nw0, F4 (&pf)[8]) {
      if (vec_ok && nw0 + kWaveBlock <= length) {
        const float* p = xs + nw0 + 4 * lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) pf[k] = *reinterpret_cast<const F4*>(p + 256 * k);
      } else {
        // ragged tail: unconditional loads from a clamped 32-bit offset (one SGPR base, no per-load
        // 64-bit address -> no register blow-up), zeroed by select
        const int len = (int)length, last = len - 1;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int n = (int)nw0 + 4 * (64 * k + lane);
          float t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = n + e;
            const float val = xs[i < last ? i : last];
            t[e] = (i < len) ? val : 0.0f;
          }
          pf[k] = F4{t[0], t[1], t[2], t[3]};
        }
      }
    };

    F4 pf[8];
    fetch((int64_t)wave * kWaveBlock, pf);
    int parity = 0;
    for (int64_t n0 = 0; n0 < length; n0 += block_len, parity ^= 1) {
      const int64_t nw = n0 + (int64_t)wave * kWaveBlock;       // first sample of this wave
      // 1. row-major pieces -> tile
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<F4*>(tile + tile_idx(4 * (64 * k + lane))) = pf[k];
      lds_fence();
      if (n0 + block_len < length) fetch(nw + block_len, pf);   // prefetch the next block
      // 2. own chunk -> registers
      float v[kCh];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const F4 t = *reinterpret_cast<const F4*>(tile + lane * kRow + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
      if (lane == 63) {   // input tail of stage 0 for the next wave / next block
        xch[xch_tail(W, n_stages, parity, 0, wave)] = v[kCh - 1];
        xch[xch_tail(W, n_stages, parity, 0, wave) + 1] = v[kCh - 2];
      }
      __syncthreads();
      // 3. stages
      for (int st = 0; st < n_stages; ++st) {
        const float* tab = tabs + st * kTabFloats;
        float hu0 = __shfl_up(v[kCh - 1], 1, 64), hu1 = __shfl_up(v[kCh - 2], 1, 64);
        if (lane == 0) {   // inputs before this wave: previous wave (this block) or last wave (previous block)
          const int src = (wave > 0) ? xch_tail(W, n_stages, parity, st, wave - 1)
                                     : xch_tail(W, n_stages, parity ^ 1, st, W - 1);
          hu0 = xch[src];
          hu1 = xch[src + 1];
        }
        float s0, s1;
        chunk_pass(tab, v, hu0, hu1, s0, s1);
        if (!(LAB & 4)) {
#pragma unroll
          for (int k = 0; k < kScanSteps; ++k) {
            const float n0s = __shfl_up(s0, 1 << k, 64), n1s = __shfl_up(s1, 1 << k, 64);
            scan_step(tab, k, lane >= (1 << k), n0s, n1s, s0, s1);
          }
        }
        if (lane == 63) {
          xch[xch_S(wave)] = s0;
          xch[xch_S(wave) + 1] = s1;
        }
        if (!(LAB & 2)) __syncthreads();                          // barrier 1: wave-end states visible
        const int cin = xch_carry(W, n_stages, parity, st);
        float e0 = 0.0f, e1 = 0.0f;
        if (!(LAB & 8)) {
          fold_entering(tab, xch, wave, xch[cin], xch[cin + 1], e0, e1);
          float p0, p1;
          mat_apply(tab + kTabPow + 4 * lane, e0, e1, p0, p1);      // M^(l+1) E_w
          s0 += p0;
          s1 += p1;                                                 // true state after chunk l
          float t0 = __shfl_up(s0, 1, 64), t1 = __shfl_up(s1, 1, 64);
          if (lane == 0) { t0 = e0; t1 = e1; }
          correct_clamp(tab, t0, t1, clamp, v);
        }
        if (lane == 63) {
          if (wave == W - 1) {   // true (unclamped) state leaving the block -> next block's carry
            const int cout = xch_carry(W, n_stages, parity ^ 1, st);
            xch[cout] = s0;
            xch[cout + 1] = s1;
          }
          // clamped output tail = input tail of the next stage
          xch[xch_tail(W, n_stages, parity, st + 1, wave)] = v[kCh - 1];
          xch[xch_tail(W, n_stages, parity, st + 1, wave) + 1] = v[kCh - 2];
        }
        if (!(LAB & 1)) __syncthreads();                          // barrier 2: tails / carries visible
      }
      // 4. chunk -> tile -> row-major pieces -> global
#pragma unroll
      for (int q = 0; q < 8; ++q)
        *reinterpret_cast<F4*>(tile + lane * kRow + 4 * q) = F4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      lds_fence();
      if (vec_ok && nw + kWaveBlock <= length) {
        float* p = ys + nw + 4 * lane;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          *reinterpret_cast<F4*>(p + 256 * k) = *reinterpret_cast<const F4*>(tile + tile_idx(4 * (64 * k + lane)));
      } else {
        const int len = (int)length;
#pragma unroll 1
        for (int i = lane; i < kWaveBlock; i += 64)      // ragged tail: element-wise, coalesced
          if ((int)nw + i < len) ys[(int)nw + i] = tile[tile_idx(i)];
      }
      lds_fence();
    }
  }
}
#endif  // __HIPCC__

}  // namespace lfw
}  // namespace aamd

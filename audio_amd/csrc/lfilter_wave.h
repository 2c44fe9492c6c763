// Biquad-class IIR filtering (order <= 2, any cascade depth <= 8): W waves cooperate on ONE sequence,
// with wave-local DPP scans and ONE workgroup barrier per stage and block (BASELINE config 5a:
// 256 sequences x 480 000 samples per GPU, fused 4-biquad cascade).
//
// Reference semantics (functional/filtering.py:1027-1099, libtorchaudio/lfilter.cpp:17-48): per stage
//   w[n] = sum_k b^[k] x[n-k];  y[n] = w[n] - sum_{k>=1} a^[k] y[n-k];  clamp(y) AFTER the recursion.
// A workgroup of W waves walks its sequence in blocks of W x 2048 samples; wave w owns samples
// [2048 w, 2048 (w+1)) of the block, lane l the 32 consecutive samples of chunk l:
//   * the NEXT block's samples travel global -> LDS by LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per instruction; rows that
//     are not 16-byte aligned: `_dword`, 256 B per instruction) while the stages of the current block run: the wave's LDS
//     tile is free then, because the chunk lives in registers, so the copy costs no registers and is not waited for
//     before it is needed (round 2: the former register prefetch of 32 values was spilled to scratch by the compiler
//     right behind the first barrier, i.e. every block waited for the loads it had just issued --
//     profiles/r02_r_lfilter_isa_notes.txt);
//   * tile layout: the DMA writes LDS linearly (wave-uniform base + lane x size), so the tile is the block in memory order
//     with the eight 16-byte pieces of chunk c stored at piece position q ^ ((c >> 1) & 7) -- the swizzle is applied to
//     the SOURCE address of each lane's piece.  The lane's own-chunk b128 accesses (stride 128 B between lanes) and the
//     row-major b128 accesses are both conflict-free under it;
//   * per stage: chunk pass from ZERO recursion state (true input history); chunk-end states combined inside the
//     wave by a 6-step DPP scan (row_shr 1/2/4/8, row_bcast 15/31: VALU moves, no LDS round trips) with the powers of
//     the 32-step transition matrix Mc; the wave-end states are published in LDS (the ONE barrier of the stage), every
//     wave scans them (16 lanes, 4 DPP steps with Mc^64, ^128, ...) into the true state E_w entering it; lane l adds
//     Mc^(l+1) E_w, takes the true state entering its chunk from lane l-1 and adds the homogeneous response
//     c[j] = -a1 c[j-1] - a2 c[j-2] started from that state (a recurrence in registers; the former table of 64
//     responses cost 64 LDS reads per lane and stage); clamp; the clamped chunk stays in registers as the next stage's
//     input;
//   * no second barrier: the input history of lane 0 for the NEXT stage is the clamped true state E_w the wave already
//     holds (E_w IS (y[-1], y[-2]) of this stage), the published states are double buffered, the block carry is written
//     a whole block before it is read; the raw-input history of stage 0 comes with the block (one more DMA instruction
//     for the 64 samples before it);
//   * a cascade reads x once and writes y once.
// Tables (a^, b^, Mc^(2^k), Mc^(l+1), Mc^(64 w)) are built per (sequence, stage) in fp64 by wave 0.
#pragma once
#include "hd.h"

namespace aamd {
namespace lfw {

constexpr int kCh = 32;                    // samples per lane and block
constexpr int kWaveBlock = 64 * kCh;       // 2048 samples per wave and block
constexpr int kHist = 64;                  // floats in front of the block: the 64 samples before it (stage 0's input history)
constexpr int kTile = kHist + kWaveBlock;  // 2112 floats per wave
constexpr int kMaxCascade = 8;
constexpr int kMaxWaves = 16;

// per-stage table (floats): ah[3] bh[3] pad[2] | Mc^(2^k), k = 0..9 [10][4] | Mc^(l+1), l = 0..63 [64][4] |
//                           Mc^(64 w), w = 0..15 [16][4]
constexpr int kPow2 = 10;
constexpr int kTabAB = 0, kTabM = 8, kTabPow = kTabM + 4 * kPow2, kTabPowW = kTabPow + 4 * 64;
// AAMD_LFW_HTAB: the homogeneous responses to the states (1, 0) and (0, 1), h0[j], h1[j], j = 0 .. 31, behind the matrices
// ({h0[j], h1[j]} interleaved: one broadcast b128 read serves two samples) -- the correction y[j] = z[j] + h0[j] t0 + h1[j] t1
// is then two independent FMAs per sample instead of a 2-term recurrence (mul + fma, a dependent chain) and an add
#ifndef AAMD_LFW_HTAB
#define AAMD_LFW_HTAB 0
#endif
constexpr bool kHTab = AAMD_LFW_HTAB != 0;
constexpr int kTabH = kTabPowW + 4 * kMaxWaves;                                   // 368
constexpr int kTabFloats = kTabH + (kHTab ? 2 * kCh : 0);                         // 368 / 432
// exchange area (floats): S[2 buffers][W][2] | carry[2 parity][stages][2]
AAMD_HD int xch_S(int W, int buf, int w) { return (buf * W + w) * 2; }
AAMD_HD int xch_carry(int W, int n_stages, int parity, int st) { return 4 * W + (parity * n_stages + st) * 2; }
AAMD_HD int xch_floats(int W, int n_stages) { return 4 * W + 4 * n_stages; }
AAMD_HD size_t lds_bytes(int W, int n_stages) {
  return ((size_t)W * kTile + (size_t)n_stages * kTabFloats + xch_floats(W, n_stages) + 4) * sizeof(float);
}

// LDS tile index of sample s of the wave's 2048 samples (s = -64 .. -1: the history row).  Piece q (4 samples) of chunk c
// sits at piece position q ^ swz(c) of the chunk's 32 floats.
AAMD_HD int swz(int c) { return (c >> 1) & 7; }
AAMD_HD int tile_idx(int s) {
  if (s < 0) return kHist + s;
  const int c = s >> 5, q = (s >> 2) & 7;
  return kHist + 32 * c + 4 * (q ^ swz(c)) + (s & 3);
}

// ---- tables of one stage (fp64) ---------------------------------------------------------------
// Entries are saturated to finite floats (NaN -> 0): lanes without a scan partner multiply them by zero, and a diverging
// filter (|pole|^2048 beyond the float range) must not turn that into NaN for the samples before it diverges.
AAMD_HD float tab_sat(double c) {
  if (!(c == c)) return 0.0f;
  return (float)(c > 3.0e38 ? 3.0e38 : (c < -3.0e38 ? -3.0e38 : c));
}
AAMD_HD void mat_mul(const double (&A)[2][2], const double (&B)[2][2], double (&C)[2][2]) {
  double T[2][2];
  for (int e = 0; e < 2; ++e)
    for (int d = 0; d < 2; ++d) T[e][d] = A[e][0] * B[0][d] + A[e][1] * B[1][d];
  for (int e = 0; e < 2; ++e)
    for (int d = 0; d < 2; ++d) C[e][d] = T[e][d];
}
AAMD_HD void mat_put(float* dst, const double (&A)[2][2]) {
  for (int e = 0; e < 2; ++e)
    for (int d = 0; d < 2; ++d) dst[2 * e + d] = tab_sat(A[e][d]);
}
AAMD_HD void build_stage(const float* a_row, const float* b_row, int n_order, float* tab) {
  const float a0 = a_row[0];
  float ah[3], bh[3];
  for (int k = 0; k < 3; ++k) {
    ah[k] = (k < n_order) ? a_row[k] / a0 : 0.0f;     // same fp32 division as the reference
    bh[k] = (k < n_order) ? b_row[k] / a0 : 0.0f;
    tab[kTabAB + k] = ah[k];
    tab[kTabAB + 3 + k] = bh[k];
  }
  tab[kTabAB + 6] = tab[kTabAB + 7] = 0.0f;
  double M[2][2];                                       // Mc: state after 32 homogeneous steps
  for (int d = 0; d < 2; ++d) {
    double h0 = (d == 0) ? 1.0 : 0.0, h1 = (d == 1) ? 1.0 : 0.0;   // y[-1], y[-2]
    for (int j = 0; j < kCh; ++j) {
      const double y = -(double)ah[1] * h0 - (double)ah[2] * h1;
      h1 = h0;
      h0 = y;
      if (kHTab) tab[kTabH + 2 * j + d] = tab_sat(y);
    }
    M[0][d] = h0;   // y[31]
    M[1][d] = h1;   // y[30]
  }
  double P[2][2] = {{M[0][0], M[0][1]}, {M[1][0], M[1][1]}};       // Mc^(l+1)
  for (int l = 0; l < 64; ++l) {
    mat_put(tab + kTabPow + 4 * l, P);
    mat_mul(M, P, P);
  }
  double Q[2][2] = {{M[0][0], M[0][1]}, {M[1][0], M[1][1]}};       // Mc^(2^k)
  double M64[2][2] = {{1.0, 0.0}, {0.0, 1.0}};
  for (int k = 0; k < kPow2; ++k) {
    mat_put(tab + kTabM + 4 * k, Q);
    if (k == 6) mat_mul(Q, M64, M64);
    mat_mul(Q, Q, Q);
  }
  double Wp[2][2] = {{1.0, 0.0}, {0.0, 1.0}};                      // Mc^(64 w)
  for (int w = 0; w < kMaxWaves; ++w) {
    mat_put(tab + kTabPowW + 4 * w, Wp);
    mat_mul(M64, Wp, Wp);
  }
}

// ---- phase 1: filter the lane's chunk from zero recursion state; (hu0, hu1) = x[-1], x[-2] -------
//      x is overwritten by the zero-state response z (in place: one register array per lane)
// the normalised coefficients of a stage, read from the table ONCE per stage and block
struct StageCoef { float a1, a2, b0, b1, b2; };
AAMD_HD StageCoef stage_coef(const float* tab) {
  return StageCoef{tab[kTabAB + 1], tab[kTabAB + 2], tab[kTabAB + 3], tab[kTabAB + 4], tab[kTabAB + 5]};
}
// AAMD_LFW_SPLIT (round 6): the two passes of a stage issue the SAME operations in an order that leaves one instruction of every
// sample on the dependent chain instead of five / four (0 = the sample-by-sample order of rounds 2 - 5, tools/lfw_ab.py)
#ifndef AAMD_LFW_SPLIT
#define AAMD_LFW_SPLIT 1
#endif
AAMD_HD void chunk_pass(const StageCoef& cf, float (&x)[kCh], float hu0, float hu1, float& s0, float& s1) {
  float hz0 = 0.0f, hz1 = 0.0f;
  if (AAMD_LFW_SPLIT) {
    // the same operations in another order: the feed-forward sums of all 32 samples first (in place, last sample first -- 96
    // independent instructions), then the recursion (one instruction of every sample on the chain, one beside it)
    auto in = [&](int j) { return j >= 0 ? x[j] : (j == -1 ? hu0 : hu1); };
#pragma unroll
    for (int j = kCh - 4; j >= 0; j -= 4) {          // samples j .. j + 3, four independent sums side by side
      float w0 = cf.b2 * in(j - 2), w1 = cf.b2 * in(j - 1), w2 = cf.b2 * in(j), w3 = cf.b2 * in(j + 1);
      w0 += cf.b1 * in(j - 1); w1 += cf.b1 * in(j); w2 += cf.b1 * in(j + 1); w3 += cf.b1 * in(j + 2);
      w0 += cf.b0 * in(j); w1 += cf.b0 * in(j + 1); w2 += cf.b0 * in(j + 2); w3 += cf.b0 * in(j + 3);
      x[j] = w0; x[j + 1] = w1; x[j + 2] = w2; x[j + 3] = w3;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    // the recursion: t = w[j + 1] - a2 y[j - 1] is worked out beside the chain instruction y[j] = t' - a1 y[j - 1]
    float t = x[0];
#pragma unroll
    for (int j = 0; j < kCh; ++j) {
      float y = t;
      if (j >= 1) y -= cf.a1 * hz0;
      if (j + 1 < kCh) {
        t = x[j + 1];
        if (j >= 1) t -= cf.a2 * hz0;      // hz0 = y[j - 1] here
      }
      hz1 = hz0; hz0 = y;
      x[j] = y;
    }
    s0 = hz0;
    s1 = hz1;
    return;
  }
#pragma unroll
  for (int j = 0; j < kCh; ++j) {
    const float u = x[j];
    float w = cf.b2 * hu1;          // oldest tap first (lfilter.cpp:40-43)
    w += cf.b1 * hu0;
    w += cf.b0 * u;
    float y = w;
    if (j >= 2) y -= cf.a2 * hz1;   // (zero recursion state: the first two samples have nothing to subtract)
    if (j >= 1) y -= cf.a1 * hz0;
    hu1 = hu0; hu0 = u;
    hz1 = hz0; hz0 = y;
    x[j] = y;
  }
  s0 = hz0;
  s1 = hz1;
}
AAMD_HD void chunk_pass(const float* tab, float (&x)[kCh], float hu0, float hu1, float& s0, float& s1) {
  chunk_pass(stage_coef(tab), x, hu0, hu1, s0, s1);
}

// (r0, r1) = M . (n0, n1) for a 2x2 M stored row-major
AAMD_HD void mat_apply(const float* M, float n0, float n1, float& r0, float& r1) {
  r0 = M[0] * n0 + M[1] * n1;
  r1 = M[2] * n0 + M[3] * n1;
}
// (s0, s1) += M . (n0, n1)
AAMD_HD void mat_acc(const float* M, float n0, float n1, float& s0, float& s1) {
  float a0, a1;
  mat_apply(M, n0, n1, a0, a1);
  s0 += a0;
  s1 += a1;
}

// ---- phase 2: the wave scan.  Steps 0..3: lane l takes lane l - 2^k of its 16-lane row (zeros outside the row) with
//      Mc^(2^k); step 4: lanes of rows 1 and 3 take lane 15 of the row before with Mc^((l & 15) + 1); step 5: lanes
//      32..63 take lane 31 with Mc^((l & 31) + 1).  `scan_src` names the source lane (-1: zeros) -- the kernel's DPP
//      controls and the CPU replay's array reads are the same map.
AAMD_HD int scan_src(int step, int lane) {
  if (step < 4) return (lane & 15) >= (1 << step) ? lane - (1 << step) : -1;
  if (step == 4) return (lane & 16) ? (lane & ~15) - 1 : -1;
  return (lane & 32) ? 31 : -1;
}
AAMD_HD const float* scan_mat(const float* tab, int step, int lane) {
  if (step < 4) return tab + kTabM + 4 * step;
  if (step == 4) return tab + kTabPow + 4 * (lane & 15);
  return tab + kTabPow + 4 * (lane & 31);
}

// ---- phase 3: true state E_w entering wave w from the published wave-end states S_0 .. S_{W-1} ---
//   E_w = sum_{i < w} Mw^(w - 1 - i) S_i + Mw^w . carry   (Mw = Mc^64; E_0 = carry).
//   Lane i (< W) of wave w multiplies S_i by the table entry Mc^(64 (w - 1 - i)) -- an entry that does not depend on the
//   published states, so it is requested before the barrier -- and the contributions are summed by a plain 4-step prefix
//   sum over the 16-lane row (source lane l - 2^k, zeros below 0); lane w - 1 holds the sum over i < w.  Round 3: the
//   former matrix scan over the W states (Mc^(64 . 2^k) per step) cost 32 instructions per stage instead of 12.
AAMD_HD int fold_entry(int w, int lane) {          // index into the Mc^(64 k) table for lane's contribution to wave w
  const int k = w - 1 - (lane & 15);
  return k > 0 ? k : 0;                            // (lanes >= w - 1 ... : only lanes < w enter the sum that is read)
}
AAMD_HD void fold_finish(const float* tab, int w, float i0, float i1, float c0, float c1, float& e0, float& e1) {
  e0 = (w > 0) ? i0 : 0.0f;
  e1 = (w > 0) ? i1 : 0.0f;
  mat_acc(tab + kTabPowW + 4 * w, c0, c1, e0, e1);
}

// ---- phase 4: homogeneous response started from the true state (t0, t1) entering the chunk, clamp ----
template <bool CLAMP>
AAMD_HD void correct_clamp_t(const StageCoef& cf, float t0, float t1, float (&z)[kCh]) {
  float c0 = t0, c1 = t1;
  if (AAMD_LFW_SPLIT) {
    // the same operations, software-pipelined: beside the chain instruction of sample j (c[j] = m - a1 c[j - 1]) stand the
    // product of sample j + 1 (- a2 c[j - 1]), the sum of sample j - 1 and the clamp of sample j - 2
    float m = -(cf.a2 * c1);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(m));
#endif
    float cp = 0.0f, yp = 0.0f;        // c[j - 1], y[j - 1] (not clamped yet)
#pragma unroll
    for (int j = 0; j <= kCh + 1; ++j) {
      float c = 0.0f;
      if (j < kCh) c = m - cf.a1 * c0;
      if (j + 1 < kCh) {
        m = -(cf.a2 * c0);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(m));     // a product of its own: contracted into the chain instruction it would put two on the chain
#endif
      }
      float y = 0.0f;
      if (j >= 1 && j <= kCh) y = z[j - 1] + cp;
      if (j >= 2) z[j - 2] = CLAMP ? fmin(fmax(yp, -1.0f), 1.0f) : yp;
      yp = y;
      cp = c;
      c0 = c;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < kCh; ++j) {
    float c = -(cf.a2 * c1);
    c -= cf.a1 * c0;
    c1 = c0;
    c0 = c;
    float y = z[j] + c;
    if (CLAMP) y = fmin(fmax(y, -1.0f), 1.0f);
    z[j] = y;
  }
}
// the same from the table of responses (kHTab): no chain
template <bool CLAMP>
AAMD_HD void correct_clamp_h(const float* tab, float t0, float t1, float (&z)[kCh]) {
#pragma unroll
  for (int j = 0; j < kCh; j += 2) {
    const F4 h = *reinterpret_cast<const F4*>(tab + kTabH + 2 * j);
    float y0 = z[j] + h.x * t0;
    float y1 = z[j + 1] + h.z * t0;
    y0 += h.y * t1;
    y1 += h.w * t1;
    if (CLAMP) { y0 = fmin(fmax(y0, -1.0f), 1.0f); y1 = fmin(fmax(y1, -1.0f), 1.0f); }
    z[j] = y0;
    z[j + 1] = y1;
  }
}
AAMD_HD void correct_clamp(const StageCoef& cf, float t0, float t1, int clamp, float (&z)[kCh]) {
  if (clamp) correct_clamp_t<true>(cf, t0, t1, z);      // (wave-uniform: two straight-line bodies instead of a select per sample)
  else correct_clamp_t<false>(cf, t0, t1, z);
}
AAMD_HD void correct_clamp(const float* tab, float t0, float t1, int clamp, float (&z)[kCh]) {
  correct_clamp(stage_coef(tab), t0, t1, clamp, z);
}
AAMD_HD float clamp1(float v, int clamp) { return clamp ? fmin(fmax(v, -1.0f), 1.0f) : v; }
// clamp argument of the launch: 0 never, 1 after every stage (n sequential lfilter calls), 2 after the LAST stage only (one
// higher-order filter factored into second-order sections)
AAMD_HD int stage_clamp(int clamp, int st, int n_stages) { return clamp == 1 || (clamp == 2 && st == n_stages - 1); }

#if defined(__HIPCC__)
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// workgroup barrier that leaves vector-memory operations (the LDS-DMA of the next block, the stores of the last one) in
// flight: __syncthreads() would drain vmcnt while a DMA is pending
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// LDS-DMA: 64 lanes fill 64 x SIZE contiguous bytes at LDS byte address `lds_dst` (wave-uniform); the source of each
// lane's piece = wave-uniform base (SGPR pair) + 32-bit lane offset, so the address arithmetic of a block's copies is scalar
// (M0 is declared clobbered instead of being saved and restored: two scalar instructions less per copy)
__device__ __forceinline__ void glds4(const float* sbase, unsigned voff_bytes, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
               : : "v"(voff_bytes), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
template <bool NT = false>
__device__ __forceinline__ void glds16(const float* sbase, unsigned voff_bytes, unsigned lds_dst) {
  if (NT)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                 : : "v"(voff_bytes), "s"(sbase), "s"(lds_dst) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 : : "v"(voff_bytes), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
typedef float f4v __attribute__((ext_vector_type(4)));
// DPP source-lane moves (VALU): lanes without a source read 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp0(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
template <int STEP>
__device__ __forceinline__ float scan_take(float v) {
  if (STEP == 0) return dpp0<0x111, 0xf>(v);      // row_shr:1
  if (STEP == 1) return dpp0<0x112, 0xf>(v);      // row_shr:2
  if (STEP == 2) return dpp0<0x114, 0xf>(v);      // row_shr:4
  if (STEP == 3) return dpp0<0x118, 0xf>(v);      // row_shr:8
  if (STEP == 4) return dpp0<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  return dpp0<0x143, 0xc>(v);                     // row_bcast:31 into rows 2 and 3
}
__device__ __forceinline__ float wave_shr1(float v) { return dpp0<0x138, 0xf>(v); }   // lane l <- lane l - 1, lane 0 <- 0

// out of line: the fp64 table builder runs once per sequence and must not raise the register
// pressure of the block loop
__device__ __attribute__((noinline)) void build_stage_call(const float* a_row, const float* b_row, int n_order,
                                                            float* tab) {
  build_stage(a_row, b_row, n_order, tab);
}

// One stage of the wave's 2048 samples (v: the lane's chunk, in place): chunk pass, wave scan, publication of the wave-end
// state, the stage's ONE barrier, fold to the true entering state, correction and clamp.  (hin0, hin1): lane 0's input
// history on entry, the next stage's on exit.
// One 2 x 2 table entry in registers
struct Mat4 { float m[4]; };
__device__ __forceinline__ Mat4 mat_load(const float* p) {
  const F4 t = *reinterpret_cast<const F4*>(p);
  return Mat4{{t.x, t.y, t.z, t.w}};
}

// PRE: the table entries of the scan are requested at the top of the stage (in flight during the chunk pass), those of the
// fold and of the state update right before the barrier (they land while the wave waits): 6 + 6 x 16 bytes, 24 registers
// at a time (the 12-wave kernels have them).  Without it the serial part of the
// stage -- scan, barrier, fold, state update, about 110 dependent instructions -- stopped six times for an LDS round trip
// (profiles/r03_zz_lfilter_issue.txt).
template <int LAB, bool PRE = false>
__device__ __forceinline__ void stage_step(const float* tab, float* xch, int W, int n_stages, int parity, int st, int sbuf,
                                           int wave, int lane, int clamp, float (&v)[kCh], float& hin0, float& hin1) {
  float hu0 = wave_shr1(v[kCh - 1]), hu1 = wave_shr1(v[kCh - 2]);
  if (lane == 0) { hu0 = hin0; hu1 = hin1; }
  float s0, s1;
  const StageCoef cf = stage_coef(tab);
  Mat4 ms[6], mf, mw, ml;
  if (PRE) {
#pragma unroll
    for (int k = 0; k < 6; ++k) ms[k] = mat_load(scan_mat(tab, k, lane));
  }
  // scheduling fences between the phases: left alone, the scheduler interleaves the chunk pass with the table reads
  // of the later phases and needs 135 registers -- 7 more than 16 waves have, and ANY scratch reload inside the stage
  // loop would wait for the LDS-DMA in flight (vmcnt counts both)
  __builtin_amdgcn_sched_barrier(0);
  chunk_pass(cf, v, hu0, hu1, s0, s1);
  __builtin_amdgcn_sched_barrier(0);
  if (!(LAB & 4)) {
#define AAMD_LFW_SCAN(STEP) mat_acc(PRE ? ms[STEP].m : scan_mat(tab, STEP, lane), scan_take<STEP>(s0), scan_take<STEP>(s1), s0, s1);
    AAMD_LFW_SCAN(0) AAMD_LFW_SCAN(1) AAMD_LFW_SCAN(2) AAMD_LFW_SCAN(3) AAMD_LFW_SCAN(4) AAMD_LFW_SCAN(5)
#undef AAMD_LFW_SCAN
  }
  if (lane == 63) *reinterpret_cast<F2*>(xch + xch_S(W, sbuf, wave)) = F2{s0, s1};
  if (PRE && !(LAB & 8)) {                                  // the tables of the fold and of the state update land while the wave waits
    mf = mat_load(tab + kTabPowW + 4 * fold_entry(wave, lane));
    mw = mat_load(tab + kTabPowW + 4 * wave);
    ml = mat_load(tab + kTabPow + 4 * lane);
  }
  if (!(LAB & 2)) lds_barrier();                            // the wave-end states of this stage are visible
  float e0 = 0.0f, e1 = 0.0f;
  if (!(LAB & 8)) {
    const F2 cin = *reinterpret_cast<const F2*>(xch + xch_carry(W, n_stages, parity, st));
    const F2 sw = *reinterpret_cast<const F2*>(xch + xch_S(W, sbuf, (lane & 15) < W ? (lane & 15) : 0));
    float f0 = (lane & 15) < W ? sw.x : 0.0f, f1 = (lane & 15) < W ? sw.y : 0.0f;
    {
      float g0, g1;
      mat_apply(PRE ? mf.m : tab + kTabPowW + 4 * fold_entry(wave, lane), f0, f1, g0, g1);
      f0 = g0;
      f1 = g1;
    }
#define AAMD_LFW_FOLD(STEP) { const float a0_ = scan_take<STEP>(f0), a1_ = scan_take<STEP>(f1); f0 += a0_; f1 += a1_; }
    AAMD_LFW_FOLD(0) AAMD_LFW_FOLD(1) AAMD_LFW_FOLD(2) AAMD_LFW_FOLD(3)
#undef AAMD_LFW_FOLD
    const int from = wave > 0 ? wave - 1 : 0;
    const float i0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f0), from));
    const float i1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f1), from));
    e0 = (wave > 0) ? i0 : 0.0f;                            // (fold_finish with the table entry in registers)
    e1 = (wave > 0) ? i1 : 0.0f;
    mat_acc(PRE ? mw.m : tab + kTabPowW + 4 * wave, cin.x, cin.y, e0, e1);
    mat_acc(PRE ? ml.m : tab + kTabPow + 4 * lane, e0, e1, s0, s1);      // + Mc^(l+1) E_w: true state after chunk l
    float t0 = wave_shr1(s0), t1 = wave_shr1(s1);
    if (lane == 0) { t0 = e0; t1 = e1; }
    __builtin_amdgcn_sched_barrier(0);
    if (kHTab) { if (clamp) correct_clamp_h<true>(tab, t0, t1, v); else correct_clamp_h<false>(tab, t0, t1, v); }
    else if (PRE) correct_clamp(cf, t0, t1, clamp, v);
    else correct_clamp(tab, t0, t1, clamp, v);             // (a1, a2 re-read: the 128-register instantiations have no room to keep them)
  }
  if (lane == 63 && wave == W - 1)   // true (unclamped) state leaving the block -> next block's carry
    *reinterpret_cast<F2*>(xch + xch_carry(W, n_stages, parity ^ 1, st)) = F2{s0, s1};
  // the outputs just before this wave ARE the state entering it: clamped, they are the next stage's input history
  hin0 = clamp1(e0, clamp);
  hin1 = clamp1(e1, clamp);
}

// LAB != 0: profiling variants (wrong results by design) -- bit 0: no LDS-DMA wait, bit 1: no barrier, bit 2: no scan,
// bit 3: no fold / correction, bit 4: no LDS-DMA issue, bit 5: no global stores; bit 7 (128): the copy of block i + 2 is issued
// BEFORE the stores of block i, bit 8 (256): non-temporal stores, bit 9 (512): non-temporal copies (these three: right results).
// Selected by the hidden AAMD_LFW_LAB environment value (tools only).
// MAXW = waves per workgroup the instantiation is compiled for (register budget: 128 VGPRs at 16 waves, 256 at 8).
// VEC: rows of x and y are 16-byte aligned (16-byte DMA pieces and stores); otherwise dword copies and element-wise stores.
template <int LAB = 0, int MAXW = kMaxWaves, bool VEC = true>
__global__ void __launch_bounds__(64 * MAXW)
lfilter_wave_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                    float* __restrict__ y, int64_t n_seq, int channels, int64_t length, int n_order,
                    int n_coeff_rows, int n_stages, int clamp) {
  constexpr bool vec_ok = VEC;
  extern __shared__ __attribute__((aligned(16))) float smem_lfw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  float* tile = smem_lfw + wave * kTile;
  float* tabs = smem_lfw + W * kTile;
  float* xch = tabs + n_stages * kTabFloats;
  const int64_t block_len = (int64_t)W * kWaveBlock;
  const unsigned tile_addr = (unsigned)(uintptr_t)tile;      // LDS byte address of the wave's tile
  // Lane-derived addresses are RECOMPUTED where they are used (a handful of VALU operations per block) from an opaque copy
  // of the lane number: hoisted out of the block loop they would be live across the stage loop, and the 128 registers of a
  // 16-wave workgroup do not hold them -- the compiler then parks per-block values in scratch and reloads them inside the
  // stage loop, and a scratch reload waits for the LDS-DMA in flight (one in-order counter for all vector-memory operations).
  auto lane_now = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  // 16-byte piece 64 i + lane of the block (memory order): chunk 8 i + (lane >> 3), piece lane & 7, swizzle
  // (4 i + (lane >> 4)) & 7 -- one lane offset for even i (odd = 0), one for odd i.  As LDS float index it is where the
  // row-major read of step 3 finds the piece; as a byte offset into the block it is the source of LDS slot 64 i + lane.
  auto piece = [](int l, int odd) { return 32 * (l >> 3) + 4 * ((l & 7) ^ (4 * odd + (l >> 4))); };

  int built_crow = -1;                                  // coefficient row of the tables currently in LDS
  for (int64_t seq = blockIdx.x; seq < n_seq; seq += gridDim.x) {
    const int ch = (int)(seq % channels);
    const int crow = (n_coeff_rows == 1) ? 0 : ch;
    __syncthreads();
    if (wave == 0) {
      // lane st builds the tables of stage st (fp64, one lane: ~10 us) -- only when the coefficient row differs from the one the
      // tables in LDS were built for: with shared coefficients a workgroup builds them ONCE, not once per sequence (65 536
      // sequences of 4 000 samples spent a quarter of their time here)
      if (lane < n_stages && crow != built_crow) {
        const int64_t coff = ((int64_t)lane * n_coeff_rows + crow) * n_order;
        build_stage_call(a + coff, b + coff, n_order, tabs + lane * kTabFloats);
      }
      for (int i = lane; i < xch_floats(W, n_stages); i += 64) xch[i] = 0.0f;   // zero initial conditions
    }
    built_crow = crow;
    __syncthreads();
    const float* xs = x + seq * length;
    float* ys = y + seq * length;

    // a wave block wholly inside the sequence: 8 (16-byte aligned rows) or 32 DMA instructions, and one more for the 64
    // samples before the block (stage 0's input history of lane 0)
    auto stage_in = [&](int64_t nw0) {
      if (LAB & 16) return;
      const float* src = xs + nw0;                              // wave-uniform
      const int l = lane_now();
      if (nw0 > 0) glds4(src - kHist, 4u * (unsigned)l, tile_addr);
      if (vec_ok) {
        const unsigned pe = 4u * (unsigned)piece(l, 0), po = 4u * (unsigned)piece(l, 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16<(LAB & 512) != 0>(src + 256 * i, (i & 1) ? po : pe, tile_addr + 4 * kHist + 1024 * i);
      } else {
        // dword copies: LDS dword 64 i + lane holds sample 64 i + 32 (lane >> 5) + 4 (((lane >> 2) & 7) ^ (i & 7)) + (lane & 3)
        const unsigned dwA = 4u * (unsigned)(32 * (l >> 5) + (l & 3));
        const unsigned dwJ = (unsigned)((l >> 2) & 7);
#pragma unroll
        for (int i = 0; i < 32; ++i)
          glds4(src + 64 * i, dwA + ((dwJ ^ (unsigned)(i & 7)) << 4), tile_addr + 4 * kHist + 256 * i);
      }
    };
    // the ragged block at the end of the sequence (or past it): zeros behind the last sample
    auto fill_ragged = [&](int64_t nw0) {
      const int64_t left = length - nw0;
#pragma unroll 1
      for (int i = lane - 64; i < kWaveBlock; i += 64) tile[tile_idx(i)] = (i < left && nw0 + i >= 0) ? xs[nw0 + i] : 0.0f;
    };
    // the lane's chunk + the two samples before the wave's block (zeros at the start of the sequence)
    auto take_chunk = [&](int64_t nw0, float (&v)[kCh], float& h0, float& h1) {
      const int l = lane_now();
      const float* own = tile + kHist + kCh * l;                // the lane's chunk: piece q at own + 4 (q ^ swz(l))
      const int own_r = swz(l);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const F4 t = *reinterpret_cast<const F4*>(own + 4 * (q ^ own_r));
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
      const F2 h = *reinterpret_cast<const F2*>(tile + tile_idx(-2));
      h0 = nw0 > 0 ? h.y : 0.0f;
      h1 = nw0 > 0 ? h.x : 0.0f;
    };

    float v[kCh], xh0, xh1;
    {
      const int64_t nw = (int64_t)wave * kWaveBlock;
      if (nw + kWaveBlock <= length) {
        stage_in(nw);
        vm_wait();
      } else {
        fill_ragged(nw);
      }
      lds_fence();
      take_chunk(nw, v, xh0, xh1);
      lds_fence();
      // the tile is free (the chunk is in registers): the second block's samples start travelling
      if (block_len < length && nw + block_len + kWaveBlock <= length) stage_in(nw + block_len);
    }
    int parity = 0, sbuf = 0;
    for (int64_t n0 = 0; n0 < length; n0 += block_len, parity ^= 1) {
      const int64_t nw = n0 + (int64_t)wave * kWaveBlock;       // first sample of this wave
      const int64_t nxt = nw + block_len;
      const bool has_next = n0 + block_len < length;            // workgroup-uniform
      const bool next_whole = has_next && nxt + kWaveBlock <= length;   // wave-uniform
      // 2. stages
      float hin0 = xh0, hin1 = xh1;                             // history of lane 0: inputs before this wave
      for (int st = 0; st < n_stages; ++st, sbuf ^= 1)
        stage_step<LAB>(tabs + st * kTabFloats, xch, W, n_stages, parity, st, sbuf, wave, lane, stage_clamp(clamp, st, n_stages), v, hin0, hin1);
      // 3. the next block's chunk -> registers (its copy has had all stages to land), then this block's chunk ->
      //    tile -> row-major pieces -> global
      float vn[kCh], nh0 = 0.0f, nh1 = 0.0f;
      if (has_next) {
        if (next_whole) {
          if (!(LAB & 1)) vm_wait();
        } else {
          fill_ragged(nxt);
        }
        lds_fence();
        take_chunk(nxt, vn, nh0, nh1);
        lds_fence();
      }
      {
        const int l = lane_now();
        float* own = tile + kHist + kCh * l;
        const int own_r = swz(l);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          *reinterpret_cast<F4*>(own + 4 * (q ^ own_r)) = F4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      }
      lds_fence();
      // 4. row-major pieces -> global; the copy of the block after the next one starts as soon as the tile has been read
      const int64_t nxt2 = nxt + block_len;
      const bool next2_whole = n0 + 2 * block_len < length && nxt2 + kWaveBlock <= length;   // wave-uniform
      if (vec_ok && nw + kWaveBlock <= length) {
        const int l = lane_now();
        f4v* p = reinterpret_cast<f4v*>(ys + nw) + l;
        const float* pe = tile + kHist + piece(l, 0);
        const float* po = tile + kHist + piece(l, 1);
        f4v r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = *reinterpret_cast<const f4v*>(((k & 1) ? po : pe) + 256 * k);
        lds_fence();
        if ((LAB & 128) && next2_whole) stage_in(nxt2);
        if (!(LAB & 32)) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (LAB & 256) __builtin_nontemporal_store(r[k], p + 64 * k);
            else p[64 * k] = r[k];
          }
        }
        if (!(LAB & 128) && next2_whole) stage_in(nxt2);
      } else {
        const int64_t left = length - nw;
        if (!(LAB & 32)) {
#pragma unroll 4
          for (int i = lane; i < kWaveBlock; i += 64)      // unaligned rows / ragged tail: element-wise, coalesced
            if (i < left) ys[nw + i] = tile[tile_idx(i)];
        }
        lds_fence();
        if (next2_whole) stage_in(nxt2);
      }
      if (has_next) {
#pragma unroll
        for (int j = 0; j < kCh; ++j) v[j] = vn[j];
        xh0 = nh0;
        xh1 = nh1;
      }
    }
  }
}
#endif  // __HIPCC__


// ---- the pipelined variant: <= 8 waves per workgroup, TWO tiles per wave ------------------------------------------------
// Measured on the cfg5a shard (profiles/r02_s_lfilter_lab.txt): with one tile per wave the stores of block i and the copy of
// block i + 2 are all issued in one burst between two stage loops; the arithmetic of a block (~16 us per CU) and the
// transfer of its 263 KB per CU (~11 us at the chip's copy rate) then overlap only partly (317 us where either alone costs
// ~235).  Here the block's input lands in an IN tile and its output is parked in an OUT tile, so both directions are issued
// a few instructions at a time at the head of every stage: the memory system sees an even stream and no wave queues behind
// a burst.  LDS: 8 waves x (2112 + 2048) floats = 133 KB.  Rows must be 16-byte aligned (the dispatcher keeps the
// one-tile kernel for the others).
constexpr int kPipeTile = kTile + kWaveBlock;
AAMD_HD size_t pipe_lds_bytes(int W, int n_stages) {
  return ((size_t)W * kPipeTile + (size_t)n_stages * kTabFloats + xch_floats(W, n_stages) + 4) * sizeof(float);
}

#if defined(__HIPCC__)
template <int LAB = 0>
__global__ void __launch_bounds__(512)
lfilter_wave_pipe_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                         float* __restrict__ y, int64_t n_seq, int channels, int64_t length, int n_order,
                         int n_coeff_rows, int n_stages, int clamp) {
  extern __shared__ __attribute__((aligned(16))) float smem_lfw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = blockDim.x >> 6;
  float* tile = smem_lfw + wave * kPipeTile;                 // IN tile: history row + block, swizzled (tile_idx)
  float* otile = tile + kTile;                               // OUT tile: block, same swizzle
  float* tabs = smem_lfw + W * kPipeTile;
  float* xch = tabs + n_stages * kTabFloats;
  const int64_t block_len = (int64_t)W * kWaveBlock;
  const unsigned tile_addr = (unsigned)(uintptr_t)tile;
  auto piece = [](int l, int odd) { return 32 * (l >> 3) + 4 * ((l & 7) ^ (4 * odd + (l >> 4))); };
  // lane constants of the copies and stores (256 registers at 8 waves: they may stay live)
  const unsigned src_e = 4u * (unsigned)piece(lane, 0), src_o = 4u * (unsigned)piece(lane, 1);
  const float* out_e = otile + piece(lane, 0);
  const float* out_o = otile + piece(lane, 1);
  float* own_in = tile + kHist + kCh * lane;
  float* own_out = otile + kCh * lane;
  const int own_r = swz(lane);

  int built_crow = -1;                                  // coefficient row of the tables currently in LDS
  for (int64_t seq = blockIdx.x; seq < n_seq; seq += gridDim.x) {
    const int ch = (int)(seq % channels);
    const int crow = (n_coeff_rows == 1) ? 0 : ch;
    __syncthreads();
    if (wave == 0) {
      if (lane < n_stages && crow != built_crow) {   // (once per coefficient row, not per sequence: see lfilter_wave_kernel)
        const int64_t coff = ((int64_t)lane * n_coeff_rows + crow) * n_order;
        build_stage_call(a + coff, b + coff, n_order, tabs + lane * kTabFloats);
      }
      for (int i = lane; i < xch_floats(W, n_stages); i += 64) xch[i] = 0.0f;
    }
    built_crow = crow;
    __syncthreads();
    const float* xs = x + seq * length;
    float* ys = y + seq * length;

    // pieces [k0, k1) of the copy of the whole wave block at nw0 (piece 0 also fetches the history row)
    auto copy_pieces = [&](int64_t nw0, int k0, int k1) {
      if (LAB & 16) return;
      const float* src = xs + nw0;
      if (k0 == 0 && k1 > 0 && nw0 > 0) glds4(src - kHist, 4u * (unsigned)lane, tile_addr);
#pragma unroll 1
      for (int i = k0; i < k1; ++i) glds16<false>(src + 256 * i, (i & 1) ? src_o : src_e, tile_addr + 4 * kHist + 1024 * i);
    };
    // pieces [k0, k1) of the stores of the whole wave block at nw0 from the OUT tile
    auto store_pieces = [&](int64_t nw0, int k0, int k1) {
      if (LAB & 32) return;
      f4v* p = reinterpret_cast<f4v*>(ys + nw0) + lane;
#pragma unroll 1
      for (int k = k0; k < k1; ++k) p[64 * k] = *reinterpret_cast<const f4v*>(((k & 1) ? out_o : out_e) + 256 * k);
    };
    auto store_ragged = [&](int64_t nw0) {
      if (LAB & 32) return;
      const int64_t left = length - nw0;
#pragma unroll 4
      for (int i = lane; i < kWaveBlock; i += 64)
        if (i < left) ys[nw0 + i] = otile[tile_idx(i) - kHist];
    };
    auto fill_ragged = [&](int64_t nw0) {
      const int64_t left = length - nw0;
#pragma unroll 1
      for (int i = lane - 64; i < kWaveBlock; i += 64) tile[tile_idx(i)] = (i < left && nw0 + i >= 0) ? xs[nw0 + i] : 0.0f;
    };
    auto take_chunk = [&](int64_t nw0, float (&v)[kCh], float& h0, float& h1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const F4 t = *reinterpret_cast<const F4*>(own_in + 4 * (q ^ own_r));
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
      const F2 h = *reinterpret_cast<const F2*>(tile + tile_idx(-2));
      h0 = nw0 > 0 ? h.y : 0.0f;
      h1 = nw0 > 0 ? h.x : 0.0f;
    };

    float v[kCh], hin0, hin1;
    {
      const int64_t nw = (int64_t)wave * kWaveBlock;
      if (nw + kWaveBlock <= length) {
        copy_pieces(nw, 0, 8);
        vm_wait();
      } else {
        fill_ragged(nw);
      }
      lds_fence();
      take_chunk(nw, v, hin0, hin1);
      lds_fence();
    }
    // the copies are finished one stage before the end of the block (they must have landed when the stages are done);
    // the stores are spread over all stages
    const int copy_stages = n_stages > 1 ? n_stages - 1 : 1;
    int parity = 0, sbuf = 0;
    for (int64_t n0 = 0; n0 < length; n0 += block_len, parity ^= 1) {
      const int64_t nw = n0 + (int64_t)wave * kWaveBlock;       // this block
      const int64_t nxt = nw + block_len, prv = nw - block_len;
      const bool has_next = n0 + block_len < length;            // workgroup-uniform
      const bool next_whole = has_next && nxt + kWaveBlock <= length;   // wave-uniform
      const bool prev_whole = n0 > 0 && prv + kWaveBlock <= length;     // the block parked in the OUT tile
      for (int st = 0; st < n_stages; ++st, sbuf ^= 1) {
        if (next_whole && st < copy_stages) copy_pieces(nxt, 8 * st / copy_stages, 8 * (st + 1) / copy_stages);
        if (prev_whole) store_pieces(prv, 8 * st / n_stages, 8 * (st + 1) / n_stages);
        stage_step<LAB>(tabs + st * kTabFloats, xch, W, n_stages, parity, st, sbuf, wave, lane, stage_clamp(clamp, st, n_stages), v, hin0, hin1);
      }
      // (the block before the last can only be ragged if this one is the last: nothing left to do for it)
      // this block's output -> OUT tile (its former content has been read: the loads of the stores are waited for at issue)
      lds_fence();
#pragma unroll
      for (int q = 0; q < 8; ++q)
        *reinterpret_cast<F4*>(own_out + 4 * (q ^ own_r)) = F4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      lds_fence();
      if (has_next) {
        if (next_whole) {
          if (!(LAB & 1)) vm_wait();
        } else {
          fill_ragged(nxt);
        }
        lds_fence();
        take_chunk(nxt, v, hin0, hin1);
        lds_fence();
      }
    }
    // the last block is still parked in the OUT tile
    {
      const int64_t last_n0 = ((length - 1) / block_len) * block_len;
      const int64_t nw = last_n0 + (int64_t)wave * kWaveBlock;
      if (nw + kWaveBlock <= length) store_pieces(nw, 0, 8);
      else store_ragged(nw);
      lds_fence();
    }
  }
}
#endif  // __HIPCC__

// ---- the mover variant (cascades of >= 3 stages): 8 filter waves + 4 MOVER waves per workgroup -------------------------
// profiles/r02_s_lfilter_lab.txt: the pipelined kernel without any memory instruction runs the cfg5a shard in 210 us, with its
// copies OR its stores in 251 us, with both in 296-305 us -- every LDS-DMA / store instruction costs the issuing wave a few
// hundred cycles (issue + LDS read latency), and with 2 waves per SIMD nothing covers them.  Here the filter waves execute no
// vector-memory instruction inside the block loop at all: four extra waves of the workgroup own the traffic.  They take part
// in the stages' barriers, which is also what orders their LDS accesses with the filter waves':
//   slot (i, st) = what a mover does between barrier (i, st - 1) and barrier (i, st) of block i;
//   round 6 (kSlot0): one more barrier per block, X(i), right behind the filter waves' exchange (block i - 1 -> OUT tiles, block
//                i + 1 <- IN tiles, after barrier (i - 1, n - 1)) orders the exchange with the movers, so that slot (i, 0) -- the
//                longest one: it holds a whole stage -- carries traffic too (cfg5a shard 238.7 -> 231.4 us, bit-identical);
//   slots (i, 0 .. n - 1): the stores of block i - 1 from the OUT tiles, an even share per slot;
//   slots (i, 0 .. n - 2): the copies of block i + 1 into the IN tiles; slot (i, n - 1) starts with vmcnt(0), so the copies have
//                landed when the filter waves pass barrier (i, n - 1) and read them.
//   (rounds 2 - 5, AAMD_LFW_SLOT0 = 0: no X(i), slot (i, 0) empty -- only barrier (i, 0) ordered the exchange with the movers --
//   stores in slots 1 .. n - 1, copies in slots 1 .. n - 2.)
constexpr int kMovers = 4;
#ifndef AAMD_LFW_PREFETCH
#define AAMD_LFW_PREFETCH 1
#endif
constexpr bool kMoverPrefetch = AAMD_LFW_PREFETCH != 0;
#ifndef AAMD_LFW_SLOT0
#define AAMD_LFW_SLOT0 1
#endif
constexpr bool kSlot0 = AAMD_LFW_SLOT0 != 0;

#if defined(__HIPCC__)
template <int LAB = 0>
#ifndef AAMD_LFW_W12TEST
#define AAMD_LFW_W12TEST 0     // lab, arithmetic-only runs: 12 filter waves whose tiles alias those of waves 0 .. 7 (wrong results)
#endif
__global__ void __launch_bounds__(64 * ((AAMD_LFW_W12TEST ? 12 : 8) + kMovers))
lfilter_wave_mover_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                          float* __restrict__ y, int64_t n_seq, int channels, int64_t length, int n_order,
                          int n_coeff_rows, int n_stages, int clamp) {
  extern __shared__ __attribute__((aligned(16))) float smem_lfw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = (blockDim.x >> 6) - kMovers;                 // filter waves
#ifndef AAMD_LFW_STAGGER
#define AAMD_LFW_STAGGER 0
#endif
  if (AAMD_LFW_STAGGER != 0) {
    // lab (tools/lfw_ab.py): every workgroup walks an equally long sequence block by block, so the whole chip issues its copies
    // and its stores in the same phase of the ~10 us block period.  Start the workgroups in 8 phases, `units` x 64 cycles apart:
    // mode 0 by XCD (block & 7), 1 within an XCD ((block >> 3) & 7), 2 both.
    constexpr int mode = AAMD_LFW_STAGGER >> 8;
    const int b_ = (int)blockIdx.x;
    const int k_ = mode == 0 ? (b_ & 7) : mode == 1 ? ((b_ >> 3) & 7) : ((b_ * 5 + (b_ >> 3)) & 7);
    for (int i = 0; i < k_; ++i) __builtin_amdgcn_s_sleep(AAMD_LFW_STAGGER & 127);
  }
  float* tabs = smem_lfw + (AAMD_LFW_W12TEST ? 8 : W) * kPipeTile;
  float* xch = tabs + n_stages * kTabFloats;
  const int64_t block_len = (int64_t)W * kWaveBlock;
  auto piece = [](int l, int odd) { return 32 * (l >> 3) + 4 * ((l & 7) ^ (4 * odd + (l >> 4))); };
  const int pc_e = piece(lane, 0), pc_o = piece(lane, 1);
  const int own_r = swz(lane);

  int built_crow = -1;                                  // coefficient row of the tables currently in LDS
  for (int64_t seq = blockIdx.x; seq < n_seq; seq += gridDim.x) {
    const int ch = (int)(seq % channels);
    const int crow = (n_coeff_rows == 1) ? 0 : ch;
    __syncthreads();
    if (wave == 0) {
      if (lane < n_stages && crow != built_crow) {   // (once per coefficient row, not per sequence: see lfilter_wave_kernel)
        const int64_t coff = ((int64_t)lane * n_coeff_rows + crow) * n_order;
        build_stage_call(a + coff, b + coff, n_order, tabs + lane * kTabFloats);
      }
      for (int i = lane; i < xch_floats(W, n_stages); i += 64) xch[i] = 0.0f;
    }
    built_crow = crow;
    __syncthreads();
    const float* xs = x + seq * length;
    float* ys = y + seq * length;

    // pieces [k0, k1) of the copy of the whole wave block at nw0 into filter wave t's IN tile; k = 0 is the history row,
    // k = 1 .. 8 the block
    auto copy_pieces = [&](int t, int64_t nw0, int k0, int k1) {
      if (LAB & 16) return;
      const float* src = xs + nw0;
      const unsigned in_addr = (unsigned)(uintptr_t)(smem_lfw + t * kPipeTile);
#pragma unroll 1
      for (int k = k0; k < k1; ++k) {
        if (k == 0) {
          if (nw0 > 0) glds4(src - kHist, 4u * (unsigned)lane, in_addr);
        } else {
          const int i = k - 1;
          glds16<false>(src + 256 * i, 4u * (unsigned)((i & 1) ? pc_o : pc_e), in_addr + 4 * kHist + 1024 * i);
        }
      }
    };
    // pieces [k0, k1) of the stores of the whole wave block at nw0 from filter wave t's OUT tile, two LDS reads in flight
    auto store_pieces = [&](int t, int64_t nw0, int k0, int k1) {
      if (LAB & 32) return;
      const float* ot = smem_lfw + t * kPipeTile + kTile;
      f4v* p = reinterpret_cast<f4v*>(ys + nw0) + lane;
      int k = k0;
#pragma unroll 1
      for (; k + 1 < k1; k += 2) {
        const f4v r0 = *reinterpret_cast<const f4v*>(ot + ((k & 1) ? pc_o : pc_e) + 256 * k);
        const f4v r1 = *reinterpret_cast<const f4v*>(ot + (((k + 1) & 1) ? pc_o : pc_e) + 256 * (k + 1));
        p[64 * k] = r0;
        p[64 * (k + 1)] = r1;
      }
      if (k < k1) p[64 * k] = *reinterpret_cast<const f4v*>(ot + ((k & 1) ? pc_o : pc_e) + 256 * k);
    };
    auto fill_ragged = [&](int t, int64_t nw0) {
      float* tl = smem_lfw + t * kPipeTile;
      const int64_t left = length - nw0;
#pragma unroll 1
      for (int i = lane - 64; i < kWaveBlock; i += 64) tl[tile_idx(i)] = (i < left && nw0 + i >= 0) ? xs[nw0 + i] : 0.0f;
    };

    if (wave >= W) {
      // ---------------------------------------------------------------- a mover wave (serves every kMovers-th filter wave)
      // (measured and dropped: two movers that only copy + two that only store, the stores issued BEHIND the slot's barrier
      // from registers -- 300 us against 290 us for this form)
      const int m = wave - W;
      // kSlot0: one more barrier per block, right behind the filter waves' exchange (block i - 1 -> OUT tiles, block i + 1 <- IN
      // tiles), orders it with the movers, so slot (i, 0) -- the longest one -- carries traffic too: stores in n slots, copies in n - 1
      const int s0 = kSlot0 ? 0 : 1;
      // (measured beside it, profiles/r06_zw_*: copies in one slot fewer +- 0, movers at a raised issue priority + 1.5 %)
      const int sslots = n_stages - s0, cslots = n_stages - 1 - s0;    // slots with stores / with copies
      for (int64_t n0 = 0; n0 < length; n0 += block_len) {
        const bool has_next = n0 + block_len < length;
        if (kSlot0 && !(LAB & 2)) lds_barrier();
        for (int st = 0; st < n_stages; ++st) {
          if (st >= s0) {
            if (st == n_stages - 1 && !(LAB & 1)) vm_wait();          // the copies of block i + 1 have landed
            for (int t = m; t < W; t += kMovers) {
              const int64_t nw = n0 + (int64_t)t * kWaveBlock;
              if (n0 > 0) store_pieces(t, nw - block_len, 8 * (st - s0) / sslots, 8 * (st - s0 + 1) / sslots);
              if (has_next) {
                const int64_t nxt = nw + block_len;
                if (nxt + kWaveBlock <= length) {
                  if (st - s0 < cslots) copy_pieces(t, nxt, 9 * (st - s0) / cslots, 9 * (st - s0 + 1) / cslots);
                } else if (st == s0) {
                  fill_ragged(t, nxt);
                }
              }
            }
          }
          if (!(LAB & 2)) lds_barrier();
        }
      }
      continue;
    }

    // ------------------------------------------------------------------ a filter wave
    float* tile = smem_lfw + (AAMD_LFW_W12TEST ? (wave & 7) : wave) * kPipeTile;
    float* otile = tile + kTile;
    float* own_in = tile + kHist + kCh * lane;
    float* own_out = otile + kCh * lane;
    auto take_chunk = [&](int64_t nw0, float (&v)[kCh], float& h0, float& h1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const F4 t = *reinterpret_cast<const F4*>(own_in + 4 * (q ^ own_r));
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      }
      const F2 h = *reinterpret_cast<const F2*>(tile + tile_idx(-2));
      h0 = nw0 > 0 ? h.y : 0.0f;
      h1 = nw0 > 0 ? h.x : 0.0f;
    };
    float v[kCh], hin0, hin1;
    {
      const int64_t nw = (int64_t)wave * kWaveBlock;
      if (nw + kWaveBlock <= length) {
        copy_pieces(wave, nw, 0, 9);
        vm_wait();
      } else {
        fill_ragged(wave, nw);
      }
      lds_fence();
      take_chunk(nw, v, hin0, hin1);
      lds_fence();
    }
    int parity = 0, sbuf = 0;
    int64_t nw_last = (int64_t)wave * kWaveBlock;
    for (int64_t n0 = 0; n0 < length; n0 += block_len, parity ^= 1) {
      const int64_t nw = n0 + (int64_t)wave * kWaveBlock;
      nw_last = nw;
      if (kSlot0 && !(LAB & 2)) lds_barrier();
      for (int st = 0; st < n_stages; ++st, sbuf ^= 1)
        stage_step<LAB, kMoverPrefetch>(tabs + st * kTabFloats, xch, W, n_stages, parity, st, sbuf, wave, lane, stage_clamp(clamp, st, n_stages), v, hin0, hin1);
      // behind barrier (i, n - 1): the movers have read block i - 1 out of the OUT tile and block i + 1 has landed in the IN tile
#pragma unroll
      for (int q = 0; q < 8; ++q)
        *reinterpret_cast<F4*>(own_out + 4 * (q ^ own_r)) = F4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      if (n0 + block_len < length) take_chunk(nw + block_len, v, hin0, hin1);
      lds_fence();
    }
    // the last block is still parked in the OUT tile: its filter wave stores it
    if (nw_last + kWaveBlock <= length) {
      store_pieces(wave, nw_last, 0, 8);
    } else if (!(LAB & 32)) {
      const int64_t left = length - nw_last;
#pragma unroll 4
      for (int i = lane; i < kWaveBlock; i += 64)
        if (i < left) ys[nw_last + i] = otile[tile_idx(i) - kHist];
    }
    lds_fence();
  }
}
#endif  // __HIPCC__

}  // namespace lfw
}  // namespace aamd

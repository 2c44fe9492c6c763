// F.fftconvolve for long impulse responses: overlap-save on a 16384-point complex FFT that lives
// entirely in LDS (functional/functional.py:2252-2258 computes irfft(rfft(x) * rfft(y)) at the full
// length nx + ny - 1; the contract is the linear-convolution result, so block-wise FFTs are free to
// differ in length).
//
//   * the taps y are cut into partitions of Pt <= 8192 taps; H_p = FFT_16384(y_p, zero padded) / N is
//     computed once per (y row, partition) into the caller's workspace, in the FFT's own
//     (digit-reversed) output order;
//   * one 1024-thread workgroup per CU walks (row, block pair) items: TWO real blocks of 16384 input
//     samples are packed as one complex sequence z = a + i b (convolution with a real h is linear, so
//     IFFT(FFT(z) H) = a*h + i b*h: no spectrum separation, no real-FFT post-pass);
//   * forward FFT = in-place decimation-in-frequency radix 16,16,16,4 (output digit-reversed), pointwise
//     product in that order, inverse = the mirrored decimation-in-time passes (input digit-reversed,
//     output natural): no reordering pass at all; the last forward radix-4 pass, the product and the first
//     inverse pass are fused in registers;
//   * a block of N = V + Pt - 1 inputs yields V valid outputs (first Pt - 1 wrap around and are dropped);
//     partitions p > 0 accumulate into the output in stream order;
//   * LDS index padding (4 complex per 64) keeps every pass' b64 accesses on distinct banks.
// Twiddles: W_N^m table (fp64-computed on device by a tiny kernel) at the head of the workspace; a
// butterfly loads w^1, w^2, w^4, w^8 and forms the other powers with <= 2 chained products.
#pragma once
#include "hd.h"

namespace aamd {
namespace fco {

constexpr int kN = 16384;                 // complex FFT length = real samples per block
constexpr int kThreads = 1024;
constexpr int kPerThread = kN / kThreads; // 16
constexpr int kLdsComplex = kN + (kN >> 6) * 4;   // padded: 17408 complex = 139 264 B
constexpr int kMaxPartTaps = 8192;

using C32 = cplx<float>;

AAMD_HD int pad_idx(int i) { return i + ((i >> 6) << 2); }

struct Geom {
  int64_t rows, nx, ny, start, out_len;   // after the operand swap: x = long operand, y = taps
  int n_part, part_taps;                  // partitions of the taps; taps per partition (last may be shorter)
  int v;                                  // valid outputs per block = kN - part_taps + 1
  int64_t n_blocks, n_pairs;              // blocks per row (per partition), block pairs per row
};

AAMD_HD void plan(int64_t ny, int64_t out_len, Geom& g) {
  g.n_part = (int)((ny + kMaxPartTaps - 1) / kMaxPartTaps);
  g.part_taps = (int)((ny + g.n_part - 1) / g.n_part);
  g.v = kN - g.part_taps + 1;
  g.n_blocks = (out_len + g.v - 1) / g.v;
  g.n_pairs = (g.n_blocks + 1) / 2;
}

template <bool conj_b>
AAMD_HD C32 cmulc(C32 a, C32 b) {   // a * b  or  a * conj(b)
  return conj_b ? C32{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}
                : C32{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}

// 4-point DFT, forward (W4 = -i) or inverse (W4 = +i), in place
template <bool inv>
AAMD_HD void dft4(C32& a, C32& b, C32& c, C32& d) {
  const C32 s0 = cadd(a, c), s1 = csub(a, c), s2 = cadd(b, d), s3 = csub(b, d);
  // forward: X1 = s1 - i s3, X3 = s1 + i s3; inverse swaps them
  const C32 m = {s3.y, -s3.x};   // -i * s3
  a = cadd(s0, s2);
  c = csub(s0, s2);
  if (!inv) { b = cadd(s1, m); d = csub(s1, m); }
  else      { b = csub(s1, m); d = cadd(s1, m); }
}

// 16-point DFT in registers: n = n1 + 4 n2, k = 4 k1 + k2 (see header of this file's pass functions)
template <bool inv>
AAMD_HD void dft16(C32 (&v)[16]) {
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, r2 = 0.70710678118654752f;
  // step 1: for each n1, DFT-4 over n2 of v[n1 + 4 n2]  -> A[n1][k2] stored at v[n1 + 4 k2]
#pragma unroll
  for (int n1 = 0; n1 < 4; ++n1) dft4<inv>(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);
  // step 2: multiply A[n1][k2] by W16^(n1 k2)
  const C32 w[10] = {{1.f, 0.f}, {c1, -s1}, {r2, -r2}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-r2, -r2},
                     {-c1, -s1}, {-1.f, 0.f}, {-c1, s1}};   // W16^0..9
#pragma unroll
  for (int n1 = 1; n1 < 4; ++n1)
#pragma unroll
    for (int k2 = 1; k2 < 4; ++k2) v[n1 + 4 * k2] = cmulc<inv>(v[n1 + 4 * k2], w[n1 * k2]);
  // step 3: for each k2, DFT-4 over n1 of v[n1 + 4 k2] -> X[4 k1 + k2] at v[k1 + 4 k2]
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) dft4<inv>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
  // reorder: X[k] for k = 4 k1 + k2 sits at v[k1 + 4 k2] -> transpose the 4x4 index
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const C32 t = v[a + 4 * b];
      v[a + 4 * b] = v[b + 4 * a];
      v[b + 4 * a] = t;
    }
}

// Hide a value from the optimiser: without it the compiler recognises that the forward and inverse
// passes of one block use the same twiddles, keeps ~150 of them live across the whole item loop and
// spills; re-loading 4 table entries per pass is far cheaper.
AAMD_HD int opaque(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// powers w^1..w^15 of w = W_N^e from the table (e * 15 < N): 4 loads + 11 products of depth <= 2
AAMD_HD void twiddle_powers(const C32* tw, int e, C32 (&p)[16]) {
  p[1] = tw[e]; p[2] = tw[2 * e]; p[4] = tw[4 * e]; p[8] = tw[8 * e];
  p[3] = cmul(p[1], p[2]); p[5] = cmul(p[4], p[1]); p[6] = cmul(p[4], p[2]); p[7] = cmul(p[4], p[3]);
  p[9] = cmul(p[8], p[1]); p[10] = cmul(p[8], p[2]); p[11] = cmul(p[8], p[3]); p[12] = cmul(p[8], p[4]);
  p[13] = cmul(p[8], p[5]); p[14] = cmul(p[8], p[6]); p[15] = cmul(p[8], p[7]);
}

// One radix-16 pass on sub-transforms of length LC (LC = 16384, 1024, 64), thread `tid` owns
// butterfly `tid`: elements base + t * m, m = LC / 16, base = (tid / m) * LC + tid % m.
//   forward (DIF): load, DFT-16, multiply output k by W_LC^(j k), store in place
//   inverse (DIT): load, multiply input k by conj(W_LC^(j k)), inverse DFT-16, store in place
template <int LC, bool inv>
AAMD_HD void pass16(int tid, C32* lds, const C32* tw) {
  constexpr int m = LC / 16;
  const int blk = tid / m, j = tid - blk * m;
  const int base = blk * LC + j;
  C32 v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) v[t] = lds[pad_idx(base + t * m)];
  C32 p[16];
  if (m > 1) twiddle_powers(tw, opaque(j * (kN / LC)), p);
  if (inv && m > 1) {
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmulc<true>(v[k], p[k]);
  }
  dft16<inv>(v);
  if (!inv && m > 1) {
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmulc<false>(v[k], p[k]);
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) lds[pad_idx(base + t * m)] = v[t];
}

// Middle step: last forward pass (radix 4 on sub-transforms of length 4: no twiddles), product with the
// tap spectrum (same digit-reversed positions), first inverse pass -- all in registers.  Thread `tid`
// owns butterflies tid + 1024 q (q = 0..3), i.e. elements 4 (tid + 1024 q) + t.
AAMD_HD void middle(int tid, C32* lds, const C32* H) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e0 = 4 * (tid + kThreads * q);
    C32 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = lds[pad_idx(e0 + t)];
    dft4<false>(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = cmul(v[t], H[e0 + t]);
    dft4<true>(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[pad_idx(e0 + t)] = v[t];
  }
}

// spectrum variant of the middle step (tap FFT): forward radix-4 only, result scaled and stored
AAMD_HD void middle_spectrum(int tid, const C32* lds, C32* H, float scale) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e0 = 4 * (tid + kThreads * q);
    C32 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = lds[pad_idx(e0 + t)];
    dft4<false>(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t) H[e0 + t] = C32{v[t].x * scale, v[t].y * scale};
  }
}

// ---- block I/O --------------------------------------------------------------------------------
// Block j of partition p covers outputs o in [j v, (j+1) v) of the slice; in full-convolution index
// n = start + o, partition-local n' = n - p * part_taps; its inputs are x[s0 .. s0 + kN), s0 = n'0 - (Pt-1).
AAMD_HD int64_t block_s0(const Geom& g, int p, int64_t j) {
  return g.start - (int64_t)p * g.part_taps + j * g.v - (g.part_taps - 1);
}

AAMD_HD void load_pair(int tid, const Geom& g, const float* xr, int p, int64_t j0, C32* lds) {
  const int64_t sa = block_s0(g, p, j0), sb = block_s0(g, p, j0 + 1);
  const bool has_b = j0 + 1 < g.n_blocks;
#pragma unroll 4
  for (int t = 0; t < kPerThread; ++t) {
    const int i = tid + kThreads * t;
    const int64_t ia = sa + i, ib = sb + i;
    C32 z;
    z.x = (ia >= 0 && ia < g.nx) ? xr[ia] : 0.0f;
    z.y = (has_b && ib >= 0 && ib < g.nx) ? xr[ib] : 0.0f;
    lds[pad_idx(i)] = z;
  }
}

AAMD_HD void store_pair(int tid, const Geom& g, const C32* lds, int p, int64_t j0, float* out_row) {
  const bool has_b = j0 + 1 < g.n_blocks;
#pragma unroll 4
  for (int t = 0; t < kPerThread; ++t) {
    const int i = tid + kThreads * t;
    if (i < g.part_taps - 1) continue;                    // wrapped-around samples
    const C32 z = lds[pad_idx(i)];
    const int64_t oa = j0 * g.v + (i - (g.part_taps - 1));
    const int64_t ob = oa + g.v;
    if (oa < g.out_len) out_row[oa] = (p == 0) ? z.x : out_row[oa] + z.x;
    if (has_b && ob < g.out_len) out_row[ob] = (p == 0) ? z.y : out_row[ob] + z.y;
  }
}

// taps of partition p of one y row -> LDS (real part; imaginary 0)
AAMD_HD void load_taps(int tid, const Geom& g, const float* yr, int p, C32* lds) {
  const int64_t t0 = (int64_t)p * g.part_taps;
#pragma unroll 4
  for (int t = 0; t < kPerThread; ++t) {
    const int i = tid + kThreads * t;
    const int64_t k = t0 + i;
    lds[pad_idx(i)] = C32{(i < g.part_taps && k < g.ny) ? yr[k] : 0.0f, 0.0f};
  }
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(256) twiddle_kernel(C32* tw) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < kN) {
    double s, c;
    sincospi(-2.0 * (double)m / (double)kN, &s, &c);
    tw[m] = C32{(float)c, (float)s};
  }
}

// H[(yrow * n_part + p) * kN + pos] = FFT(y_p)[digit-reversed pos] / kN
__global__ void __launch_bounds__(kThreads)
spectrum_kernel(Geom g, const float* __restrict__ y, const C32* __restrict__ tw, C32* __restrict__ H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_fco[];
  C32* lds = reinterpret_cast<C32*>(smem_fco);
  const int tid = threadIdx.x;
  const int64_t yrow = blockIdx.x / g.n_part;
  const int p = blockIdx.x - (int)yrow * g.n_part;
  load_taps(tid, g, y + yrow * g.ny, p, lds);
  __syncthreads();
  pass16<16384, false>(tid, lds, tw);
  __syncthreads();
  pass16<1024, false>(tid, lds, tw);
  __syncthreads();
  pass16<64, false>(tid, lds, tw);
  __syncthreads();
  middle_spectrum(tid, lds, H + (int64_t)blockIdx.x * kN, 1.0f / (float)kN);
}

// one partition of the taps: out (+)= conv(x, y_p) on the slice
__global__ void __launch_bounds__(kThreads)
overlap_save_kernel(Geom g, int p, const float* __restrict__ x, const C32* __restrict__ tw,
                    const C32* __restrict__ H, const int64_t* __restrict__ x_row_of,
                    const int64_t* __restrict__ y_row_of, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_fco[];
  C32* lds = reinterpret_cast<C32*>(smem_fco);
  const int tid = threadIdx.x;
  const int64_t n_items = g.rows * g.n_pairs;
#pragma unroll 1
  for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t row = item / g.n_pairs;
    const int64_t j0 = 2 * (item - row * g.n_pairs);
    const int64_t rx = x_row_of ? x_row_of[row] : row;
    const int64_t ry = y_row_of ? y_row_of[row] : row;
    load_pair(tid, g, x + rx * g.nx, p, j0, lds);
    __syncthreads();
    pass16<16384, false>(tid, lds, tw);
    __syncthreads();
    pass16<1024, false>(tid, lds, tw);
    __syncthreads();
    pass16<64, false>(tid, lds, tw);
    __syncthreads();
    middle(tid, lds, H + (ry * g.n_part + p) * (int64_t)kN);
    __syncthreads();
    pass16<64, true>(tid, lds, tw);
    __syncthreads();
    pass16<1024, true>(tid, lds, tw);
    __syncthreads();
    pass16<16384, true>(tid, lds, tw);
    __syncthreads();
    store_pair(tid, g, lds, p, j0, out + row * g.out_len);
    __syncthreads();
  }
}
#endif  // __HIPCC__

}  // namespace fco
}  // namespace aamd

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
//     the partitions of one output block pair are accumulated IN THE FREQUENCY DOMAIN, in registers
//     (16 complex per thread, at the positions the fused middle step owns): n_part forward FFTs, one
//     inverse FFT and one plain store per output pair -- no read-modify-write of the output;
//   * the first forward pass takes its 16 inputs per thread straight from global memory and the last
//     inverse pass stores its 16 outputs straight to global memory (thread tid owns elements
//     tid + 1024 t in both), so neither the input nor the output is staged through LDS; the loads of the
//     next partition are issued before the middle step of the current one;
//   * LDS index padding (4 complex per 64) keeps every pass' b64 accesses on distinct banks.
// Twiddles: W_N^m table (fp64-computed on device by a tiny kernel) at the head of the workspace.  Gathering
// W^e per lane from that 128 KB table costs one cache line per lane (it was the bottleneck of every pass), so
// each workgroup keeps a two-level copy in LDS -- W^(128 a), a < 128 and W^b, b < 128 (2 KB) -- and forms
// W^e = W^(128 (e >> 7)) W^(e & 127); a butterfly looks up w^1, w^2, w^4, w^8 that way and forms the other
// powers with <= 2 chained products.
#pragma once
#include "hd.h"

namespace aamd {
namespace fco {

constexpr int kN = 16384;                 // complex FFT length = real samples per block
constexpr int kThreads = 1024;
constexpr int kPerThread = kN / kThreads; // 16
constexpr int kLdsData = kN + (kN >> 6) * 4;      // padded: 17408 complex = 139 264 B
constexpr int kTwB = 256;                         // tl + kTwB: W_1024^(j k) table of pass B, [k - 1][j], j < 64
constexpr int kTwC = kTwB + 15 * 64;              // tl + kTwC: W_64^(j k) table of pass C, [k - 1][j], j < 4
constexpr int kLdsComplex = kLdsData + kTwC + 15 * 4;   // data + twiddle tables = 149 632 B
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

// two-level twiddle tables in LDS: tl[a] = W^(128 a), tl[128 + b] = W^b
// ... and the complete twiddles of the two inner passes: tl[kTwB + (k-1) 64 + j] = W_1024^(j k) = W_N^(16 j k),
// tl[kTwC + (k-1) 4 + j] = W_64^(j k) = W_N^(256 j k)   (call with tid = 0 .. 1023 once per workgroup)
AAMD_HD void twiddle_tables(int tid, const C32* tw, C32* tl) {
  if (tid < 128) tl[tid] = tw[128 * tid];
  else if (tid < 256) tl[tid] = tw[tid - 128];
  if (tid < 15 * 64) { const int k = tid / 64 + 1, j = tid % 64; tl[kTwB + tid] = tw[16 * j * k]; }
  else if (tid < 15 * 64 + 60) { const int u = tid - 15 * 64, k = u / 4 + 1, j = u % 4; tl[kTwC + u] = tw[256 * j * k]; }
}
AAMD_HD C32 tw_at(const C32* tl, int e) { return cmul(tl[e >> 7], tl[128 + (e & 127)]); }

// v[k] *= w^k (or conj) for k = 1..15, w = W_N^e (8 e < N): 4 look-ups, 11 products of depth <= 2.  Written
// low-to-high so that only w^1..w^8 are ever live next to v.
template <bool conj_w>
AAMD_HD void mul_twiddles(C32 (&v)[16], const C32* tl, int e) {
  C32 p[9];
  p[1] = tw_at(tl, e); p[2] = tw_at(tl, 2 * e); p[4] = tw_at(tl, 4 * e); p[8] = tw_at(tl, 8 * e);
  p[3] = cmul(p[1], p[2]); p[5] = cmul(p[4], p[1]); p[6] = cmul(p[4], p[2]); p[7] = cmul(p[4], p[3]);
#pragma unroll
  for (int k = 1; k <= 8; ++k) v[k] = cmulc<conj_w>(v[k], p[k]);
#pragma unroll
  for (int k = 1; k <= 7; ++k) v[8 + k] = cmulc<conj_w>(v[8 + k], cmul(p[8], p[k]));
}

// One radix-16 pass on sub-transforms of length LC (LC = 16384, 1024, 64), thread `tid` owns
// butterfly `tid`: elements base + t * m, m = LC / 16, base = (tid / m) * LC + tid % m.
//   forward (DIF): load, DFT-16, multiply output k by W_LC^(j k), store in place
//   inverse (DIT): load, multiply input k by conj(W_LC^(j k)), inverse DFT-16, store in place
template <int LC, bool inv>
AAMD_HD void pass16(int tid, C32* lds, const C32* tw) {
  constexpr int m = LC / 16;
  // pad_idx(base + t m) = pad_idx(base) + t ms, ms = m + 4 (m >> 6): exact for m = 1024 and 64 (multiples of 64) and for m = 4
  // (base % 64 = j < 4, so base + 4 t stays inside its group of 64).  ONE address per pass, the rest immediate offsets; with
  // the 16 padded indices computed separately the compiler kept every one of them live across the step loop and spilled --
  // and each reload of a spilled LDS address is an s_waitcnt vmcnt(0) that also waits for the prefetched inputs.
  constexpr int ms = m + 4 * (m >> 6);
  tid = opaque(tid);
  const int blk = tid / m, j = tid - blk * m;
  const int base = blk * LC + j;
  C32* cell = lds + pad_idx(base);
  C32 v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) v[t] = cell[t * ms];
  // inner passes read their 15 twiddles from the LDS tables (no products to form the powers)
  const C32* tab = tw + (LC == 1024 ? kTwB : kTwC) + j;
  if (inv && (LC == 1024 || LC == 64)) {
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmulc<true>(v[k], tab[(k - 1) * m]);
  } else if (inv && m > 1) {
    mul_twiddles<true>(v, tw, j * (kN / LC));
  }
  dft16<inv>(v);
  if (!inv && (LC == 1024 || LC == 64)) {
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmulc<false>(v[k], tab[(k - 1) * m]);
  } else if (!inv && m > 1) {
    mul_twiddles<false>(v, tw, j * (kN / LC));
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) cell[t * ms] = v[t];
}

// Middle step = last forward pass (radix 4 on sub-transforms of length 4: no twiddles), product with the
// tap spectrum (same digit-reversed positions), first inverse pass -- all in registers.  Thread `tid` owns, for
// q = 0..3, the 4 elements from middle_e0(tid, q): they lie inside the 1024-element sub-transform of the
// thread's own wave, like everything passes B and C touch, so B -> C -> middle (and back) need no workgroup
// barrier, only program order within the wave.
AAMD_HD int middle_e0(int tid, int q) { return 1024 * (tid >> 6) + 4 * ((tid & 63) + 64 * q); }

AAMD_HD void wave_sync() {     // orders this wave's LDS writes before its later LDS reads (LDS is in-order per wave)
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// spectrum variant of the middle step (tap FFT): forward radix-4 only, result scaled and stored
AAMD_HD void middle_spectrum(int tid, const C32* lds, C32* H, float scale) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e0 = middle_e0(tid, q);
    C32 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = lds[pad_idx(e0) + t];
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

// ---- register-resident first / last pass ---------------------------------------------------------
// pass16<16384, .> has m = 1024: thread tid owns elements tid + 1024 t -- the same elements a coalesced
// global load / store of the block gives it.
AAMD_HD void first_pass_from_regs(int tid, C32 (&v)[16], C32* lds, const C32* tw) {
  dft16<false>(v);
  tid = opaque(tid);
  mul_twiddles<false>(v, tw, tid);
  C32* cell = lds + pad_idx(tid);               // pad_idx(tid + 1024 t) = pad_idx(tid) + 1088 t
#pragma unroll
  for (int t = 0; t < 16; ++t) cell[t * 1088] = v[t];
}

AAMD_HD void last_pass_to_regs(int tid, const C32* lds, const C32* tw, C32 (&v)[16]) {
  tid = opaque(tid);
  const C32* cell = lds + pad_idx(tid);
#pragma unroll
  for (int t = 0; t < 16; ++t) v[t] = cell[t * 1088];
  mul_twiddles<true>(v, tw, tid);
  dft16<true>(v);
}

// inputs of two blocks starting at x[sa], x[sb] (either may reach outside [0, nx): zeros), element tid + 1024 t -> v[t] = a + i b
AAMD_HD void load_blocks(int tid, int64_t nx, const float* xr, int64_t sa, int64_t sb, bool has_b, C32 (&v)[16]) {
  // uniform block bases + one 32-bit lane offset (opaque: keeps the address arithmetic out of the prologue)
  const float* pa = xr + sa;
  const float* pb = xr + sb;
  const unsigned lane = (unsigned)opaque(tid);
  if (has_b && sa >= 0 && sb >= 0 && sa + kN <= nx && sb + kN <= nx) {   // interior pair (uniform branch): no bounds checks
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = C32{(pa + 1024 * t)[lane], (pb + 1024 * t)[lane]};
    return;
  }
  // edge pair: valid element ranges [lo, hi) of the two blocks as 32-bit block-local indices
  const int lo_a = (int)(sa < 0 ? (-sa < kN ? -sa : kN) : 0);
  const int hi_a = (int)(nx - sa < kN ? (nx - sa > 0 ? nx - sa : 0) : kN);
  const int lo_b = (int)(sb < 0 ? (-sb < kN ? -sb : kN) : 0);
  const int hi_b = !has_b ? 0 : (int)(nx - sb < kN ? (nx - sb > 0 ? nx - sb : 0) : kN);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = (int)lane + 1024 * t;
    v[t].x = (i >= lo_a && i < hi_a) ? (pa + 1024 * t)[lane] : 0.0f;
    v[t].y = (i >= lo_b && i < hi_b) ? (pb + 1024 * t)[lane] : 0.0f;
  }
}

// inputs of the block pair (j0, j0 + 1) of partition p
AAMD_HD void load_pair_regs(int tid, const Geom& g, const float* xr, int p, int64_t j0, C32 (&v)[16]) {
  load_blocks(tid, g.nx, xr, block_s0(g, p, j0), block_s0(g, p, j0 + 1), j0 + 1 < g.n_blocks, v);
}

// element i of block a -> oa[i] for skip <= i < hi_a (b likewise): uniform bases and limits
AAMD_HD void store_blocks(int tid, const C32 (&v)[16], float* oa, float* ob, int skip, int hi_a, int hi_b) {
  const unsigned lane = (unsigned)opaque(tid);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int i = (int)lane + 1024 * t;
    if (i >= skip && i < hi_a) (oa + 1024 * t)[lane] = v[t].x;
    if (i >= skip && i < hi_b) (ob + 1024 * t)[lane] = v[t].y;
  }
}

AAMD_HD void store_pair_regs(int tid, const Geom& g, const C32 (&v)[16], int64_t j0, float* out_row) {
  const bool has_b = j0 + 1 < g.n_blocks;
  const int skip = g.part_taps - 1;                       // wrapped-around samples
  float* oa = out_row + (j0 * g.v - skip);                // element i of block a -> oa[i]
  const int64_t left_a = g.out_len - (j0 * g.v - skip), left_b = left_a - g.v;
  const int hi_a = (int)(left_a < kN ? (left_a > 0 ? left_a : 0) : kN);
  const int hi_b = !has_b ? 0 : (int)(left_b < kN ? (left_b > 0 ? left_b : 0) : kN);
  store_blocks(tid, v, oa, oa + g.v, skip, hi_a, hi_b);
}

// middle step with the partitions accumulated in registers: acc[4 q + t] += DFT4(lds[4 (tid + 1024 q) + .])[t] * H
AAMD_HD void middle_accumulate(int tid, const C32* lds, const C32* H, C32 (&acc)[16]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e0 = middle_e0(tid, q);
    C32 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = lds[pad_idx(e0) + t];
    dft4<false>(v[0], v[1], v[2], v[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const C32 h = H[e0 + t];
      acc[4 * q + t].x += v[t].x * h.x - v[t].y * h.y;
      acc[4 * q + t].y += v[t].x * h.y + v[t].y * h.x;
    }
  }
}

// ... and the first inverse pass on the accumulated spectrum
AAMD_HD void middle_finish(int tid, C32 (&acc)[16], C32* lds) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e0 = middle_e0(tid, q);
    dft4<true>(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[pad_idx(e0) + t] = acc[4 * q + t];
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

// ---- frequency-domain delay line (uniform partitions) ---------------------------------------------------------------
// With block hop = partition length = kHop = kN / 2 the spectrum Z_j of the input segment x[start + (j - 1) kHop .. + kN) serves
// n_part consecutive output blocks:  Y_j = sum_p H_p Z_(j - p)  (H_p = spectrum of taps [p kHop, (p + 1) kHop)).  A workgroup
// walks consecutive blocks of one row segment and keeps the last n_part - 1 spectra -- the latest in registers, the older
// ones in a ring in global memory (each thread re-reads exactly the 32 elements it wrote: thread-private, coalesced, no
// synchronisation): ONE forward and one inverse FFT per step instead of n_part + 1.  The two real sequences packed into the complex FFT are the two halves of the segment
// (blocks j and j + hn), so both see the same delay.  A segment starts with n_part - 1 forward-only steps that fill the ring.
constexpr int kHop = kN / 2;
constexpr int kMaxFdlParts = 4;

struct FdlGeom {
  int64_t rows, nx, ny, start, out_len;
  int n_part;              // ceil(ny / kHop)
  int segs;                // segments per row: one work item each
  int64_t n_blocks;        // ceil(out_len / kHop)
  int64_t seg_blocks;      // blocks per segment (the last one may be shorter)
};

// cost in FFTs on the critical path of the busiest workgroup; returns true when the delay line is the cheaper plan
AAMD_HD bool plan_fdl(int64_t rows, int64_t ny, int64_t out_len, int cu_count, FdlGeom& f) {
  f.n_part = (int)((ny + kHop - 1) / kHop);
  f.n_blocks = (out_len + kHop - 1) / kHop;
  f.segs = 1;
  f.seg_blocks = f.n_blocks;
  if (f.n_part < 2 || f.n_part > kMaxFdlParts || rows < 1 || f.n_blocks < 2 || cu_count < 1) return false;
  Geom g{};
  plan(ny, out_len, g);
  const int64_t old_cost = ((rows * g.n_pairs + cu_count - 1) / cu_count) * (g.n_part + 1);
  int64_t best = -1;
  for (int segs = 1; segs <= 1024 && segs * 2 <= f.n_blocks; ++segs) {   // (a single very long row still fills the chip)
    const int64_t sb = (f.n_blocks + segs - 1) / segs, hn = (sb + 1) / 2;
    const int64_t used = (f.n_blocks + sb - 1) / sb;                      // segments that own blocks
    const int64_t cost = ((rows * used + cu_count - 1) / cu_count) * (2 * hn + f.n_part - 1);
    if (best < 0 || cost < best) { best = cost; f.segs = (int)used; f.seg_blocks = sb; }
  }
  return best > 0 && best * 10 <= old_cost * 9;
}

// the middle step of one delay-line step for logical thread tid: Z = radix-4 of the LDS data; with `produce`,
// Y = sum_p H_p Z_(-p) and the first inverse radix-4 back into LDS.  Z_(-1) stays in registers (zprev: the thread's own 16
// elements); Z_(-2) .. Z_(-(NP-1)) come from the ring (NP - 2 slots; Z_(-1) goes to `wslot`, the slot of the oldest, after
// that one has been read).  Ring traffic per step: one spectrum written, NP - 2 read (none at all for NP = 2).
// ... in two halves per q so that the kernel can request the operands of q + 1 before it works on q:
// fdl_fetch = the tap spectra H_0 .. H_(NP-1) and the ring spectra Z_(-2) .. of the 4 elements of q
template <int NP>
AAMD_HD void fdl_fetch(int tid, const C32* H, const C32* ring, int wslot, int q, C32 (&h)[NP][4], C32 (&zr)[NP][4]) {
  constexpr int R = NP - 2;
  // (opaque: the 2 x 4 x (2 NP - 2) element addresses are loop invariants of the step loop; hoisted, they were spilled)
  const int e0 = middle_e0(opaque(tid), q);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int t = 0; t < 4; ++t) h[p][t] = H[(int64_t)p * kN + e0 + t];
    if (p >= 2) {
      int sp = wslot - (p - 1);
      if (sp < 0) sp += R;
#pragma unroll
      for (int t = 0; t < 4; ++t) zr[p][t] = ring[(int64_t)sp * kN + e0 + t];
    }
  }
}
template <int NP>
AAMD_HD void fdl_apply(int tid, C32* lds, C32* ring, int wslot, bool produce, int q, const C32 (&h)[NP][4],
                       const C32 (&zr)[NP][4], C32 (&zprev)[16]) {
  constexpr int R = NP - 2;
  const int e0 = middle_e0(opaque(tid), q);
  C32 v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = lds[pad_idx(e0) + t];
  dft4<false>(v[0], v[1], v[2], v[3]);
  if (produce) {
    C32 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const C32 z1 = zprev[4 * q + t];
      acc[t] = C32{v[t].x * h[0][t].x - v[t].y * h[0][t].y, v[t].x * h[0][t].y + v[t].y * h[0][t].x};
      acc[t].x += z1.x * h[1][t].x - z1.y * h[1][t].y;
      acc[t].y += z1.x * h[1][t].y + z1.y * h[1][t].x;
#pragma unroll
      for (int p = 2; p < NP; ++p) {
        acc[t].x += zr[p][t].x * h[p][t].x - zr[p][t].y * h[p][t].y;
        acc[t].y += zr[p][t].x * h[p][t].y + zr[p][t].y * h[p][t].x;
      }
    }
    dft4<true>(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t) lds[pad_idx(e0) + t] = acc[t];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (R > 0) ring[(int64_t)wslot * kN + e0 + t] = zprev[4 * q + t];
    zprev[4 * q + t] = v[t];
  }
}
// the whole middle step; the operands of q + 1 are requested before q is worked on (the ring comes from HBM / the
// Infinity Cache: with the requests at the head of each q the step waited four round trips per half)
template <int NP>
AAMD_HD void middle_fdl(int tid, C32* lds, const C32* H, C32* ring, int wslot, bool produce, C32 (&zprev)[16]) {
  C32 ha[NP][4], za[NP][4], hb[NP][4], zb[NP][4];
  if (produce) fdl_fetch<NP>(tid, H, ring, wslot, 0, ha, za);
  if (produce) fdl_fetch<NP>(tid, H, ring, wslot, 1, hb, zb);
  fdl_apply<NP>(tid, lds, ring, wslot, produce, 0, ha, za, zprev);
  if (produce) fdl_fetch<NP>(tid, H, ring, wslot, 2, ha, za);
  fdl_apply<NP>(tid, lds, ring, wslot, produce, 1, hb, zb, zprev);
  if (produce) fdl_fetch<NP>(tid, H, ring, wslot, 3, hb, zb);
  fdl_apply<NP>(tid, lds, ring, wslot, produce, 2, ha, za, zprev);
  fdl_apply<NP>(tid, lds, ring, wslot, produce, 3, hb, zb, zprev);
}

// blocks of step s of an item: a = j_lo + s, b = j_lo + hn + s
AAMD_HD int64_t fdl_seg_start(const FdlGeom& f, int64_t j) { return f.start + (j - 1) * (int64_t)kHop; }
AAMD_HD void fdl_load(int tid, const FdlGeom& f, const float* xr, int64_t ja, int64_t jb, C32 (&v)[16]) {
  load_blocks(tid, f.nx, xr, fdl_seg_start(f, ja), fdl_seg_start(f, jb), true, v);
}
AAMD_HD void fdl_store(int tid, const FdlGeom& f, const C32 (&v)[16], int64_t ja, int64_t jb, int64_t j_hi, float* out_row) {
  // element i >= kHop of block j is output j kHop + i - kHop
  const int64_t left_a = f.out_len - (ja - 1) * (int64_t)kHop, left_b = f.out_len - (jb - 1) * (int64_t)kHop;
  const int hi_a = (int)(left_a < kN ? (left_a > 0 ? left_a : 0) : kN);
  const int hi_b = jb >= j_hi ? 0 : (int)(left_b < kN ? (left_b > 0 ? left_b : 0) : kN);
  store_blocks(tid, v, out_row + (ja - 1) * (int64_t)kHop, out_row + (jb - 1) * (int64_t)kHop, kHop, hi_a, hi_b);
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
  C32* tl = lds + kLdsData;
  twiddle_tables(tid, tw, tl);
  load_taps(tid, g, y + yrow * g.ny, p, lds);
  __syncthreads();
  pass16<16384, false>(tid, lds, tl);
  __syncthreads();
  pass16<1024, false>(tid, lds, tl);
  __syncthreads();
  pass16<64, false>(tid, lds, tl);
  __syncthreads();
  middle_spectrum(tid, lds, H + (int64_t)blockIdx.x * kN, 1.0f / (float)kN);
}

// out = conv(x, y) on the slice: one item = one pair of output blocks, all tap partitions.  512 threads, each
// playing TWO of the 1024 logical butterfly owners (tid and tid + 512, i.e. logical waves w and w + 8): 2 waves
// per SIMD with a 256-register budget, so the frequency-domain accumulators (64 registers) and the
// prefetched inputs of the next partition (64) stay in registers next to a pass' working set.
// Barriers: only around the first / last pass (which exchange data between waves); passes B, C and the
// middle step of a logical wave stay inside its own 1024-element sub-transform.
constexpr int kPhys = 512;
__global__ void __launch_bounds__(kPhys)
overlap_save_kernel(Geom g, const float* __restrict__ x, const C32* __restrict__ tw,
                    const C32* __restrict__ H, const int64_t* __restrict__ x_row_of,
                    const int64_t* __restrict__ y_row_of, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_fco[];
  C32* lds = reinterpret_cast<C32*>(smem_fco);
  const int t0 = threadIdx.x, t1 = threadIdx.x + kPhys;
  const int64_t n_items = g.rows * g.n_pairs;
  C32* tl = lds + kLdsData;
  twiddle_tables(t0, tw, tl);
  twiddle_tables(t1, tw, tl);
  __syncthreads();
  C32 v0[16], v1[16];
  if ((int64_t)blockIdx.x < n_items) {
    const int64_t row = blockIdx.x / g.n_pairs;
    const float* xr = x + (x_row_of ? x_row_of[row] : row) * g.nx;
    load_pair_regs(t0, g, xr, 0, 2 * (blockIdx.x - row * g.n_pairs), v0);
    load_pair_regs(t1, g, xr, 0, 2 * (blockIdx.x - row * g.n_pairs), v1);
  }
#pragma unroll 1
  for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t row = item / g.n_pairs;
    const int64_t j0 = 2 * (item - row * g.n_pairs);
    const int64_t rx = x_row_of ? x_row_of[row] : row;
    const int64_t ry = y_row_of ? y_row_of[row] : row;
    const float* xr = x + rx * g.nx;
    const C32* Hr = H + ry * g.n_part * (int64_t)kN;
    C32 acc0[16], acc1[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc0[t] = acc1[t] = C32{0.0f, 0.0f};
#pragma unroll 1
    for (int p = 0; p < g.n_part; ++p) {
      first_pass_from_regs(t0, v0, lds, tl);
      first_pass_from_regs(t1, v1, lds, tl);
      __syncthreads();
      if (p + 1 < g.n_part) {                                  // in flight during the remaining passes
        load_pair_regs(t0, g, xr, p + 1, j0, v0);
        load_pair_regs(t1, g, xr, p + 1, j0, v1);
      }
      const C32* Hp = Hr + (int64_t)p * kN;
      pass16<1024, false>(t0, lds, tl);
      wave_sync();
      pass16<64, false>(t0, lds, tl);
      wave_sync();
      middle_accumulate(t0, lds, Hp, acc0);
      pass16<1024, false>(t1, lds, tl);
      wave_sync();
      pass16<64, false>(t1, lds, tl);
      wave_sync();
      middle_accumulate(t1, lds, Hp, acc1);
      __syncthreads();                                          // the next first pass overwrites everything
    }
    middle_finish(t0, acc0, lds);
    wave_sync();
    pass16<64, true>(t0, lds, tl);
    wave_sync();
    pass16<1024, true>(t0, lds, tl);
    middle_finish(t1, acc1, lds);
    wave_sync();
    pass16<64, true>(t1, lds, tl);
    wave_sync();
    pass16<1024, true>(t1, lds, tl);
    __syncthreads();
    C32 w[16];
    last_pass_to_regs(t0, lds, tl, w);
    store_pair_regs(t0, g, w, j0, out + row * g.out_len);
    const int64_t nitem = item + gridDim.x;                   // next item's first inputs: in flight during the stores
    const int64_t nrow = nitem / g.n_pairs;
    const float* nxr = x + (x_row_of && nitem < n_items ? x_row_of[nrow] : nrow) * g.nx;
    if (nitem < n_items) load_pair_regs(t0, g, nxr, 0, 2 * (nitem - nrow * g.n_pairs), v0);
    last_pass_to_regs(t1, lds, tl, w);
    store_pair_regs(t1, g, w, j0, out + row * g.out_len);
    if (nitem < n_items) load_pair_regs(t1, g, nxr, 0, 2 * (nitem - nrow * g.n_pairs), v1);
    __syncthreads();
  }
}
// the delay-line variant of overlap_save_kernel: one item = one row segment, walked block by block (see above)
template <int NP>
__global__ void __launch_bounds__(kPhys)
overlap_save_fdl_kernel(FdlGeom f, const float* __restrict__ x, const C32* __restrict__ tw,
                        const C32* __restrict__ H, C32* __restrict__ ring, const int64_t* __restrict__ x_row_of,
                        const int64_t* __restrict__ y_row_of, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_fco[];
  C32* lds = reinterpret_cast<C32*>(smem_fco);
  const int t0 = threadIdx.x, t1 = threadIdx.x + kPhys;
  const int64_t n_items = f.rows * f.segs;
  C32* tl = lds + kLdsData;
  twiddle_tables(t0, tw, tl);
  twiddle_tables(t1, tw, tl);
  __syncthreads();
  constexpr int R = NP - 2;
  C32* my_ring = ring + (int64_t)blockIdx.x * (R > 0 ? R : 1) * kN;
  C32 v0[16], v1[16], zp0[16], zp1[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) zp0[t] = zp1[t] = C32{0.0f, 0.0f};
#pragma unroll 1
  for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t row = item / f.segs;
    const int64_t j_lo = (item - row * f.segs) * f.seg_blocks;
    const int64_t j_hi = j_lo + f.seg_blocks < f.n_blocks ? j_lo + f.seg_blocks : f.n_blocks;
    const int64_t hn = (j_hi - j_lo + 1) / 2;
    const float* xr = x + (x_row_of ? x_row_of[row] : row) * f.nx;
    const C32* Hr = H + (y_row_of ? y_row_of[row] : row) * NP * (int64_t)kN;
    float* out_row = out + row * f.out_len;
    int wslot = 0;
    fdl_load(t0, f, xr, j_lo - (NP - 1), j_lo + hn - (NP - 1), v0);
    fdl_load(t1, f, xr, j_lo - (NP - 1), j_lo + hn - (NP - 1), v1);
#pragma unroll 1
    for (int64_t s = -(NP - 1); s < hn; ++s) {
      const bool produce = s >= 0;
      first_pass_from_regs(t0, v0, lds, tl);
      first_pass_from_regs(t1, v1, lds, tl);
      __syncthreads();
#define AAMD_FDL_FWD(T, ZP)                                                                          \
      pass16<1024, false>(T, lds, tl);                                                               \
      wave_sync();                                                                                   \
      pass16<64, false>(T, lds, tl);                                                                 \
      wave_sync();                                                                                   \
      middle_fdl<NP>(T, lds, Hr, my_ring, wslot, produce, ZP);
#define AAMD_FDL_INV(T)                                                                              \
      if (produce) {                                                                                 \
        wave_sync();                                                                                 \
        pass16<64, true>(T, lds, tl);                                                                \
        wave_sync();                                                                                 \
        pass16<1024, true>(T, lds, tl);                                                              \
      }
      AAMD_FDL_FWD(t0, zp0)
      AAMD_FDL_INV(t0)
      AAMD_FDL_FWD(t1, zp1)
      if (s + 1 < hn) {   // the next step's inputs: in flight during the inverse passes and the stores (their 64 registers
                          // are the ones the middle step's operands have just left)
        fdl_load(t0, f, xr, j_lo + s + 1, j_lo + hn + s + 1, v0);
        fdl_load(t1, f, xr, j_lo + s + 1, j_lo + hn + s + 1, v1);
      }
      AAMD_FDL_INV(t1)
#undef AAMD_FDL_FWD
#undef AAMD_FDL_INV
      __syncthreads();
      if (produce) {
        C32 w[16];
        last_pass_to_regs(t0, lds, tl, w);
        fdl_store(t0, f, w, j_lo + s, j_lo + hn + s, j_hi, out_row);
        last_pass_to_regs(t1, lds, tl, w);
        fdl_store(t1, f, w, j_lo + s, j_lo + hn + s, j_hi, out_row);
        __syncthreads();
      }
      if (R > 0) wslot = wslot + 1 == R ? 0 : wslot + 1;
    }
  }
}
#endif  // __HIPCC__

}  // namespace fco
}  // namespace aamd

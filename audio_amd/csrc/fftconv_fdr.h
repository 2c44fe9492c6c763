// F.fftconvolve for 193 .. 32768 taps (4 ms .. 0.68 s impulse responses at 48 kHz; BASELINE config 5b): overlap-save and, beyond
// 8192 taps, the frequency-domain delay line on REAL blocks, the whole state of a row on one CU (functional/functional.py:2252-2258 computes
// irfft(rfft(x) * rfft(y)); the contract is the linear convolution, so block-wise FFTs of another length are free).
//
// What it replaces (fco::overlap_save_fdl_kernel, fftconv_os.h): a 16384-point COMPLEX FFT in LDS (two real blocks per
// transform, 150 of 160 KB), 512 threads at 256 registers = 2 waves per SIMD -- a wave64 VALU instruction issues every ~4
// cycles there (profiles/r03_zz_valu_issue.txt) -- and a spectrum ring in global memory (256 KB of ring + 384 KB of tap
// spectra through L2 per step and CU for 128 KB of algorithmic I/O).  VERDICT r3: 0.138 of the HBM roofline.
//
// Formulation (MI355X: 160 KB LDS, 512 KB of registers per CU, 4 waves per SIMD at <= 128 registers):
//   * one step = ONE real block of N = 16384 input samples (hop B = 8192) as an M = 8192-point complex FFT of
//     z[n] = s[2n] + i s[2n+1]; the spectrum of the real block follows from the split  Z[k] = E + T, Z[M-k] = conj(E - T),
//     E = (C[k] + conj C[M-k]) / 2, T = -i W_N^k (C[k] - conj C[M-k]) / 2 (and back by the merge) -- 68 KB of LDS;
//   * 1024 threads; thread t owns 8 spectrum bins for the whole launch, so the delay line Z_(j-1), Z_(j-2) is 2 x 16 REGISTERS
//     per thread: no ring in memory, no ring in LDS;  Y_j = H_0 Z_j + H_1 Z_(j-1) + H_2 Z_(j-2), tap spectra H_p (8192-tap
//     partitions) read from L2 in the thread-owned layout (coalesced, half the bytes per output of the complex kernel);
//   * radices (8, 8, 8, 8, 2), decimation in frequency forward / mirrored decimation in time inverse, no reordering pass:
//     the first pass takes its 8 inputs per thread straight from global memory (n = t + 1024 r: coalesced 8-byte loads), the
//     passes of length 128 and 16 stay inside a wave's own 512 elements (no workgroup barrier), the final radix-2 is a DPP
//     exchange between neighbouring lanes; the spectrum sits digit-reversed in LDS: frequency k = r1 + 8 r2 + 64 r3 + 512 r4
//     + 4096 r5 at position 1024 r1 + 128 r2 + 16 r3 + 2 r4 + r5;
//   * in that order bins k and k + 4096 are neighbours and so are their mirror bins 4096 - k and 8192 - k: a thread reads the
//     two pairs of its "quad" with two 16-byte LDS reads, splits, multiplies, merges and writes them back in place.
// Everything is AAMD_HD and replayed on the CPU by tests/cpu_sim (sim_fftconv_fdr).
#pragma once
#include "hd.h"
#include "fftconv_os.h"      // fco::C32, cmulc, dft4, opaque, wave_sync, block geometry helpers

namespace aamd {
namespace fdr {

using fco::C32;
constexpr int kM = 8192;                   // complex FFT length
constexpr int kN = 16384;                  // real samples per block
constexpr int kHop = 8192;                 // taps per partition; outputs per block of the 2 / 3-partition delay line
constexpr int kThreads = 1024;
constexpr int kMaxParts = 4;               // Z_j in flight + up to three delayed spectra in registers (round 5: 24 577 .. 32 768 taps too)
AAMD_HD int pad(int i) { return i + ((i >> 5) << 1); }            // 2 complex per 32: see the bank notes at each pass
constexpr int kLdsData = kM + (kM >> 5) * 2;                      // 8704 complex = 69 632 B
// complete twiddle tables behind the data (W_M = e^(-2 pi i / 8192)), [k - 1][j] so that the lanes of a wave read consecutive
// entries: 65 KB of the 90 KB this kernel leaves free -- forming w^2 .. w^7 from three look-ups cost a pass 28 of its ~130 vector
// instructions (first build: two-level table + products), a table entry costs one conflict-free ds_read_b64
//   kTw1 + 1024 (k - 1) + t: W_8192^(t k), t < 1024;   kTw2 + 128 (k - 1) + j: W_1024^(j k), j < 128;
//   kTw3 + 16 (k - 1) + j: W_128^(j k), j < 16;        kTw4 + 2 (k - 1) + j: W_16^(j k), j < 2
constexpr int kTw1 = 0, kTw2 = kTw1 + 7 * 1024, kTw3 = kTw2 + 7 * 128, kTw4 = kTw3 + 7 * 16, kTwEnd = kTw4 + 7 * 2;
constexpr int kFlags = kLdsData + kTwEnd;                         // 16 arrival counters (one dword per wave) of pair_sync
constexpr int kLdsComplex = kFlags + 8;                           // 16 902 complex = 135 216 B

// tw16k: the W_16384^m table of fco::twiddle_kernel (m < 16384); W_M^e = tw16k[2 e]
AAMD_HD void twiddle_tables(int tid, const C32* tw16k, C32* tl) {
  for (int u = tid; u < kTwEnd; u += kThreads) {
    int e;                                                        // exponent of W_M
    if (u < kTw2) { const int k = u / 1024 + 1, t = u % 1024; e = t * k; }
    else if (u < kTw3) { const int v = u - kTw2, k = v / 128 + 1, j = v % 128; e = 8 * j * k; }
    else if (u < kTw4) { const int v = u - kTw3, k = v / 16 + 1, j = v % 16; e = 64 * j * k; }
    else { const int v = u - kTw4, k = v / 2 + 1, j = v % 2; e = 512 * j * k; }
    tl[u] = tw16k[2 * e];                                         // 2 e < 16384 for every entry
  }
}
// ---- LDS access of a thread's 8 elements ---------------------------------------------------------------------------------
// hipcc pairs neighbouring 8-byte LDS accesses into ds_read2_b64 / ds_write2_b64, and on this part a ds_read2_b64 costs the
// LDS 8 cycles per wave-instruction against 2 for a ds_read_b64 (MI355X_MICROARCH.md, LDS table): the first build of this
// kernel issued 60 of them per thread and step and measured SQ_LDS_IDX_ACTIVE = 15 k cycles per step and CU (56 % of the step,
// profiles/r04_d_pmc_fftconv_fdr.txt).  The eight accesses of a pass are therefore written out as single ds_read_b64 /
// ds_write_b64: an empty asm statement with a memory clobber between two accesses keeps the load / store optimiser from pairing
// them (inline-asm ds_read_b64 with register-pair constraints made the allocator spill around every block).
// O0 .. O7: offsets in complex elements.
#if defined(__HIP_DEVICE_COMPILE__)
#define AAMD_FDR_FENCE asm volatile("" ::: "memory");     /* nothing moves (or merges) across it; costs no instruction */
#else
#define AAMD_FDR_FENCE
#endif
template <int O0, int O1, int O2, int O3, int O4, int O5, int O6, int O7>
AAMD_HD void lds_read8(const C32* cell, C32 (&v)[8]) {
  v[0] = cell[O0]; AAMD_FDR_FENCE v[1] = cell[O1]; AAMD_FDR_FENCE v[2] = cell[O2]; AAMD_FDR_FENCE v[3] = cell[O3]; AAMD_FDR_FENCE
  v[4] = cell[O4]; AAMD_FDR_FENCE v[5] = cell[O5]; AAMD_FDR_FENCE v[6] = cell[O6]; AAMD_FDR_FENCE v[7] = cell[O7]; AAMD_FDR_FENCE
}
template <int O0, int O1, int O2, int O3, int O4, int O5, int O6, int O7>
AAMD_HD void lds_write8(C32* cell, const C32 (&v)[8]) {
  cell[O0] = v[0]; AAMD_FDR_FENCE cell[O1] = v[1]; AAMD_FDR_FENCE cell[O2] = v[2]; AAMD_FDR_FENCE cell[O3] = v[3]; AAMD_FDR_FENCE
  cell[O4] = v[4]; AAMD_FDR_FENCE cell[O5] = v[5]; AAMD_FDR_FENCE cell[O6] = v[6]; AAMD_FDR_FENCE cell[O7] = v[7]; AAMD_FDR_FENCE
}
#define AAMD_FDR_STRIDE(S) 0, (S), 2 * (S), 3 * (S), 4 * (S), 5 * (S), 6 * (S), 7 * (S)
#define AAMD_FDR_M16 0, 16, 34, 50, 68, 84, 102, 118          /* 16 r + 2 (r >> 1) */
// the two neighbouring complex numbers at p (16-byte aligned): one ds_read_b128
AAMD_HD void lds_read_pair(const C32* p, C32& a, C32& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const F4 q = *reinterpret_cast<const F4*>(p);
  a = C32{q.x, q.y}; b = C32{q.z, q.w};
#else
  a = p[0]; b = p[1];
#endif
}

// v[k] *= tab[stride (k - 1)] (or its conjugate), k = 1 .. 7
template <bool conj_w, int STRIDE>
AAMD_HD void mul_table8(C32 (&v)[8], const C32* tab) {
  C32 w[8];
  lds_read8<0, STRIDE, 2 * STRIDE, 3 * STRIDE, 4 * STRIDE, 5 * STRIDE, 6 * STRIDE, 6 * STRIDE>(tab, w);    // (7 entries; the 8th re-reads the 7th)
#pragma unroll
  for (int k = 1; k < 8; ++k) v[k] = fco::cmulc<conj_w>(v[k], w[k - 1]);
}

// 8-point DFT in registers, natural order in and out; forward (e^-) or inverse (e^+, unnormalised)
template <bool inv>
AAMD_HD void dft8(C32 (&v)[8]) {
  constexpr float r2 = 0.70710678118654752f;
  fco::dft4<inv>(v[0], v[2], v[4], v[6]);      // E[k] -> v[0], v[2], v[4], v[6]
  fco::dft4<inv>(v[1], v[3], v[5], v[7]);      // O[k] -> v[1], v[3], v[5], v[7]
  // O[k] *= W8^k (forward: 1, (1 - i) / sqrt2, -i, (-1 - i) / sqrt2; inverse: conjugates)
  const C32 o1 = v[3], o2 = v[5], o3 = v[7];
  if (!inv) {
    v[3] = C32{(o1.x + o1.y) * r2, (o1.y - o1.x) * r2};
    v[5] = C32{o2.y, -o2.x};
    v[7] = C32{(o3.y - o3.x) * r2, -(o3.x + o3.y) * r2};
  } else {
    v[3] = C32{(o1.x - o1.y) * r2, (o1.y + o1.x) * r2};
    v[5] = C32{-o2.y, o2.x};
    v[7] = C32{-(o3.x + o3.y) * r2, (o3.x - o3.y) * r2};
  }
  // X[k] = E[k] + O'[k], X[k + 4] = E[k] - O'[k]
  const C32 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], p0 = v[1], p1 = v[3], p2 = v[5], p3 = v[7];
  v[0] = cadd(e0, p0); v[4] = csub(e0, p0);
  v[1] = cadd(e1, p1); v[5] = csub(e1, p1);
  v[2] = cadd(e2, p2); v[6] = csub(e2, p2);
  v[3] = cadd(e3, p3); v[7] = csub(e3, p3);
}

// ---- pass 1 (length 8192, m = 1024): thread tid owns elements tid + 1024 r -- what a coalesced load of the block gives it ----
AAMD_HD void first_pass_from_regs(int tid, C32 (&v)[8], C32* lds, const C32* tl) {
  dft8<false>(v);
  tid = fco::opaque(tid);
  mul_table8<false, 1024>(v, tl + kTw1 + tid);
  C32* cell = lds + pad(tid);                  // pad(tid + 1024 r) = pad(tid) + 1088 r; consecutive lanes, consecutive cells
  lds_write8<AAMD_FDR_STRIDE(1088)>(cell, v);
}
AAMD_HD void last_pass_to_regs(int tid, const C32* lds, const C32* tl, C32 (&v)[8]) {
  tid = fco::opaque(tid);
  const C32* cell = lds + pad(tid);
  lds_read8<AAMD_FDR_STRIDE(1088)>(cell, v);
  mul_table8<true, 1024>(v, tl + kTw1 + tid);
  dft8<true>(v);
}

// ---- pass 2 (length 1024, m = 128): butterfly j of block blk owns 1024 blk + j + 128 r; crosses waves (two per block half) ----
template <bool inv>
AAMD_HD void pass_m128(int tid, C32* lds, const C32* tl) {
  tid = fco::opaque(tid);
  const int blk = tid >> 7, j = tid & 127;
  C32* cell = lds + pad(1024 * blk + j);       // pad(base + 128 r) = pad(base) + 136 r; lanes read consecutive cells
  C32 v[8];
  lds_read8<AAMD_FDR_STRIDE(136)>(cell, v);
  if (inv) mul_table8<true, 128>(v, tl + kTw2 + j);
  dft8<inv>(v);
  if (!inv) mul_table8<false, 128>(v, tl + kTw2 + j);
  lds_write8<AAMD_FDR_STRIDE(136)>(cell, v);
}

// ---- pass 3 (length 128, m = 16): 128 blk + j + 16 r, inside the wave's own 512 elements ----
template <bool inv>
AAMD_HD void pass_m16(int tid, C32* lds, const C32* tl) {
  tid = fco::opaque(tid);
  // the wave's 64 butterflies are its four blocks x 16 j.  ds_read_b64 is served in groups of 32 lanes over 64 banks, ds_write_b64
  // in groups of 16 lanes over 32 banks (MI355X_MICROARCH.md): lanes 16 g .. 16 g + 15 take block (0, 2, 1, 3)[g], j = l & 15 --
  // a group of 16 writes 16 consecutive cells, a group of 32 reads cells 8 b + j (mod 32) of blocks b and b + 2 (136 = 8 mod 32,
  // so 16 apart): conflict-free both ways (model: tools/lab/fdr_lds_bank_model.py)
  const int sub = (tid >> 4) & 3;
  const int blk = ((tid >> 6) << 2) + (((sub & 1) << 1) | (sub >> 1)), j = tid & 15;
  C32* cell = lds + pad(128 * blk + j);        // offsets of r: 16 r + 2 (r >> 1)  (j < 16: the run of 32 changes every second r)
  const C32* tab = tl + kTw3 + j;
  C32 v[8];
  lds_read8<AAMD_FDR_M16>(cell, v);
  if (inv) mul_table8<true, 16>(v, tab);
  dft8<inv>(v);
  if (!inv) mul_table8<false, 16>(v, tab);
  lds_write8<AAMD_FDR_M16>(cell, v);
}

// The same pass with the first NR of its 7 twiddles W_128^(j k) held in registers: they depend on the lane only (j = tid & 15),
// so 2 NR registers replace NR ds_read_b64 in each direction of every step (round 5; the values are the table's own entries:
// bit-identical results).  w[k - 1] = tl[kTw3 + 16 (k - 1) + j]; twiddles NR + 1 .. 7 still come from the table.
template <int NR>
AAMD_HD void load_tw3(int tid, const C32* tl, C32 (&w)[7]) {
#pragma unroll
  for (int k = 0; k < 7; ++k) w[k] = k < NR ? tl[kTw3 + 16 * k + (tid & 15)] : C32{0.0f, 0.0f};
}
template <bool inv, int NR>
AAMD_HD void pass_m16_regtw(int tid, C32* lds, const C32* tl, const C32 (&wr)[7]) {
  tid = fco::opaque(tid);
  const int sub = (tid >> 4) & 3;
  const int blk = ((tid >> 6) << 2) + (((sub & 1) << 1) | (sub >> 1)), j = tid & 15;
  C32* cell = lds + pad(128 * blk + j);
  C32 v[8], w[7];
  lds_read8<AAMD_FDR_M16>(cell, v);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    if (k < NR) w[k] = wr[k];
    else { w[k] = tl[kTw3 + 16 * k + j]; AAMD_FDR_FENCE }
  }
  if (inv) {
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = fco::cmulc<true>(v[k], w[k - 1]);
  }
  dft8<inv>(v);
  if (!inv) {
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = fco::cmulc<false>(v[k], w[k - 1]);
  }
  lds_write8<AAMD_FDR_M16>(cell, v);
}

// ---- pass 4 (length 16, m = 2) + the final radix-2 (length 2): thread (blk, j) owns 16 blk + j + 2 r.  The radix-2 pairs
// positions (2 q, 2 q + 1) = output r of lane j = 0 with output r of lane j = 1: a neighbour exchange (DPP on the device; the
// CPU replay hands over the neighbour's array).  Forward: o = twiddled DFT-8 outputs; lane 0 keeps o0 + o1, lane 1 o0 - o1. ----
// lane map of the length-16 pass: lane l of wave w takes block 32 w + 16 (l >> 5) + 2 ((l >> 1) & 7) + ((l >> 4) & 1), j = l & 1
// (the radix-2 partners j = 0, 1 of a block stay neighbours): 16 consecutive lanes write cells j + 2 i (mod 16), 32 consecutive
// lanes read cells 16 odd + j + 2 i (mod 32) -- conflict-free both ways; (block l >> 1) collided two-way on every store
AAMD_HD int m2_block(int tid) {
  return ((tid >> 6) << 5) + (((tid >> 5) & 1) << 4) + (((tid >> 1) & 7) << 1) + ((tid >> 4) & 1);
}
AAMD_HD void pass_m2_fwd_a(int tid, const C32* lds, const C32* tl, C32 (&o)[8]) {
  tid = fco::opaque(tid);
  const int blk = m2_block(tid), j = tid & 1;
  const C32* cell = lds + pad(16 * blk + j);   // 16 blk + j + 2 r stays inside one run of 32: offsets 2 r
  lds_read8<AAMD_FDR_STRIDE(2)>(cell, o);
  dft8<false>(o);
  const C32* tab = tl + kTw4 + j;
  mul_table8<false, 2>(o, tab);
}
// Round 6: the radix-2 is ONE instruction per float on the device -- v_fmac_f32_dpp own, dpp(own), s = own + s nb, s = +1 in lane
// j = 0 and -1 in lane j = 1 -- so lane 1 holds o1 - o0: the NEGATED value.  The sign is carried, not repaired: every ODD
// position of the digit-reversed spectrum in LDS (the bins k + 4096 and 8192 - k of a quad) holds minus its bin.  The middle step
// reads and writes the pairs with that sign (free: other add / sub operands), and the inverse radix-2 -- own + s' nb with
// s' = -s -- turns (x0, -x1) into the true x0 + x1 and x0 - x1.  Rounds 3-5: v_mov (initialise) + v_mov_b32_dpp + add + sub +
// v_cndmask per float; the step is VALU-issue bound (profiles/r06_e_fdr_lab_radix32.txt).
AAMD_HD float radix2_sign(int tid) { return (tid & 1) ? -1.0f : 1.0f; }
AAMD_HD void radix2_level(C32 (&own)[8], const C32 (&nb)[8], float s) {      // the CPU replay's form of the DPP level
#pragma unroll
  for (int r = 0; r < 8; ++r) own[r] = C32{own[r].x + s * nb[r].x, own[r].y + s * nb[r].y};
}
AAMD_HD void pass_m2_fwd_store(int tid, const C32 (&w)[8], C32* lds) {
  tid = fco::opaque(tid);
  C32* cell = lds + pad(16 * m2_block(tid) + (tid & 1));
  lds_write8<AAMD_FDR_STRIDE(2)>(cell, w);
}
AAMD_HD void pass_m2_fwd_b(int tid, const C32 (&o)[8], const C32 (&nb)[8], C32* lds) {
  C32 w[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) w[r] = o[r];
  radix2_level(w, nb, radix2_sign(tid));
  pass_m2_fwd_store(tid, w, lds);
}
// inverse: x = the values at 16 blk + j + 2 r; lane 0 forms x0 + x1, lane 1 x0 - x1, then the inverse radix-8 pass
AAMD_HD void pass_m2_inv_a(int tid, const C32* lds, C32 (&x)[8]) {
  tid = fco::opaque(tid);
  const C32* cell = lds + pad(16 * m2_block(tid) + (tid & 1));
  lds_read8<AAMD_FDR_STRIDE(2)>(cell, x);
}
// inverse: the stored (x0, -x1) -> x0 + x1 in lane 0 (own - nb), x0 - x1 in lane 1 (own + nb): true values again
AAMD_HD void pass_m2_inv_finish(int tid, C32 (&v)[8], C32* lds, const C32* tl) {
  tid = fco::opaque(tid);
  const int blk = m2_block(tid), j = tid & 1;
  const C32* tab = tl + kTw4 + j;
  mul_table8<true, 2>(v, tab);
  dft8<true>(v);
  C32* cell = lds + pad(16 * blk + j);
  lds_write8<AAMD_FDR_STRIDE(2)>(cell, v);
}
AAMD_HD void pass_m2_inv_b(int tid, const C32 (&x)[8], const C32 (&nb)[8], C32* lds, const C32* tl) {
  C32 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = x[r];
  radix2_level(v, nb, -radix2_sign(tid));
  pass_m2_inv_finish(tid, v, lds, tl);
}

// ---- the middle step: split, delay line, merge ------------------------------------------------------------------------
// Position of frequency k (< 8192) in the digit-reversed spectrum.
AAMD_HD int rev_pos(int k) {
  return ((k & 7) << 10) + (((k >> 3) & 7) << 7) + (((k >> 6) & 7) << 4) + (((k >> 9) & 7) << 1) + (k >> 12);
}
// Thread t owns two QUADS s = 0, 1: q = 8 c + 2 h + s with (c, h) = mid_owner(t) (t = 0: c = h = 0), i.e. the bins k, k + 4096 (positions 2 q, 2 q + 1) with
// k = k_of_q(q) < 2048, and their mirror bins 4096 - k, 8192 - k (positions 2 q', 2 q' + 1).  Thread 0's quad 0 is the one
// exception: k = 0 -- bins 0 and 4096 mirror themselves (DC / Nyquist of the real block, and the bin M / 2) -- and it also
// takes the pair (2048, 6144) that no k < 2048 mirrors (positions 8, 9).
AAMD_HD int k_of_q(int q) { return (q >> 9) + 8 * ((q >> 6) & 7) + 64 * ((q >> 3) & 7) + 512 * (q & 7); }
struct MidConst {
  int a0, a1[2];         // padded LDS index of position 2 q of quad 0 (quad 1: + 2, same run of 32) and of position 2 q' of both
  C32 w0;                // W_N^k of quad 0 (the pair (k + 4096, 4096 - k) uses -i w); quad 1 has k + 512: w0 W_N^512, formed
                         // per step (4 instructions) instead of held (2 registers: the 128 of four waves per SIMD are all in use)
  AAMD_HD int a0s(int s) const { return a0 + 2 * s; }
  AAMD_HD C32 w(int s, int tid) const {
    const C32 c = C32{0.98078528040323043f, -0.19509032201612825f};       // W_16384^512 = e^(-i pi / 16)
    if (s == 0) return w0;
    return tid == 0 ? c : cmul(w0, c);                                    // (thread 0 holds W^2048 of its extra pair in w0)
  }
};
// Which quads a thread owns is free (the tap spectra are stored per owner); lane bits are permuted -- h = l0, c = 32 w +
// (l1, l2, l5, l3, l4) -- the best of the 720 bit permutations under the bank model of the 16-byte quad reads and 8-byte
// write-backs (conflict cycles 642 -> 336 per wave and step against 192 ideal; the plain order t -> (t >> 1, t & 1) was 3.3 x)
AAMD_HD int mid_owner_q0(int tid) {
  const int l = tid & 63;
  const int cl = ((l >> 1) & 3) | (((l >> 5) & 1) << 2) | (((l >> 3) & 1) << 3) | (((l >> 4) & 1) << 4);
  return 8 * (((tid >> 6) << 5) + cl) + 2 * (l & 1);
}
AAMD_HD void mid_init(int tid, const C32* tw16k, MidConst& mc) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = mid_owner_q0(tid) + s;
    const int k = k_of_q(q);
    if (s == 0) mc.a0 = pad(2 * q);
    const bool special = (tid == 0 && s == 0);
    mc.a1[s] = pad(special ? 8 : rev_pos(4096 - k));
    if (s == 0) mc.w0 = tw16k[special ? 2048 : k];
  }
}
// un-halved split of the pair (C[k], C[M - k]) with w = W_N^k:  zk = 2 Z[k], zm = 2 Z[M - k]
AAMD_HD void split_pair(C32 ck, C32 cm, C32 w, C32& zk, C32& zm) {
  const C32 e = C32{ck.x + cm.x, ck.y - cm.y};          // C[k] + conj C[M-k]
  const C32 d = C32{ck.x - cm.x, ck.y + cm.y};          // C[k] - conj C[M-k]
  const C32 wd = cmul(w, d);
  const C32 t = C32{wd.y, -wd.x};                       // -i w d
  zk = cadd(e, t);
  zm = C32{e.x - t.x, -(e.y - t.y)};                    // conj(e - t)
}
// un-halved merge: (Y[k], Y[M - k]) -> 2 C'[k], 2 C'[M - k]
AAMD_HD void merge_pair(C32 yk, C32 ym, C32 w, C32& ck, C32& cm) {
  const C32 e = C32{yk.x + ym.x, yk.y - ym.y};
  const C32 t = C32{yk.x - ym.x, yk.y + ym.y};
  const C32 wt = fco::cmulc<true>(t, w);                // t conj(w)
  const C32 d = C32{-wt.y, wt.x};                       // i conj(w) t
  ck = cadd(e, d);
  cm = C32{e.x - d.x, -(e.y - d.y)};
}
// the thread's 8 spectrum bins of the block in LDS -> z[0 .. 7] (twice the real block's spectrum):
//   z[4 s + 0] = Z[k], + 1: Z[k + 4096], + 2: Z[4096 - k], + 3: Z[8192 - k]
//   thread 0, s = 0: z[0] = (Z[0], Z[8192]) (both real), z[1] = Z[4096], z[2] = Z[2048], z[3] = Z[6144]
// (the ODD cell of every pair read here holds MINUS its bin -- see radix2_level: c1 and c3 are negated on the way in, which the
// compiler folds into the operands of the first additions)
AAMD_HD C32 cneg(C32 a) { return C32{-a.x, -a.y}; }
AAMD_HD void mid_split(int tid, const C32* lds, const MidConst& mc, C32 (&z)[8]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    C32 c0, c1, c2, c3;
    lds_read_pair(lds + mc.a0s(s), c0, c1);
    lds_read_pair(lds + mc.a1[s], c2, c3);
    c1 = cneg(c1); c3 = cneg(c3);
    if (tid == 0 && s == 0) {
      z[0] = C32{2.0f * (c0.x + c0.y), 2.0f * (c0.x - c0.y)};
      z[1] = C32{2.0f * c1.x, -2.0f * c1.y};
      split_pair(c2, c3, mc.w0, z[2], z[3]);
    } else {
      const C32 w = mc.w(s, tid);
      split_pair(c0, c3, w, z[4 * s], z[4 * s + 3]);
      split_pair(c1, c2, C32{w.y, -w.x}, z[4 * s + 1], z[4 * s + 2]);     // -i w = W_N^(k + 4096)
    }
  }
}
AAMD_HD void mid_merge(int tid, const C32 (&y)[8], const MidConst& mc, C32* lds) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    C32 c0, c1, c2, c3;
    if (tid == 0 && s == 0) {
      c0 = C32{y[0].x + y[0].y, y[0].x - y[0].y};
      c1 = C32{2.0f * y[1].x, -2.0f * y[1].y};
      merge_pair(y[2], y[3], mc.w0, c2, c3);
    } else {
      const C32 w = mc.w(s, tid);
      merge_pair(y[4 * s], y[4 * s + 3], w, c0, c3);
      merge_pair(y[4 * s + 1], y[4 * s + 2], C32{w.y, -w.x}, c1, c2);
    }
    // (the odd cells are stored negated: what the inverse radix-2 expects)
    lds[mc.a0s(s)] = c0; lds[mc.a0s(s) + 1] = cneg(c1); lds[mc.a1[s]] = c2; lds[mc.a1[s] + 1] = cneg(c3);
  }
}
// acc += h * z bin by bin (complex), except thread 0's slot 0 = two REAL bins packed into one complex number
AAMD_HD void mid_mac(int tid, const C32 (&h)[8], const C32 (&z)[8], C32 (&acc)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i == 0 && tid == 0) {
      acc[0].x += h[0].x * z[0].x;
      acc[0].y += h[0].y * z[0].y;
    } else {
      acc[i].x += z[i].x * h[i].x - z[i].y * h[i].y;
      acc[i].y += z[i].x * h[i].y + z[i].y * h[i].x;
    }
  }
}
// tap spectra in the thread-owned layout.  AAMD_FDR_H16 = 0: complex index ((p * 8 + i) * 1024 + tid) of a y row's table -- eight
// coalesced 8-byte loads per partition and thread; 1 (round 6): bins i and i + 1 of a thread lie side by side,
// ((p * 4 + i / 2) * 1024 + tid) * 2 + (i & 1) -- four 16-byte loads: the 192 KB of tap spectra a step reads are a throughput cost
// of the vector-memory path (the kernel without them ran 13 % faster, profiles/r05_c_fdr_lab_barriers_hloads.txt), and an 8-byte
// access moves bytes at 0.54-0.70 of the 16-byte rate (MI355X_MICROARCH.md)
#ifndef AAMD_FDR_H16
#define AAMD_FDR_H16 1
#endif
AAMD_HD int64_t h_index(int p, int i, int tid) {
#if AAMD_FDR_H16
  return (((int64_t)p * 4 + (i >> 1)) * kThreads + tid) * 2 + (i & 1);
#else
  return ((int64_t)p * 8 + i) * kThreads + tid;
#endif
}
constexpr int64_t kHPerPart = 8 * kThreads;            // complex numbers per partition (= 8192)

// ---- geometry --------------------------------------------------------------------------------------------------------
// Block j covers outputs [j hop, (j + 1) hop) of the slice; its 16384 input samples start at  start + j hop - skip, and sample
// i >= skip of the transformed block is output j hop + (i - skip):
//   2 / 3 partitions of 8192 taps: hop = skip = 8192 (the delay line needs hop = partition length);
//   1 partition (<= 8192 taps):    skip = taps - 1, hop = 16384 - skip -- every sample the circular convolution leaves valid
struct Geom {
  int64_t rows, nx, ny, start, out_len;
  int n_part;              // 1 .. 3
  int hop, skip;
  int segs;                // segments per row (one work item each)
  int64_t n_blocks;        // ceil(out_len / hop)
  int64_t seg_blocks;      // blocks per segment (the last may be shorter)
};
// cost = steps on the busiest workgroup (a forward-only warm-up step counts half); returns false when the shape is not served
AAMD_HD bool plan(int64_t rows, int64_t ny, int64_t out_len, int cu_count, Geom& g) {
  g.n_part = (int)((ny + kHop - 1) / kHop);
  if (g.n_part < 1 || g.n_part > kMaxParts || ny < 2) return false;
  g.skip = g.n_part == 1 ? (int)ny - 1 : kHop;
  g.hop = kN - g.skip;
  g.n_blocks = (out_len + g.hop - 1) / g.hop;
  g.segs = 1;
  g.seg_blocks = g.n_blocks;
  if (rows < 1 || g.n_blocks < 1 || cu_count < 1) return false;
  int64_t best = -1;
  for (int segs = 1; segs <= 1024 && segs <= g.n_blocks; ++segs) {
    const int64_t sb = (g.n_blocks + segs - 1) / segs;
    const int64_t used = (g.n_blocks + sb - 1) / sb;
    const int64_t cost = ((rows * used + cu_count - 1) / cu_count) * (2 * sb + (g.n_part - 1));
    if (best < 0 || cost < best) { best = cost; g.segs = (int)used; g.seg_blocks = sb; }
  }
  return true;
}
AAMD_HD int64_t seg_start(const Geom& g, int64_t j) { return g.start + j * (int64_t)g.hop - g.skip; }

// inputs of block j: z[n] = (x[s0 + 2 n], x[s0 + 2 n + 1]) for n = tid + 1024 r; zeros outside [0, nx)
AAMD_HD void load_block(int tid, const Geom& g, const float* xr, int64_t j, bool vec_ok, C32 (&v)[8]) {
  const int64_t s0 = seg_start(g, j);
  const unsigned lane = (unsigned)fco::opaque(tid);
  if (vec_ok && !(s0 & 1) && s0 >= 0 && s0 + kN <= g.nx) {     // interior block on an 8-byte aligned offset (uniform branch)
    const C32* p = reinterpret_cast<const C32*>(xr + s0);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = (p + 1024 * r)[lane];
    return;
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int64_t i = s0 + 2 * ((int64_t)lane + 1024 * r);
    v[r].x = (i >= 0 && i < g.nx) ? xr[i] : 0.0f;
    v[r].y = (i + 1 >= 0 && i + 1 < g.nx) ? xr[i + 1] : 0.0f;
  }
}
// outputs of block j: sample i = 2 n (+ 1) >= skip of the block is output j hop + i - skip of the slice.  vec_ok: the ROW starts
// on an even float offset; a pair (2 n, 2 n + 1) is then one 8-byte store when skip and j hop are even
AAMD_HD void store_block(int tid, const Geom& g, const C32 (&v)[8], int64_t j, int64_t j_hi, bool vec_ok, float* out_row) {
  if (j >= j_hi) return;
  const unsigned lane = (unsigned)fco::opaque(tid);
  const int64_t o0 = j * (int64_t)g.hop - g.skip;              // output index of block sample 0 (may be negative)
  if (g.skip == kHop && vec_ok && o0 + kN <= g.out_len) {       // delay-line blocks: the upper half, aligned, inside the row
    C32* p = reinterpret_cast<C32*>(out_row + o0);
#pragma unroll
    for (int r = 4; r < 8; ++r) (p + 1024 * r)[lane] = v[r];
    return;
  }
  const bool pair_ok = vec_ok && !(o0 & 1);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int i = 2 * ((int)lane + 1024 * r);
    const int64_t o = o0 + i;
    if (pair_ok && i >= g.skip && o + 1 < g.out_len) {
      *reinterpret_cast<C32*>(out_row + o) = v[r];
    } else {
      if (i >= g.skip && o < g.out_len) out_row[o] = v[r].x;
      if (i + 1 >= g.skip && o + 1 < g.out_len) out_row[o + 1] = v[r].y;
    }
  }
}
// taps of partition p of one y row as the packed real block (imaginary lane = odd samples)
AAMD_HD void load_taps(int tid, int64_t ny, const float* yr, int p, C32 (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int n = tid + 1024 * r;                        // block-local sample 2 n, 2 n + 1; taps occupy samples [0, kHop)
    const int64_t k = (int64_t)p * kHop + 2 * n;
    v[r].x = (2 * n < kHop && k < ny) ? yr[k] : 0.0f;
    v[r].y = (2 * n + 1 < kHop && k + 1 < ny) ? yr[k + 1] : 0.0f;
  }
}
constexpr float kSpectrumScale = 1.0f / (8.0f * (float)kM);      // un-halved split (2) x un-halved merge (2) x split of H (2) x M

#if defined(__HIPCC__)
// lab switches (tools/fdr_lab.py; timing only, the results are wrong): AAMD_FDR_LAB_NOBAR bit 0 turns the barriers around the
// length-1024 pass into wave-local fences, bit 1 those around the middle step, bit 2 the ones of the first / last pass --
// an upper bound on what any restructuring of the barriers could win; AAMD_FDR_LAB_NOH reads no tap spectra
#ifndef AAMD_FDR_LAB_NOBAR
#define AAMD_FDR_LAB_NOBAR 0
#endif
// The length-1024 pass runs on the 1024 elements of TWO neighbouring waves (w, w ^ 1) and the passes on either side of it are
// wave-local: the two workgroup barriers around it are replaced by a rendezvous of the pair (AAMD_FDR_PAIRSYNC, default on) --
// a wave waits for its neighbour only, not for the slowest of sixteen (the lab build without those two barriers: - 6.5 %).
// gfx950 has no named barriers: each wave publishes an arrival counter in LDS and polls its neighbour's.  LDS executes a
// wave's instructions in order, and the s_waitcnt in front of the flag write has every data write of the pass performed
// before the counter moves; the reads behind the poll are issued after the neighbour's counter has been seen.
#ifndef AAMD_FDR_PAIRSYNC
#define AAMD_FDR_PAIRSYNC 1
#endif
#ifndef AAMD_FDR_WAR_BARRIER
#define AAMD_FDR_WAR_BARRIER 0
#endif
__device__ __forceinline__ void pair_sync(unsigned* flags, unsigned gen) {
  // (release / acquire spelled out instead of relying on `volatile` and in-order LDS.  The fences name the LDS address
  // space only -- "local" -- so they lower to s_waitcnt lgkmcnt(0): a plain workgroup-scope release would also wait for the
  // vector-memory loads in flight, the next block's samples and the tap spectra, which have nothing to do with this hand-over.
  // The flag itself is a relaxed workgroup-scope atomic: never cached in a register, never torn.)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(flags + wave, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (;;) {
    const unsigned f = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)__hip_atomic_load(flags + (wave ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if ((int)(f - gen) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
#define AAMD_FDR_PAIR(BIT)                                                                \
  do {                                                                                    \
    if (AAMD_FDR_LAB_NOBAR & (BIT)) fco::wave_sync();                                      \
    else if (AAMD_FDR_PAIRSYNC) pair_sync(pair_flags, ++pair_gen);                         \
    else __syncthreads();                                                                 \
  } while (0)
#define AAMD_FDR_BARRIER(BIT) do { if (AAMD_FDR_LAB_NOBAR & (BIT)) fco::wave_sync(); else __syncthreads(); } while (0)
// the radix-2 between neighbouring lanes (lane ^ 1: DPP quad_perm [1, 0, 3, 2]) on all 16 floats of a thread, in place:
// u += s * dpp(u), one v_fmac_f32_dpp each (radix2_level).  One asm block: the s_nop in front covers the two wait states a DPP read
// needs behind a VALU write of the same register; inside the block every instruction touches its own register.
__device__ __forceinline__ void radix2_dpp(C32 (&u)[8], float s) {
#define AAMD_FDR_QP " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
  asm("s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %16" AAMD_FDR_QP "v_fmac_f32_dpp %1, %1, %16" AAMD_FDR_QP
      "v_fmac_f32_dpp %2, %2, %16" AAMD_FDR_QP "v_fmac_f32_dpp %3, %3, %16" AAMD_FDR_QP
      "v_fmac_f32_dpp %4, %4, %16" AAMD_FDR_QP "v_fmac_f32_dpp %5, %5, %16" AAMD_FDR_QP
      "v_fmac_f32_dpp %6, %6, %16" AAMD_FDR_QP "v_fmac_f32_dpp %7, %7, %16" AAMD_FDR_QP
      "v_fmac_f32_dpp %8, %8, %16" AAMD_FDR_QP "v_fmac_f32_dpp %9, %9, %16" AAMD_FDR_QP
      "v_fmac_f32_dpp %10, %10, %16" AAMD_FDR_QP "v_fmac_f32_dpp %11, %11, %16" AAMD_FDR_QP
      "v_fmac_f32_dpp %12, %12, %16" AAMD_FDR_QP "v_fmac_f32_dpp %13, %13, %16" AAMD_FDR_QP
      "v_fmac_f32_dpp %14, %14, %16" AAMD_FDR_QP "v_fmac_f32_dpp %15, %15, %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "+v"(u[0].x), "+v"(u[0].y), "+v"(u[1].x), "+v"(u[1].y), "+v"(u[2].x), "+v"(u[2].y), "+v"(u[3].x), "+v"(u[3].y),
        "+v"(u[4].x), "+v"(u[4].y), "+v"(u[5].x), "+v"(u[5].y), "+v"(u[6].x), "+v"(u[6].y), "+v"(u[7].x), "+v"(u[7].y)
      : "v"(s));
#undef AAMD_FDR_QP
}
// forward transform of the block in v (registers) to the digit-reversed spectrum in LDS; ends with a workgroup barrier
// how many of the 7 twiddles of the length-128 pass live in registers, by partition count (the 3-partition kernel holds two
// delayed spectra and has fewer registers to spare: tools/fdr_lab.py reports registers and scratch per setting)
#ifndef AAMD_FDR_TW3_N12
#define AAMD_FDR_TW3_N12 7
#endif
#ifndef AAMD_FDR_TW3_N3
#define AAMD_FDR_TW3_N3 5
#endif
#ifndef AAMD_FDR_TW3_N4
#define AAMD_FDR_TW3_N4 0
#endif
template <int NP> struct Tw3Regs { static constexpr int n = NP >= 4 ? AAMD_FDR_TW3_N4 : NP == 3 ? AAMD_FDR_TW3_N3 : AAMD_FDR_TW3_N12; };
template <int NR>
__device__ __forceinline__ void forward_block(int tid, C32 (&v)[8], C32* lds, const C32* tl, const C32 (&tw3)[7]) {
  first_pass_from_regs(tid, v, lds, tl);
  AAMD_FDR_BARRIER(4);
  pass_m128<false>(tid, lds, tl);
  AAMD_FDR_BARRIER(1);
  if (NR > 0) pass_m16_regtw<false, NR>(tid, lds, tl, tw3); else pass_m16<false>(tid, lds, tl);
  fco::wave_sync();
  C32 o[8];
  pass_m2_fwd_a(tid, lds, tl, o);
  radix2_dpp(o, radix2_sign(tid));
  pass_m2_fwd_store(tid, o, lds);
  AAMD_FDR_BARRIER(2);
}
// ... and back: LDS spectrum (digit-reversed) -> the block's samples in v.  The first half (inverse radix-2 + length-16 and
// length-128 passes, wave-local) is written out in the kernel; this is the second (the kernel requests the next block's samples
// between the two: during the first half a thread holds the exchanged radix-2 operands on top of its delay line)
__device__ __forceinline__ void inverse_block_b(int tid, C32* lds, const C32* tl, C32 (&v)[8]) {
  pass_m128<true>(tid, lds, tl);
  AAMD_FDR_BARRIER(4);
  last_pass_to_regs(tid, lds, tl, v);
}

__device__ __forceinline__ int64_t uniform64(int64_t v) {      // a wave-uniform 64-bit value through two scalar registers
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (int64_t)(((unsigned long long)hi << 32) | lo);
}
// H[(yrow * n_part + p) * 8192 + h_index] = scale * (twice the spectrum of the taps of partition p), thread-owned layout
__global__ void __launch_bounds__(kThreads)
spectrum_kernel(int64_t ny, int n_part, const float* __restrict__ y, const C32* __restrict__ tw16k, C32* __restrict__ H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_fdr[];
  C32* lds = reinterpret_cast<C32*>(smem_fdr);
  C32* tl = lds + kLdsData;
  const int tid = threadIdx.x;
  const int64_t yrow = blockIdx.x / n_part;
  const int p = blockIdx.x - (int)yrow * n_part;
  twiddle_tables(tid, tw16k, tl);
  MidConst mc;
  mid_init(tid, tw16k, mc);
  C32 v[8];
  load_taps(tid, ny, y + yrow * ny, p, v);
  __syncthreads();
  C32 tw3[7];
  load_tw3<7>(tid, tl, tw3);
  forward_block<7>(tid, v, lds, tl, tw3);
  C32 z[8];
  mid_split(tid, lds, mc, z);
  C32* Hp = H + (int64_t)blockIdx.x * kHPerPart;
#pragma unroll
  for (int i = 0; i < 8; ++i) Hp[h_index(0, i, tid)] = C32{z[i].x * kSpectrumScale, z[i].y * kSpectrumScale};
}

// One work item = one row segment, walked block by block; NP - 1 forward-only steps fill the delay line first.
template <int NP>
__global__ void __launch_bounds__(kThreads, 4)
delay_line_kernel(Geom g, const float* __restrict__ x, const C32* __restrict__ tw16k, const C32* __restrict__ H,
                  const int64_t* __restrict__ x_row_of, const int64_t* __restrict__ y_row_of, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_fdr[];
  C32* lds = reinterpret_cast<C32*>(smem_fdr);
  C32* tl = lds + kLdsData;
  const int tid = threadIdx.x;
  twiddle_tables(tid, tw16k, tl);
  unsigned* pair_flags = reinterpret_cast<unsigned*>(lds + kFlags);
  unsigned pair_gen = 0;                  // wave-uniform: rendezvous executed so far (the arrival counters start at 0)
  if (tid < 16) pair_flags[tid] = 0u;
  MidConst mc;
  mid_init(tid, tw16k, mc);
  __syncthreads();
  constexpr int kTw3 = Tw3Regs<NP>::n;
  C32 tw3[7];
  load_tw3<kTw3>(tid, tl, tw3);
  const float r2s = radix2_sign(tid);
#ifdef AAMD_FDR_PRIO
  // lab (tools/fdr_lab.py): static issue priority per wave -- the four waves of a SIMD (w, w + 4, w + 8, w + 12) get distinct
  // priorities so that they leave a pass one after the other instead of together
  {
#if AAMD_FDR_PRIO == 1
    const int pr = __builtin_amdgcn_readfirstlane((threadIdx.x >> 8) & 3);
#elif AAMD_FDR_PRIO == 2
    const int pr = __builtin_amdgcn_readfirstlane(3 - ((threadIdx.x >> 8) & 3));
#else
    const int pr = __builtin_amdgcn_readfirstlane((threadIdx.x >> 9) & 1);
#endif
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
#endif
  const unsigned n_items = (unsigned)(g.rows * g.segs);          // < 2^31 (checked by the launcher): 32-bit item arithmetic
#pragma unroll 1
  for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t row = (int64_t)(item / (unsigned)g.segs);
    // block numbers fit 32 bits (n_blocks <= out_len / hop; the launcher checks rows * segs < 2^31): scalar compares in the loop
    const int j_lo = (int)((int64_t)(item - (unsigned)row * (unsigned)g.segs) * g.seg_blocks);
    const int j_hi = (int)(j_lo + g.seg_blocks < g.n_blocks ? j_lo + g.seg_blocks : g.n_blocks);
    // row numbers are uniform over the workgroup: taken through scalar registers, so that every row pointer below is a scalar
    // base (global_load v, v_offset, s[base]) instead of eight 64-bit vector addresses per tap-spectrum partition
    const int64_t rx = uniform64(x_row_of ? x_row_of[row] : row);
    const int64_t ry = uniform64(y_row_of ? y_row_of[row] : row);
    const float* xr = x + rx * g.nx;
    const C32* Hr = H + ry * NP * kHPerPart;
    float* out_row = out + row * g.out_len;
    // 8-byte paths: the row's first sample / first output on an even float offset (block starts are multiples of 8192)
    const bool vin = (reinterpret_cast<uintptr_t>(xr) & 7) == 0;        // (load_block adds the parity of the block's own offset)
    const bool vout = (reinterpret_cast<uintptr_t>(out_row) & 7) == 0;
    C32 z1[8], z2[8], z3[8];               // Z_(j-1), Z_(j-2), Z_(j-3): this thread's 8 bins (as many as NP - 1 are live)
#pragma unroll
    for (int i = 0; i < 8; ++i) z1[i] = z2[i] = z3[i] = C32{0.0f, 0.0f};
    C32 v[8];
    load_block(tid, g, xr, j_lo - (NP - 1), vin, v);
#pragma unroll 1
    for (int j = j_lo - (NP - 1); j < j_hi; ++j) {
      const bool produce = j >= j_lo;
      C32 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = C32{0.0f, 0.0f};
      C32 h0[8];                           // tap spectra of this thread's bins, one partition at a time
      auto h_load = [&](int p) {           // partition p: 8 coalesced 8-byte loads from L2 (the thread-owned layout)
        // ONE scalar base (the row's table) + a 32-bit vector byte offset per load: global_load_dwordx2 v, v_off, s[base].
        // Left to itself the compiler forms eight 64-bit vector addresses per partition (16 registers and 16 carry chains)
        // or, with per-load scalar bases, hoists 24 base pairs out of the step loop and spills scalar registers.
#if AAMD_FDR_H16
        const unsigned off = (unsigned)fco::opaque(tid) * 16u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned o = off + (unsigned)((p * 4 + i) * kThreads * 16);
          asm volatile("" : "+v"(o));
          const F4 q = *reinterpret_cast<const F4*>(reinterpret_cast<const char*>(Hr) + o);
          h0[2 * i] = C32{q.x, q.y};
          h0[2 * i + 1] = C32{q.z, q.w};
        }
#else
        const unsigned off = (unsigned)fco::opaque(tid) * (unsigned)sizeof(C32);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          unsigned o = off + (unsigned)((p * 8 + i) * kThreads * (int)sizeof(C32));
          asm volatile("" : "+v"(o));
          h0[i] = *reinterpret_cast<const C32*>(reinterpret_cast<const char*>(Hr) + o);
        }
#endif
#ifdef AAMD_FDR_LAB_NOH
#pragma unroll
        for (int i = 0; i < 8; ++i) h0[i] = C32{1.0f + (float)i, 0.5f};
#endif
      };
      // the delayed partitions do not wait for this block's spectrum: H_1 Z_(j-1) + H_2 Z_(j-2) is formed BEFORE the forward
      // transform and only H_0 Z_j is left for the middle step
      if (produce) {
        if (NP > 1) { h_load(1); mid_mac(tid, h0, z1, acc); }
        if (NP > 2) { h_load(2); mid_mac(tid, h0, z2, acc); }
        if (NP > 3) { h_load(3); mid_mac(tid, h0, z3, acc); }
      }
      // (Round 5 tried to hide the tap-spectrum loads behind the passes -- H_1 in flight during the first pass, H_2 during the
      // length-1024 pass, H_0 during the wave-local ones, unconditional so that the compiler's vmcnt stays exact: 0.710-0.724 ms
      // against 0.700 for this order on the cfg5b shard, although the kernel WITHOUT any tap-spectrum read runs 0.61: the
      // 192 KB per step are a throughput cost of the vector-memory path, not an exposed latency.  profiles/r05_c_fdr_lab_barriers_hloads.txt)
      first_pass_from_regs(tid, v, lds, tl);
      AAMD_FDR_BARRIER(4);
      pass_m128<false>(tid, lds, tl);
      AAMD_FDR_PAIR(1);
      if (kTw3 > 0) pass_m16_regtw<false, kTw3>(tid, lds, tl, tw3); else pass_m16<false>(tid, lds, tl);
      fco::wave_sync();
      {
        C32 o[8];
        pass_m2_fwd_a(tid, lds, tl, o);
        radix2_dpp(o, r2s);
        pass_m2_fwd_store(tid, o, lds);
      }
      AAMD_FDR_BARRIER(2);
      if (produce) h_load(0);              // H_0: requested before the split reads LDS (an L2 round trip)
      C32 z0[8];
      mid_split(tid, lds, mc, z0);
      if (produce) {
        mid_mac(tid, h0, z0, acc);
        mid_merge(tid, acc, mc, lds);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (NP > 3) z3[i] = z2[i];
        z2[i] = z1[i]; z1[i] = z0[i];
      }
      AAMD_FDR_BARRIER(2);
      // the next block's samples: in flight during the inverse passes and the stores (requested only now: during the middle
      // step the thread holds three spectra, the accumulators and a partition of tap spectra -- with these 16 registers on top
      // the 128-register budget of four waves per SIMD spilled)
      if (produce) {                       // inverse_block_a with the pair rendezvous in place of its closing barrier
        C32 xq[8];
        pass_m2_inv_a(tid, lds, xq);
        radix2_dpp(xq, -r2s);
        pass_m2_inv_finish(tid, xq, lds, tl);
        fco::wave_sync();
        if (kTw3 > 0) pass_m16_regtw<true, kTw3>(tid, lds, tl, tw3); else pass_m16<true>(tid, lds, tl);
        AAMD_FDR_PAIR(1);
      }
      if (j + 1 < j_hi) load_block(tid, g, xr, j + 1, vin, v);
      if (produce) {
        C32 w[8];
        inverse_block_b(tid, lds, tl, w);
        store_block(tid, g, w, j, j_hi, vout, out_row);
        // No barrier here (round 5; rounds 3-4 had one: "the next first pass overwrites what the last pass has just read"):
        // the last pass READS the cells pad(tid) + 1088 r and the next first pass WRITES exactly those cells from the same
        // thread -- program order of one thread is all the hazard needs, and every other access to them lies behind the
        // barrier that follows the first pass.  AAMD_FDR_WAR_BARRIER=1 restores it for A/B runs.
#if AAMD_FDR_WAR_BARRIER
        AAMD_FDR_BARRIER(4);
#endif
      }
    }
  }
}
// (Round 4 also built the PIPELINED walk -- two spectrum buffers, the inverse of block j between the same barriers as the forward
// of block j + 1, 4 barriers per block instead of 7, pass-1 twiddles formed from a two-level table because the 57 KB table no
// longer fits -- correct (3e-7 of the peak) and 10 % SLOWER on the cfg5b shard, 0.84-0.86 against 0.76-0.78 ms on the same box:
// profiles/r04_g_fdr_pipelined_walk_ab.txt, code at commit b5384a0.  Two passes per barrier interval do not overlap better
// than one, and the formed twiddles cost what the saved barriers gave.  Removed.)
#endif  // __HIPCC__

}  // namespace fdr
}  // namespace aamd

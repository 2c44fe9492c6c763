// F.fftconvolve for 193 .. 32768 taps (4 ms .. 0.68 s impulse responses at 48 kHz; BASELINE config 5b): overlap-save and, beyond
// 8192 taps, the frequency-domain delay line on REAL blocks, the whole state of a row on one CU (functional/functional.py:2252-2258 computes
// irfft(rfft(x) * rfft(y)); the contract is the linear convolution, so block-wise FFTs of another length are free).
//
// Formulation (MI355X: 160 KB LDS, 512 KB of registers per CU, 4 waves per SIMD at <= 128 registers):
//   * one step = ONE real block of N = 16384 input samples (hop B = 8192) as an M = 8192-point complex FFT of
//     z[n] = s[2n] + i s[2n+1]; the spectrum of the real block follows from the split  Z[k] = E + T, Z[M-k] = conj(E - T),
//     E = (C[k] + conj C[M-k]) / 2, T = -i W_N^k (C[k] - conj C[M-k]) / 2 (and back by the merge);
//   * 1024 threads; a thread owns 8 spectrum bins for the whole launch, so the delay line Z_(j-1), Z_(j-2) is 2 x 16 REGISTERS
//     per thread: no ring in memory, no ring in LDS;  Y_j = H_0 Z_j + H_1 Z_(j-1) + H_2 Z_(j-2), tap spectra H_p (8192-tap
//     partitions) read from L2 in the thread-owned layout (16-byte loads).
//
// Round 6: radices (8, 32, 32) -- THREE trips through LDS per transform instead of five (8, 8, 8, 8, 2).  Rounds 3-5 measured the
// step as VALU issue 5.3 us + LDS pipe 4.4 us + barriers, ADDITIVE, and 64 % of the LDS time as ds_write_b64 (6 cycles per wave
// instruction, MI355X_MICROARCH.md): the only lever left on the LDS side was fewer trips.
//   n = 1024 n1 + 32 n2 + n3,  k = k1 + 8 k2 + 256 k3   (n1, k1 < 8;  n2, k2, n3, k3 < 32)
//   pass A: DFT-8 over n1 in registers, straight from the coalesced global loads (n = t + 1024 r), x W_8192^(k1 t);
//   pass B: DFT-32 over n2, pass C: DFT-32 over n3.  A DFT-32 = a DFT-8 in the registers of each of the FOUR lanes of a quad
//     (index 4 r + q, q = lane & 3), a lane twiddle W_32^(q kappa1), and a DFT-4 ACROSS the quad on DPP quad_perm exchanges
//     (two radix-2 levels, v_mov_dpp + fma each; lane q ends up with kappa2 = bitrev(q), index kappa1 + 8 kappa2);
//   * pass C hands its outputs to the middle step IN REGISTERS.  A quad holds the 32 bins k0 + 256 k3 of one k0 = k1 + 8 k2; their
//     mirror bins M - k are exactly the 32 bins of quad 256 - k0 with k3 -> 31 - k3, i.e. lane (q) <-> lane (3 - q), register
//     kappa1 <-> 7 - kappa1.  Pass C places quad k0 and quad 256 - k0 in the two halves of one OCTET of lanes: the mirror partner
//     of a lane is DPP row_half_mirror away, and split, delay line, merge exchange their 4 + 4 complex numbers per lane without
//     touching LDS.  (Quads 0 and 128 mirror into themselves: the one octet they share goes through 64 cells of LDS scratch.)
//   * the spectrum sits in LDS as cell(k1, a, b) = 1024 k1 + 32 a + (b ^ psi(a)) -- (a, b) = (n2, n3) -> (k2, n3) -> back -- with
//     psi(a) = a2 << 4 | (a1 ^ a4) << 3 | (a0 ^ a4 ^ a3) << 2: every 8-byte access of every pass is bank-conflict free without
//     padding (16-lane write groups over 32 banks, 32-lane read groups over 64; the two k1 = 0 waves of pass C: two-way), and
//     a register's address is one XOR away from the thread's base (tools/lab/fdr_lds_bank_model.py);
//   * barriers per step: A | B | C ... C' | B' | A' = 4 (rounds 4-5: 4 + 2 pair rendezvous + wave fences); LDS instructions per
//     thread and step 32 stores + ~62 loads (round 5: 64 + ~102).
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
constexpr int kMaxParts = 4;               // Z_j in flight + up to three delayed spectra in registers (24 577 .. 32 768 taps too)
constexpr int kLdsData = kM;               // 8192 cells, no padding (the columns are XOR-swizzled)
// twiddle tables behind the data (W_M = e^(-2 pi i / 8192)), [register][thread] so that the lanes of a wave read consecutive entries:
//   kTwA + 1024 (k1 - 1) + t:  W_8192^(k1 t), t < 1024 (pass A / A');
//   kTwB + 128 kappa1 + p:     sigma W_1024^(b k2) of pass B / B', p = thread within the wave pair, b = b_of_thread(p), k2 = kappa1 +
//                              8 bitrev(p & 3), sigma = the sign the quad's DFT-4 leaves in lane p & 3 (xl_level)
constexpr int kTwA = 0, kTwB = kTwA + 7 * 1024, kTwEnd = kTwB + 8 * 128;
constexpr int kScratch = kLdsData + kTwEnd;                       // 64 cells: the self-mirrored octet's hand-over
constexpr int kLaneTw = kScratch + 64;                            // [k - 1][q]: W_32^(q k), the lane twiddles of a DFT-32 (28 cells)
constexpr int kLdsComplex = kLaneTw + 28;                         // 16 476 complex = 131 808 B

#if defined(__HIP_DEVICE_COMPILE__)
#define AAMD_FDR_FENCE asm volatile("" ::: "memory");     /* nothing moves (or merges) across it; costs no instruction */
#else
#define AAMD_FDR_FENCE
#endif
// the sign the quad's DFT-4 leaves in a lane (see xl_level)
AAMD_HD float sigma_of(int tid) { return ((tid & 3) == 1 || (tid & 3) == 2) ? -1.0f : 1.0f; }
// what a thread's z of the middle step is multiplied by to give the true spectrum (the self-mirrored octet works on true values)
AAMD_HD float spectrum_sign(int tid) { return tid < 8 ? 1.0f : sigma_of(tid); }
AAMD_HD int brev2(int q) { return ((q & 1) << 1) | (q >> 1); }
// column swizzle of row a (see the header): bits 4, 3, 2 of the column
AAMD_HD int psi(int a) { return (((a >> 2) & 1) << 4) | ((((a >> 1) ^ (a >> 4)) & 1) << 3) | (((a ^ (a >> 4) ^ (a >> 3)) & 1) << 2); }
AAMD_HD int cell(int k1, int a, int b) { return (k1 << 10) + (a << 5) + (b ^ psi(a)); }
// pass B / B': the wave pair (w >> 1) = k1; its 32 quads are the 32 values of b.  Lane bits -> b so that a 16-lane group's quads
// differ in b1 b0 and a 32-lane group's additionally in b4 (what the swizzle's conflict analysis asks for)
AAMD_HD int b_of_thread(int tid) {
  const int x = (tid & 63) >> 2, om = (tid >> 6) & 1;
  return (x & 1) | (((x >> 1) & 1) << 1) | (((x >> 3) & 1) << 2) | (om << 3) | (((x >> 2) & 1) << 4);
}
// pass C / C': octet o = 8 w + (lane >> 3) holds quad k0 (lanes 0 .. 3) and its mirror quad 256 - k0 (lanes 4 .. 7); octet 0 holds
// the two self-mirrored quads 0 and 128.  k0 = k1 + 8 k2 with k1 = w & 7 and k2 = (j1, j0, j2, w3) from the octet's number j in the
// wave -- neighbouring octets differ in k2's low bits, which the swizzle turns into distinct banks
AAMD_HD int k0_of_thread(int tid) {
  const int w = tid >> 6, l = tid & 63, j = l >> 3, h = (l >> 2) & 1;
  const int k2 = ((j >> 1) & 1) | ((j & 1) << 1) | (((j >> 2) & 1) << 2) | ((w >> 3) << 3);
  const int k0 = (w & 7) + 8 * k2;
  if (!h) return k0;
  return tid < 8 ? 128 : 256 - k0;
}

// tw16k: the W_16384^m table of fco::twiddle_kernel (m < 16384); W_M^e = tw16k[2 e]
AAMD_HD void twiddle_tables(int tid, const C32* tw16k, C32* tl) {
  for (int u = tid; u < kTwEnd; u += kThreads) {
    int e;                                                        // exponent of W_M
    if (u < kTwB) { const int k = u / 1024 + 1, t = u % 1024; e = t * k; }
    else {
      const int v = u - kTwB, kap1 = v / 128, p = v % 128;
      const int k2 = kap1 + 8 * brev2(p & 3);
      e = 8 * ((b_of_thread(p) * k2) & 1023);
      if (sigma_of(p) < 0.0f) e = (e + kM / 2) & (kM - 1);        // sigma W = W_M^(e + M / 2): the sign the quad's DFT-4 leaves (xl_level)
    }
    tl[u] = tw16k[2 * e];                                         // 2 e < 16384 for every entry
  }
}
// the lane twiddles of a DFT-32: L[k - 1] = W_32^(q k), q = lane & 3, k = 1 .. 7 -- the same for passes B, C and (conjugated) B',
// C'.  14 registers that the middle step (two delayed spectra, accumulators, tap spectra, this block's spectrum) does not have:
// they live in a 28-cell LDS table and are re-read TWICE per step, in front of the forward passes B, C and in front of the inverse
// ones (7 broadcast reads each), so that nothing of them is live across the middle step.
AAMD_HD void lane_twiddle_table(int tid, const C32* tw16k, C32* tab) {
  if (tid < 28) tab[tid] = tw16k[512 * (tid & 3) * ((tid >> 2) + 1)];
}
AAMD_HD void lane_twiddles(int tid, const C32* tab, C32 (&L)[7]) {
  tid = fco::opaque(tid);
#pragma unroll
  for (int k = 0; k < 7; ++k) { L[k] = tab[4 * k + (tid & 3)]; AAMD_FDR_FENCE }
}

// ---- LDS access of a thread's 8 elements ---------------------------------------------------------------------------------
// hipcc pairs neighbouring 8-byte LDS accesses into ds_read2_b64 / ds_write2_b64, and on this part a ds_read2_b64 costs the
// LDS 8 cycles per wave-instruction against 2 for a ds_read_b64 (MI355X_MICROARCH.md, LDS table): the accesses are written out
// as single ds_read_b64 / ds_write_b64 with an empty asm statement (memory clobber) between them, which keeps the load / store
// optimiser from pairing them.  Offsets in complex elements.
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
// element r of a thread at BYTE offset (base ^ X(r)) + D(r): one v_xor_b32 per access, D(r) in the instruction's offset field
AAMD_HD const C32* at_bytes(const C32* lds, unsigned off) { return reinterpret_cast<const C32*>(reinterpret_cast<const char*>(lds) + off); }
AAMD_HD C32* at_bytes(C32* lds, unsigned off) { return reinterpret_cast<C32*>(reinterpret_cast<char*>(lds) + off); }
// a = 4 r + q ("low": the lanes are the LOW bits of the row): psi(a) = xlow(r) ^ (q1 << 3 | q0 << 2)
constexpr unsigned xlow(int r) { return (unsigned)(((r & 1) << 4) | (((r >> 2) & 1) << 3) | ((((r >> 2) ^ (r >> 1)) & 1) << 2)); }
// a = kappa + 8 kappa2 ("high": the lanes are the HIGH bits of the row, a4 a3 = kappa2): psi(a) = xhigh(kappa) ^ (a4 << 3 | (a4 ^ a3) << 2)
constexpr unsigned xhigh(int k) { return (unsigned)((((k >> 2) & 1) << 4) | (((k >> 1) & 1) << 3) | ((k & 1) << 2)); }
template <typename F>
AAMD_HD void read8_at(const C32* lds, unsigned base, F off, C32 (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) { v[r] = *at_bytes(lds, off(base, r)); AAMD_FDR_FENCE }
}
template <typename F>
AAMD_HD void write8_at(C32* lds, unsigned base, F off, const C32 (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) { *at_bytes(lds, off(base, r)) = v[r]; AAMD_FDR_FENCE }
}
struct OffLow { AAMD_HD unsigned operator()(unsigned base, int r) const { return (base ^ (xlow(r) << 3)) + ((unsigned)r << 10); } };     // rows 4 r + q: 128 r cells
struct OffHigh { AAMD_HD unsigned operator()(unsigned base, int k) const { return (base ^ (xhigh(k) << 3)) + ((unsigned)k << 8); } };   // rows kappa + 8 kappa2: 32 kappa cells
struct OffCol { AAMD_HD unsigned operator()(unsigned base, int r) const { return base ^ ((unsigned)r << 5); } };                        // columns 4 r + q of one row
AAMD_HD unsigned base_low(int tid) {        // pass B reads / pass B' writes: cell(k1, 4 r + q, b)
  const int k1 = tid >> 7, q = tid & 3, b = b_of_thread(tid);
  return (unsigned)((k1 << 10) + (q << 5) + (b ^ (((q >> 1) << 3) | ((q & 1) << 2)))) << 3;
}
AAMD_HD unsigned base_high(int tid) {       // pass B writes / pass B' reads: cell(k1, kappa + 8 kappa2, b), kappa2 = bitrev(q)
  const int k1 = tid >> 7, kap2 = brev2(tid & 3), b = b_of_thread(tid), a3 = kap2 & 1, a4 = kap2 >> 1;
  return (unsigned)((k1 << 10) + (kap2 << 8) + (b ^ ((a4 << 3) | ((a4 ^ a3) << 2)))) << 3;
}
AAMD_HD unsigned base_col(int tid) {        // pass C reads / pass C' writes: cell(k1, k2, 4 r + q) of the thread's quad
  const int k0 = k0_of_thread(tid), k1 = k0 & 7, k2 = k0 >> 3, q = tid & 3;
  return (unsigned)((k1 << 10) + (k2 << 5) + (q ^ psi(k2))) << 3;
}

// v[k] *= tab[stride (k - 1)] (or its conjugate), k = 1 .. 7
template <bool conj_w, int STRIDE>
AAMD_HD void mul_table8(C32 (&v)[8], const C32* tab) {
  C32 w[8];
  lds_read8<0, STRIDE, 2 * STRIDE, 3 * STRIDE, 4 * STRIDE, 5 * STRIDE, 6 * STRIDE, 6 * STRIDE>(tab, w);    // (7 entries; the 8th re-reads the 7th)
#pragma unroll
  for (int k = 1; k < 8; ++k) v[k] = fco::cmulc<conj_w>(v[k], w[k - 1]);
}
// v[k] *= tab[stride k] (or its conjugate), k = 0 .. 7
template <bool conj_w, int STRIDE>
AAMD_HD void mul_table8_all(C32 (&v)[8], const C32* tab) {
  C32 w[8];
  lds_read8<AAMD_FDR_STRIDE(STRIDE)>(tab, w);
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = fco::cmulc<conj_w>(v[k], w[k]);
}
template <bool conj_w>
AAMD_HD void mul_lane8(C32 (&v)[8], const C32 (&L)[7]) {
#pragma unroll
  for (int k = 1; k < 8; ++k) v[k] = fco::cmulc<conj_w>(v[k], L[k - 1]);
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

// ---- the DFT-4 across a quad of lanes -------------------------------------------------------------------------------------
// One radix-2 level between a lane and its partner is ONE instruction per float on the device: v_fmac_f32_dpp own, dpp(own), s =
// own + s nb (the partner's value arrives through the DPP operand, s = +-1 is a lane constant).  A lane that should hold
// "partner - own" holds own - partner instead: the NEGATED value.  The signs are carried, not repaired:
//   forward (lane q in: x_q):  level lane ^ 2 with s = (+, +, -, -), lane 3 times -i, level lane ^ 1 with s = (+, -, +, -)
//            -> lane q holds sigma_q y_(bitrev q), sigma = (+, -, -, +);
//   inverse (lane q in: sigma_q y_(bitrev q)):  level lane ^ 1 with s = (-, +, -, +), lane 3 times +i, level lane ^ 2 with
//            s = (-, -, +, +) -> lane q holds x_q, all signs true.
// sigma is the same for a lane and its mirror partner (q <-> 3 - q), every step between a forward and an inverse DFT-4 is linear
// (twiddles, split, multiplication by the TRUE tap spectra, merge), so the values simply travel as sigma times themselves: pass B's
// twiddle table carries sigma (LDS holds true values), the spectrum kernel stores sigma z (= the true tap spectra), and the
// self-mirrored octet -- which mixes lanes -- multiplies by sigma on its way through the scratch cells.
// (v_mov_b32_dpp + v_fma_f32 + the v_mov that initialises the DPP destination: 3 instructions per float and level, 19 % slower than
// the five-pass kernel on the cfg5b shard -- the step is VALU-issue bound; profiles/r06_c_fdr_lab_radix32_first.txt)
AAMD_HD void xl_level(const C32 (&own)[8], const C32 (&nb)[8], float s, C32 (&out)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) out[r] = C32{own[r].x + s * nb[r].x, own[r].y + s * nb[r].y};
}

// the one twiddle inside a DFT-4 that is not +-1: lane 3 multiplies by -i (forward) / +i (inverse) between the two levels
template <bool inv>
AAMD_HD void xl_rot(C32 (&u)[8], bool rot) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const C32 a = u[r];
    u[r].x = rot ? (inv ? -a.y : a.y) : a.x;
    u[r].y = rot ? (inv ? a.x : -a.x) : a.y;
  }
}
// forward: level (lane ^ 2, s_hi), rotation in lane 3, level (lane ^ 1, s_lo); inverse: level (lane ^ 1, -s_lo), rotation, level
// (lane ^ 2, -s_hi)
AAMD_HD float sign_hi(int tid) { return (tid & 2) ? -1.0f : 1.0f; }
AAMD_HD float sign_lo(int tid) { return (tid & 1) ? -1.0f : 1.0f; }

// ---- pass A (length 8192, stride 1024): thread tid owns elements tid + 1024 r -- what a coalesced load of the block gives it ----
AAMD_HD void passA_fwd(int tid, C32 (&v)[8], C32* lds, const C32* tl) {
  dft8<false>(v);
  tid = fco::opaque(tid);
  mul_table8<false, 1024>(v, tl + kTwA + tid);
  lds_write8<AAMD_FDR_STRIDE(1024)>(lds + cell(0, tid >> 5, tid & 31), v);
}
AAMD_HD void passA_inv(int tid, const C32* lds, const C32* tl, C32 (&v)[8]) {
  tid = fco::opaque(tid);
  lds_read8<AAMD_FDR_STRIDE(1024)>(lds + cell(0, tid >> 5, tid & 31), v);
  mul_table8<true, 1024>(v, tl + kTwA + tid);
  dft8<true>(v);
}
// ---- passes B and C, forward: part a = loads + DFT-8 + lane twiddles (then the quad's DFT-4), part b of pass B = twiddles + stores ----
AAMD_HD void passB_fwd_a(int tid, const C32* lds, const C32 (&L)[7], C32 (&v)[8]) {
  tid = fco::opaque(tid);
  read8_at(lds, base_low(tid), OffLow(), v);
  dft8<false>(v);
  mul_lane8<false>(v, L);
}
AAMD_HD void passB_fwd_b(int tid, C32 (&v)[8], C32* lds, const C32* tl) {
  tid = fco::opaque(tid);
  mul_table8_all<false, 128>(v, tl + kTwB + (tid & 127));
  write8_at(lds, base_high(tid), OffHigh(), v);
}
AAMD_HD void passC_fwd_a(int tid, const C32* lds, const C32 (&L)[7], C32 (&v)[8]) {
  tid = fco::opaque(tid);
  read8_at(lds, base_col(tid), OffCol(), v);
  dft8<false>(v);
  mul_lane8<false>(v, L);
}
// ---- ... and inverse: (the quad's inverse DFT-4, then) lane twiddles + DFT-8 + stores ----
AAMD_HD void passC_inv_b(int tid, C32 (&v)[8], const C32 (&L)[7], C32* lds) {
  tid = fco::opaque(tid);
  mul_lane8<true>(v, L);
  dft8<true>(v);
  write8_at(lds, base_col(tid), OffCol(), v);
}
AAMD_HD void passB_inv_a(int tid, const C32* lds, const C32* tl, C32 (&v)[8]) {
  tid = fco::opaque(tid);
  read8_at(lds, base_high(tid), OffHigh(), v);
  mul_table8_all<true, 128>(v, tl + kTwB + (tid & 127));
}
AAMD_HD void passB_inv_b(int tid, C32 (&v)[8], const C32 (&L)[7], C32* lds) {
  tid = fco::opaque(tid);
  mul_lane8<true>(v, L);
  dft8<true>(v);
  write8_at(lds, base_low(tid), OffLow(), v);
}

// ---- the middle step: split, delay line, merge ------------------------------------------------------------------------
// After pass C lane (quad k0, q) holds c[kap] = C[k0 + 256 (kap + 8 kappa2)], kappa2 = bitrev(q); its mirror partner (the lane
// row_half_mirror away: quad 256 - k0, lane 3 - q) holds C[M - that bin] in register 7 - kap.  A lane OWNS the four pairs
// (kA_i, M - kA_i), kA_i = k0 + 2048 kappa2 + 256 i, i = 0 .. 3: cA[i] = its own c[i], cM[i] = the partner's c[7 - i]; the
// partner owns the pairs of ITS registers 0 .. 3 -- this lane's registers 7 .. 4.  8 bins per lane: z[2 i] = Z[kA_i],
// z[2 i + 1] = Z[M - kA_i] (twice the real block's spectrum).
// The octet of lanes 0 .. 7 holds the self-mirrored quads 0 (lanes 0 .. 3) and 128 (lanes 4 .. 7): their 2 x 32 values go through
// LDS scratch and lane (h, lam) owns the pairs j = 4 i + lam of its own quad:
//   quad 128: bins 128 + 256 j and M - (128 + 256 j) = 128 + 256 (31 - j);
//   quad 0:   bins 256 j and M - 256 j = 256 (32 - j), j >= 1;  j = 0 (thread 0, i = 0): z[0] = (Z[0], Z[8192]) (both real),
//             z[1] = Z[4096].
struct MidConst {
  C32 w0;                // W_N^(kA_0)
  C32 step;              // W_N^(kA_(i+1) - kA_i): W_N^256 (W_N^1024 in the self-mirrored octet); w_i formed per step, 3 products
};
AAMD_HD void mid_init(int tid, const C32* tw16k, MidConst& mc) {
  if (tid < 8) {
    mc.w0 = tw16k[128 * (tid >> 2) + 256 * (tid & 3)];
    mc.step = tw16k[1024];
  } else {
    mc.w0 = tw16k[k0_of_thread(tid) + 2048 * brev2(tid & 3)];
    mc.step = tw16k[256];
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
AAMD_HD void mid_split(int tid, const C32 (&cA)[4], const C32 (&cM)[4], const MidConst& mc, C32 (&z)[8]) {
  C32 w = mc.w0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (tid == 0 && i == 0) {
      z[0] = C32{2.0f * (cA[0].x + cA[0].y), 2.0f * (cA[0].x - cA[0].y)};
      z[1] = C32{2.0f * cM[0].x, -2.0f * cM[0].y};
    } else {
      split_pair(cA[i], cM[i], w, z[2 * i], z[2 * i + 1]);
    }
    if (i < 3) w = cmul(w, mc.step);
  }
}
AAMD_HD void mid_merge(int tid, const C32 (&y)[8], const MidConst& mc, C32 (&ck)[4], C32 (&cm)[4]) {
  C32 w = mc.w0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (tid == 0 && i == 0) {
      ck[0] = C32{y[0].x + y[0].y, y[0].x - y[0].y};
      cm[0] = C32{2.0f * y[1].x, -2.0f * y[1].y};
    } else {
      merge_pair(y[2 * i], y[2 * i + 1], w, ck[i], cm[i]);
    }
    if (i < 3) w = cmul(w, mc.step);
  }
}
// the self-mirrored octet (threads 0 .. 7): registers <-> scratch, scratch <-> owned pairs
// (opaque thread numbers: hoisted out of the step loop, the 24 cell addresses of these four functions were spilled to scratch
// memory, and a scratch reload waits for every vector-memory load in flight)
AAMD_HD int scratch_cell(int tid, int kap) { return 32 * (tid >> 2) + kap + 8 * brev2(tid & 3); }
AAMD_HD void special_put(int tid, const C32 (&c)[8], C32* scratch) {       // (the cells hold TRUE values: sigma c)
  tid = fco::opaque(tid);
  const float sg = sigma_of(tid);
#pragma unroll
  for (int k = 0; k < 8; ++k) scratch[scratch_cell(tid, k)] = C32{sg * c[k].x, sg * c[k].y};
}
AAMD_HD void special_get(int tid, const C32* scratch, C32 (&c)[8]) {
  tid = fco::opaque(tid);
  const float sg = sigma_of(tid);
#pragma unroll
  for (int k = 0; k < 8; ++k) { const C32 v = scratch[scratch_cell(tid, k)]; c[k] = C32{sg * v.x, sg * v.y}; }
}
AAMD_HD void special_pair_cells(int tid, int i, int& ia, int& im) {
  const int h = tid >> 2, j = 4 * i + (tid & 3);
  if (h) { ia = 32 + j; im = 32 + 31 - j; }
  else if (j == 0) { ia = 0; im = 16; }
  else { ia = j; im = 32 - j; }
}
AAMD_HD void special_get_pairs(int tid, const C32* scratch, C32 (&cA)[4], C32 (&cM)[4]) {
  tid = fco::opaque(tid);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ia, im;
    special_pair_cells(tid, i, ia, im);
    cA[i] = scratch[ia]; cM[i] = scratch[im];
  }
}
AAMD_HD void special_put_pairs(int tid, const C32 (&ck)[4], const C32 (&cm)[4], C32* scratch) {
  tid = fco::opaque(tid);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ia, im;
    special_pair_cells(tid, i, ia, im);
    scratch[ia] = ck[i]; scratch[im] = cm[i];
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
// lab switches (tools/fdr_lab.py; timing only, the results are wrong): AAMD_FDR_LAB_NOBAR bit 0 turns the barriers around
// pass B / B' (behind pass A, in front of pass A') into wave-local fences, bit 1 those between passes B and C / C' and B' --
// an upper bound on what any restructuring of the barriers could win; AAMD_FDR_LAB_NOH reads no tap spectra
#ifndef AAMD_FDR_LAB_NOBAR
#define AAMD_FDR_LAB_NOBAR 0
#endif
#define AAMD_FDR_BARRIER(BIT) do { if (AAMD_FDR_LAB_NOBAR & (BIT)) fco::wave_sync(); else __syncthreads(); } while (0)
// lane exchanges of 8 complex registers on DPP: quad_perm [2, 3, 0, 1] (0x4E: lane ^ 2), [1, 0, 3, 2] (0xB1: lane ^ 1);
// row_half_mirror (0x141: lane 7 - l of each group of 8) for the mirror partner of the middle step
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {      // (bound_ctrl: no instruction to initialise the destination; every lane is active)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// one level on all 16 floats of a thread, in place: u += s * dpp(u).  (One asm block: the s_nop in front covers the two wait
// states a DPP read needs behind a VALU write of the same register; inside the block every instruction touches its own register.)
#define AAMD_FDR_FMAC16(QP)                                                                                                          \
  asm("s_nop 1\n\t"                                                                                                                  \
      "v_fmac_f32_dpp %0, %0, %16 " QP " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, %1, %16 " QP " row_mask:0xf bank_mask:0xf\n\t"   \
      "v_fmac_f32_dpp %2, %2, %16 " QP " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, %3, %16 " QP " row_mask:0xf bank_mask:0xf\n\t"   \
      "v_fmac_f32_dpp %4, %4, %16 " QP " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %5, %5, %16 " QP " row_mask:0xf bank_mask:0xf\n\t"   \
      "v_fmac_f32_dpp %6, %6, %16 " QP " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %7, %7, %16 " QP " row_mask:0xf bank_mask:0xf\n\t"   \
      "v_fmac_f32_dpp %8, %8, %16 " QP " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %9, %9, %16 " QP " row_mask:0xf bank_mask:0xf\n\t"   \
      "v_fmac_f32_dpp %10, %10, %16 " QP " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %11, %11, %16 " QP " row_mask:0xf bank_mask:0xf\n\t" \
      "v_fmac_f32_dpp %12, %12, %16 " QP " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %13, %13, %16 " QP " row_mask:0xf bank_mask:0xf\n\t" \
      "v_fmac_f32_dpp %14, %14, %16 " QP " row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %15, %15, %16 " QP " row_mask:0xf bank_mask:0xf"      \
      : "+v"(u[0].x), "+v"(u[0].y), "+v"(u[1].x), "+v"(u[1].y), "+v"(u[2].x), "+v"(u[2].y), "+v"(u[3].x), "+v"(u[3].y),              \
        "+v"(u[4].x), "+v"(u[4].y), "+v"(u[5].x), "+v"(u[5].y), "+v"(u[6].x), "+v"(u[6].y), "+v"(u[7].x), "+v"(u[7].y)               \
      : "v"(s))
__device__ __forceinline__ void level_xor2(C32 (&u)[8], float s) { AAMD_FDR_FMAC16("quad_perm:[2,3,0,1]"); }
__device__ __forceinline__ void level_xor1(C32 (&u)[8], float s) { AAMD_FDR_FMAC16("quad_perm:[1,0,3,2]"); }
__device__ __forceinline__ void quad_dft4_fwd(C32 (&u)[8], float s_hi, float s_lo, bool rot) {
  level_xor2(u, s_hi);
  xl_rot<false>(u, rot);
  level_xor1(u, s_lo);
}
__device__ __forceinline__ void quad_dft4_inv(C32 (&u)[8], float s_hi, float s_lo, bool rot) {
  level_xor1(u, -s_lo);
  xl_rot<true>(u, rot);
  level_xor2(u, -s_hi);
}
// the partner's registers 7 .. 4 (as cM[0 .. 3]) / the partner's cm[0 .. 3] into registers 7 .. 4
__device__ __forceinline__ void mirror4(const C32& a7, const C32& a6, const C32& a5, const C32& a4, C32 (&m)[4]) {
  m[0] = C32{dpp_f<0x141>(a7.x), dpp_f<0x141>(a7.y)};
  m[1] = C32{dpp_f<0x141>(a6.x), dpp_f<0x141>(a6.y)};
  m[2] = C32{dpp_f<0x141>(a5.x), dpp_f<0x141>(a5.y)};
  m[3] = C32{dpp_f<0x141>(a4.x), dpp_f<0x141>(a4.y)};
}
struct LaneConst {
  float s_hi, s_lo;      // signs of the two radix-2 levels of the quad's DFT-4
  bool rot;              // lane 3 of the quad
};
__device__ __forceinline__ void lane_init(int tid, LaneConst& lc) {
  lc.s_hi = sign_hi(tid); lc.s_lo = sign_lo(tid); lc.rot = (tid & 3) == 3;
}
// forward transform of the block in v (registers): passes A, B in LDS, pass C into registers c (the digit-reversed spectrum of
// the thread's quad); two workgroup barriers
__device__ __forceinline__ void forward_block(int tid, C32 (&v)[8], C32* lds, const C32* tl, const LaneConst& lc, C32 (&c)[8]) {
  passA_fwd(tid, v, lds, tl);
  AAMD_FDR_BARRIER(1);
  C32 L[7];
  lane_twiddles(tid, lds + kLaneTw, L);
  passB_fwd_a(tid, lds, L, c);
  quad_dft4_fwd(c, lc.s_hi, lc.s_lo, lc.rot);
  passB_fwd_b(tid, c, lds, tl);
  AAMD_FDR_BARRIER(2);
  passC_fwd_a(tid, lds, L, c);
  quad_dft4_fwd(c, lc.s_hi, lc.s_lo, lc.rot);
}
// the owned pairs of the thread from its pass-C registers (and the partner's)
__device__ __forceinline__ void gather_pairs(int tid, const C32 (&c)[8], C32* lds, C32 (&cA)[4], C32 (&cM)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) cA[i] = c[i];
  mirror4(c[7], c[6], c[5], c[4], cM);
  if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {         // wave 0 holds the self-mirrored octet
    if (tid < 8) special_put(tid, c, lds + kScratch);
    fco::wave_sync();
    if (tid < 8) special_get_pairs(tid, lds + kScratch, cA, cM);
  }
}
__device__ __forceinline__ void scatter_pairs(int tid, const C32 (&ck)[4], const C32 (&cm)[4], C32* lds, C32 (&c)[8]) {
  C32 m[4];
  mirror4(cm[0], cm[1], cm[2], cm[3], m);                     // the partner's cm[i] -> register 7 - i
#pragma unroll
  for (int i = 0; i < 4; ++i) { c[i] = ck[i]; c[7 - i] = m[i]; }
  if (__builtin_amdgcn_readfirstlane(tid >> 6) == 0) {
    fco::wave_sync();                                          // (the pair reads of gather_pairs are done: program order)
    if (tid < 8) special_put_pairs(tid, ck, cm, lds + kScratch);
    fco::wave_sync();
    if (tid < 8) special_get(tid, lds + kScratch, c);
  }
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
  lane_twiddle_table(tid, tw16k, lds + kLaneTw);
  MidConst mc;
  mid_init(tid, tw16k, mc);
  LaneConst lc;
  lane_init(tid, lc);
  C32 v[8];
  load_taps(tid, ny, y + yrow * ny, p, v);
  __syncthreads();
  C32 c[8], cA[4], cM[4], z[8];
  forward_block(tid, v, lds, tl, lc, c);
  gather_pairs(tid, c, lds, cA, cM);
  mid_split(tid, cA, cM, mc, z);
  C32* Hp = H + (int64_t)blockIdx.x * kHPerPart;
  const float sc = kSpectrumScale * spectrum_sign(tid);      // the TRUE tap spectra (see xl_level)
#pragma unroll
  for (int i = 0; i < 8; ++i) Hp[h_index(0, i, tid)] = C32{z[i].x * sc, z[i].y * sc};
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
  lane_twiddle_table(tid, tw16k, lds + kLaneTw);
  MidConst mc;
  mid_init(tid, tw16k, mc);
  LaneConst lc;
  lane_init(tid, lc);
  __syncthreads();
  const unsigned n_items = (unsigned)(g.rows * g.segs);          // < 2^31 (checked by the launcher): 32-bit item arithmetic
#pragma unroll 1
  for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t row = (int64_t)(item / (unsigned)g.segs);
    // block numbers fit 32 bits (n_blocks <= out_len / hop; the launcher checks rows * segs < 2^31): scalar compares in the loop
    const int j_lo = (int)((int64_t)(item - (unsigned)row * (unsigned)g.segs) * g.seg_blocks);
    const int j_hi = (int)(j_lo + g.seg_blocks < g.n_blocks ? j_lo + g.seg_blocks : g.n_blocks);
    // row numbers are uniform over the workgroup: taken through scalar registers, so that every row pointer below is a scalar
    // base (global_load v, v_offset, s[base]) instead of 64-bit vector addresses per tap-spectrum partition
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
      auto h_load = [&](int p) {           // partition p: coalesced loads from L2 (the thread-owned layout)
        // ONE scalar base (the row's table) + a 32-bit vector byte offset per load: global_load_dwordx4 v, v_off, s[base].
        // Left to itself the compiler forms 64-bit vector addresses per load (registers and carry chains) or, with per-load
        // scalar bases, hoists the base pairs out of the step loop and spills scalar registers.
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
      // (Round 5 tried to hide the tap-spectrum loads behind the passes: slower -- the reads are a throughput cost of the
      // vector-memory path, not an exposed latency.  profiles/r05_c_fdr_lab_barriers_hloads.txt)
      if (produce) {
        if (NP > 1) { h_load(1); mid_mac(tid, h0, z1, acc); }
        if (NP > 2) { h_load(2); mid_mac(tid, h0, z2, acc); }
        if (NP > 3) { h_load(3); mid_mac(tid, h0, z3, acc); }
      }
      C32 c[8];
      passA_fwd(tid, v, lds, tl);
      AAMD_FDR_BARRIER(1);
      {
        C32 L[7];
        lane_twiddles(tid, lds + kLaneTw, L);
        passB_fwd_a(tid, lds, L, c);
        quad_dft4_fwd(c, lc.s_hi, lc.s_lo, lc.rot);
        passB_fwd_b(tid, c, lds, tl);
        AAMD_FDR_BARRIER(2);
        passC_fwd_a(tid, lds, L, c);
      }
      quad_dft4_fwd(c, lc.s_hi, lc.s_lo, lc.rot);
      if (produce) h_load(0);              // H_0: an L2 round trip behind the quad's DFT-4 (requested earlier its 16 registers spill)
      C32 z0[8];
      {
        C32 cA[4], cM[4];
        gather_pairs(tid, c, lds, cA, cM);
        mid_split(tid, cA, cM, mc, z0);
      }
      if (produce) {
        mid_mac(tid, h0, z0, acc);
        C32 ck[4], cm[4];
        mid_merge(tid, acc, mc, ck, cm);
        scatter_pairs(tid, ck, cm, lds, c);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (NP > 3) z3[i] = z2[i];
        z2[i] = z1[i]; z1[i] = z0[i];
      }
      if (produce) {
        C32 L[7];
        quad_dft4_inv(c, lc.s_hi, lc.s_lo, lc.rot);
        lane_twiddles(tid, lds + kLaneTw, L);
        passC_inv_b(tid, c, L, lds);       // (pass C' rewrites the cells pass C read: the same quad, program order)
        AAMD_FDR_BARRIER(2);
        passB_inv_a(tid, lds, tl, c);
        quad_dft4_inv(c, lc.s_hi, lc.s_lo, lc.rot);
        passB_inv_b(tid, c, L, lds);
        AAMD_FDR_BARRIER(1);
      } else {
        // a forward-only warm-up step: the next pass A rewrites cells that a slower wave's pass C may still be reading
        AAMD_FDR_BARRIER(2);
      }
      // the next block's samples: in flight during the last pass and the stores
      if (j + 1 < j_hi) load_block(tid, g, xr, j + 1, vin, v);
      if (produce) {
        C32 w[8];
        passA_inv(tid, lds, tl, w);
        store_block(tid, g, w, j, j_hi, vout, out_row);
        // No barrier here: pass A' READS the cells cell(k1, tid >> 5, tid & 31) and the next pass A WRITES exactly those cells
        // from the same thread -- program order of one thread is all the hazard needs, and every other access to them lies
        // behind the barrier that follows pass A.
      }
    }
  }
}
#endif  // __HIPCC__

}  // namespace fdr
}  // namespace aamd

// Polyphase windowed-sinc resampling on the matrix cores (heavy filters: kaiser_best 44.1k -> 16k
// has 815 taps x 160 phases; BASELINE config 3).
//
//   y[q*new + p] = sum_k h[p][k] * xpad[q*orig + k]          (functional/functional.py:1405-1432)
//
// is, per waveform, the product  Y[p][q] = sum_k H[p][k] X[k][q]  with the Hankel-structured
// X[k][q] = xpad[q*orig + k].  That contraction is FLOP-bound on this chip (~100
// FLOP/B), and the tap table is BANDED: phase p only has non-negligible taps in a window that
// slides with p.  So:
//   * phases are cut into tiles of 16; tile t uses taps [tap_lo[t], tap_lo[t] + 4 KS) only
//     (host-provided band table; KS = k-steps of v_mfma_f32_16x16x4_f32);
//   * one compute wave per (phase tile, q-group) keeps its H fragment (KS registers per lane)
//     resident for the whole launch and streams 16-q tiles of X through the MFMA: the only
//     per-MFMA traffic is ONE conflict-free ds_read_b32 of the X fragment;
//   * the waveform chunk (32 q per q-group: (32 qg - 1) orig + taps samples) lives in LDS,
//     double buffered; two loader waves fetch chunk c+1 (16-B loads, zero padding resolved by
//     index) while the compute waves work on chunk c -- one barrier per chunk;
//   * the contraction order is permuted (k-slot (kk, g) <-> tap tap_lo + kk + KS*g): with
//     KS = 16 (mod 32) the 32 lanes of a ds_read_b32 group (16 q x 2 g) hit 32 distinct banks
//     for odd `orig`, and the per-k-step address is an immediate offset;
//   * C layout gives each lane 4 consecutive phases of one q = one 16-B store.
// fp32 MFMA is an exact fp32 FMA chain (same numerics class as the reference's conv1d).
#pragma once
#include <cstring>
#include <cstdint>
#include "hd.h"

namespace aamd {
namespace rsm {

constexpr int kLoaderWaves = 2;
constexpr int kMaxPhaseTiles = 14;   // compute waves per workgroup <= 14 (+ 2 loaders = 16 waves)
constexpr int kQPerGroup = 32;       // two 16-q MFMA tiles per compute wave and chunk (two accumulators)

struct Geom {
  int64_t rows, length, row_stride, out_len;
  int orig, new_, width, taps;
  int pt0, n_pt;             // phase tiles [pt0, pt0 + n_pt) are produced by this launch
  int qg;                    // q-groups per workgroup; compute waves = n_pt * qg
  int rounds;                // q-groups a compute wave works through per chunk (f16 kernel; 1 for the fp32 kernel)
  int n_loaders;             // loader waves (f16 kernel: 2 or 4; the fp32 kernel has kLoaderWaves)
  int chunks_per_row;
  int64_t n_chunks;
  int chunks_per_block;
  int buf_floats;            // floats per LDS chunk buffer (multiple of 4)
  int vec_in, vec_out;       // 16-B global loads / stores are legal
  int tap_lo[kMaxPhaseTiles];
  int lab;                   // tools only (AAMD_RSM_LAB): 1 no conversion, 2 no MFMA, 4 no LDS operand reads, 8 no global loads, 16 no tap fragments, 32 no stores
  const uint32_t* frag;      // f16 kernel: the packed tap fragments prepared once per filter (frag_piece), or null: every compute wave
                             // splits its 224 taps itself (224 scattered loads + 448 binary16 conversions per lane and launch)
};

// k-steps needed for a band of `span` taps, from the supported set; 0 = too wide.  16 / 48 / 80 / 112 are = 16 mod 32 (the
// 4-byte operand reads are conflict-free for them).  104 (round 6) exists for the 8-byte operand layout only, i.e. for odd
// `orig`: its odd lane groups rotate by 7 steps (b64_rot) -- and it is exactly what BASELINE config 3 needs: the 16-phase
// tiles of kaiser_best 441 : 160 span 414 taps = 104 k-steps, so 112 multiplied 7 % of zeros.
AAMD_HD int pick_ks(int span, int orig = 0) {
  const int need = (span + 3) / 4;
  if (need <= 16) return 16;
  if (need <= 48) return 48;
  if (need <= 80) return 80;
  if (need <= 104 && (orig & 1)) return 104;
  if (need <= 112) return 112;
  return 0;
}
AAMD_HD int max_compute_waves(int ks) { return ks >= 80 ? 10 : 14; }
// 16-byte pieces per lane that a loader wave of the f16 kernel keeps in flight in registers (a whole chunk: the launcher
// sizes the chunk to at most 128 of these); the 16-wave instantiations (KS < 80) have 128 registers per thread
AAMD_HD constexpr int loader_pieces_per_lane(int ks) { return ks >= 80 ? 30 : 20; }

AAMD_HD int chunk_q(const Geom& g) { return kQPerGroup * g.qg * g.rounds; }

// floats one chunk buffer must hold: every index a compute wave can read, + 3 for the 16-B phase
AAMD_HD int buf_floats_needed(int qc, int orig, int taps, int max_tap_lo, int ks) {
  int reach = max_tap_lo + 4 * ks;
  if (reach < taps) reach = taps;
  return (((qc - 1) * orig + reach + 3) + 3) & ~3;
}

// Chunk geometry of one launch over g.n_pt phase tiles: q-groups (compute waves = n_pt * qg fill the workgroup, bounded by the
// row), and for the f16 kernel the loader waves and the rounds -- a chunk is as long as the LDS double buffer and the
// registers of the loader waves (64 x loader_pieces_per_lane pieces each: the whole chunk is in flight) allow, because a
// chunk period costs ~4 us of fetch / stage / barrier whatever its length: the 10-tile 160 : 147 pair (48 k -> 44.1 k) with
// 32 q per chunk spent 5.6 us per chunk on 12 MFMAs per wave.  Four loader waves when the 16 wave slots have room.
// Returns false when not even one q-group fits the LDS (huge orig: scalar kernel).
AAMD_HD bool plan_chunk(Geom& g, int ks, bool f16, int64_t nq, int max_lo, size_t lds_cap) {
  const int max_cw = max_compute_waves(ks);
  int qg = max_cw / g.n_pt;
  while (qg > 1 && (int64_t)kQPerGroup * (qg - 1) >= nq) --qg;
  g.rounds = 1;
  g.n_loaders = kLoaderWaves;
  if (f16 && ks < 80 && g.n_pt * qg + 4 <= 16) g.n_loaders = 4;
  auto fits = [&](int q_groups) {
    const int64_t fl = buf_floats_needed(kQPerGroup * q_groups, g.orig, g.taps, max_lo, ks);
    if (2 * (size_t)fl * sizeof(float) + 48 > lds_cap) return false;
    return !f16 || fl <= 4ll * 64 * g.n_loaders * loader_pieces_per_lane(ks);
  };
  while (qg > 1 && !fits(qg)) --qg;
  if (f16)
    while (g.rounds < 8 && (int64_t)kQPerGroup * qg * g.rounds < nq && fits(qg * (g.rounds + 1))) ++g.rounds;
  g.qg = qg;
  g.buf_floats = buf_floats_needed(chunk_q(g), g.orig, g.taps, max_lo, ks);
  // Round 6: on long rows the chunk buffer of the wide f16 instantiations is padded to EXACTLY what the two loader waves hold in
  // registers (64 x 2 x 30 pieces of 16 bytes): every lane then owns a whole piece in every one of its 30 slots -- no index clamp
  // in front of the loads and the LDS writes, immediate offsets (the few samples fetched beyond the chunk's reach are the next
  // chunk's, i.e. L2 hits; they take part in the chunk maximum, which any scale >= the needed one tolerates)
  if (f16 && ks >= 80 && g.n_loaders == kLoaderWaves) {
    const int64_t full = 4ll * 64 * kLoaderWaves * loader_pieces_per_lane(ks);
    if (g.buf_floats <= full && nq * g.orig >= 8 * full && 2 * (size_t)full * sizeof(float) + 48 <= lds_cap) g.buf_floats = (int)full;
  }
  return 2 * (size_t)g.buf_floats * sizeof(float) + (f16 ? 48 : 0) <= lds_cap;
}

// sample index of LDS float 0 of chunk (qc0): 16-B phase aligned with global memory when vec_in
AAMD_HD int64_t chunk_a0(const Geom& g, int64_t qc0) {
  const int64_t s = qc0 * g.orig - g.width;          // sample index of padded index qc0*orig
  if (!g.vec_in) return s;
  return s - (((s % 4) + 4) % 4);
}

// ---- loader: one 16-B piece of chunk buffer <- waveform samples (zero outside [0, length)) ----
AAMD_HD F4 load_piece(const Geom& g, const float* row, int64_t a0, int piece) {
  const int64_t i = a0 + 4 * (int64_t)piece;
  if (g.vec_in && i >= 0 && i + 4 <= g.length) return *reinterpret_cast<const F4*>(row + i);
  F4 v;
  v.x = (i >= 0 && i < g.length) ? row[i] : 0.0f;
  v.y = (i + 1 >= 0 && i + 1 < g.length) ? row[i + 1] : 0.0f;
  v.z = (i + 2 >= 0 && i + 2 < g.length) ? row[i + 2] : 0.0f;
  v.w = (i + 3 >= 0 && i + 3 < g.length) ? row[i + 3] : 0.0f;
  return v;
}

// ---- fragment index math (shared with tests/cpu_sim) ----------------------------------------
// A fragment of lane l, k-step kk, phase tile pt: H[16 pt + l%16][tap_lo + kk + KS * (l/16)]
AAMD_HD float a_frag(const Geom& g, const float* kern, int pt, int tap_lo, int ks, int kk, int lane) {
  const int p = 16 * pt + (lane & 15), tap = tap_lo + kk + ks * (lane >> 4);
  return (p < g.new_ && tap < g.taps) ? kern[(int64_t)p * g.taps + tap] : 0.0f;
}
// LDS float index of the B fragment of lane l for q-tile `qt` (0-based inside the chunk), k-step 0
AAMD_HD int b_base(const Geom& g, int qt, int tap_lo, int ks, int shift, int lane) {
  return (16 * qt + (lane & 15)) * g.orig + tap_lo + ks * (lane >> 4) + shift;
}
// store the C fragment (4 consecutive phases of one q) of q-tile qt
AAMD_HD void store_c(const Geom& g, float* out_row, int64_t qc0, int qt, int pt, int lane,
                     float c0, float c1, float c2, float c3) {
  const int64_t q = qc0 + 16 * qt + (lane & 15);
  const int p0 = 16 * pt + 4 * (lane >> 4);
  const int64_t oi = q * g.new_ + p0;
  if (g.vec_out && p0 + 4 <= g.new_ && oi + 4 <= g.out_len) {
    *reinterpret_cast<F4*>(out_row + oi) = F4{c0, c1, c2, c3};
    return;
  }
  // (a 4-byte-aligned dwordx4 store for odd `new` -- q new + p0 is not 16-byte aligned then -- aborts the process on this
  // stack with a memory fault: measured in round 2; the four guarded dword stores stay)
  const float c[4] = {c0, c1, c2, c3};
  for (int i = 0; i < 4; ++i)
    if (p0 + i < g.new_ && oi + i < g.out_len) out_row[oi + i] = c[i];
}

// store_c for an explicit output group q (the 8-byte operand-read layout below deals q to tiles by parity)
AAMD_HD void store_c_at(const Geom& g, float* out_row, int64_t q, int pt, int lane, float c0, float c1, float c2, float c3) {
  const int p0 = 16 * pt + 4 * (lane >> 4);
  const int64_t oi = q * g.new_ + p0;
  if (g.vec_out && p0 + 4 <= g.new_ && oi + 4 <= g.out_len) {
    *reinterpret_cast<F4*>(out_row + oi) = F4{c0, c1, c2, c3};
    return;
  }
  const float c[4] = {c0, c1, c2, c3};
  for (int i = 0; i < 4; ++i)
    if (p0 + i < g.new_ && oi + i < g.out_len) out_row[oi + i] = c[i];
}

// ---- 8-byte operand reads (round 5; f16 kernel, odd `orig`, KS = 80 / 112) -------------------------------------------------
// The f16 kernel's MFMA loop is bound by its LDS operand reads: a compute wave issues 8 ds_read_b32 per operand tile (a lane
// needs 8 consecutive dwords, and q orig + tap is odd for every second q), 2 240 of them per chunk and CU = 4 480 LDS cycles
// against 1 680 cycles of matrix pipe per SIMD.  ds_read_b64 moves twice the bytes per LDS cycle (MI355X_MICROARCH.md, LDS
// table) but wants 8-byte aligned addresses and is served in groups of 32 lanes.  So, for odd `orig`:
//   * the 32 output groups of a q-group are dealt to the two MFMA tiles BY PARITY -- tile h holds q = 2 n + h (n = lane & 15):
//     the dword address (32 grp + 2 n + h) orig + tap_lo + KS g + shift + 8 s then has the same parity in all 64 lanes, and
//     exactly one of the two tiles is 8-byte aligned, wave-uniformly: h = (tap_lo + shift) & 1.  That tile reads 4 pairs per
//     step; the other one reads the 5 aligned pairs that start one dword earlier and regroups from dword 1 on (the
//     v_perm_b32 of the hi / lo regrouping takes any two registers: no extra instruction);
//   * bank model: lanes n of one group are 2 orig dwords apart -- 2 (orig n mod 32) mod 64, sixteen distinct EVEN banks whose
//     complement among the even banks is the same set + 32; lane group g + 1 sits KS = 16 (mod 32) dwords further, so the odd
//     lane groups walk the contraction steps ROTATED by r steps with KS + 8 r = 32 (mod 64) (r = 6 for KS = 112, 7 for 104, 2 for 80;
//     none exists for 16 / 48): step s of an odd group multiplies taps tap_lo + KS g + 8 ((s + r) mod NS) + e.  The 32 lanes a
//     ds_read_b64 is served with then cover 32 even banks + their odd neighbours: conflict-free (the sum over the contraction
//     does not care about the order; A fragments are loaded in the same rotated order).
// Round 6: the rotation WITH WRAP of round 5 kept that promise only in front of the wrap -- behind it the odd groups sat 8 r dwords
// from the even ones, not 32 (mod 64): SQ_LDS_BANK_CONFLICT was 31 % of the kernel's LDS cycles (profiles/r06_zl_pmc_resample.txt;
// the bank model below reproduces the count: 7 of the 13 steps of KS = 104 two-way).  What is needed is a PERMUTATION sigma of the
// steps with  KS + 8 (sigma(s) - s) = 32 (mod 64)  for as many s as possible, in runs of consecutive steps (tile B carries a
// dword from step to step inside a run).  The congruence fixes sigma(s) - s modulo 8, and a perfect permutation does not exist
// for these KS; the tables leave ONE step of 13 (KS = 104) and TWO of 14 / 10 (112 / 80) two-way conflicted:
//   104: sigma - s = 7 (mod 8): s = 1 .. 4 -> 0 .. 3, s = 5 -> 12, s = 6 .. 12 -> 5 .. 11, and step 0 takes the block left (4)
//   112: sigma - s = 6 (mod 8): s = 2 .. 5 -> 0 .. 3, s = 6, 7 -> 12, 13, s = 8 .. 13 -> 6 .. 11, steps 0, 1 take 4, 5
//    80: sigma - s = 2 (mod 8): the rotation by 2 of round 5 (its last two steps are the conflicted ones)
AAMD_HD constexpr int b64_rot(int ks) { return ks == 112 ? 6 : ks == 104 ? 7 : ks == 80 ? 2 : 0; }   // != 0: the layout exists for this KS
AAMD_HD constexpr int b64_sigma(int ks, int s) {
  if (ks == 104) return s == 0 ? 4 : s <= 4 ? s - 1 : s == 5 ? 12 : s - 1;
  if (ks == 112) return s <= 1 ? s + 4 : s <= 5 ? s - 2 : s <= 7 ? s + 6 : s - 2;
  if (ks == 80) return s < 8 ? s + 2 : s - 8;
  return s;
}
// a step where the odd groups' run of consecutive table steps starts (tile B reads its carried dword there)
AAMD_HD constexpr bool b64_run_start(int ks, int s) { return s == 0 || b64_sigma(ks, s) != b64_sigma(ks, s - 1) + 1; }
// what the odd groups' operand pointer moves by in front of step s, beyond the 8 dwords every step advances
AAMD_HD constexpr int b64_jump(int ks, int s) {
  return 8 * ((b64_sigma(ks, s) - s) - (s == 0 ? 0 : b64_sigma(ks, s - 1) - (s - 1)));
}
AAMD_HD constexpr bool b64_sigma_is_permutation(int ks) {
  unsigned seen = 0;
  for (int s = 0; s < ks / 8; ++s) seen |= 1u << b64_sigma(ks, s);
  return seen == (1u << (ks / 8)) - 1u;
}
AAMD_HD constexpr int b64_conflicted_steps(int ks) {
  int n = 0;
  for (int s = 0; s < ks / 8; ++s) n += ((ks + 8 * (b64_sigma(ks, s) - s)) % 64 + 64) % 64 != 32;
  return n;
}
static_assert(b64_sigma_is_permutation(80) && b64_sigma_is_permutation(104) && b64_sigma_is_permutation(112), "every table step once");
static_assert(b64_conflicted_steps(104) == 1 && b64_conflicted_steps(112) == 2 && b64_conflicted_steps(80) == 2,
              "odd lane groups must land 32 banks from the even ones in all but these steps");
AAMD_HD bool b64_ok(int ks, int orig) { return (orig & 1) != 0 && b64_rot(ks) != 0; }
// tap-table step that lane group `grp4` (= lane >> 4) multiplies at loop step s
AAMD_HD int b64_step(int ks, int s, int grp4) { return (grp4 & 1) ? b64_sigma(ks, s) : s; }
// LDS dword index of the B fragment of lane l, tile h (q = 2 n + h) of 32-q group `grp`, table step 0
AAMD_HD int b_base64(const Geom& g, int grp, int h, int tap_lo, int ks, int shift, int lane) {
  return (32 * grp + 2 * (lane & 15) + h) * g.orig + tap_lo + ks * (lane >> 4) + shift;
}

// ---- prepared tap fragments (round 5) --------------------------------------------------------------------------------------
// The f16 kernel's A operand of (phase tile pt, table step s, lane) is 8 binary16 hi parts + 8 lo parts of the taps
// tap_lo[pt] + KS (lane >> 4) + 8 s + e of phase 16 pt + (lane & 15), scaled by 2^15: constants of the filter.  Every compute wave of
// every workgroup used to form them in its prologue -- 224 scattered loads and 448 conversions per lane; the lab switch that skips
// them says 0.557 -> 0.485 ms on the cfg3 shard (profiles/r05_n_rsm_lab_ablation_prepared_fragments.txt), 13 % of the launch.  Prepared ONCE per filter
// (frag_build_kernel, the host keeps the table with the tap tensor) they are 2 NS coalesced 16-byte loads per lane.  The table is in
// natural step order; the 8-byte operand layout's rotated lane groups (b64_step) pick their rows from it.
AAMD_HD int64_t frag_piece(int pt, int ns, int s, int hl, int lane) { return (((int64_t)pt * ns + s) * 2 + hl) * 64 + lane; }   // 16-B pieces
AAMD_HD int64_t frag_bytes(int n_tiles, int ks) { return (int64_t)n_tiles * (ks / 8) * 2 * 64 * 16; }

// ---- the f16 matrix-pipe variant: fp32-class accuracy from three f16 MFMAs per product --------------------------------
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the f16 rate of the matrix cores (157 against 2 500 TFLOP/s), and cfg3 is bound by
// exactly that.  Every operand is therefore split into two binary16 numbers, v = hi + lo with hi = f16(v), lo = f16(v - hi)
// (22 significant bits together), and a product of sums is evaluated as  hi*hi + hi*lo + lo*hi  on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation (the dropped lo*lo is 2^-22 of the product): three instructions that do
// 8 x the contraction depth of one fp32 instruction in half its time.
//   * range: binary16 ends at 65504 and loses precision under 6e-5, so operands are scaled by powers of two (exact): taps by
//     2^15 (|h| <= 1), the samples of a chunk by 2^(14 - e) with e the binary exponent of the chunk's LARGEST |sample| --
//     found by the loader waves (LDS atomic max) while they fetch the chunk; the result is scaled back by 2^(e - 29).
//     A low part that falls under the binary16 normal range is worth < 2^-31 of the chunk's peak.
//   * the chunk sits in LDS as one dword per sample, (lo << 16) | hi: same footprint and the same conflict-free addresses as
//     the float image; the loaders store raw floats, and whichever wave is free converts the buffer in place, batch by
//     batch (LDS counters; one barrier per chunk -- see the timeline in the kernel).
//   * contraction slot (step s, lane group g, element e) <-> tap tap_lo + KS g + 8 s + e: a lane reads 8 consecutive dwords
//     per step and q-tile and regroups them with v_perm_b32 into the hi and the lo operand.
AAMD_HD uint16_t f16_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint16_t, static_cast<_Float16>(f));
#else
  uint32_t u;                                   // round to nearest even, subnormals kept (what v_cvt_f16_f32 does)
  std::memcpy(&u, &f, 4);
  const uint32_t sign = u & 0x80000000u;
  u ^= sign;
  uint16_t o;
  if (u >= ((127u + 16u) << 23)) {
    o = (u > (255u << 23)) ? 0x7e00 : 0x7c00;
  } else if (u < (113u << 23)) {
    const uint32_t magic = ((127u - 15u) + (23u - 10u) + 1u) << 23;
    float t, mf;
    std::memcpy(&t, &u, 4);
    std::memcpy(&mf, &magic, 4);
    t += mf;
    uint32_t tu;
    std::memcpy(&tu, &t, 4);
    o = (uint16_t)(tu - magic);
  } else {
    const uint32_t odd = (u >> 13) & 1u;
    u += ((15u - 127u) << 23) + 0xfffu;
    u += odd;
    o = (uint16_t)(u >> 13);
  }
  return (uint16_t)(o | (sign >> 16));
#endif
}
AAMD_HD float f16_value(uint16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
  return static_cast<float>(__builtin_bit_cast(_Float16, h));
#else
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 31u, m = h & 1023u, u;
  if (e == 0) {
    if (m == 0) { u = sign; }
    else {
      int sh = 0;
      while (!(m & 1024u)) { m <<= 1; ++sh; }
      u = sign | ((uint32_t)(113 - sh) << 23) | ((m & 1023u) << 13);
    }
  } else if (e == 31) {
    u = sign | 0x7f800000u | (m << 13);
  } else {
    u = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  std::memcpy(&f, &u, 4);
  return f;
#endif
}
// (lo << 16) | hi of an already scaled value
AAMD_HD uint32_t pack_hl(float xs) {
  const uint16_t hi = f16_bits(xs);
  const uint16_t lo = f16_bits(xs - f16_value(hi));
  return (uint32_t)hi | ((uint32_t)lo << 16);
}
// The SAMPLES' split (round 5; the taps keep pack_hl): hi = the value with its mantissa CUT to binary16's 10 bits (one v_and_b32,
// exact in binary16), lo = the remainder rounded to nearest -- so that ONE v_cvt_pk_f16_f32 (hi, lo) of gfx950 forms the packed dword:
// 4 operations per sample (scale, and, subtract, convert-and-pack) instead of 6 (scale, convert, convert back, subtract, convert,
// pack); the conversion is 16 % of a BASELINE config-3 launch.  The pair carries 2^-21 of the sample instead of 2^-22 (a cut hi leaves
// a remainder twice as large), unbiased.  (The first build used v_cvt_pkrtz_f16_f32: a lo rounded towards zero is a BIAS that the
// filter's DC gain adds up coherently -- one reference fixture failed its element-wise bound, 1.25e-7 of the peak.)
AAMD_HD uint32_t pack_hl_cut(float xs) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
  typedef float f2_ __attribute__((ext_vector_type(2)));
  const float hi = __uint_as_float(__float_as_uint(xs) & 0xffffe000u);
  const f2_ v = {hi, xs - hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2_));
#else
  uint32_t u;
  std::memcpy(&u, &xs, 4);
  u &= 0xffffe000u;
  float hi;
  std::memcpy(&hi, &u, 4);
  return (uint32_t)f16_bits(hi) | ((uint32_t)f16_bits(xs - hi) << 16);
#endif
}
constexpr int kTapShift = 15;                       // taps are scaled by 2^15
// scale of a chunk from the bits of its largest |sample|: the largest scaled sample lies in [2^14, 2^15)
AAMD_HD void chunk_scale(uint32_t max_bits, float& scale, float& inv) {
  int e = (int)(max_bits >> 23) - 127;
  if (max_bits == 0 || e > 90 || e < -90) e = 14;   // silence, inf / nan, absurd magnitudes: unit scale
  const uint32_t su = (uint32_t)(127 + 14 - e) << 23, iu = (uint32_t)(127 + e - 14 - kTapShift) << 23;
  std::memcpy(&scale, &su, 4);
  std::memcpy(&inv, &iu, 4);
}
// A operand dword d (elements 2 d, 2 d + 1) of step s: the hi parts and the lo parts
// (a_pack16_at: the same for an explicit tap index `tap` of element 2 d)
AAMD_HD void a_pack16_at(const Geom& g, const float* kern, int pt, int tap, int lane, uint32_t& hi, uint32_t& lo);
AAMD_HD void a_pack16(const Geom& g, const float* kern, int pt, int tap_lo, int ks, int s, int d, int lane,
                      uint32_t& hi, uint32_t& lo) {
  a_pack16_at(g, kern, pt, tap_lo + ks * (lane >> 4) + 8 * s + 2 * d, lane, hi, lo);
}
AAMD_HD void a_pack16_at(const Geom& g, const float* kern, int pt, int tap, int lane, uint32_t& hi, uint32_t& lo) {
  const int p = 16 * pt + (lane & 15);
  // unconditional loads from clamped indices, zeroed by select: the 2 x 112 loads of a lane are then issued back to back
  // (behind a branch each they were serialised: 0.08 ms per launch on the cfg3 shard)
  const float* row = kern + (int64_t)(p < g.new_ ? p : 0) * g.taps;
  const float r0 = row[tap < g.taps ? tap : g.taps - 1], r1 = row[tap + 1 < g.taps ? tap + 1 : g.taps - 1];
  const float h0 = (p < g.new_ && tap < g.taps) ? r0 : 0.0f;
  const float h1 = (p < g.new_ && tap + 1 < g.taps) ? r1 : 0.0f;
  const uint32_t a = pack_hl(h0 * (float)(1 << kTapShift)), b = pack_hl(h1 * (float)(1 << kTapShift));
  hi = (a & 0xffffu) | (b << 16);
  lo = (a >> 16) | (b & 0xffff0000u);
}

#if defined(__HIPCC__)
template <int KS>
__global__ void __launch_bounds__(KS >= 80 ? 768 : 1024)
resample_mfma_kernel(Geom g, const float* __restrict__ wav, const float* __restrict__ kern,
                     float* __restrict__ out) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  extern __shared__ __attribute__((aligned(16))) float smem_rsm[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ncw = g.n_pt * g.qg;
  const bool loader = wave >= ncw;
  const int pt_l = loader ? 0 : wave % g.n_pt;
  const int qgi = loader ? 0 : wave / g.n_pt;
  const int pt = g.pt0 + pt_l;
  const int tap_lo = g.tap_lo[pt_l];
  const int qc = chunk_q(g);

  const int64_t first = (int64_t)blockIdx.x * g.chunks_per_block;
  int64_t end = first + g.chunks_per_block;
  if (end > g.n_chunks) end = g.n_chunks;

  // The two roles run separate loops with the same barrier count (the branch is wave-uniform), so
  // the loader's staging registers and the compute waves' H fragment never share a live range.
  if (loader) {
    const int pieces = g.buf_floats >> 2;
    const int lt = threadIdx.x - 64 * ncw;          // loader thread id
    auto load_chunk = [&](int64_t cid, float* buf) {
      const int64_t row = cid / g.chunks_per_row;
      const int64_t qc0 = (cid - row * g.chunks_per_row) * qc;
      const float* wrow = wav + row * g.row_stride;
      const int64_t a0 = chunk_a0(g, qc0);
      // half a chunk in flight at once (U x 16 B per lane; the loader branch owns its registers):
      // with 8 the two loader waves were latency-bound and the compute waves idled at the barrier
      constexpr int U = 16;
      // interior chunk (every sample exists, 16-B aligned): branch-free batches of U loads per lane
      const bool interior = g.vec_in && a0 >= 0 && a0 + g.buf_floats <= g.length;
      for (int j0 = lt; j0 < pieces; j0 += 64 * kLoaderWaves * U) {
        F4 v[U];
        if (interior) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            int j = j0 + 64 * kLoaderWaves * u;
            if (j >= pieces) j = pieces - 1;
            v[u] = *reinterpret_cast<const F4*>(wrow + a0 + 4 * j);
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = j0 + 64 * kLoaderWaves * u;
            if (j < pieces) v[u] = load_piece(g, wrow, a0, j);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + 64 * kLoaderWaves * u;
          if (j < pieces) *reinterpret_cast<F4*>(buf + 4 * j) = v[u];
        }
      }
    };
    if (first < end) load_chunk(first, smem_rsm);
    __syncthreads();
    for (int64_t cid = first; cid < end; ++cid) {
      if (cid + 1 < end) load_chunk(cid + 1, smem_rsm + ((cid + 1 - first) & 1) * g.buf_floats);
      __syncthreads();
    }
    return;
  }

  float a[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) a[kk] = a_frag(g, kern, pt, tap_lo, KS, kk, lane);
  __syncthreads();
  for (int64_t cid = first; cid < end; ++cid) {
    const float* buf = smem_rsm + ((cid - first) & 1) * g.buf_floats;
    const int64_t row = cid / g.chunks_per_row;
    const int64_t qc0 = (cid - row * g.chunks_per_row) * qc;
    const int shift = (int)((qc0 * g.orig - g.width) - chunk_a0(g, qc0));
    const int qt0 = 2 * qgi, qt1 = 2 * qgi + 1;
    const float* b0 = buf + b_base(g, qt0, tap_lo, KS, shift, lane);
    const float* b1 = buf + b_base(g, qt1, tap_lo, KS, shift, lane);
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b0[kk], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b1[kk], acc1, 0, 0, 0);
    }
    float* out_row = out + row * g.out_len;
    store_c(g, out_row, qc0, qt0, pt, lane, acc0[0], acc0[1], acc0[2], acc0[3]);
    store_c(g, out_row, qc0, qt1, pt, lane, acc1[0], acc1[1], acc1[2], acc1[3]);
    __syncthreads();
  }
}

// lab bit 64: timestamps (100 MHz wall clock) of workgroup 0, chunks 8 .. 39: [wave 16][chunk 32][stamp 8] (tools only)
__device__ long long g_rsm_census[16 * 32 * 8];
#define AAMD_RSM_STAMP(K, I)                                                                                       \
  if ((lab & 64) && blockIdx.x == 0 && lane == 0 && (K) >= 8 && (K) < 40)                                        \
    g_rsm_census[(wave * 32 + ((K) - 8)) * 8 + (I)] = wall_clock64();
// the f16 variant (see the block comment above a_pack16): same geometry, same band tables, same LDS addresses
// LABM: the tools-only switches live in instantiations of their own -- 1 = the time stamps (AAMD_RSM_LAB=64), 2 = every lab
// switch (their branches inside the MFMA loop cut it into basic blocks the scheduler cannot move the LDS reads across:
// the product kernel carried them until the round-2 census, profiles/r02_v)
// RD: 0 = ds_read_b32 operand reads (any geometry); 1 = the 8-byte operand reads described above b64_rot (odd orig, KS >= 80)
// FULL: 1 = the chunk buffer is plan_chunk's padded one (64 x 2 x U pieces: every lane of the two loader waves owns a whole piece in
// every slot) -- an instantiation of its own, because both addressing schemes in one kernel spilled
template <int KS, int LABM, int RD = 0, int FULL = 0>
__global__ void __launch_bounds__(KS >= 80 ? 768 : 1024)
resample_f16_kernel(Geom g, const float* __restrict__ wav, const float* __restrict__ kern, float* __restrict__ out) {
  static_assert(RD == 0 || b64_rot(KS) != 0, "8-byte operand reads need a conflict-free rotation (KS = 80 / 104 / 112)");
  const int lab = LABM ? g.lab : 0;
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  using h8 = __attribute__((ext_vector_type(8))) _Float16;
  using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
  constexpr int NS = KS / 8;                       // MFMA steps: 8 taps per lane group and step
  extern __shared__ __attribute__((aligned(16))) float smem_rsm[];
#ifndef AAMD_RSM_FLAGS
#define AAMD_RSM_FLAGS 0
#endif
  // AAMD_RSM_FLAGS = 1 (round 6): no workgroup barrier in the chunk loop.  The two LDS buffers are handed over by counters -- `fullc[b]`:
  // loader waves that have written their packed pieces of a chunk into buffer b; `donec[b]`: compute waves that have finished reading
  // it -- so a wave that is done with chunk k goes on to chunk k + 1 as soon as the loaders have it, instead of idling at the barrier
  // until the slowest wave of the workgroup arrives (the oldest wave of a SIMD finishes its matrix loop 2 us before the youngest of a
  // 5.3 us period, profiles/r06_zi).  Every wave of the workgroup is resident: the waits are bounded.  Four maximum slots instead of
  // three; a slot is zeroed by the first loader wave one chunk before its next use (see publish).
  constexpr int kSlots = AAMD_RSM_FLAGS ? 4 : 3;
#if AAMD_RSM_FLAGS && defined(AAMD_RSM_SHARED_CONV)
#error "the flag hand-over needs the loader-side conversion (the batch counters of AAMD_RSM_SHARED_CONV share its LDS words)"
#endif
  unsigned* mx = reinterpret_cast<unsigned*>(smem_rsm + 2 * g.buf_floats);     // [kSlots]: largest |sample| bits, chunk k -> slot k % kSlots
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ncw = g.n_pt * g.qg;
  const int nld = g.n_loaders;
  const bool loader = wave >= ncw;
  const int pt_l = loader ? 0 : wave % g.n_pt;
  const int qgi = loader ? 0 : wave / g.n_pt;
  const int pt = g.pt0 + pt_l;
  const int tap_lo = g.tap_lo[pt_l];
  const int qc = chunk_q(g);
  const int pieces = g.buf_floats >> 2;

  const int64_t first = (int64_t)blockIdx.x * g.chunks_per_block;
  int64_t end = first + g.chunks_per_block;
  if (end > g.n_chunks) end = g.n_chunks;
  if (threadIdx.x < 12) mx[threadIdx.x] = 0u;         // maxima + arrival counters + (3 batch counters | 2 + 2 hand-over counters): 48 bytes
  __syncthreads();

  // Timeline (one barrier per chunk; round 6):
  //   loaders   fetch(f) stage(f) fetch(f+1)   A0 | stage(f+1) fetch(f+2)                  A(f) | stage(f+2) ...
  //   the rest  tap fragments                  A0 | compute(f) (all rounds), stores, header(f+1) A(f) | compute(f+1) ...
  // fetch = the chunk's global loads into registers: issued a whole chunk period before the LDS buffer they go to is free
  // (two chunks in LDS + one in the loaders' registers = the HBM latency of a 60 KB burst per CU is off the critical path);
  // stage = wait for the data, publish the wave's largest |sample| (max3 chain + DPP reduction + one LDS atomic max), count the
  // wave in, wait for the other loader wave(s), then split the pieces IN THE REGISTERS with the chunk's scale and write the
  // packed (lo << 16 | hi) dwords to the free buffer.  (Rounds 2-5, kept as -DAAMD_RSM_SHARED_CONV for A/Bs: the loaders staged
  // the raw floats and every wave that found itself free converted batches of them in place -- `convert` below; with the
  // 6-operation split and the staged path of round 2 the two loader waves were the critical path, profiles/r02_v.  With the
  // 4-operation split, buffer loads and immediate offsets a loader wave's chunk is ~650 instructions, profiles/r06_zi.)
  unsigned* cnt = mx + kSlots;                         // [kSlots]: loader waves whose maximum is in LDS, monotonic
  unsigned* fullc = mx + 8;                            // [2] (AAMD_RSM_FLAGS): loader waves that have filled buffer b, monotonic
  unsigned* donec = mx + 10;                           // [2] (AAMD_RSM_FLAGS): compute waves that have consumed buffer b, monotonic
  (void)fullc; (void)donec;
  auto pack4 = [](const F4& t, float scale) {
    u32x4 o;
#if defined(AAMD_RSM_RNE_SAMPLES)    /* lab: the round-to-nearest split of the samples (6 operations each; the taps always use it) */
    o.x = pack_hl(t.x * scale); o.y = pack_hl(t.y * scale); o.z = pack_hl(t.z * scale); o.w = pack_hl(t.w * scale);
#else
    o.x = pack_hl_cut(t.x * scale); o.y = pack_hl_cut(t.y * scale); o.z = pack_hl_cut(t.z * scale); o.w = pack_hl_cut(t.w * scale);
#endif
    return o;
  };
  // convert(k): raw image of chunk k -> packed dwords, in batches of kGrab x 64 pieces handed out by an LDS counter to whichever
  // wave is free (the early finishers of the MFMA loop and the loaders behind their fetches take most of them; the last
  // compute waves of a period find nothing left), once all loader waves have counted themselves in (slot k % 3 is used for
  // the (k / 3 + 1)-th time; every wave of the workgroup is resident, so the wait is bounded)
  unsigned* grab = cnt + 3;                            // [3]: batches handed out (AAMD_RSM_SHARED_CONV builds only)
  constexpr int kGrab = 3;
  auto convert = [&](int k) {
    const unsigned want = (unsigned)nld * (unsigned)(k / kSlots + 1);
    while (__atomic_load_n(&cnt[k % kSlots], __ATOMIC_RELAXED) < want) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (!loader) { AAMD_RSM_STAMP(k - 1, 3) }
    if (lab & 1) return;
#if defined(AAMD_RSM_PRIO_CONV)
    __builtin_amdgcn_s_setprio(AAMD_RSM_PRIO_CONV);     // lab: the conversion yields its issue slots to the MFMA loops still running
#endif
    float scale, inv;
    chunk_scale(__atomic_load_n(&mx[k % kSlots], __ATOMIC_RELAXED), scale, inv);
    float* buf = smem_rsm + (k & 1) * g.buf_floats;
    for (;;) {
      unsigned b = 0u;
      if (lane == 0) b = atomicAdd(&grab[k % 3], 1u);
      const int j0 = (int)__builtin_amdgcn_readfirstlane(b) * (64 * kGrab) + lane;
      if (j0 - lane >= pieces) break;
      F4 t[kGrab];
#pragma unroll
      for (int i = 0; i < kGrab; ++i)
        if (j0 + 64 * i < pieces) t[i] = *reinterpret_cast<const F4*>(buf + 4 * (j0 + 64 * i));
#pragma unroll
      for (int i = 0; i < kGrab; ++i)
        if (j0 + 64 * i < pieces) *reinterpret_cast<u32x4*>(buf + 4 * (j0 + 64 * i)) = pack4(t[i], scale);
    }
#if defined(AAMD_RSM_PRIO_CONV)
    if (loader) __builtin_amdgcn_s_setprio(3);
#endif
  };
  if (loader) {
    __builtin_amdgcn_s_setprio(3);                  // the producers: first pick of the issue slots of their SIMDs
    const int lt = threadIdx.x - 64 * ncw;          // loader thread id
    // the whole chunk in flight at once (U x 16 B per lane; 120 registers that only this branch owns): with the fp32
    // kernel's 16 the two loader waves needed two HBM round trips per chunk and the f16 compute waves waited for them
    // (the 16-wave instantiations, KS < 80, have 128 registers: 30 x 16 B there spilled 536 B per thread, and a scratch
    // reload waits for every load in flight -- the default-quality rate pairs ran 4-5 x slower than they do now;
    // a chunk of more than 128 U pieces takes the staged path)
    constexpr int U = loader_pieces_per_lane(KS);
    F4 v[U];
    const float* wrow = wav;
    int64_t a0 = 0;
    bool interior = false;
    auto chunk_src = [&](int64_t cid) {
      const int64_t row = (int64_t)((uint32_t)cid / (uint32_t)g.chunks_per_row);   // n_chunks < 2^31 (checked by the launcher)
      const int64_t qc0 = (cid - row * g.chunks_per_row) * qc;
      wrow = wav + row * g.row_stride;
      a0 = chunk_a0(g, qc0);
      interior = g.vec_in && a0 >= 0 && a0 + g.buf_floats <= g.length && pieces <= 64 * nld * U && !(lab & 8);
    };
    constexpr bool full = FULL != 0;   // (the launcher: pieces == 64 * kLoaderWaves * U, two loader waves)
    // largest |sample| so far, as the bits of a non-negative float (ordered like unsigned integers): two v_max3_f32 with |.| source
    // modifiers per piece (rounds 2-6: four v_and_b32 + three integer maxima).  A NaN sample no longer forces the unit scale -- the
    // maximum skips it; the outputs it reaches are NaN either way
    auto absmax = [](unsigned m, const F4& t) {
#if defined(AAMD_RSM_INT_ABSMAX)
      const unsigned a = __float_as_uint(t.x) & 0x7fffffffu, bb = __float_as_uint(t.y) & 0x7fffffffu;
      const unsigned cc = __float_as_uint(t.z) & 0x7fffffffu, d = __float_as_uint(t.w) & 0x7fffffffu;
      return max(max(m, max(a, bb)), max(cc, d));
#else
      asm("v_max3_f32 %0, |%1|, |%2|, %0\n\tv_max3_f32 %0, |%3|, |%4|, %0" : "+v"(m) : "v"(t.x), "v"(t.y), "v"(t.z), "v"(t.w));   // (one statement: hipcc pads between two)
      return m;
#endif
    };
    // five pieces per statement (hipcc pads an s_nop between two asm statements; U = 20 / 30)
    auto absmax5 = [](unsigned m, const F4& a, const F4& b, const F4& c, const F4& d, const F4& e) {
      asm("v_max3_f32 %0, |%1|, |%2|, %0\n\tv_max3_f32 %0, |%3|, |%4|, %0\n\t"
          "v_max3_f32 %0, |%5|, |%6|, %0\n\tv_max3_f32 %0, |%7|, |%8|, %0\n\t"
          "v_max3_f32 %0, |%9|, |%10|, %0\n\tv_max3_f32 %0, |%11|, |%12|, %0\n\t"
          "v_max3_f32 %0, |%13|, |%14|, %0\n\tv_max3_f32 %0, |%15|, |%16|, %0\n\t"
          "v_max3_f32 %0, |%17|, |%18|, %0\n\tv_max3_f32 %0, |%19|, |%20|, %0"
          : "+v"(m) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w),
                      "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w), "v"(e.x), "v"(e.y), "v"(e.z), "v"(e.w));
      return m;
    };
    static_assert(U % 5 == 0, "absmax5 takes the loader's pieces five at a time");
    // the wave's maximum by DPP moves, ONE LDS atomic per wave (64 same-address LDS atomics per wave cost 3.4 us per chunk),
    // then -- behind the wave's own LDS writes, which the LDS executes in order -- the arrival count
    auto publish = [&](int k, unsigned m) {
#define AAMD_RSM_MAX_STEP(CTRL, ROWS) m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, CTRL, ROWS, 0xf, false));
      AAMD_RSM_MAX_STEP(0x111, 0xf) AAMD_RSM_MAX_STEP(0x112, 0xf) AAMD_RSM_MAX_STEP(0x114, 0xf) AAMD_RSM_MAX_STEP(0x118, 0xf)
      AAMD_RSM_MAX_STEP(0x142, 0xa) AAMD_RSM_MAX_STEP(0x143, 0xc)
#undef AAMD_RSM_MAX_STEP
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 63) {
        atomicMax(&mx[k % kSlots], m);
        // (flags: the slot of chunk k + 1 was last used by chunk k - 3, whose readers -- the compute waves' headers -- ran before
        // their loops of chunk k - 3, and this wave has waited for the end of those loops in its stage of chunk k - 1; the other
        // loader waves publish into it only behind this chunk's rendezvous, i.e. behind the atomicAdd below)
        if (AAMD_RSM_FLAGS && wave == ncw) __atomic_store_n(&mx[(k + 1) % kSlots], 0u, __ATOMIC_RELAXED);
        atomicAdd(&cnt[k % kSlots], 1u);
      }
    };
#define AAMD_RSM_FETCH(CID)                                                                                        \
    {                                                                                                              \
      chunk_src(CID);                                                                                              \
      if (interior) {   /* wave-uniform base + one 32-bit lane offset per load; edge chunks are fetched in STAGE */ \
        const F4* base4 = reinterpret_cast<const F4*>(wrow + a0);                                                  \
        int lt_ = lt;                                                                                              \
        asm volatile("" : "+v"(lt_));   /* offsets recomputed here: hoisted out of the chunk loop they get spilled */ \
        if (full) {   /* the base in SGPRs (said so: the compiler formed thirty 64-bit lane addresses and spilled) */  \
          const uint64_t b_ = reinterpret_cast<uint64_t>(base4);                                                   \
          const uint32_t blo_ = __builtin_amdgcn_readfirstlane((uint32_t)b_);                                      \
          const uint32_t bhi_ = __builtin_amdgcn_readfirstlane((uint32_t)(b_ >> 32));                              \
          const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
              reinterpret_cast<void*>(((uint64_t)bhi_ << 32) | blo_), 0, 64 * kLoaderWaves * U * 16, 0x00020000);  \
          const int vo_ = lt_ * 16;   /* buffer loads: descriptor + ONE lane offset + a constant per load */        \
          _Pragma("unroll") for (int u = 0; u < U; ++u)                                                            \
            v[u] = __builtin_bit_cast(F4, __builtin_amdgcn_raw_buffer_load_b128(rs_, vo_, 64 * kLoaderWaves * 16 * u, 0)); \
        } else {                                                                                                   \
          _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                          \
            const int j = lt_ + 64 * nld * u;                                                                      \
            v[u] = base4[(unsigned)(j < pieces ? j : pieces - 1)];                                                 \
          }                                                                                                        \
        }                                                                                                          \
      }                                                                                                            \
    }
#if defined(AAMD_RSM_SHARED_CONV)   /* lab: rounds 2-6 -- raw floats staged, every free wave converts them in place (convert()) */
#define AAMD_RSM_STAGE(CID)                                                                                        \
    {                                                                                                              \
      const int k_ = (int)((CID) - first);                                                                         \
      float* buf_ = smem_rsm + (k_ & 1) * g.buf_floats;                                                            \
      unsigned m_ = 0u;                                                                                            \
      AAMD_RSM_STAMP(k_, 0)                                                                                        \
      if (interior) {   /* from the registers; a lane past the end holds a copy of the last piece and rewrites it */ \
        int lt_ = lt;                                                                                              \
        asm volatile("" : "+v"(lt_));   /* as in FETCH: nothing of this hoisted out of the chunk loop */            \
        F4* dst_ = reinterpret_cast<F4*>(buf_);                                                                    \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                            \
          const int j = lt_ + 64 * nld * u;                                                               \
          dst_[j < pieces ? j : pieces - 1] = v[u];                                                                \
          m_ = absmax(m_, v[u]);                                                                                   \
        }                                                                                                          \
      } else if (!(lab & 8)) {   /* edge chunks, very long chunks */                                               \
        _Pragma("unroll 4") for (int j = lt; j < pieces; j += 64 * nld) {                                 \
          const F4 t = load_piece(g, wrow, a0, j);                                                                 \
          *reinterpret_cast<F4*>(buf_ + 4 * j) = t;                                                                \
          m_ = absmax(m_, t);                                                                                      \
        }                                                                                                          \
      }                                                                                                            \
      AAMD_RSM_STAMP(k_, 1)                                                                                        \
      publish(k_, m_);                                                                                             \
      AAMD_RSM_STAMP(k_, 2)                                                                                        \
    }
#else
    // Round 6: the loader waves convert THEIR pieces from the registers the fetch left them in -- largest |sample| of the wave
    // (DPP) -> LDS atomic max + arrival count -> wait for the other loader wave(s) -> the chunk's power-of-two scale -> split
    // and pack in the registers -> ONE 16-byte LDS write per piece.  (Rounds 2-6 staged the raw floats, and every wave of the
    // workgroup that found itself free converted batches of them in place: a write, a read and a write per piece, a batch
    // counter, and the compute waves' matrix loops could not start before the last batch.)  Same maximum, same scale, same
    // arithmetic per sample: bit-identical results.  Edge chunks (zero padding by index) and chunks longer than the loaders'
    // registers go through LDS: raw pieces written, then -- behind the same rendezvous -- converted in place by the lane that
    // wrote them (the LDS executes one wave's accesses in order).
#define AAMD_RSM_STAGE(CID)                                                                                        \
    {                                                                                                              \
      const int k_ = (int)((CID) - first);                                                                         \
      float* buf_ = smem_rsm + (k_ & 1) * g.buf_floats;                                                            \
      unsigned m_ = 0u;                                                                                            \
      AAMD_RSM_STAMP(k_, 0)                                                                                        \
      if (interior) {   /* ONE wait: the pieces were requested a chunk period ago (left alone: a vmcnt(N) per piece) */ \
        __builtin_amdgcn_s_waitcnt(0x0F70);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        _Pragma("unroll") for (int u = 0; u < U; u += 5) m_ = absmax5(m_, v[u], v[u + 1], v[u + 2], v[u + 3], v[u + 4]); \
      } else if (!(lab & 8)) {   /* edge chunks, very long chunks */                                               \
        _Pragma("unroll 4") for (int j = lt; j < pieces; j += 64 * nld) {                                          \
          const F4 t = load_piece(g, wrow, a0, j);                                                                 \
          *reinterpret_cast<F4*>(buf_ + 4 * j) = t;                                                                \
          m_ = absmax(m_, t);                                                                                      \
        }                                                                                                          \
      }                                                                                                            \
      AAMD_RSM_STAMP(k_, 1)                                                                                        \
      publish(k_, m_);                                                                                             \
      {                                                                                                            \
        const unsigned want_ = (unsigned)nld * (unsigned)(k_ / kSlots + 1);                                             \
        while (__atomic_load_n(&cnt[k_ % kSlots], __ATOMIC_RELAXED) < want_) __builtin_amdgcn_s_sleep(1);               \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                                                     \
      }                                                                                                            \
      AAMD_RSM_STAMP(k_, 2)                                                                                        \
      if (AAMD_RSM_FLAGS) {   /* the buffer's previous chunk (k - 2) consumed by every compute wave */              \
        const unsigned wd_ = (unsigned)ncw * (unsigned)(k_ / 2);                                                   \
        while (__atomic_load_n(&donec[k_ & 1], __ATOMIC_RELAXED) < wd_) __builtin_amdgcn_s_sleep(1);               \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                                                     \
      }                                                                                                            \
      if (!(lab & 1)) {                                                                                            \
        float scale_, inv_;                                                                                        \
        chunk_scale(__atomic_load_n(&mx[k_ % kSlots], __ATOMIC_RELAXED), scale_, inv_);                                 \
        if (interior) {   /* a lane past the end holds a copy of the last piece and rewrites it */                 \
          int lt_ = lt;                                                                                            \
          asm volatile("" : "+v"(lt_));   /* as in FETCH: nothing of this hoisted out of the chunk loop */          \
          u32x4* dst_ = reinterpret_cast<u32x4*>(buf_);                                                            \
          if (full) {   /* one address register, immediate offsets */                                              \
            u32x4* dl_ = dst_ + lt_;                                                                               \
            _Pragma("unroll") for (int u = 0; u < U; ++u) dl_[64 * kLoaderWaves * u] = pack4(v[u], scale_);        \
          } else {                                                                                                 \
            _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                        \
              const int j = lt_ + 64 * nld * u;                                                                    \
              dst_[j < pieces ? j : pieces - 1] = pack4(v[u], scale_);                                             \
            }                                                                                                      \
          }                                                                                                        \
        } else if (!(lab & 8)) {                                                                                   \
          _Pragma("unroll 4") for (int j = lt; j < pieces; j += 64 * nld) {                                        \
            const F4 t = *reinterpret_cast<const F4*>(buf_ + 4 * j);                                               \
            *reinterpret_cast<u32x4*>(buf_ + 4 * j) = pack4(t, scale_);                                            \
          }                                                                                                        \
        }                                                                                                          \
      }                                                                                                            \
      if (AAMD_RSM_FLAGS) {   /* behind the wave's own LDS writes (the LDS executes a wave's accesses in order) */   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                                                     \
        if (lane == 63) atomicAdd(&fullc[k_ & 1], 1u);                                                             \
      }                                                                                                            \
    }
#endif
    if (first < end) {
      AAMD_RSM_FETCH(first)
      AAMD_RSM_STAGE(first)
    }
    if (first + 1 < end) AAMD_RSM_FETCH(first + 1)
#if defined(AAMD_RSM_SHARED_CONV)
    if (first < end) convert(0);
#endif
    if (!AAMD_RSM_FLAGS) __syncthreads();                // A0
    for (int64_t cid = first; cid < end; ++cid) {
      if (cid + 1 < end) AAMD_RSM_STAGE(cid + 1)          // fetched a whole chunk period ago; its buffer is free since A(cid - 1)
      AAMD_RSM_STAMP((int)(cid + 1 - first), 3)
      if (cid + 2 < end) AAMD_RSM_FETCH(cid + 2)          // stays in registers until the next round
#if defined(AAMD_RSM_SHARED_CONV)
      if (cid + 1 < end) convert((int)(cid + 1 - first));
#endif
      AAMD_RSM_STAMP((int)(cid + 1 - first), 4)
      if (!AAMD_RSM_FLAGS) __syncthreads();              // A(cid)
      AAMD_RSM_STAMP((int)(cid + 1 - first), 5)
    }
#undef AAMD_RSM_FETCH
#undef AAMD_RSM_STAGE
    return;
  }

  uint32_t ah[NS * 4], al[NS * 4];
  if (g.frag != nullptr && !(lab & 16)) {
    // prepared fragments: 2 NS coalesced 16-byte loads per lane (natural step order in the table; the rotated lane groups of the
    // 8-byte layout take the row of THEIR table step)
    const u32x4* ft = reinterpret_cast<const u32x4*>(g.frag);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const int sig = RD ? b64_step(KS, s_, lane >> 4) : s_;
      const u32x4 h = ft[frag_piece(pt, NS, sig, 0, lane)], l = ft[frag_piece(pt, NS, sig, 1, lane)];
#pragma unroll
      for (int d = 0; d < 4; ++d) { ah[4 * s_ + d] = h[d]; al[4 * s_ + d] = l[d]; }
    }
  } else if constexpr (RD == 1) {
    // odd lane groups hold the steps rotated (b64_step): one per-lane tap offset, the wrap a compile-time choice per step.
    // The per-lane base goes through an opaque move every AAMD_RSM_PRO_BATCH steps: left alone, the 224 tap indices are all
    // formed up front, and with the 8-byte loop's registers on top the fragment loads were spilled one by one behind vmcnt(0)
#ifndef AAMD_RSM_PRO_BATCH
#define AAMD_RSM_PRO_BATCH 2
#endif
    int tap_base = tap_lo + KS * (lane >> 4);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      if (s_ % AAMD_RSM_PRO_BATCH == 0) asm volatile("" : "+v"(tap_base));
      if (b64_jump(KS, s_) != 0) tap_base += ((lane >> 4) & 1) * b64_jump(KS, s_);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int i = 4 * s_ + d;
        ah[i] = 0x3c003c00u; al[i] = 0x1c001c00u + i;
        if (!(lab & 16)) a_pack16_at(g, kern, pt, tap_base + 8 * s_ + 2 * d, lane, ah[i], al[i]);
      }
      // the packed fragments of the step exist HERE: the split arithmetic was otherwise sunk behind convert(0)'s wait loop and
      // the barrier, with all 224 raw taps alive across both
      asm volatile("" : "+v"(ah[4 * s_]), "+v"(ah[4 * s_ + 1]), "+v"(ah[4 * s_ + 2]), "+v"(ah[4 * s_ + 3]),
                        "+v"(al[4 * s_]), "+v"(al[4 * s_ + 1]), "+v"(al[4 * s_ + 2]), "+v"(al[4 * s_ + 3]));
    }
  } else {
#pragma unroll
    for (int i = 0; i < NS * 4; ++i) {
      ah[i] = 0x3c003c00u; al[i] = 0x1c001c00u + i;
      if (!(lab & 16)) a_pack16(g, kern, pt, tap_lo, KS, i >> 2, i & 3, lane, ah[i], al[i]);
    }
  }
#if defined(AAMD_RSM_SHARED_CONV)
  if (first < end) convert(0);
#endif
  if (!AAMD_RSM_FLAGS) __syncthreads();                  // A0
  // The scalars of a chunk -- its row, first output group, 16-byte phase, output row, and the scale that undoes the chunk's
  // power-of-two (from the loaders' maximum) -- are ~75 dependent scalar instructions and an LDS round trip.  Round 6: they are
  // worked out for chunk k + 1 IN FRONT of barrier A(k), where eleven of the twelve waves wait anyway; behind the barrier every wave
  // of the workgroup did them at the same moment and no matrix instruction issued for ~0.3 us of a 5.3 us chunk period.  The maximum
  // of chunk k + 1 is final once all loader waves have counted themselves in (cnt: it is, long before; checked, with the
  // acquire fence the barrier used to supply).
  struct ChunkHeader { float inv; int shift; int64_t qc0; float* out_row; };
  auto chunk_header = [&](int64_t cid_) {
    ChunkHeader h;
    const int k_ = (int)(cid_ - first);
    const unsigned want_ = (unsigned)nld * (unsigned)(k_ / kSlots + 1);
    while (__atomic_load_n(&cnt[k_ % kSlots], __ATOMIC_RELAXED) < want_) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float scale_;
    chunk_scale(__atomic_load_n(&mx[k_ % kSlots], __ATOMIC_RELAXED), scale_, h.inv);
    const int64_t row_ = (int64_t)((uint32_t)cid_ / (uint32_t)g.chunks_per_row);   // n_chunks < 2^31 (checked by the launcher)
    h.qc0 = (cid_ - row_ * g.chunks_per_row) * qc;
    h.shift = (int)((h.qc0 * g.orig - g.width) - chunk_a0(g, h.qc0));
    h.out_row = out + row_ * g.out_len;
    return h;
  };
  ChunkHeader hdr{};
  for (int64_t cid = first; cid < end; ++cid) {
    const int k = (int)(cid - first);
    AAMD_RSM_STAMP(k, 0)
#if defined(AAMD_RSM_PRIO_LOOP)
#if AAMD_RSM_PRIO_LOOP == 9                              /* lab: the younger wave of a SIMD first */
    if ((wave >> 2) == 0) __builtin_amdgcn_s_setprio(1); else if ((wave >> 2) == 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
#else
    __builtin_amdgcn_s_setprio(AAMD_RSM_PRIO_LOOP);
#endif
#endif
    const uint32_t* buf = reinterpret_cast<const uint32_t*>(smem_rsm + (k & 1) * g.buf_floats);
#if defined(AAMD_RSM_HEADER_BEHIND_BARRIER)   /* lab: rounds 2-6 -- every wave works out the chunk's scalars right behind the barrier */
    if (true) hdr = chunk_header(cid);
#else
    if (k == 0) hdr = chunk_header(cid);
#endif
    if (AAMD_RSM_FLAGS) {   // the chunk is in its buffer: every loader wave has written its pieces
      const unsigned wf = (unsigned)nld * (unsigned)(k / 2 + 1);
      while (__atomic_load_n(&fullc[k & 1], __ATOMIC_RELAXED) < wf) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const float inv = hdr.inv;
    const int64_t qc0 = hdr.qc0;
    const int shift = hdr.shift;
    float* out_row = hdr.out_row;
#pragma unroll 1
    for (int r = 0; r < g.rounds; ++r) {                 // (indentation of the body kept: one more level would not fit the lines)
    if constexpr (RD == 1) {
      // 8-byte operand reads (see b64_rot): tile A = the parity of q whose addresses are 8-byte aligned for this wave and chunk,
      // tile B the other parity, read from one dword earlier; odd lane groups walk the steps rotated by ROT
      using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
      constexpr int ROT = b64_rot(KS);
      const int grp = qgi + g.qg * r;
      int ln = lane;                                        // lane-derived addresses are formed per round: hoisted out of the chunk loop they were spilled
      asm volatile("" : "+v"(ln));
      const int hA = (tap_lo + shift) & 1, hB = hA ^ 1;
      const int odd_g = (ln >> 4) & 1;
      const uint32_t* pa = buf + b_base64(g, grp, hA, tap_lo, KS, shift, ln);
      const uint32_t* pb = buf + b_base64(g, grp, hB, tap_lo, KS, shift, ln) - 1;
      // steps s >= NS - ROT of the odd lane groups read table step s + ROT - NS: both pointers step back KS dwords there
      // (once per tile and round, formed from the lane number: nothing more to hold across the loop)
      f32x4 accA = {0.0f, 0.0f, 0.0f, 0.0f}, accB = {0.0f, 0.0f, 0.0f, 0.0f};
      // single ds_read_b64 each: left alone, hipcc pairs two of them into a ds_read2_b64 (8 LDS cycles instead of 2 x 2; fftconv_fdr.h)
#define AAMD_RSM_PAIRS(P, N, R)                                                                                    \
      _Pragma("unroll") for (int j = 0; j < (N); ++j) {                                                            \
        const u32x2 t_ = *reinterpret_cast<const u32x2*>((P) + 2 * j);                                             \
        asm volatile("" ::: "memory");                                                                             \
        R[2 * j] = t_.x; R[2 * j + 1] = t_.y;                                                                      \
      }
#define AAMD_RSM_READ_A(S, R) { if (b64_jump(KS, S) != 0) pa += odd_g * b64_jump(KS, S); AAMD_RSM_PAIRS(pa + 8 * (S), 4, R) }
      // tile B: its dwords 0 .. 7 of a step are (carry, pairs (1, 2) (3, 4) (5, 6), first half of (7, 8)) and the second half of
      // (7, 8) is dword 0 of the NEXT table step: four pairs per step, every register used (five pairs with two half-used
      // ones were narrowed to ds_read_b32 by the compiler: two-way bank conflicts in this layout).  Dword 0 is read on its own
      // where a run of table steps starts: step 0, and step NS - ROT where the odd lane groups wrap to table step 0
#define AAMD_RSM_READ_B(S, R)                                                                                      \
      {                                                                                                            \
        if (b64_jump(KS, S) != 0) pb += odd_g * b64_jump(KS, S);                                                   \
        if (b64_run_start(KS, S)) { cf = pb[8 * (S) + 1]; asm volatile("" ::: "memory"); }                         \
        AAMD_RSM_PAIRS(pb + 8 * (S) + 2, 4, R)                                                                     \
      }
#if defined(AAMD_RSM_LAB_NOMFMA)   /* lab, timing only (wrong results): the loop without its matrix instructions (operands kept alive) */
#define AAMD_RSM_MFMA3(ACC, HV, LV) asm volatile("" :: "v"(HV), "v"(LV), "v"(ahv), "v"(alv));
#else
#define AAMD_RSM_MFMA3(ACC, HV, LV)                                                                                \
      if (!((AAMD_RSM_LAB_T1MASK >> s) & 1)) {                                                                     \
      ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, __builtin_bit_cast(h8, HV), ACC, 0, 0, 0);                 \
      ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, __builtin_bit_cast(h8, LV), ACC, 0, 0, 0);                 \
      }                                                                                                            \
      ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, __builtin_bit_cast(h8, HV), ACC, 0, 0, 0);
#endif
#ifndef AAMD_RSM_LAB_T1MASK      /* lab, timing only (wrong results): steps that issue hi x hi alone */
#define AAMD_RSM_LAB_T1MASK 0
#endif
#ifndef AAMD_RSM_ONE_WAIT
#define AAMD_RSM_ONE_WAIT 1
#endif
      // the eight regrouping instructions of a tile in front of its three matrix instructions (the scheduler put two of them behind
      // the first MFMA and paid an s_nop for the hazard in front of the second)
#if defined(AAMD_RSM_NO_ORDER) || defined(AAMD_RSM_LAB_NOPERM) || defined(AAMD_RSM_LAB_NOMFMA)
#define AAMD_RSM_ORDER_PERM_MFMA
#else
#define AAMD_RSM_ORDER_PERM_MFMA if (!((AAMD_RSM_LAB_T1MASK >> s) & 1)) { __builtin_amdgcn_sched_group_barrier(0x002, 8, 0); __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); }
#endif
#if defined(AAMD_RSM_LAB_NOREAD)   /* lab, timing only (wrong results): operand tiles read once per round, re-used (opaquely) by every step */
      constexpr bool kNoRead = true;
#else
      constexpr bool kNoRead = false;
#endif
      // operand tiles one after the other as in the 4-byte loop (A(s), B(s), A(s + 1), ...), the reads of the tile after the
      // next in flight: 3 x 8 registers + the carry
      uint32_t ra[2][8], rb[2][8], cf = 0u;
#if defined(AAMD_RSM_DEEP_A)
      uint32_t ra3[3][8];
#define AAMD_RSM_RA_CUR ra3[s % 3]
      AAMD_RSM_READ_A(0, ra3[0])
#else
#define AAMD_RSM_RA_CUR ra[cur]
      AAMD_RSM_READ_A(0, ra[0])
#endif
      AAMD_RSM_READ_B(0, rb[0])
      __builtin_amdgcn_sched_barrier(0);
#if defined(AAMD_RSM_B64_INTERLEAVE)
      // lab: both tiles of a step regrouped first, their six MFMAs interleaved (no MFMA waits for its predecessor's accumulator)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const h8 ahv = __builtin_bit_cast(h8, u32x4{ah[4 * s], ah[4 * s + 1], ah[4 * s + 2], ah[4 * s + 3]});
        const h8 alv = __builtin_bit_cast(h8, u32x4{al[4 * s], al[4 * s + 1], al[4 * s + 2], al[4 * s + 3]});
        const uint32_t d0 = b64_run_start(KS, s) ? cf : rb[(s + 1) & 1][7];
        u32x4 hv, lv, hw, lw;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          hv[d] = __builtin_amdgcn_perm(ra[s & 1][2 * d + 1], ra[s & 1][2 * d], 0x05040100u);
          lv[d] = __builtin_amdgcn_perm(ra[s & 1][2 * d + 1], ra[s & 1][2 * d], 0x07060302u);
        }
        hw[0] = __builtin_amdgcn_perm(rb[s & 1][0], d0, 0x05040100u);
        lw[0] = __builtin_amdgcn_perm(rb[s & 1][0], d0, 0x07060302u);
#pragma unroll
        for (int d = 1; d < 4; ++d) {
          hw[d] = __builtin_amdgcn_perm(rb[s & 1][2 * d], rb[s & 1][2 * d - 1], 0x05040100u);
          lw[d] = __builtin_amdgcn_perm(rb[s & 1][2 * d], rb[s & 1][2 * d - 1], 0x07060302u);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < NS) { AAMD_RSM_READ_A(s + 1, ra[(s + 1) & 1]) AAMD_RSM_READ_B(s + 1, rb[(s + 1) & 1]) }
        __builtin_amdgcn_sched_barrier(0);
        accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, __builtin_bit_cast(h8, hv), accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, __builtin_bit_cast(h8, hw), accB, 0, 0, 0);
        accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, __builtin_bit_cast(h8, lv), accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, __builtin_bit_cast(h8, lw), accB, 0, 0, 0);
        accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, __builtin_bit_cast(h8, hv), accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, __builtin_bit_cast(h8, hw), accB, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int cur = kNoRead ? 0 : (s & 1);
        const h8 ahv = __builtin_bit_cast(h8, u32x4{ah[4 * s], ah[4 * s + 1], ah[4 * s + 2], ah[4 * s + 3]});
        const h8 alv = __builtin_bit_cast(h8, u32x4{al[4 * s], al[4 * s + 1], al[4 * s + 2], al[4 * s + 3]});
#if defined(AAMD_RSM_DEEP_A)   /* lab: tile A's pairs requested two steps ahead (three register sets) */
        if (s == 0 && NS > 1) AAMD_RSM_READ_A(1, ra3[1])
        if (s + 2 < NS) AAMD_RSM_READ_A(s + 2, ra3[(s + 2) % 3])
#else
        if (s + 1 < NS && !kNoRead) AAMD_RSM_READ_A(s + 1, ra[(s + 1) & 1])
#endif
        if (kNoRead) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { asm volatile("" : "+v"(ra[0][j])); asm volatile("" : "+v"(rb[0][j])); }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ONE wait for the four pairs of the tile (left alone, hipcc waits in front of each regrouping instruction: lgkmcnt(11),
        // (10), (9), (8) -- four issue slots of a loop that is bound by its issue slots: 19.5 instructions around 3 MFMAs of 4 slots
        // each, profiles/r06_zd); where the carried dword's read is among the outstanding ones the compiler's own waits stand
        // (outstanding behind tile A(s): B(s) -- with its carried dword where a run starts -- and A(s + 1))
        if (AAMD_RSM_ONE_WAIT && s + 1 < NS) {
          if (b64_run_start(KS, s)) __builtin_amdgcn_s_waitcnt(0xC07F | (9 << 8)); else __builtin_amdgcn_s_waitcnt(0xC07F | (8 << 8));
          __builtin_amdgcn_sched_barrier(0);
        }
        u32x4 hv, lv;
#if defined(AAMD_RSM_LAB_NOPERM)   /* lab, timing only (wrong results): what the 16 v_perm_b32 per step cost */
#pragma unroll
        for (int d = 0; d < 4; ++d) { hv[d] = AAMD_RSM_RA_CUR[2 * d]; lv[d] = AAMD_RSM_RA_CUR[2 * d + 1]; }
#else
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          hv[d] = __builtin_amdgcn_perm(AAMD_RSM_RA_CUR[2 * d + 1], AAMD_RSM_RA_CUR[2 * d], 0x05040100u);
          lv[d] = __builtin_amdgcn_perm(AAMD_RSM_RA_CUR[2 * d + 1], AAMD_RSM_RA_CUR[2 * d], 0x07060302u);
        }
#endif
        AAMD_RSM_MFMA3(accA, hv, lv)
        AAMD_RSM_ORDER_PERM_MFMA
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t d0 = b64_run_start(KS, s) ? cf : rb[kNoRead ? 0 : ((s + 1) & 1)][7];     // (before the next reads land there)
        if (s + 1 < NS && !kNoRead) AAMD_RSM_READ_B(s + 1, rb[(s + 1) & 1])
        __builtin_amdgcn_sched_barrier(0);
        // (behind tile B(s): A(s + 1) and B(s + 1))
        if (AAMD_RSM_ONE_WAIT && s + 1 < NS) {
          if (b64_run_start(KS, s + 1)) __builtin_amdgcn_s_waitcnt(0xC07F | (9 << 8)); else __builtin_amdgcn_s_waitcnt(0xC07F | (8 << 8));
          __builtin_amdgcn_sched_barrier(0);
        }
#if defined(AAMD_RSM_LAB_NOPERM)
        hv[0] = d0; lv[0] = rb[cur][0];
#pragma unroll
        for (int d = 1; d < 4; ++d) { hv[d] = rb[cur][2 * d - 1]; lv[d] = rb[cur][2 * d]; }
#else
        hv[0] = __builtin_amdgcn_perm(rb[cur][0], d0, 0x05040100u);
        lv[0] = __builtin_amdgcn_perm(rb[cur][0], d0, 0x07060302u);
#pragma unroll
        for (int d = 1; d < 4; ++d) {
          hv[d] = __builtin_amdgcn_perm(rb[cur][2 * d], rb[cur][2 * d - 1], 0x05040100u);
          lv[d] = __builtin_amdgcn_perm(rb[cur][2 * d], rb[cur][2 * d - 1], 0x07060302u);
        }
#endif
        AAMD_RSM_MFMA3(accB, hv, lv)
        AAMD_RSM_ORDER_PERM_MFMA
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
#undef AAMD_RSM_MFMA3
#undef AAMD_RSM_RA_CUR
#undef AAMD_RSM_ORDER_PERM_MFMA
#undef AAMD_RSM_READ_A
#undef AAMD_RSM_READ_B
#undef AAMD_RSM_PAIRS
      asm volatile("" : "+v"(accA), "+v"(accB));
      if (!(lab & 32) || accA[0] == 12345.0f) {
        const int64_t q0 = qc0 + 32 * grp + 2 * (ln & 15);
        store_c_at(g, out_row, q0 + hA, pt, ln, accA[0] * inv, accA[1] * inv, accA[2] * inv, accA[3] * inv);
        store_c_at(g, out_row, q0 + hB, pt, ln, accB[0] * inv, accB[1] * inv, accB[2] * inv, accB[3] * inv);
      }
      continue;
    }
    const int qt0 = 2 * (qgi + g.qg * r), qt1 = qt0 + 1;
    const uint32_t* b0 = buf + b_base(g, qt0, tap_lo, KS, shift, lane);
    const uint32_t* b1 = buf + b_base(g, qt1, tap_lo, KS, shift, lane);
    f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    // 2 NS operand tiles (step s = t >> 1, q-tile t & 1), one after the other (both tiles' operands at once do not fit the
    // 168 registers beside the 112 of the taps).  The 8 packed dwords of tile t + 2 are requested before tile t is
    // regrouped and multiplied: with three waves per SIMD a look-ahead of ONE tile (what the scheduler does on its own)
    // is shorter than the LDS round trip and every tile stalled on its reads (round-2 census, profiles/r02_v).
    uint32_t raw[3][8];
#define AAMD_RSM_READ(T, R)                                                                                        \
    {                                                                                                              \
      const uint32_t* bp_ = ((T) & 1) ? b1 : b0;                                                                   \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                              \
        R[j] = 0x3c003c00u + j;                                                                                    \
        if (!(LABM == 2 && (lab & 4))) R[j] = bp_[8 * ((T) >> 1) + j];                                                  \
      }                                                                                                            \
    }
    AAMD_RSM_READ(0, raw[0])
    AAMD_RSM_READ(1, raw[1])
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2 * NS; ++t) {
      const int s = t >> 1;
      const h8 ahv = __builtin_bit_cast(h8, u32x4{ah[4 * s], ah[4 * s + 1], ah[4 * s + 2], ah[4 * s + 3]});
      const h8 alv = __builtin_bit_cast(h8, u32x4{al[4 * s], al[4 * s + 1], al[4 * s + 2], al[4 * s + 3]});
      if (t + 2 < 2 * NS) AAMD_RSM_READ(t + 2, raw[(t + 2) % 3])
      __builtin_amdgcn_sched_barrier(0);                 // (sched_group_barrier pipelines picked the reads of tile t itself)
      u32x4 hv, lv;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        hv[d] = __builtin_amdgcn_perm(raw[t % 3][2 * d + 1], raw[t % 3][2 * d], 0x05040100u);
        lv[d] = __builtin_amdgcn_perm(raw[t % 3][2 * d + 1], raw[t % 3][2 * d], 0x07060302u);
      }
      f32x4& acc = (t & 1) ? acc1 : acc0;
      if (LABM == 2 && (lab & 2)) {
        acc[0] += __uint_as_float(hv[0] ^ lv[1]); acc[1] += __uint_as_float(hv[2] ^ lv[3]);
      } else {                                          // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(alv, __builtin_bit_cast(h8, hv), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, __builtin_bit_cast(h8, lv), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahv, __builtin_bit_cast(h8, hv), acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef AAMD_RSM_READ
    asm volatile("" : "+v"(acc0), "+v"(acc1));
    if (!(lab & 32) || acc0[0] == 12345.0f) {
      store_c(g, out_row, qc0, qt0, pt, lane, acc0[0] * inv, acc0[1] * inv, acc0[2] * inv, acc0[3] * inv);
      store_c(g, out_row, qc0, qt1, pt, lane, acc1[0] * inv, acc1[1] * inv, acc1[2] * inv, acc1[3] * inv);
    }
    }
    if (AAMD_RSM_FLAGS) {   // this wave's operand reads of the chunk have all returned (the matrix instructions used them)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) atomicAdd(&donec[k & 1], 1u);
    }
    AAMD_RSM_STAMP(k, 1)                                 // (the MFMA loops and the stores of all rounds)
    AAMD_RSM_STAMP(k, 2)
#if defined(AAMD_RSM_SHARED_CONV)
#if defined(AAMD_RSM_CONV_SIMD23)                         /* lab: the waves of SIMDs 0 / 1 (three MFMA loops each) leave the conversion to the others */
    if (cid + 1 < end && (wave & 3) >= 2) convert(k + 1);
#else
    if (cid + 1 < end) convert(k + 1);                   // (stamp 3 inside: both loaders have arrived)
#endif
#else
    AAMD_RSM_STAMP(k, 3)
#endif
    AAMD_RSM_STAMP(k, 4)
#if !defined(AAMD_RSM_HEADER_BEHIND_BARRIER)
    if (cid + 1 < end) hdr = chunk_header(cid + 1);
#endif
    if (!AAMD_RSM_FLAGS) __syncthreads();                // A(cid): chunk cid is consumed, chunk cid + 1 is in LDS
    AAMD_RSM_STAMP(k, 5)
    if (!AAMD_RSM_FLAGS && threadIdx.x == 0) { mx[k % kSlots] = 0u; grab[k % 3] = 0u; }   // read by everybody before A(cid); next written behind A(cid + 1)
  }
}
// the prepared tap fragments of the phase tiles [g.pt0, g.pt0 + g.n_pt) of one filter: one thread per (tile, step, lane)
__global__ void __launch_bounds__(256)
frag_build_kernel(Geom g, int ks, const float* __restrict__ kern, uint32_t* __restrict__ frag) {
  using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
  const int ns = ks / 8;
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= g.n_pt * ns * 64) return;
  const int lane = i & 63, s = (i >> 6) % ns, pt_l = (i >> 6) / ns, pt = g.pt0 + pt_l;
  u32x4 h, l;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t hi, lo;
    a_pack16(g, kern, pt, g.tap_lo[pt_l], ks, s, d, lane, hi, lo);
    h[d] = hi; l[d] = lo;
  }
  u32x4* ft = reinterpret_cast<u32x4*>(frag);
  ft[frag_piece(pt, ns, s, 0, lane)] = h;
  ft[frag_piece(pt, ns, s, 1, lane)] = l;
}

// the instantiation with 8-byte operand reads where one exists (the launcher asks for it only for those KS)
template <int KS, int LABM, int FULL = 0>
inline auto kernel_rd64() {
  if constexpr (b64_rot(KS) != 0) return resample_f16_kernel<KS, LABM, 1, FULL>;
  else return resample_f16_kernel<KS, LABM, 0, 0>;
}
// does the launch's chunk qualify for the FULL instantiations? (plan_chunk pads exactly then)
AAMD_HD bool chunk_is_full(const Geom& g, int ks) {
  return ks >= 80 && g.n_loaders == kLoaderWaves && g.buf_floats == 4 * 64 * kLoaderWaves * loader_pieces_per_lane(ks);
}
#endif  // __HIPCC__

}  // namespace rsm
}  // namespace aamd

// Headline kernel: fused MelSpectrogram for n_fft = 400, hop = 160 (the RNN-T / Whisper
// style front-end, BASELINE.json config 2), power = 2, centre + reflect padding.  The same kernel is
// instantiated for hop = 100 and 200 (template parameter H = hop / 20, struct Hop<H>); the text below
// describes H = 8.
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
//     padded weights) lives in LDS, read as b64 (2 weights) + b128 (2 bins x 2 frames);
//   * HBM traffic is all 16 B per lane: the waveform tile (1200 contiguous samples) is copied
//     global -> LDS by LDS-DMA one tile ahead (no VGPR round trip, overlaps the second half of
//     the previous tile), and the tile's 6 x n_mels outputs (contiguous in memory) leave through
//     an LDS transposition as dwordx4 stores.
//
// Reference semantics: transforms/_transforms.py:612-622 (MelSpectrogram.forward),
// functional/functional.py:112-145, torch/functional.py:675-681; framing is bit-exact
// (index reflection i<0 -> -i, i>=L -> 2(L-1)-i).
#pragma once
#include <type_traits>
#include "hd.h"
#include "stft_generic.h"
#include "resample_mfma.h"   // rsm::f16_bits / f16_value: binary16 conversion shared with the CPU replay

namespace aamd {
namespace m400 {

constexpr int kN = 400;
constexpr int kPad = 200;
constexpr int kFramesPerWave = 6;
#ifndef AAMD_M400_WAVES
#define AAMD_M400_WAVES 12
#define AAMD_M400_MINWAVES 3
#endif
constexpr int kWavesPerBlock = AAMD_M400_WAVES;              // 768 threads = 3 waves per SIMD: ONE block per CU (a second block of a
                                               // smaller size is not admitted: its waves land on the same SIMDs)
// LDS strides picked with tools/lds_conflicts.py (bank model of MI355X_MICROARCH.md):
// Transposition buffer, round-2 layout: one ROW per transposed value and one COLUMN per writer lane.
//   row (s, re) at dword kTOff[s], row (s, im) at kTOff[s] + 64; element of writer lane l at row + l.
// A wave writes a value with ds_write_addtid_b32 (LDS address = M0 + offset + 4 * lane: no address VGPR, 2 LDS cycles per
// wave-instruction against 6 for the ds_write_b64 of the first layout: 80 instead of 120 cycles per tile), and the reader
// lane (pair p, column c) fetches the 20 values of its column from lanes 20 p .. 20 p + 19 as 5 + 5 ds_read_b128 at
// kTOff[c] + 20 p (+ 64).  The row starts are 64 * rank + 4 * a(c) with slots a(c) found by tools/lds_conflicts.py-style
// search so that every b128 read of a wave touches 16 distinct 4-bank slots in each of its lane groups (40 cycles for
// the 10 reads = the conflict-free minimum; a uniform row stride cannot do better than 90).
constexpr int kTOff[20] = {0, 260, 520, 652, 784, 916, 1444, 1576, 128, 388,
                           2104, 2232, 2364, 1708, 1840, 1048, 1180, 1312, 1972, 2492};
constexpr int kTBufDwords = 2492 + 128;        // 2620
constexpr int kTRow = 44;                      // (first layout; still sizes the wave's region: 3 * 20 * 44 = 2640 >= 2620)
                                               // rows -> conflict-free ds_read_b128; column writes 6 array
                                               // cycles = the ds_write_b64 issue cost
constexpr int kTPair = 20 * kTRow;             // 880 dwords per pair
constexpr int kLdsDwordsPerWave = 3 * kTPair;  // 2640 dwords = 10560 B (the P rows alias it)
constexpr int kPK = 208;                       // readable bins per P row: 201 + zeroed tail, even
constexpr int kPPair = 416;                    // dwords per pair of P rows (>= 2 * kPK); = 32 mod 64, so the b128
                                               // band reads of neighbouring pairs land on opposite bank halves
constexpr int kSOff = 1344;                    // staging area for the NEXT tile's samples (LDS-DMA)

// Everything that depends on the hop.  The kernel serves hop = 20 H for H = 5, 8, 10 (hop 100 = n_fft/4,
// 160 = the 10 ms ASR hop of the headline config, 200 = torchaudio's default n_fft/2): frame b of a pair
// starts H x 20 samples after frame a, so a lane gathers 20 + H strided samples for the pair.
template <int H>
struct Hop {
  static constexpr int hop = 20 * H;
  static constexpr int nx = 20 + H;                             // samples gathered per lane and pair
  static constexpr int tile_samples = (kFramesPerWave - 1) * hop + kN;   // 900 / 1200 / 1400
  static constexpr int pair_stride = 2 * hop;                   // samples between the pairs of a tile
  // staging layout: `pad` dwords after every pair_stride samples put the three pairs' strided rows on
  // disjoint banks: (pair_stride + pad) = 20 (mod 32); always a multiple of 4 = whole 16-B pieces
  static constexpr int pad = ((20 - pair_stride % 32) + 32) % 32;
  static constexpr int blk_data = pair_stride / 4, blk_pieces = blk_data + pad / 4;
  static constexpr int data_pieces = tile_samples / 4;
  static constexpr int pieces = data_pieces + (pad / 4) * ((tile_samples - 1) / pair_stride);
  static constexpr int ndma = (pieces + 63) / 64;               // LDS-DMA instructions per tile: 4 / 5 / 6
  static constexpr int lds_dwords = (kSOff + 256 * ndma > kLdsDwordsPerWave) ? kSOff + 256 * ndma : kLdsDwordsPerWave;
  static_assert(pad % 4 == 0 && (pair_stride + pad) % 32 == 20, "staging pad");
};
// Staging layout of the NEXT tile's samples for an input element type TIn (float, or int16 PCM: SURVEY 8(f) rank 4,
// the upstream half -- the waveform is read as the 16-bit samples the decoder produced and converted in the gather, so
// the separate int16 -> float pass and half of the input bytes disappear).  16-B pieces hold 4 or 8 samples; `pad`
// dwords after every pair_stride samples put the three pairs' strided rows on disjoint banks:
//   float: (pair dwords + pad) = 20 (mod 32);  int16: two lanes share a dword, a pair's row spans 10 dwords, and
//   (pair dwords + pad) = 44 (mod 64) puts the three rows at banks 0.., 44.., 24..
// Input element types.  float and int16_t are planar rows (one channel per row).  PcmStereo is one SAMPLE TIME of
// interleaved 16-bit stereo, (L, R) in one dword -- what a decoder hands over before the reference transposes it to
// (channel, time) (torchaudio/_torchcodec.py:150-152): the kernel's rows are then (clip, channel) pairs, both rows of a clip
// stage the same dwords (the float layout: one dword per sample time, conflict-free as for float) and the gather
// takes the row's half-word.  Channel de-interleave, int16 -> float and the 1/32768 (folded into the window) cost nothing
// beyond the conversion the planar int16 path already does.
struct PcmStereo { uint32_t lr; };
template <typename TIn> struct InTraits {
  static constexpr int chans = 1;
  AAMD_HD static float get(const TIn* p, int) { return (float)*p; }
};
template <> struct InTraits<PcmStereo> {
  static constexpr int chans = 2;
  AAMD_HD static float get(const PcmStereo* p, int ch) { return (float)(int16_t)(uint16_t)(p->lr >> (16 * ch)); }
};

template <int H, typename TIn>
struct Stage {
  using HC = Hop<H>;
  static constexpr int spp = 16 / (int)sizeof(TIn);                                  // samples per piece
  static constexpr int pair_dwords = HC::pair_stride * (int)sizeof(TIn) / 4;
  static constexpr int pad = sizeof(TIn) == 4 ? HC::pad : ((44 - pair_dwords % 64) + 64) % 64;   // dwords
  static constexpr int pad_elems = pad * 4 / (int)sizeof(TIn);
  static constexpr int blk_data = HC::pair_stride / spp, blk_pieces = blk_data + pad / 4;
  static constexpr int data_pieces = HC::tile_samples / spp;
  static constexpr int pieces = data_pieces + (pad / 4) * ((HC::tile_samples - 1) / HC::pair_stride);
  static constexpr int ndma = (pieces + 63) / 64;
  static constexpr bool ok = (HC::pair_stride % spp == 0) && (HC::tile_samples % spp == 0) && (pad % 4 == 0) &&
                             (kSOff + 256 * ndma <= HC::lds_dwords);
};
static_assert(Stage<8, float>::blk_data == Hop<8>::blk_data && Stage<8, float>::blk_pieces == Hop<8>::blk_pieces &&
              Stage<8, float>::pieces == Hop<8>::pieces && Stage<8, float>::ndma == Hop<8>::ndma, "float staging = Hop");
static_assert(Stage<8, int16_t>::ok && Stage<8, int16_t>::pad == 12 && Stage<8, int16_t>::ndma == 3, "int16 staging, hop 160");
static_assert(Stage<10, int16_t>::ok && Stage<5, float>::ok && Stage<10, float>::ok, "staging geometry");
static_assert(sizeof(PcmStereo) == 4 && Stage<8, PcmStereo>::ok && Stage<10, PcmStereo>::ok &&
              Stage<8, PcmStereo>::pieces == Stage<8, float>::pieces, "interleaved stereo stages like float");

constexpr int kMelSlots = 20;                  // mels per round
constexpr int kMelMaxRounds = 8;               // n_mels <= 160
constexpr int kMelMaxTaps = 64;                // widest padded band (taps)

static_assert(3 * kPPair <= kLdsDwordsPerWave && 2 * kPK <= kPPair, "P rows must fit in the transposition buffer");
static_assert(kTBufDwords <= kLdsDwordsPerWave, "transposition rows must fit in the wave's region");
static_assert(3 * (kPK - 201) <= 64, "one lane per tail bin");
static_assert(Hop<8>::lds_dwords == kLdsDwordsPerWave && Hop<8>::ndma == 5 && Hop<8>::pad == 20, "headline geometry");

// Epilogues of the tile loop (compile-time variants of the same kernel):
//   EPI400_MEL     banded mel of |X|^2                       -> out[rows][T][n_mels]
//   EPI400_MEL_DB  ... then amplitude_to_DB (functional.py:390-391) fused, plus the running
//                  maximum per cut-off group (the top_db reduction of :393-402)
//   EPI400_SPEC    no mel: |X|^power for the 201 one-sided bins -> out[rows][T][201]
//   EPI400_MEL_NORM  the RNN-T front-end's feature post-processing fused (pipelines/rnnt_pipeline.py:16-47, 319-326):
//                  y = piecewise_linear_log(mel * gain), (y - mean[m]) * invstddev[m], rows of out_frames >= T frames
//   EPI400_MFCC    MEL_DB + the DCT-II product on the matrix cores: out[rows][T][n_mfcc] WITHOUT the top_db cut-off, the
//                  group maximum and the minimum of every tile's dB values on the side; a second launch (`fixup`) redoes
//                  only the tiles whose minimum lies under the cut-off, clamped (transforms/_transforms.py:692-709)
enum { EPI400_MEL = 0, EPI400_MEL_DB = 1, EPI400_SPEC = 2, EPI400_MEL_NORM = 3, EPI400_MFCC = 4 };
struct Epi400 {
  float multiplier, amin, db_sub;   // MEL_DB: y = multiplier * log10(max(x, amin)) - db_sub
  float* group_max;                 // MEL_DB: [n_groups] running max of y (float bit pattern), may be null
  int64_t rows_per_group;           // MEL_DB: waveform rows per cut-off group
  float power;                      // SPEC: 2 -> |X|^2, 1 -> |X|, else |X|^power
  float gain;                       // MEL_NORM: y = plog(mel * gain)
  const float* mean;                // MEL_NORM: [n_mels]
  const float* invstd;              // MEL_NORM: [n_mels]
  int64_t out_frames;               // MEL_NORM: frames per clip in `out` (>= n_frames; the tail is the caller's)
  const float* dct_frag;            // MFCC: the DCT matrix as binary16 hi / lo MFMA A fragments (mfcc_frag_piece), [kMfccFragFloats]
  int n_mfcc;                       // MFCC: coefficients (<= 48, multiple of 4)
  float top_db;                     // MFCC fix-up: cut-off = group_max[g] - top_db
  float* tile_min;                  // MFCC: [n_tiles] minimum dB value of each tile (written by pass 0, read by the fix-up)
  int* fix_count;                   // MFCC: number of tiles the fix-up pass redoes (pass 0 zeroes it, the fix-up workgroups add theirs)
  int* fix_list;                    // MFCC fix-up: [n_tiles] scratch -- every workgroup keeps the list of ITS flagged tiles in a
                                    //   private run of it (any order)
  int frag_in_lds;                  // MFCC: the workgroup's LDS has room for the fragment table (hop 100 / 160; not hop 200)
  int fixup;                        // MFCC: 0 = first pass, 1 = fix-up pass
  int lab;                          // MFCC (tools only): 1 no fragment loads, 2 no MFMA, 4 no tile minimum, 8 no stores
  unsigned* pool;                   // tail pools (round 5; null = static runs only): one ticket counter per pool, kPoolStride dwords apart
  int pool_p;                       // tiles every workgroup leaves in its pool (0 < pool_p <= tiles_per_block when pool != null)
};
constexpr int kSpecBins = 201;

// ---- tail pools (round 5) ----------------------------------------------------------------------------------------------
// A workgroup owns a contiguous run of tiles_per_block tiles and its waves claim them from an LDS queue; the launch ends when
// the SLOWEST workgroup has finished its run (round 3: workgroups finish 63.9 .. 71.4 us, the XCDs differ by up to 7 % and
// which one is slow changes from box to box).  So every workgroup leaves the last P tiles of its run in a POOL that it shares
// with the workgroups of the other XCDs (pool p = workgroups p, p + NP, p + 2 NP, ...: with the XCD-aware numbering one per
// XCD): a wave that finds its workgroup's queue empty takes tickets from the pool's counter in memory (one device-scope
// atomic per tile, requested a phase ahead) until a ticket lies past the pool's tiles.  Every wave of every member ends with
// exactly ONE such ticket, so a launch draws exactly members * (P + waves) tickets from a pool: the wave that draws the last one
// puts the counter back to zero for the next launch on the stream (no memset, no second counter; the host hands out one
// counter block per stream).  Which wave computes a tile never changes the tile's result: outputs are bit-identical.
#ifndef AAMD_M400_POOLS
#define AAMD_M400_POOLS 0      /* built, verified bit-identical, measured - 0.8 % on the headline batch (profiles/r05_i_mel400_pool_sweep.txt): the
                                  tile loop's part is compiled out of the product; -DAAMD_M400_POOLS=1 builds it */
#endif
constexpr int kPoolStride = 64;                               // dwords between two pools' counters (256 B)
constexpr unsigned kNoTile = 0xffffffffu;
AAMD_HD int pool_count(int nb) { return nb >= 8 ? nb / 8 : 1; }                       // NP
AAMD_HD int pool_members(int nb, int np, int p) { return p < nb ? (nb - p + np - 1) / np : 0; }
// tiles a workgroup hands to its pool; 0 = no pools for this launch
// (measured on the cfg2 / Spectrogram / cfg4 batches, profiles/r05_i_mel400_pool_sweep.txt: 2, 4 and 8 tiles per workgroup all
// give - 0.5 .. - 1.3 us per launch, 24 nothing, 40 costs 0.8 us: the tickets are device-scope atomics)
AAMD_HD int pool_share(int tiles_per_block) { const int p = tiles_per_block / 20; return tiles_per_block < 48 ? 0 : p > 8 ? 8 : p; }
// ticket -> global tile; kNoTile with exhausted = true behind the pool's last tile, kNoTile with exhausted = false for a slot
// of the last workgroup's run that lies past the end of the batch (the caller draws again)
AAMD_HD unsigned pool_tile(int np, int P, int tiles_per_block, unsigned n_tiles, int p, int members, unsigned ticket,
                           bool& exhausted) {
  exhausted = ticket >= (unsigned)(members * P);
  if (exhausted) return kNoTile;
  const unsigned m = ticket / (unsigned)P, off = ticket - m * (unsigned)P;
  const unsigned long long t = (unsigned long long)(m * (unsigned)np + (unsigned)p) * (unsigned)tiles_per_block +
                               (unsigned)(tiles_per_block - P) + off;
  return t < n_tiles ? (unsigned)t : kNoTile;
}
static_assert(kFramesPerWave * kSpecBins + 3 <= kSOff && 3 * 2 * kSpecBins + 3 <= kSOff, "SPEC rows must fit below the staging area");

// column held by the lane at position pi of a 20-lane group (pass 2), and its inverse:
// positions (0,1) = columns (0,10), then (2j, 2j+1) = (j, 20-j): lane ^ 1 holds column 20 - c.
AAMD_HD constexpr int col_of_pos(int pi) {
  return pi == 0 ? 0 : pi == 1 ? 10 : (pi & 1) ? 20 - (pi >> 1) : (pi >> 1);
}
AAMD_HD constexpr int pos_of_col(int c) {
  return c == 0 ? 0 : c == 10 ? 1 : c < 10 ? 2 * c : 2 * (20 - c) + 1;
}

// ---- MFCC epilogue: the DCT product on the f16 matrix pipe, operands split into two binary16 numbers ----------------------
//   out[f][c] = sum_m Y[f][m] D[m][c]:  A = D^T tile (16 coefficients x K mels), B = Y^T (K mels x 16 frames, 6 live),
//   C row 4 (l >> 4) + r = coefficient, column l & 15 = frame: a lane ends with 4 consecutive coefficients of one frame.
//   Round 2 ran this on v_mfma_f32_16x16x4_f32 (exact fp32 chains): 60 instructions per tile that hold the SIMD for 32
//   cycles each IN SERIES with the kernel's fp32 VALU work -- 81 us on the cfg4 batch, which made the one-kernel MFCC slower
//   than two kernels.  Now (round 3) every operand is v = hi + lo, hi = f16(v), lo = f16(v - hi) (22 significant bits), and
//   a product of sums is hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 / 16x16x16_f16 with fp32 accumulation (the
//   dropped lo*lo is 2^-22 of the product): 80 mels = two K = 32 steps + one K = 16 step, 3 coefficient tiles, 3 terms =
//   27 instructions of 8-16 cycles on a pipe the rest of the kernel does not use.
//   Round 6: 18 instructions -- the six live frames fill 6 of the 16 B columns, so the hi plane of the dB rows goes to
//   columns 0 .. 5 and the lo plane to columns 8 .. 13 of ONE operand; A_lo x B and A_hi x B leave (A_hi + A_lo) B_hi in lane j
//   and (A_hi + A_lo) B_lo in lane j + 8 of every row of 16 lanes (all four terms) and one DPP add per accumulator register
//   joins them.  Ranges: dB values are scaled by 2^-4 (|y| < 770 for any float32 power and multiplier <= 20 -> < 48; the lo
//   part of a typical 50 dB value is a NORMAL binary16 number -- the 2^-8 of rounds 3-5 left it subnormal, 2.4e-5 dB of
//   absolute error, now ~4e-6), DCT weights (|d| <= 0.16 for "ortho", <= 2 unnormalised) by 2^4 (their lo parts normal as
//   well): the product needs no scaling on the way out.  n_mels = 80 exactly (every K slot is a real mel), n_mfcc <= 48.
constexpr int kMfccMT = 3, kMfccMels = 80, kMfccSteps = 3;
constexpr float kMfccYScale = 1.0f / 16.0f, kMfccDScale = 16.0f;
static_assert(kMfccYScale * kMfccDScale == 1.0f, "the accumulators leave the matrix pipe unscaled");
// fragment table: [t][s][hi / lo][lane][8 halves] as 16-byte pieces -> floats
constexpr int kMfccFragFloats = kMfccMT * kMfccSteps * 2 * 64 * 4;      // 4608 floats = 18 432 B
AAMD_HD int mfcc_frag_piece(int t, int s, int hl, int lane) { return ((t * kMfccSteps + s) * 2 + hl) * 64 + lane; }   // 16-B pieces
// mel of contraction slot j of lane `lane` in step s (steps 0, 1: K = 32, 8 per lane; step 2: K = 16, 4 per lane), -1 = none
AAMD_HD int mfcc_slot_mel(int s, int lane, int j) {
  if (s < 2) return 32 * s + 8 * (lane >> 4) + j;
  return j < 4 ? 64 + 4 * (lane >> 4) + j : -1;
}
AAMD_HD float mfcc_frag_value(const float* dct, int n_mels, int n_mfcc, int t, int s, int lane, int j) {
  const int mel = mfcc_slot_mel(s, lane, j), coef = 16 * t + (lane & 15);
  return (mel >= 0 && mel < n_mels && coef < n_mfcc) ? dct[mel * n_mfcc + coef] * kMfccDScale : 0.0f;
}
// index (in halves) of the lane's B values of step s in the staged binary16 planes.  Column n = lane & 15 of the B operand:
// n = 0 .. 5 the hi plane of frame n, n = 8 .. 13 the lo plane of frame n - 8 (columns 6, 7, 14, 15: any live row; their
// products are never stored).  The lo plane starts kMfccPlaneHalves behind the hi plane.
AAMD_HD int mfcc_b_col_frame(int lane) { const int j = lane & 7; return j < kFramesPerWave ? j : kFramesPerWave - 1; }
AAMD_HD int mfcc_b_col_plane(int lane) { return (lane >> 3) & 1; }
AAMD_HD int mfcc_b_index(int lane, int s) {
  return mfcc_b_col_plane(lane) * (kFramesPerWave * kMfccMels) + mfcc_b_col_frame(lane) * kMfccMels +
         (s < 2 ? 32 * s + 8 * (lane >> 4) : 64 + 4 * (lane >> 4));
}
constexpr int kMfccPlaneHalves = kFramesPerWave * kMfccMels;     // 480: the lo plane starts here (960 B, 16-B aligned)

// ---- banded filterbank in LDS (built once per launch by every workgroup) ------------------
//   Bands are processed in chunks of 4 taps = one b128 of weights + two b128 of P (2 bins x 2
//   frames each), so band starts are floored to even bins and rows are zero padded.
//   Table ROWS are lane assignments: row 20 r + pi is evaluated by the lane at position pi in round r.
//   Which mel a row holds is free (MelBandsDev::order, chosen on the host so that the b128 band reads of
//   each 16-lane group fall on distinct 4-bank slots); row_mel[] maps a row back to its mel for the store.
struct MelTab {
  const float* w;     // [rows][ws]: weight of bin lo2[row] + j, zero outside the band
  const int* lo2;     // [rows]: even band start (<= first non-zero bin)
  const int* rc;      // [n_rounds]: 4-tap chunks of the round (wave-uniform trip count)
  const int* row_mel; // [rows]: mel evaluated by the row, -1 = none
  int n_mels, ws, n_rounds;
};

AAMD_HD int mel_rounds(int n_mels) { return (n_mels + kMelSlots - 1) / kMelSlots; }
// row stride: covers the widest even-aligned band rounded up to 4 taps; ws / 4 odd so that the
// b128 reads of 16 different rows hit 16 different 4-bank slots
AAMD_HD int mel_ws(int max_width) {
  const int w4 = (max_width + 1 + 3) & ~3;
  return ((w4 >> 2) & 1) ? w4 : w4 + 4;
}
AAMD_HD int mel_rows(int n_mels) { return mel_rounds(n_mels) * kMelSlots; }
AAMD_HD int mel_tab_dwords(int n_mels, int max_width) {
  return mel_rows(n_mels) * (mel_ws(max_width) + 2) + kMelMaxRounds;
}
AAMD_HD int row_to_mel(const MelBandsDev& mb, int row) {
  const int m = mb.order ? mb.order[row] : row;
  return (m >= 0 && m < mb.n_mels) ? m : -1;
}

// pointers of the table image that starts at `base` (no memory access)
AAMD_HD void mel_tab_layout(const MelBandsDev& mb, float* base, MelTab& mt) {
  mt.n_mels = mb.n_mels;
  mt.ws = mel_ws(mb.max_width);
  mt.n_rounds = mel_rounds(mb.n_mels);
  const int rows = mel_rows(mb.n_mels);
  mt.w = base;
  int* lo2 = reinterpret_cast<int*>(base + rows * mt.ws);
  mt.lo2 = lo2;
  mt.row_mel = lo2 + rows;
  mt.rc = lo2 + 2 * rows;
}

// phase 1 (then a workgroup barrier): per-round chunk counts; phase 2: weights and band starts
AAMD_HD void mel_tab_rounds(int tid, int nthr, const MelBandsDev& mb, float* base, MelTab& mt) {
  mt.n_mels = mb.n_mels;
  mt.ws = mel_ws(mb.max_width);
  mt.n_rounds = mel_rounds(mb.n_mels);
  const int rows = mel_rows(mb.n_mels);
  mt.w = base;
  int* lo2 = reinterpret_cast<int*>(base + rows * mt.ws);
  int* row_mel = lo2 + rows;
  int* rc = row_mel + rows;
  mt.lo2 = lo2;
  mt.row_mel = row_mel;
  mt.rc = rc;
  for (int r = tid; r < mt.n_rounds; r += nthr) {
    int rw = 4;
    for (int row = r * kMelSlots; row < (r + 1) * kMelSlots; ++row) {
      const int q = row_to_mel(mb, row);
      if (q < 0) continue;
      const int e = (mb.width[q] + (mb.lo[q] & 1) + 3) & ~3;
      rw = e > rw ? e : rw;
    }
    rc[r] = rw >> 2;
  }
}

AAMD_HD void mel_tab_fill(int tid, int nthr, const MelBandsDev& mb, float* base, const MelTab& mt) {
  float* w = base;
  const int rows = mel_rows(mb.n_mels);
  int* lo2 = reinterpret_cast<int*>(base + rows * mt.ws);
  int* row_mel = lo2 + rows;
  for (int i = tid; i < rows * mt.ws; i += nthr) {
    const int row = i / mt.ws, j = i - row * mt.ws;
    const int m = row_to_mel(mb, row);
    const int rw = 4 * mt.rc[row / kMelSlots];
    const int lo = m >= 0 ? mb.lo[m] : 0, wd = m >= 0 ? mb.width[m] : 0;
    int l2 = lo & ~1;
    if (l2 + rw > kPK) l2 = kPK - rw;
    const int off = lo - l2;
    w[i] = (j >= off && j - off < wd) ? mb.weights[m * mb.max_width + (j - off)] : 0.0f;
    if (j == 0) {
      lo2[row] = l2;
      row_mel[row] = m;
    }
  }
}

// Per-workgroup constant tables in LDS (re-read every tile: 15 b128 reads per lane instead of
// 58 live registers -> one more wave per SIMD):
//   twiddles  [20 b][44]: W400^(b c) as (re, im) pairs, c = 0..19, row padded to 44 dwords
//   window    [20 b][20]: 0.5 * scale * window[b + 20 q]
constexpr int kTwRow = 42;    // 42 b mod 64 is distinct (and even) for the 20 rows: conflict-free ds_read_b64 of the twiddle pairs (44 made rows 16..19 collide with 0..3)
constexpr int kConstDwords = 20 * kTwRow + 20 * 20;   // 1280

AAMD_HD void const_tab_build(int tid, int nthr, const float* window, const float* tw400, float scale,
                             float* base) {
  for (int i = tid; i < 20 * 20; i += nthr) {
    const int b = i / 20, s = i - 20 * b;
    const int idx = (b * s) % kN;
    base[kTwRow * b + 2 * s] = tw400[2 * idx];
    base[kTwRow * b + 2 * s + 1] = tw400[2 * idx + 1];
    base[20 * kTwRow + i] = window[b + 20 * s] * (0.5f * scale);   // (b, q = s)
  }
}

// dynamic LDS of one workgroup: per-wave regions, constant tables, mel table, tile queue
AAMD_HD size_t lds_bytes(int n_mels, int max_width, int wave_dwords = kLdsDwordsPerWave, bool mfcc = false) {
  return ((size_t)kWavesPerBlock * wave_dwords + kConstDwords + mel_tab_dwords(n_mels, max_width) + 4 +
          (mfcc ? kMfccFragFloats : 0)) * sizeof(float);
}

struct LaneConst {
  const float* tw;         // this lane's twiddle row (pass-1 role b = pi)
  const float* win;        // this lane's window row
  int p, pi, col;          // pair 0..2, position in the 20-lane group, pass-2 column
  int active;              // lanes 60..63 shadow lanes 40..43 but never store
  int troff;               // kTOff[col] + 20 p: where this lane's column starts in the transposition buffer (dwords)
};

AAMD_HD void lane_init(int lane, const float* const_tab, LaneConst& c) {
  c.active = lane < 60;
  const int l = c.active ? lane : lane - 20;
  c.p = l / 20;
  c.pi = l - 20 * c.p;
  c.col = col_of_pos(c.pi);
  c.tw = const_tab + kTwRow * c.pi;
  c.win = const_tab + 20 * kTwRow + 20 * c.pi;
  int toff = 0;
#pragma unroll
  for (int k = 0; k < 20; ++k) toff = (c.col == k) ? kTOff[k] : toff;   // once per launch: selects, no table in memory
  c.troff = toff + 20 * c.p;
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
//   lane (p, b) owns samples n = b + 20 q of frames a = t0 + 2p and a + 1; the 28 gathered
//   samples X[q] sit at signal index a*160 - 200 + b + 20 q: frame a uses X[0..19], frame
//   a + 1 uses X[8..27].
//
// Staged tiles: the tile's 1200 contiguous samples were copied global -> LDS by LDS-DMA
// (coalesced 16-B pieces, issued one tile ahead); sample i of the tile lives at staging dword
// i + 20 * (i / 320): the 20-dword pad per 320 samples moves the three pairs' rows onto disjoint
// banks, so the strided gather below is conflict free.
template <int H, typename TIn = float>
AAMD_HD int stage_src_piece(int u) {   // staging piece u (16 B) <- tile piece (4 or 8 samples)
  using SG = Stage<H, TIn>;
  const int blk = u / SG::blk_pieces, r = u - SG::blk_pieces * blk;
  const int s = SG::blk_data * blk + (r < SG::blk_data ? r : SG::blk_data - 1);
  return s < SG::data_pieces ? s : SG::data_pieces - 1;
}

template <int H, typename TIn = float>
AAMD_HD void gather_lds(const LaneConst& c, const TIn* S, float (&X)[Hop<H>::nx], int chan = 0) {
  using HC = Hop<H>;
  using SG = Stage<H, TIn>;
  const TIn* src = S + (HC::pair_stride + SG::pad_elems) * c.p + c.pi;
#pragma unroll
  for (int q = 0; q < HC::nx; ++q) X[q] = InTraits<TIn>::get(src + 20 * q + SG::pad_elems * (q / (2 * H)), chan);
}

// Unstaged tiles (clip edges: reflect padding; or inputs that are not 16-B aligned): direct loads.
template <int H, typename TIn = float>
AAMD_HD void gather_global(const LaneConst& c, const TIn* wav_row, int64_t length, int64_t t0,
                           int n_frames, float (&X)[Hop<H>::nx], int chan = 0) {
  const int64_t ta = t0 + 2 * c.p;
  const int64_t i0 = ta * Hop<H>::hop - kPad + c.pi;
#pragma unroll
  for (int q = 0; q < Hop<H>::nx; ++q) {
    const int64_t i = i0 + 20 * q;
    // q < 20 belongs to frame a (and to a + 1 when q >= H); q >= 20 only to frame a + 1
    const bool need = (q < 20) ? (ta < n_frames) : (ta + 1 < n_frames);
    X[q] = need ? InTraits<TIn>::get(wav_row + reflect_idx(i, length), chan) : 0.0f;
  }
}

// DFT-20 over q of the lane's 20 complex inputs, twiddle by W400^(b s), transposed write (shared by the forward
// kernel and the inverse / adjoint kernel of istft400.h, whose inputs are spectrum bins instead of windowed samples)
#if defined(__HIP_DEVICE_COMPILE__)
// five transposed complex values (columns S0 .. S0 + 4) of every lane -> their rows; M0 = LDS byte address of the
// wave's region (declared clobbered: the compiler keeps nothing in M0 here, so there is nothing to save and restore --
// every scalar instruction costs the wave an issue slot of ~5 cycles, profiles/r03_zz_valu_issue.txt)
template <int S0>
__device__ __forceinline__ void tr_store5(unsigned lds_base, const float (&wr)[5], const float (&wi)[5]) {
  asm volatile(
      "s_mov_b32 m0, %[b]\n\t"
      "s_nop 0\n\t"
      "ds_write_addtid_b32 %[r0] offset:%[o0]\n\t"
      "ds_write_addtid_b32 %[i0] offset:%[q0]\n\t"
      "ds_write_addtid_b32 %[r1] offset:%[o1]\n\t"
      "ds_write_addtid_b32 %[i1] offset:%[q1]\n\t"
      "ds_write_addtid_b32 %[r2] offset:%[o2]\n\t"
      "ds_write_addtid_b32 %[i2] offset:%[q2]\n\t"
      "ds_write_addtid_b32 %[r3] offset:%[o3]\n\t"
      "ds_write_addtid_b32 %[i3] offset:%[q3]\n\t"
      "ds_write_addtid_b32 %[r4] offset:%[o4]\n\t"
      "ds_write_addtid_b32 %[i4] offset:%[q4]"
      :
      : [b] "s"(lds_base), [r0] "v"(wr[0]), [i0] "v"(wi[0]), [r1] "v"(wr[1]), [i1] "v"(wi[1]), [r2] "v"(wr[2]),
        [i2] "v"(wi[2]), [r3] "v"(wr[3]), [i3] "v"(wi[3]), [r4] "v"(wr[4]), [i4] "v"(wi[4]),
        [o0] "i"(4 * kTOff[S0]), [q0] "i"(4 * kTOff[S0] + 256), [o1] "i"(4 * kTOff[S0 + 1]), [q1] "i"(4 * kTOff[S0 + 1] + 256),
        [o2] "i"(4 * kTOff[S0 + 2]), [q2] "i"(4 * kTOff[S0 + 2] + 256), [o3] "i"(4 * kTOff[S0 + 3]),
        [q3] "i"(4 * kTOff[S0 + 3] + 256), [o4] "i"(4 * kTOff[S0 + 4]), [q4] "i"(4 * kTOff[S0 + 4] + 256)
      : "memory", "m0");
}
#endif

// DFT-20 over q of the lane's 20 complex inputs, twiddle by W400^(b s), transposed write (shared by the forward
// kernel and the inverse / adjoint kernel of istft400.h, whose inputs are spectrum bins instead of windowed samples)
template <int TREG = 0>      // leading batches of five columns whose twiddles are in registers (`twr`); the rest comes from the LDS table
AAMD_HD void phase_a_core(const LaneConst& c, const float (&xr)[20], const float (&xi)[20], float* __restrict__ lds,
                          const float* twr = nullptr) {
  // The twiddle table and the transposition rows live in the same LDS array but never overlap.  The reads of batch
  // k + 1 are issued BEFORE the multiplies and row writes of batch k (the row writes are asm volatile with a memory
  // clobber, which nothing can be hoisted across), and batch 0's before the DFT: one LDS round trip per tile is
  // exposed here instead of ten in the first version (round-2 ISA review: read -> wait -> multiply -> write, ten times).
  // Batches of five columns (10 VGPRs, double buffered) keep this inside the 168-register budget of 3 waves per SIMD.
  const float* __restrict__ twp = c.tw;
  float twn[10];
  auto tw_load = [&](int blk) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const F2 t = *reinterpret_cast<const F2*>(twp + 2 * (5 * blk + j));   // 8-byte aligned (row start is 16-B aligned)
      twn[2 * j] = t.x;
      twn[2 * j + 1] = t.y;
    }
  };
  if (TREG < 4) tw_load(TREG);
  float yr[20], yi[20];
  dft20(xr, xi, yr, yi);
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned lds_base = (unsigned)(uintptr_t)lds;      // wave-uniform LDS byte address
#endif
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    float tw[10];      // W^(b s) for s = 5 blk .. 5 blk + 4 as (re, im) pairs
    if (blk >= TREG) {
#pragma unroll
      for (int j = 0; j < 10; ++j) tw[j] = twn[j];
      if (blk < 3) tw_load(blk + 1);
    } else {
#pragma unroll
      for (int j = 0; j < 10; ++j) tw[j] = twr[10 * blk + j];
    }
    float wr[5], wi[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int s = 5 * blk + j;
      if (s == 0) {
        wr[j] = yr[0];
        wi[j] = yi[0];
      } else {
        wr[j] = yr[s] * tw[2 * j] - yi[s] * tw[2 * j + 1];
        wi[j] = yr[s] * tw[2 * j + 1] + yi[s] * tw[2 * j];
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (blk == 0) tr_store5<0>(lds_base, wr, wi);
    else if (blk == 1) tr_store5<5>(lds_base, wr, wi);
    else if (blk == 2) tr_store5<10>(lds_base, wr, wi);
    else tr_store5<15>(lds_base, wr, wi);
#else
    const int l = 20 * c.p + c.pi;       // CPU replay: lanes 60..63 rewrite what lanes 40..43 wrote (same values)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      lds[kTOff[5 * blk + j] + l] = wr[j];
      lds[kTOff[5 * blk + j] + 64 + l] = wi[j];
    }
#endif
  }
}

//   A frame b beyond the end of the clip is NOT zeroed here (that cost 20 selects per tile): its
//   spectrum is garbage that no store path writes (store_direct / store_wide / store_spec mask by
//   frame) and that the dB epilogue excludes from the running maximum.
template <int H, bool WREG = false, int TREG = 0>
AAMD_HD void phase_a(const LaneConst& c, const float (&X)[Hop<H>::nx], float* lds, const float* winr = nullptr,
                     const float* twr = nullptr) {
  float xr[20], xi[20];
  if (WREG) {                           // lab: window taps held in registers instead of the LDS table
#pragma unroll
    for (int q = 0; q < 20; ++q) { xr[q] = X[q] * winr[q]; xi[q] = X[q + H] * winr[q]; }
  } else {
#pragma unroll
    for (int q4 = 0; q4 < 5; ++q4) {
      const F4 w = *reinterpret_cast<const F4*>(c.win + 4 * q4);
      const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = 4 * q4 + e;
        xr[q] = X[q] * wv[e];
        xi[q] = X[q + H] * wv[e];
      }
    }
  }
  phase_a_core<TREG>(c, xr, xi, lds, twr);
}

// ---- phase B1: read own row (then DFT-20 over b  ->  Z[col + 20 d] in registers) ----------
AAMD_HD void phase_b1_load(const LaneConst& c, const float* lds, float (&vr)[20], float (&vi)[20]) {
  const float* col = lds + c.troff;      // row (column c, re) at the pair's 20 writer lanes; (c, im) 64 dwords on
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const F4 r = *reinterpret_cast<const F4*>(col + 4 * j);
    const F4 i = *reinterpret_cast<const F4*>(col + 64 + 4 * j);
    vr[4 * j] = r.x; vr[4 * j + 1] = r.y; vr[4 * j + 2] = r.z; vr[4 * j + 3] = r.w;
    vi[4 * j] = i.x; vi[4 * j + 1] = i.y; vi[4 * j + 2] = i.z; vi[4 * j + 3] = i.w;
  }
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
//   x = the exchanged values: q of lane ^ 1 (DPP swap), except on the self-paired columns 0 and 10,
//   which keep their own q (exchange_partner below).  conj(Z[400-k]) for k = col + 20u is x[9 - u].
AAMD_HD void phase_b2(const LaneConst& c, const float (&zr)[20], const float (&zi)[20],
                      const float (&xr)[10], const float (&xi)[10], float* lds) {
  if (!c.active) return;
  float* P = lds + kPPair * c.p + 2 * c.col;
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    const float cr = xr[9 - u], ci = xi[9 - u];
    const float ar = zr[u] + cr, ai = zi[u] - ci;   // 2*A = Z[k] + conj(Z[N-k])
    const float br = zr[u] - cr, bi = zi[u] + ci;   // |2*B|: Z[k] - conj(Z[N-k])
    *reinterpret_cast<F2*>(P + 40 * u) = F2{ar * ar + ai * ai, br * br + bi * bi};
  }
  if (c.col == 0) {  // k = 200 (Nyquist) is its own conjugate partner: A = Re, B = Im
    const float ar = zr[10] + zr[10], bi = zi[10] + zi[10];
    *reinterpret_cast<F2*>(P + 400) = F2{ar * ar, bi * bi};
  }
}

// ---- SPEC epilogue: |A|^power, |B|^power as frame-contiguous rows -------------------------
//   The tile's 6 x 201 outputs are contiguous in memory (frame-major spectrogram).  They are
//   staged at LDS float index `phase + 201 * frame + k`, phase = (global float offset of the
//   tile) mod 4, so that 16-B LDS pieces line up with 16-B global pieces.
AAMD_HD float spec_pow(float m2, float power) {
  if (power == 2.0f) return m2;
  const float m = sqrt(m2);
  if (power == 1.0f) return m;
  return pow(m, power);
}

AAMD_HD void phase_b2_spec(const LaneConst& c, const float (&zr)[20], const float (&zi)[20],
                           const float (&xr)[10], const float (&xi)[10], float power, int phase, float* lds) {
  if (!c.active) return;
  float* Ra = lds + phase + kSpecBins * 2 * c.p + c.col;
  float* Rb = Ra + kSpecBins;
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    const float cr = xr[9 - u], ci = xi[9 - u];
    const float ar = zr[u] + cr, ai = zi[u] - ci;
    const float br = zr[u] - cr, bi = zi[u] + ci;
    Ra[20 * u] = spec_pow(ar * ar + ai * ai, power);
    Rb[20 * u] = spec_pow(br * br + bi * bi, power);
  }
  if (c.col == 0) {
    const float ar = zr[10] + zr[10], bi = zi[10] + zi[10];
    Ra[200] = spec_pow(ar * ar, power);
    Rb[200] = spec_pow(bi * bi, power);
  }
}

// Complex spectrogram (power = None): interleaved (re, im) rows of 402 floats.  Six of them do not fit
// below the staging area, so the tile is written in two halves of 3 frames (frames 3 h .. 3 h + 2):
//   A = (Z[k] + conj Z[N-k]) / 2 = (ar, ai) / 2,   B = (Z[k] - conj Z[N-k]) / (2i) = (bi, -br) / 2
// (the 1/2 is folded into the window table).
AAMD_HD void phase_b2_spec_complex(const LaneConst& c, const float (&zr)[20], const float (&zi)[20],
                                   const float (&xr)[10], const float (&xi)[10], int half, int phase, float* lds) {
  if (!c.active) return;
  const int fa = 2 * c.p - 3 * half, fb = fa + 1;          // row of frame a / b inside this half, or outside [0, 3)
  const bool wa = fa >= 0 && fa < 3, wb = fb >= 0 && fb < 3;
  float* Ra = lds + phase + 2 * kSpecBins * fa + 2 * c.col;
  float* Rb = lds + phase + 2 * kSpecBins * fb + 2 * c.col;
#pragma unroll
  for (int u = 0; u < 10; ++u) {
    const float cr = xr[9 - u], ci = xi[9 - u];
    if (wa) { Ra[40 * u] = zr[u] + cr; Ra[40 * u + 1] = zi[u] - ci; }
    if (wb) { Rb[40 * u] = zi[u] + ci; Rb[40 * u + 1] = cr - zr[u]; }
  }
  if (c.col == 0) {   // Nyquist bin: real for both frames
    if (wa) { Ra[400] = zr[10] + zr[10]; Ra[401] = 0.0f; }
    if (wb) { Rb[400] = zi[10] + zi[10]; Rb[401] = 0.0f; }
  }
}

// coalesced copy of the staged rows: 16-B pieces where a piece lies fully inside the run,
// single floats at the ragged ends
AAMD_HD void store_spec(int lane, const float* lds, float* out, int64_t a0, int n_floats) {
  const int phase = (int)(a0 & 3);
  float* base = out + (a0 - phase);   // 16-B aligned when `out` is
  const int n_pieces = (phase + n_floats + 3) >> 2;
  for (int j = lane; j < n_pieces; j += 64) {
    const int i0 = 4 * j;
    if (i0 >= phase && i0 + 4 <= phase + n_floats) {
      *reinterpret_cast<F4*>(base + i0) = *reinterpret_cast<const F4*>(lds + i0);
    } else {
      for (int e = 0; e < 4; ++e)
        if (i0 + e >= phase && i0 + e < phase + n_floats) base[i0 + e] = lds[i0 + e];
    }
  }
}

// The same copy for a FULL tile (6 x 201 floats; 166 of a 10 s clip's 167): the five 16-byte pieces of a lane are read
// together and stored from registers; only the run's first and last piece can be ragged (phase != 0, (phase + 1206) % 4 != 0)
// and their two lanes finish with single floats out of the registers they already hold.  (The loop above reads, waits and
// stores piece by piece and walks four guarded single-float blocks, each with an LDS read of its own, in its first and last
// iteration.)
#ifndef AAMD_M400_SPEC_FULL
#define AAMD_M400_SPEC_FULL 1
#endif
AAMD_HD void store_spec_full(int lane, const float* lds, float* out, int64_t a0) {
  constexpr int kN = kFramesPerWave * kSpecBins;          // 1206
  const int phase = (int)(a0 & 3);
  float* base = out + (a0 - phase);
  const int n_pieces = (phase + kN + 3) >> 2;             // 302 or 303
  F4 v[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int j = lane + 64 * u;
    if (u < 4 || j < n_pieces) v[u] = *reinterpret_cast<const F4*>(lds + 4 * j);
  }
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int j = lane + 64 * u, i0 = 4 * j;
    const bool full = i0 >= phase && i0 + 4 <= phase + kN;
    if ((u < 4 || j < n_pieces) && full) *reinterpret_cast<F4*>(base + i0) = v[u];
  }
  // the ragged ends: piece 0 (lane 0 of u = 0) and piece n_pieces - 1 (u = 4)
  if (lane == 0 && phase != 0) {
    const float e[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k >= phase) base[k] = e[k];
  }
  const int jl = n_pieces - 1, il = 4 * jl;               // (jl >= 256: a lane of u = 4)
  if (lane + 256 == jl && il + 4 > phase + kN) {
    const float e[4] = {v[4].x, v[4].y, v[4].z, v[4].w};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (il + k < phase + kN) base[il + k] = e[k];
  }
}

// amplitude_to_DB of one value (functional.py:390-391); log10 through the hardware log2
// (v_log_f32, ~1 ulp of log2 x => < 2e-5 dB absolute)
AAMD_HD float fast_log10(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __log2f(x) * 0.30102999566398120f;
#else
  return log10(x);
#endif
}
// _piecewise_linear_log (pipelines/rnnt_pipeline.py:20-23) EXACTLY as the reference evaluates it: two in-place
// masked assignments, `x[x > e] = log(x[x > e])` then `x[x <= e] = x[x <= e] / e` -- the second mask is taken
// AFTER the first assignment, so values in (e, e^e] (whose log is <= e) are divided by e as well:
//   x <= e: x / e;   e < x <= e^e: ln(x) / e;   x > e^e: ln(x).   The trained models expect exactly this.
AAMD_HD float epi_plog(float x) {
  const float kE = 2.718281828459045f;
#if defined(__HIP_DEVICE_COMPILE__)
  const float lg = __log2f(x) * 0.69314718055994531f;
#else
  const float lg = log(x);
#endif
  const float t = x > kE ? lg : x;
  return t <= kE ? t / kE : t;
}
AAMD_HD float epi_db(float x, const Epi400& e) {
  return e.multiplier * fast_log10(fmax(x, e.amin)) - e.db_sub;
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
//   lane (p, slot) computes mel m = slot + 20 r of frames 2p and 2p + 1 in round r; the chunk
//   count rc[r] is wave-uniform.  All LDS reads of a round are issued before its FMAs.
template <int NC>
AAMD_HD void mel_chunks(const float* wt, const float* P, float& sa, float& sb) {
  F4 w[NC], q0[NC], q1[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    w[k] = *reinterpret_cast<const F4*>(wt + 4 * k);
    q0[k] = *reinterpret_cast<const F4*>(P + 8 * k);       // (a, b) of bins 4k, 4k + 1
    q1[k] = *reinterpret_cast<const F4*>(P + 8 * k + 4);   // (a, b) of bins 4k + 2, 4k + 3
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    sa += w[k].x * q0[k].x;
    sb += w[k].x * q0[k].y;
    sa += w[k].y * q0[k].z;
    sb += w[k].y * q0[k].w;
    sa += w[k].z * q1[k].x;
    sb += w[k].z * q1[k].y;
    sa += w[k].w * q1[k].z;
    sb += w[k].w * q1[k].w;
  }
}

// Per-lane copies of the table's control words, read ONCE per launch (round-2 ISA review: per round and tile the kernel
// re-read chunk count, band start and mel index from LDS and waited for each -- 12 dependent LDS round trips per tile):
//   nc[r]   chunks of round r (wave-uniform, an SGPR)      poff[r]  dword offset of the lane's band start in a P row pair
//   mel[r]  the mel the lane evaluates in round r, -1 = none
template <int NR>
struct MelRegs {
  int nc[NR];     // wave-uniform
  int pm[NR];     // poff | (mel & 0xffff) << 16: one VGPR per round (the 168-register budget of 3 waves/SIMD is tight)
  AAMD_HD int poff(int r) const { return pm[r] & 0xffff; }
  AAMD_HD int mel(int r) const { return pm[r] >> 16; }      // arithmetic shift: 0xffff.... -> -1
};

template <int NR>
AAMD_HD void mel_regs_load(const LaneConst& c, const MelTab& mt, MelRegs<NR>& h) {
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const bool on = r < mt.n_rounds;
    const int row = r * kMelSlots + c.pi;
#if defined(__HIP_DEVICE_COMPILE__)
    h.nc[r] = on ? __builtin_amdgcn_readfirstlane(mt.rc[r]) : 0;
#else
    h.nc[r] = on ? mt.rc[r] : 0;
#endif
    const int poff = on ? 2 * mt.lo2[row] : 0;
    const int mel = on ? mt.row_mel[row] : -1;
    h.pm[r] = (int)(((unsigned)mel << 16) | (unsigned)poff);
  }
}

// SIG != 0: the filterbank's shape is a compile-time constant -- nibble r = 4-tap chunks of round r, exactly NR rounds, every
// table row holds a mel (n_mels = 20 NR).  The four rounds are then straight-line code (no switch on a chunk count, no
// `round exists` / `row holds a mel` tests), so the reads of all rounds are in flight together: - 3 % on the headline batch
// (profiles/r03_g_mel400_lab_rcsig.txt).  The launcher picks the instantiation when the band table's signature matches.
AAMD_HD constexpr int sig_chunks(int sig, int r) { return (sig >> (4 * r)) & 15; }
constexpr int kSigHtk80 = 0x4221, kSigSlaney80 = 0x4211;      // melscale_fbanks(201, 0, 8000, 80, 16000): htk / slaney

template <int NR = kMelMaxRounds, int SIG = 0, bool FENCED = false>
AAMD_HD void phase_c(const LaneConst& c, const MelTab& mt, const float* lds,
                     float (&acc_a)[NR], float (&acc_b)[NR], const MelRegs<NR>* h = nullptr) {
  const float* Pp = lds + kPPair * c.p;
  const float* wt0 = mt.w + c.pi * mt.ws;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    float sa = 0.0f, sb = 0.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    // FENCED: the straight-line rounds of a signature instantiation one after the other (the scheduler otherwise puts the reads
    // of all four rounds in flight together: up to 108 registers, which the MFCC instantiation does not have)
    if (FENCED && SIG != 0 && r > 0 && (r & 1) == 0) __builtin_amdgcn_sched_barrier(0);
#endif
    if (SIG != 0 || r < mt.n_rounds) {
      const float* wt = wt0 + r * kMelSlots * mt.ws;
      const float* P = Pp + (h ? h->poff(r) : 2 * mt.lo2[r * kMelSlots + c.pi]);
      const int nc = SIG != 0 ? sig_chunks(SIG, r) : (h ? h->nc[r] : mt.rc[r]);
      switch (nc) {
        case 1: mel_chunks<1>(wt, P, sa, sb); break;
        case 2: mel_chunks<2>(wt, P, sa, sb); break;
        case 3: mel_chunks<3>(wt, P, sa, sb); break;
        case 4: mel_chunks<4>(wt, P, sa, sb); break;
        default:
          for (int k = 0; k < nc; ++k) mel_chunks<1>(wt + 4 * k, P + 8 * k, sa, sb);
      }
    }
    acc_a[r] = sa;
    acc_b[r] = sb;
  }
}

// narrow store path (any n_mels / alignment): 4-byte stores straight from the accumulators
//   Addressing: ONE wave-uniform tile pointer (SGPRs) + a small per-lane 32-bit offset, so every
//   store is `global_store_dword voffset, data, s[base]` with one v_add -- no per-lane 64-bit pointers.
template <int NR = kMelMaxRounds, int SIG = 0>
AAMD_HD void store_direct(const LaneConst& c, const MelTab& mt, const float (&acc_a)[NR],
                          const float (&acc_b)[NR], float* out_row, int64_t t0, int n_frames,
                          const MelRegs<NR>* h = nullptr) {
  const int left = (int)(n_frames - t0);               // frames of this tile inside the clip (wave-uniform)
  const bool va = c.active && 2 * c.p < left, vb = c.active && 2 * c.p + 1 < left;
  float* out_tile = out_row + t0 * (int64_t)mt.n_mels;   // wave-uniform
  const unsigned oa = 2u * (unsigned)c.p * (unsigned)mt.n_mels;
  if (SIG != 0) {      // every round and every table row is live: one EXEC region per frame instead of one per store
    if (va) {
#pragma unroll
      for (int r = 0; r < NR; ++r) out_tile[oa + (unsigned)h->mel(r)] = acc_a[r];
    }
    if (vb) {
#pragma unroll
      for (int r = 0; r < NR; ++r) out_tile[oa + (unsigned)mt.n_mels + (unsigned)h->mel(r)] = acc_b[r];
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if (r < mt.n_rounds) {
      const int m = h ? h->mel(r) : mt.row_mel[r * kMelSlots + c.pi];
      if (m >= 0) {
        if (va) out_tile[oa + (unsigned)m] = acc_a[r];
        if (vb) out_tile[oa + (unsigned)mt.n_mels + (unsigned)m] = acc_b[r];
      }
    }
  }
}

// wide store path (n_mels % 4 == 0, 16-B aligned output): the tile's 6 x n_mels outputs are
// contiguous in memory; stage them in LDS (over the dead P rows) and write 16 B per lane.
template <int NR = kMelMaxRounds>
AAMD_HD void store_stage(const LaneConst& c, const MelTab& mt, const float (&acc_a)[NR],
                         const float (&acc_b)[NR], float* lds, const MelRegs<NR>* h = nullptr) {
  if (!c.active) return;
  float* oa = lds + 2 * c.p * mt.n_mels;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    if (r < mt.n_rounds) {
      const int m = h ? h->mel(r) : mt.row_mel[r * kMelSlots + c.pi];
      if (m >= 0) {
        oa[m] = acc_a[r];
        oa[mt.n_mels + m] = acc_b[r];
      }
    }
  }
}

AAMD_HD void store_wide(int lane, const MelTab& mt, const float* lds, float* out_row, int64_t t0,
                        int n_frames) {
  const int64_t left = n_frames - t0;
  const int n_valid = left < kFramesPerWave ? (int)left : kFramesPerWave;
  const int pieces = n_valid * mt.n_mels / 4;
  float* dst = out_row + t0 * (int64_t)mt.n_mels;
  for (int j = lane; j < pieces; j += 64)
    *reinterpret_cast<F4*>(dst + 4 * j) = *reinterpret_cast<const F4*>(lds + 4 * j);
}

#if defined(__HIPCC__)
// v_max_f32 / v_min_f32 on operands the caller knows to be canonical (results of arithmetic): fmaxf / fminf compile to
// llvm.maxnum / minnum, which under the IEEE mode of compute kernels get a quieting `v_max_f32 x, x, x` per operand whose
// producer the compiler cannot see.  (The instruction itself quiets a signalling NaN and returns the other operand for a quiet
// one, as fmaxf does.)
__device__ __forceinline__ float vmax_raw(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ float vmin_raw(float x, float y) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ float vmax3_raw(float x, float y, float z) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}
__device__ __forceinline__ float vmin3_raw(float x, float y, float z) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}
__device__ __forceinline__ void wave_lds_fence() {
  // Hand-off between lanes of ONE wave through LDS.  The hardware executes a wave's LDS
  // instructions in order, so no s_waitcnt is needed between the writes and the dependent
  // reads; this only stops the COMPILER from moving LDS accesses across the hand-off
  // (a release/acquire fence would also drain vmcnt/lgkmcnt and stall the wave 6x per tile).
  asm volatile("" ::: "memory");
}

// In-place exchange with lane ^ 1 (DPP quad_perm [1,0,3,2], a VALU move) of 20 registers, with the
// self-paired lanes (columns 0 and 10: wave-uniform mask `self_mask`) switched off in EXEC so that
// they keep their own values: replaces 20 v_mov_dpp + 20 v_cndmask by 20 v_mov_dpp.  Lanes 0/1 of a
// quad are each other's partners, so no enabled lane reads a disabled one.  s_nop: VALU write ->
// DPP read of the same VGPR needs 2 wait states, an EXEC write before a DPP op 5.
__device__ __forceinline__ void exchange_partner(float (&qr)[10], float (&qi)[10], unsigned long long self_mask) {
  unsigned long long saved;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_andn2_b64 exec, exec, %[m]\n\t"
      "s_nop 4\n\t"
      "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %12, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %13, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %14, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %16, %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %17, %17 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %18, %18 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_mov_b32_dpp %19, %19 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b64 exec, %[sv]"
      : "+v"(qr[0]), "+v"(qr[1]), "+v"(qr[2]), "+v"(qr[3]), "+v"(qr[4]), "+v"(qr[5]), "+v"(qr[6]), "+v"(qr[7]),
        "+v"(qr[8]), "+v"(qr[9]), "+v"(qi[0]), "+v"(qi[1]), "+v"(qi[2]), "+v"(qi[3]), "+v"(qi[4]), "+v"(qi[5]),
        "+v"(qi[6]), "+v"(qi[7]), "+v"(qi[8]), "+v"(qi[9]), [sv] "=&s"(saved)
      : [m] "s"(self_mask));
}

// LDS-DMA of one 16-B piece per lane: 64 lanes fill 1 KiB at LDS byte address `lds_dst`
// (wave-uniform).  hipcc does not count this load: the consumer waits with stage_wait().
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

// the same with the source as wave-uniform base (SGPR pair) + 32-bit lane offset in bytes: no per-lane 64-bit address
#ifndef AAMD_M400_DMA_SADDR
#define AAMD_M400_DMA_SADDR 1
#endif
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff_bytes, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff_bytes), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct TileInfo {
  int64_t row, t0;
  bool staged;   // fully inside the clip and 16-B aligned: gathered through the LDS staging area
};

// LAB != 0 builds profiling variants for tools/ubench/mel400_lab.hip (wrong results by design):
//   bit 0: no wait for the staged tile   bit 1: no global stores   bit 2: no phase C
//   bit 3: no LDS-DMA issue              bit 4: no phase B (second DFT, separation, P rows)
//   bit 15 (32768): interior tiles gather their samples with plain global loads, the NEXT tile's 28 samples per lane
//                   prefetched into registers right after phase A (no LDS staging at all)
//   bit 22 (4194304): the stores of a wave always hit the same (cache-resident) 1920 bytes: store ISSUE cost without HBM writes
//   bit 20 (1048576): every wave records (cycle counter, 100 MHz wall clock) at entry and exit into epi.fix_count
//   bit 17 (131072): tiles handed out chip-wide in chunks of kLabChunk from ONE global counter (epi.group_max, zeroed by
//                   the lab before each launch) instead of static per-workgroup ranges
// float max through integer atomics (target initialised to -inf or any float)
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// The DCT matrix as MFMA A fragments, split into binary16 hi / lo planes (once per DCT matrix; aamd_mfcc_frag_build).
__global__ void __launch_bounds__(256) mfcc_frag_build_kernel(const float* __restrict__ dct, int n_mels, int n_mfcc,
                                                              float* __restrict__ frag) {
  uint16_t* fh = reinterpret_cast<uint16_t*>(frag);
  const int n = kMfccMT * kMfccSteps * 64 * 8;                        // (t, s, lane, j)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int j = i & 7, lane = (i >> 3) & 63, g = i >> 9, s = g % kMfccSteps, t = g / kMfccSteps;
    const float v = mfcc_frag_value(dct, n_mels, n_mfcc, t, s, lane, j);
    const uint16_t hi = rsm::f16_bits(v);
    const uint16_t lo = rsm::f16_bits(v - rsm::f16_value(hi));
    fh[mfcc_frag_piece(t, s, 0, lane) * 8 + j] = hi;
    fh[mfcc_frag_piece(t, s, 1, lane) * 8 + j] = lo;
  }
}

// One workgroup lays out the band table image in global memory (once per filterbank; aamd_mel400_table_build).
__global__ void __launch_bounds__(256) mel_tab_build_kernel(MelBandsDev mb, float* __restrict__ out) {
  MelTab mt{};
  mel_tab_rounds(threadIdx.x, blockDim.x, mb, out, mt);
  __syncthreads();   // the chunk counts written above are read below by other threads of this workgroup
  mel_tab_fill(threadIdx.x, blockDim.x, mb, out, mt);
}

// NR = rounds of 20 mels the per-lane arrays are sized for: 4 (n_mels <= 80, the common front-ends: the control words of
// the band table then live in registers, MelRegs) or kMelMaxRounds (up to 160 mels, control words re-read from LDS).
template <int LAB, int EPI, int H = 8, typename TIn = float, int NR = kMelMaxRounds, int SIG = 0>
__global__ void __launch_bounds__(64 * kWavesPerBlock, AAMD_M400_MINWAVES)
melspec400_kernel(const TIn* __restrict__ wav, const float* __restrict__ window,
                  const float* __restrict__ tw400, MelBandsDev mb, float* __restrict__ out,
                  int64_t rows, int64_t length, int64_t row_stride, int n_frames, float scale,
                  int tiles_per_row, int64_t n_tiles, int tiles_per_block, int in_aligned,
                  int out_wide, Epi400 epi) {
  extern __shared__ __attribute__((aligned(16))) float smem400[];
  // Fix-up pass of the fused MFCC.  The workgroup's candidates are the tiles b, b + nb, b + 2 nb, ... (clamped tiles cluster by
  // clip -- silence --, a strided share spreads any run of them over the whole chip): it checks their minima against the
  // cut-offs -- final now: the caller reduced group_max over the ranks between the passes --, keeps the flagged ones in its
  // private run of the scratch list and leaves when there are none (the common case), before any table is built.  (Rounds
  // 3-4 compacted ONE chip-wide list in a kernel of its own between the passes; un-profiled that launch cost 3-5 us per call,
  // profiles/r04_p_mfcc_one_launch.txt.)
  // (Round 6 measured what this launch costs on the cfg4 noise batch, 7 us per call, and where: the bare launch of the grid is
  // ~1 us -- a build that returns here at once --, reading the minima in coalesced blocks of 16 instead of one cache line per
  // lane changes nothing, and the batch is not "nothing flagged": 2 of its 85 504 tiles lie under the cut-off, so two workgroups
  // build their tables and redo one tile each -- a wave-tile's latency.  The launch costs what redoing ONE tile costs.
  // profiles/r06_n_mfcc_fixup_floor.txt)
  unsigned fix_base = 0, fix_cnt = 0;
#if defined(AAMD_LAB) && defined(AAMD_MFCC_FIXUP_RETURNS)      /* lab, timing only: what the bare launch of the fix-up grid costs */
  if (EPI == EPI400_MFCC && epi.fixup != 0) return;
#endif
  if (EPI == EPI400_MFCC && epi.fixup != 0) {
    int* const cnt = reinterpret_cast<int*>(smem400);     // (wave 0's region: nothing lives there yet)
    if (threadIdx.x == 0) *cnt = 0;
    __syncthreads();
    const unsigned nt = (unsigned)n_tiles, nbk = gridDim.x, b = blockIdx.x;
    const unsigned n_cand = b < nt ? (nt - b + nbk - 1u) / nbk : 0u;
    fix_base = b * (nt / nbk) + (b < nt % nbk ? b : nt % nbk);
    for (unsigned k0 = 0; k0 < n_cand; k0 += blockDim.x) {
      const unsigned k = k0 + threadIdx.x;
      bool hit = false;
      unsigned t = 0;
      if (k < n_cand) {
        t = b + k * nbk;
        const unsigned row = t / (unsigned)tiles_per_row;
        hit = epi.tile_min[t] < epi.group_max[row / epi.rows_per_group] - epi.top_db;
      }
      const unsigned long long m = __ballot(hit);
      int slot = 0;
      if ((threadIdx.x & 63) == 0 && m)
        slot = __hip_atomic_fetch_add(cnt, __popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      slot = __builtin_amdgcn_readfirstlane(slot);
      if (hit) epi.fix_list[fix_base + slot + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = (int)t;
    }
    __syncthreads();                                      // (also drains the list stores of every wave)
    fix_cnt = (unsigned)__builtin_amdgcn_readfirstlane(*cnt);
    if (fix_cnt == 0) return;
    if (threadIdx.x == 0) atomicAdd(epi.fix_count, (int)fix_cnt);
    __syncthreads();                                      // *cnt is read: the tile loop may reuse the word
  }
  // pass 0 of the fused MFCC resets the counter of redone tiles that the fix-up workgroups add to (one launch less per call
  // than a memset; nothing touches it before the fix-up launch, which is ordered behind this kernel on the stream)
  // (every instantiation does it, the tools-only ones included: the caller hands over an uninitialised counter -- ADVICE r3)
  if (EPI == EPI400_MFCC && epi.fixup == 0 && epi.fix_count != nullptr && !(LAB & (1048576 | 8388608)) && blockIdx.x == 0 &&
      threadIdx.x == 0)
    *epi.fix_count = 0;
  // the tools-only switches of the MFCC epilogue (AAMD_MFCC_LAB) are honoured by an instantiation of their own: as run-time
  // branches in the product kernel they cut its MFMA section into basic blocks (the lesson of the resampler's census)
  const int elab = (LAB & 524288) ? epi.lab : 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef AAMD_M400_PRIO
  // lab (tools/mel400_lab.py): a static issue priority per wave; the waves w, w + 4, w + 8 of a SIMD get distinct ones
  {
    const int pr = AAMD_M400_PRIO == 1 ? (wave >> 2) : AAMD_M400_PRIO == 2 ? 2 - (wave >> 2) : (wave >> 2) == 0 ? 1 : 0;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
  }
#endif
  long long lab_t0 = 0;
  if (LAB & 1024) lab_t0 = wall_clock64();
  if ((LAB & 1048576) && lane == 0) {  // lab: ... and at its entry (shader clock of the launch = cycles / wall time)
    long long* rec = reinterpret_cast<long long*>(epi.fix_count);
    const int w = blockIdx.x * kWavesPerBlock + wave;
    rec[4 * w] = (long long)clock64();
    rec[4 * w + 1] = wall_clock64();
  }
  using HC = Hop<H>;
  constexpr int kHop = HC::hop;
  float* lds = smem400 + wave * HC::lds_dwords;
  const unsigned s_addr = (unsigned)(uintptr_t)(lds + kSOff);   // LDS byte address of the staging area

  float* const_tab = smem400 + kWavesPerBlock * HC::lds_dwords;
  const_tab_build(threadIdx.x, blockDim.x, window, tw400, scale, const_tab);
  MelTab mt{};
  const int tab_dwords = (EPI != EPI400_SPEC) ? mel_tab_dwords(mb.n_mels, mb.max_width) : 0;
  // tile queue of this workgroup: the next unclaimed tile (waves start on tiles 0 .. W-1)
  int* queue = reinterpret_cast<int*>(const_tab + kConstDwords + tab_dwords);
  // lab bit 24 (16777216, tools/mel400_lab.py): UPPER BOUND of a wave-specialised variant, emulated without
  // any hand-off: the last wave of the workgroup is an I/O wave that issues every LDS-DMA of the workgroup's tiles (into the
  // staging areas, 8 tiles in flight) and every output store (16-byte stores from LDS), the other 11 waves claim all the
  // tiles and run the arithmetic with no vector-memory instruction at all (build with bits 1 | 2 | 8 and launch with the wide
  // store path, so phase C still ends in the LDS staging writes a real hand-off needs).  Outputs are garbage -- nothing
  // orders the two kinds of waves -- the TIME is what a perfect, free hand-off would approach.  Bit 25: the I/O wave idles.
  // Bit 26: TWO I/O waves (tiles dealt alternately, 10 compute waves).  Bit 27: the I/O waves only fetch; the compute waves
  // keep their own (narrow) stores -- build without bit 2 and launch the narrow path.
  constexpr bool kIoWave = (LAB & 16777216) != 0;
  constexpr int kNio = kIoWave ? ((LAB & 67108864) ? 2 : 1) : 0;
  if (threadIdx.x == 0) *queue = kWavesPerBlock - kNio;
  // MFCC: the DCT fragments (18 KB) live in LDS behind the queue -- fetched from memory per tile they cost 18 global
  // loads per lane whose registers (prefetch) pushed the kernel into scratch, and a scratch reload waits for the LDS-DMA in
  // flight (one in-order vmcnt)
  float* frag_lds = const_tab + kConstDwords + tab_dwords + 4;
  // MFCC: the DCT fragments live in LDS for hop 100 / 160 and are read from the cache-resident table for hop 200 (whose wave
  // regions leave no room) -- decided by the INSTANTIATION, not at run time: with both paths in one kernel the compiler kept the
  // twelve 64-bit piece addresses of the global path in registers for the whole launch (24 VGPRs of a 168-register budget)
  constexpr bool kFragLds = (H != 10);
  if (EPI == EPI400_MFCC && (epi.frag_in_lds != 0) != kFragLds) __builtin_trap();      // (a launcher bug)
  if (EPI == EPI400_MFCC && kFragLds)
    for (int i = threadIdx.x; i < kMfccFragFloats; i += blockDim.x) frag_lds[i] = epi.dct_frag[i];
  if (EPI != EPI400_SPEC && mb.table400 != nullptr) {
    // the band table was laid out once per filterbank (mel_tab_build_kernel): one round of independent loads here
    // instead of the three dependent ones (lane order -> band start -> weight) of the in-kernel build
    float* dst = const_tab + kConstDwords;
    for (int i = threadIdx.x; i < tab_dwords; i += blockDim.x) dst[i] = mb.table400[i];
    mel_tab_layout(mb, dst, mt);
    __syncthreads();
  } else {
    if (EPI != EPI400_SPEC) mel_tab_rounds(threadIdx.x, blockDim.x, mb, const_tab + kConstDwords, mt);
    __syncthreads();
    if (EPI != EPI400_SPEC) mel_tab_fill(threadIdx.x, blockDim.x, mb, const_tab + kConstDwords, mt);
    __syncthreads();
  }

  long long lab_t1 = 0;
  if (LAB & 1024) lab_t1 = wall_clock64();
  LaneConst c;
  lane_init(lane, const_tab, c);
  // The 20 window taps of this lane live in registers for the whole launch (round 2: -5 us on the headline batch once
  // the buffers rotate over more than the 256 MiB Infinity Cache; 5 b128 LDS reads per tile less).  Lab bit 8192 forces
  // it on, bit 262144 forces the LDS table; the twiddles (38 more registers) stay in LDS (bit 16384: spills at 3 waves/SIMD).
  // (the MFCC instantiation kept its window taps in the LDS table until round 6: its 163 registers were 30 too many because both
  // fragment paths lived in one kernel -- kFragLds above; now 162 with the window and two twiddle batches in registers)
  // (the hop-200 MFCC instantiation reads its fragments from global memory and holds their addresses: window in the LDS table there)
  constexpr bool kWinRegs = (EPI != EPI400_MFCC || H != 10) && (((LAB & 8192) != 0) || !(LAB & 262144));
  float winr[20], twr[40];
  if (kWinRegs) {
#pragma unroll
    for (int q = 0; q < 20; ++q) winr[q] = c.win[q];
  }
#ifndef AAMD_M400_TWREG
#define AAMD_M400_TWREG 3
#endif
  // Twiddle batches (of five columns) held in registers instead of re-read from the LDS table every tile: all four under lab
  // bit 14 (spills).  The signature instantiation runs at 136 registers, so three batches fit its 168 (164; - 3.0 % on the
  // headline batch, 67.3 -> 65.3 us, profiles/r03_i_mel400_lab_twiddle_regs.txt): 15 of the 20 ds_read_b64 per tile gone.
  // The other float-input hop-160 instantiations get what their register count leaves (tests/test_no_spills.py keeps every
  // one of them out of scratch): Spectrogram 144 -> 2 batches, generic mel 149 / 135 (NR 4 / 8) -> 1 / 3, mel + dB 154 -> 1.
#ifndef AAMD_M400_TWREG_DB
#define AAMD_M400_TWREG_DB 1
#endif
#ifndef AAMD_M400_TWREG_MFCC
#define AAMD_M400_TWREG_MFCC 2
#endif
  constexpr bool kPlain = H == 8 && std::is_same<TIn, float>::value && LAB == 0;
  constexpr int kTwRegBatches = (LAB & 16384) ? 4
                                : (SIG != 0 && EPI == EPI400_MEL) ? AAMD_M400_TWREG
                                : !kPlain ? 0
                                : EPI == EPI400_SPEC ? 2
                                : (EPI == EPI400_MEL && SIG == 0) ? (NR <= 4 ? 1 : 3)
                                : (EPI == EPI400_MFCC && NR <= 4) ? AAMD_M400_TWREG_MFCC
                                : (EPI == EPI400_MEL_DB && NR <= 4) ? AAMD_M400_TWREG_DB : 0;
  if (kTwRegBatches > 0) {
#pragma unroll
    for (int q = 0; q < 10 * kTwRegBatches; ++q) twr[q] = c.tw[q];
  }
  constexpr bool kHoist = NR <= 4;
  MelRegs<NR> mregs;
  const MelRegs<NR>* mh = nullptr;
  if (EPI != EPI400_SPEC && kHoist) {
    mel_regs_load<NR>(c, mt, mregs);
    mh = &mregs;
  }
  if (SIG != 0) {
    // the launcher chose this instantiation from the caller's `table_sig`: it must describe the table that was just copied
    // into LDS (exactly NR rounds, every row a mel, these chunk counts) -- anything else is a caller bug, and a wrong
    // answer is the one thing this library does not return
    static_assert(SIG == 0 || NR == 4, "filterbank signatures are 4 rounds of 20 mels");
    const int sig_rt = mregs.nc[0] | (mregs.nc[1 % NR] << 4) | (mregs.nc[2 % NR] << 8) | (mregs.nc[3 % NR] << 12);
    if (mt.n_rounds != NR || mt.n_mels != kMelSlots * NR || sig_rt != SIG) __builtin_trap();
  }
  // the one-kernel MFCC serves 80 mels = exactly NR rounds of 20 table rows, each holding a mel (c_api: mfcc_fused_ok); its
  // epilogue is written for that
  if (EPI == EPI400_MFCC && (mt.n_rounds != NR || mt.n_mels != kMelSlots * NR)) __builtin_trap();
  float nrm_mean[NR], nrm_inv[NR];   // MEL_NORM: statistics of this lane's mels
  if (EPI == EPI400_MEL_NORM) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int m = r < mt.n_rounds ? mt.row_mel[r * kMelSlots + c.pi] : -1;
      nrm_mean[r] = m >= 0 ? epi.mean[m] : 0.0f;
      nrm_inv[r] = m >= 0 ? epi.invstd[m] : 0.0f;
    }
  }
  const unsigned long long self_mask = __ballot((c.col == 0) || (c.col == 10));   // wave-uniform (SGPR pair)
  using SG = Stage<H, TIn>;
  static_assert(SG::ok, "no staging layout for this hop / input type");
  int spiece[SG::ndma];   // first sample of the tile piece fetched by this lane in DMA instruction k
#pragma unroll
  for (int k = 0; k < SG::ndma; ++k) spiece[k] = SG::spp * stage_src_piece<H, TIn>(64 * k + lane);

  // XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous
  // range of tiles so the frame-overlap re-reads stay inside one L2.
  const int nb = gridDim.x;
  int lb = blockIdx.x;
  if ((nb & 7) == 0) lb = (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3);
  // this workgroup's run of tiles; its waves claim them one at a time from the LDS queue, so a
  // wave that the scheduler favours simply takes more tiles (static shares left 40 % idle tails)
  unsigned blk_first = (unsigned)lb * (unsigned)tiles_per_block;
  unsigned blk_count = 0;
  if (blk_first < (unsigned)n_tiles) {
    blk_count = (unsigned)n_tiles - blk_first;
    if (blk_count > (unsigned)tiles_per_block) blk_count = (unsigned)tiles_per_block;
  }
  if (EPI == EPI400_MFCC && epi.fixup != 0) {             // the entries of this workgroup's own list
    blk_first = 0;
    blk_count = fix_cnt;
  }
  auto claim = [&]() {   // wave-uniform
    int v = 0;
    if (lane == 0) v = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return (unsigned)__builtin_amdgcn_readfirstlane(v);
  };
#if AAMD_M400_POOLS
  // tail pools (see pool_tile): with a pool, blk_count shrinks to the workgroup's OWN tiles; every later queue index is a ticket of
  // the pool.  Everything else about the pool is formed where a ticket is drawn (once or twice per wave and launch) from an opaque
  // copy of the workgroup number: held across the tile loop, those ten scalars cost the MFCC instantiation 45 more spilled SGPRs.
#define AAMD_M400_POOLED (AAMD_M400_POOLS && (LAB == 0) && epi.pool != nullptr && !(EPI == EPI400_MFCC && epi.fixup != 0))   /* (not held in a register pair) */
  if (AAMD_M400_POOLED && blk_count > (unsigned)(tiles_per_block - epi.pool_p)) blk_count = (unsigned)(tiles_per_block - epi.pool_p);
  unsigned pool_ticket = 0;      // lane 0: the ticket in flight
  auto pool_request = [&]() {
    int lbo = (int)(blk_first / (unsigned)tiles_per_block);    // = lb (not kept across the loop)
    asm volatile("" : "+s"(lbo));
    unsigned* const ctr = epi.pool + (lbo % pool_count(nb)) * kPoolStride;
    if (lane == 0) pool_ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // the ticket requested earlier -> queue index of its tile (global tile - blk_first, modulo 2^32) and whether there is one
  auto pool_take = [&]() {
    int lbo = (int)(blk_first / (unsigned)tiles_per_block);
    asm volatile("" : "+s"(lbo));
    const int np = pool_count(nb), pid = lbo % np, mem = pool_members(nb, np, pid);
    for (;;) {
      const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)pool_ticket);
      if (v + 1u == (unsigned)(mem * (epi.pool_p + kWavesPerBlock)) && lane == 0)      // the launch's last ticket of this pool
        __hip_atomic_store(epi.pool + pid * kPoolStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool exhausted;
      const unsigned t = pool_tile(np, epi.pool_p, tiles_per_block, (unsigned)n_tiles, pid, mem, v, exhausted);
      if (t != kNoTile) return t - blk_first;
      if (exhausted) return 0x80000000u;          // (= kBadIdx below)
      pool_request();            // a slot past the end of the batch (last workgroup's run only): draw again
    }
  };
  // queue indices: idx < blk_count = the workgroup's own tiles; a pool tile t travels as t - blk_first (modulo 2^32); kBadIdx = no
  // tile, kPendIdx = a pool ticket is to be drawn -- neither can be a difference of two tile numbers (n_tiles < 2^31)
  constexpr unsigned kBadIdx = 0x80000000u, kPendIdx = 0x80000001u;
#define AAMD_M400_IDX_OK(I) ((((I) ^ 0x80000000u)) > 1u)
#define AAMD_M400_LOCAL(I) ((I) < blk_count ? (I) : kBadIdx)
#else
#define AAMD_M400_IDX_OK(I) ((I) < blk_count)
#define AAMD_M400_LOCAL(I) (I)
#endif
  // lab bit 17: chip-wide queue.  A wave owns a chunk of kLabChunk consecutive tiles; the ticket of its NEXT chunk is
  // requested when a chunk starts and read when it ends (the atomic's round trip hides behind kLabChunk tiles).
  constexpr unsigned kLabChunk = 8;
  unsigned* gq = reinterpret_cast<unsigned*>(epi.group_max);
  int g_pending = 0;            // lane 0: ticket of the next chunk (in flight)
  unsigned g_base = 0, g_pos = 0;
  auto g_request = [&]() {
    if (lane == 0) g_pending = (int)__hip_atomic_fetch_add(gq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (LAB & 131072) {
    blk_count = (unsigned)n_tiles;                       // indices below are global tile numbers
    g_request();
    g_base = (unsigned)__builtin_amdgcn_readfirstlane(g_pending) * kLabChunk;
    g_request();
  }
  auto g_next = [&]() {          // global index of the tile after the current one
    if (++g_pos < kLabChunk) return g_base + g_pos;
    g_base = (unsigned)__builtin_amdgcn_readfirstlane(g_pending) * kLabChunk;
    g_pos = 0;
    g_request();
    return g_base;
  };

  auto tile_info = [&](unsigned idx) {
    TileInfo ti;
    unsigned t = ((LAB & 131072) ? 0u : blk_first) + idx;
    if (EPI == EPI400_MFCC && epi.fixup != 0)   // (an L1-bypassing load: other waves of this workgroup wrote the entry)
      t = idx < blk_count ? (unsigned)__hip_atomic_load(epi.fix_list + fix_base + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const unsigned row = t / (unsigned)tiles_per_row;
    ti.row = row;
    ti.t0 = (int64_t)(t - row * (unsigned)tiles_per_row) * kFramesPerWave;
    ti.staged = AAMD_M400_IDX_OK(idx) && in_aligned && (ti.t0 * kHop - kPad >= 0) &&
                ((ti.t0 + kFramesPerWave - 1) * kHop + (kN - kPad) <= length) &&
                (ti.t0 + kFramesPerWave <= n_frames);
    return ti;
  };
  auto stage_issue = [&](const TileInfo& ti) {
    const TIn* src = wav + (ti.row / InTraits<TIn>::chans) * row_stride + (ti.t0 * kHop - kPad);   // stereo: rows 2 i, 2 i + 1 = clip i
    if (LAB & 32) src = wav + 6 * kHop;                 // lab: always the same (cache-resident) tile
#pragma unroll
    for (int k = 0; k < SG::ndma; ++k)
      if (!(LAB & 64) || k == 0) {                      // lab bit 6: one piece only
#if AAMD_M400_DMA_SADDR
        glds16s(src, (unsigned)spiece[k] * (unsigned)sizeof(TIn), s_addr + 1024 * k);
#else
        glds16(src + spiece[k], s_addr + 1024 * k);
#endif
      }
  };

  // MEL_DB: running maximum of this wave's dB values, flushed whenever the cut-off group changes
  float wmax = -INFINITY;
  int64_t wgroup = -1, wrow = -1;
  constexpr bool kDb = (EPI == EPI400_MEL_DB) || (EPI == EPI400_MFCC);
  const bool fix = (EPI == EPI400_MFCC) && epi.fixup != 0;      // kernel-uniform: the fix-up pass of the fused MFCC
  auto flush_max = [&]() {
    if (kDb && !fix && epi.group_max != nullptr && wgroup >= 0) {
      float m = wmax;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if (lane == 0) atomic_max_f32(epi.group_max + wgroup, m);
    }
    wmax = -INFINITY;
  };

  if (kIoWave && wave >= kWavesPerBlock - kNio) {
    constexpr unsigned D = 8;                              // tiles in flight per I/O wave
    constexpr bool kFetchOnly = (LAB & 134217728) != 0;
    if (!(LAB & 33554432)) {
      const unsigned n_mine = (blk_count + (unsigned)kNio - 1u - (unsigned)(wave - (kWavesPerBlock - kNio))) / (unsigned)kNio;
      for (unsigned i = 0; i < n_mine + D; ++i) {
        const unsigned t = i * (unsigned)kNio + (unsigned)(wave - (kWavesPerBlock - kNio));
        if (i < n_mine) {
          const TileInfo ti = tile_info(AAMD_M400_LOCAL(t));
          const unsigned dst = (unsigned)(uintptr_t)(smem400 + (t % (unsigned)kWavesPerBlock) * HC::lds_dwords + kSOff);
          if (ti.staged) {
            const TIn* src = wav + (ti.row / InTraits<TIn>::chans) * row_stride + (ti.t0 * kHop - kPad);
#pragma unroll
            for (int k = 0; k < SG::ndma; ++k) glds16(src + spiece[k], dst + 1024 * k);
          }
        }
        if (i >= D) {
          if (kFetchOnly) {
            asm volatile("s_waitcnt vmcnt(40)" ::: "memory");   // 5 D younger operations
          } else {
            asm volatile("s_waitcnt vmcnt(56)" ::: "memory");   // 7 D younger operations: the pieces of the tile D back have landed
            const unsigned told = t - D * (unsigned)kNio;
            const TileInfo to = tile_info(AAMD_M400_LOCAL(told));
            const float* stg = smem400 + (told % (unsigned)kWavesPerBlock) * HC::lds_dwords;
            store_wide(lane, mt, stg, out + to.row * (int64_t)n_frames * (int64_t)mb.n_mels, to.t0, n_frames);
          }
        }
      }
    }
    blk_count = 0;                                         // (this wave's copy: it runs no tile)
  }
  unsigned cur_idx = (LAB & 131072) ? g_base : (unsigned)wave;
#if AAMD_M400_POOLS
  if (cur_idx >= blk_count) cur_idx = kBadIdx;
  if (AAMD_M400_POOLED && cur_idx == kBadIdx) {       // fewer own tiles than waves (the last workgroups of a launch): straight to the pool
    pool_request();
    cur_idx = pool_take();
  }
#endif
  TileInfo cur = tile_info(cur_idx);
  if (LAB & 128) {   // lab: stagger the waves of a SIMD by thirds of a tile time.  Interleaved A/B runs
    // (tools/ubench/mel400_lab) put it within noise of the lock-step start (74-78 us either way): off.
    const int k = (wave + 6 * lb) % 3;
    for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(112);
  }
  if (LAB & 2048) {  // lab: 12 distinct phase offsets, 1/12 of a tile time apart
    const int k = 4 * (wave & 3) + (wave >> 2) % 3;   // waves w, w+4, w+8 share a SIMD: a third apart
    for (int i = 0; i < (k % 12); ++i) __builtin_amdgcn_s_sleep(17);
  }
  if (LAB & 4096) {  // lab: thirds within a SIMD + a twelfth between SIMDs
    const int k = 4 * ((wave >> 2) % 3) + (wave & 3);
    for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(17);
  }
  float Xn[HC::nx];   // lab bit 15: the next tile's samples, in flight
  auto gload = [&](const TileInfo& ti) {
    const TIn* src = wav + (ti.row / InTraits<TIn>::chans) * row_stride + (ti.t0 * kHop - kPad) + (HC::pair_stride * c.p + c.pi);
#pragma unroll
    for (int q = 0; q < HC::nx; ++q) Xn[q] = InTraits<TIn>::get(src + 20 * q, (int)(ti.row % InTraits<TIn>::chans));
  };
  if (LAB & 32768) {
#pragma unroll
    for (int q = 0; q < HC::nx; ++q) Xn[q] = 0.0f;
    if (cur.staged) gload(cur);
  } else if (cur.staged && !(LAB & 8) && !fix) {
    stage_issue(cur);
  }

  // lab bit 23 (tools/mel400_lab.py phases): shader cycles per phase, summed over the wave's tiles.  The stamps sit where the
  // kernel drains its LDS counter anyway (s_memtime returns through the same counter), so they move little.
  long long ph_acc[6] = {0, 0, 0, 0, 0, 0}, ph_t = 0;
  int ph_tiles = 0;
#define AAMD_M400_STAMP(K)                                  \
  if (LAB & 8388608) {                                      \
    const long long now_ = (long long)clock64();            \
    ph_acc[K] += now_ - ph_t;                               \
    ph_t = now_;                                            \
  }
  // MFCC: accumulators of the DCT product (C row 4 (l >> 4) + r = coefficient, column l & 15 = frame / plane)
  using mfcc_f32x4 = __attribute__((ext_vector_type(4))) float;
  mfcc_f32x4 cf[kMfccMT];
  const int mfcc_elab = (LAB & 524288) ? epi.lab : 0;
  auto mfcc_finish = [&](int64_t row, int64_t t0) {
#pragma unroll
    for (int t = 0; t < kMfccMT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {     // lane j += lane j + 8 of its row (row_ror:8)
        cf[t][i] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(cf[t][i]), 0x128, 0xf, 0xf, true));
        asm volatile("" : "+v"(cf[t][i]));     // keeps the add beside its DPP move (one v_add_f32_dpp): without it the add sinks
                                               // into the EXEC-masked store blocks and the move stays behind as an instruction
      }
    // stores: lane (j = l & 15 < 6, g = l >> 4) holds coefficients 16 t + 4 g .. + 3 of frame j.  One wave-uniform tile
    // pointer + a 32-bit lane offset that does not change from tile to tile (no per-lane 64-bit pointer arithmetic).
    const int64_t left64 = n_frames - t0;
    const int left = left64 < kFramesPerWave ? (int)left64 : kFramesPerWave;     // live frames of this tile (wave-uniform)
    if (!(LAB & 2) && !(mfcc_elab & 8) && (lane & 15) < left) {
      float* otile = out + (row * (int64_t)n_frames + t0) * (int64_t)epi.n_mfcc;       // wave-uniform
      const unsigned ooff = (unsigned)(lane & 15) * (unsigned)epi.n_mfcc + 4u * (unsigned)(lane >> 4);
#pragma unroll
      for (int t = 0; t < kMfccMT; ++t) {
        const int k0 = 16 * t + 4 * (lane >> 4);
        if (k0 < epi.n_mfcc) *reinterpret_cast<F4*>(otile + (ooff + 16u * (unsigned)t)) = F4{cf[t][0], cf[t][1], cf[t][2], cf[t][3]};
      }
    }
  };
  while (AAMD_M400_IDX_OK(cur_idx)) {
    if (LAB & 8388608) { ph_t = (long long)clock64(); ++ph_tiles; }
    // claim the tile after this one now: it is prefetched while this one is in its second half
    // (issuing the LDS atomic here and reading its ticket behind the column reads' wait moved 70 cycles from phase A to
    // phase B and nothing else: profiles/r03_zz_mel400_phase_census.txt)
#if AAMD_M400_POOLS
    unsigned nxt_idx = (LAB & 131072) ? g_next() : claim();
    if (nxt_idx >= blk_count) nxt_idx = AAMD_M400_POOLED ? kPendIdx : kBadIdx;   // (the queue is empty: with pools, a ticket is drawn below)
#else
    const unsigned nxt_idx = (LAB & 131072) ? g_next() : claim();
#endif
    TileInfo nxt = tile_info(nxt_idx);
    float fix_cut = -INFINITY;
    if (fix) {
      // fix-up pass: the tiles of the compacted list (smallest dB value under the cut-off) are redone, clamped; their samples
      // are staged now, without prefetch: flagged tiles are the exception
      fix_cut = epi.group_max[cur.row / epi.rows_per_group] - epi.top_db;
      if (cur.staged) stage_issue(cur);
    }

    float X[HC::nx];
    if ((LAB & 32768) && cur.staged) {
#pragma unroll
      for (int q = 0; q < HC::nx; ++q) X[q] = Xn[q];
    } else if (cur.staged) {
      if (!(LAB & 1)) stage_wait();
      if (LAB & 256) {   // lab: no gather (samples from a register expression)
#pragma unroll
        for (int q = 0; q < HC::nx; ++q) X[q] = (float)(q + lane) * scale;
      } else {
        gather_lds<H, TIn>(c, reinterpret_cast<const TIn*>(lds + kSOff), X, (int)(cur.row % InTraits<TIn>::chans));
      }
    } else {
      gather_global<H, TIn>(c, wav + (cur.row / InTraits<TIn>::chans) * row_stride, length, cur.t0, n_frames, X,
                            (int)(cur.row % InTraits<TIn>::chans));
    }
#if AAMD_M400_POOLS
    // (behind stage_wait's vmcnt(0): the atomic flies during phase A and the column reads and is read in front of the next DMA)
    if (nxt_idx == kPendIdx) pool_request();
#endif
    phase_a<H, kWinRegs, kTwRegBatches>(c, X, lds, winr, twr);
    if ((LAB & 32768) && nxt.staged) gload(nxt);        // X is dead: the next tile's samples fly during phases B and C
    wave_lds_fence();
    AAMD_M400_STAMP(0)   // claim, gather, window, first DFT-20, twiddles, transposed writes
    float vr[20], vi[20], zr[20], zi[20], qr[10], qi[10];
    phase_b1_load(c, lds, vr, vi);
    if (!(LAB & 32768)) {
      // every transposition row has been read: the staging area (it aliases rows) is free again
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      AAMD_M400_STAMP(1)   // column reads
#if AAMD_M400_POOLS
      if (nxt_idx == kPendIdx) {
        nxt_idx = pool_take();
        nxt = tile_info(nxt_idx);
      }
#endif
      if (nxt.staged && !(LAB & 8) && !fix) stage_issue(nxt);
    }
    if (LAB & 16) { wave_lds_fence(); cur = nxt; cur_idx = nxt_idx; continue; }
    dft20(vr, vi, zr, zi);
    phase_b2_send(c, zr, zi, qr, qi);
    exchange_partner(qr, qi, self_mask);
    wave_lds_fence();
    AAMD_M400_STAMP(2)     // DMA issue, second DFT-20, partner exchange
    if (EPI == EPI400_SPEC) {
      const int64_t left = n_frames - cur.t0;
      const int n_valid = left < kFramesPerWave ? (int)left : kFramesPerWave;
      if (epi.power > 0.0f) {
        const int64_t a0 = (cur.row * n_frames + cur.t0) * (int64_t)kSpecBins;
        phase_b2_spec(c, zr, zi, qr, qi, epi.power, (int)(a0 & 3), lds);
        wave_lds_fence();
        if (!(LAB & 2)) {
          if (AAMD_M400_SPEC_FULL && n_valid == kFramesPerWave) store_spec_full(lane, lds, out, a0);
          else store_spec(lane, lds, out, a0, n_valid * kSpecBins);
        }
        wave_lds_fence();
      } else {   // complex output, two halves of 3 frames
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int64_t a0 = (cur.row * n_frames + cur.t0 + 3 * half) * (int64_t)(2 * kSpecBins);
          const int nv = n_valid - 3 * half < 3 ? n_valid - 3 * half : 3;
          phase_b2_spec_complex(c, zr, zi, qr, qi, half, (int)(a0 & 3), lds);
          wave_lds_fence();
          if (!(LAB & 2) && nv > 0) store_spec(lane, lds, out, a0, nv * 2 * kSpecBins);
          wave_lds_fence();
        }
      }
      cur = nxt;
      cur_idx = nxt_idx;
      continue;
    }
    phase_b2(c, zr, zi, qr, qi, lds);
    phase_b2_pad(lane, lds);
    wave_lds_fence();
    AAMD_M400_STAMP(3)     // power, P rows
    float acc_a[NR], acc_b[NR];
    if (LAB & 4) { wave_lds_fence(); cur = nxt; cur_idx = nxt_idx; continue; }
    phase_c<NR, SIG, EPI == EPI400_MFCC>(c, mt, lds, acc_a, acc_b, mh);
    if (LAB & 8388608) { asm volatile("" : "+v"(acc_a[0]), "+v"(acc_b[NR - 1])); }
    AAMD_M400_STAMP(4)     // band reduction
    if (kDb) {
      if (cur.row != wrow) {        // (a 64-bit division: once per row of the wave's run, not once per tile)
        wrow = cur.row;
        const int64_t g = cur.row / epi.rows_per_group;   // wave-uniform
        if (g != wgroup) {
          flush_max();
          wgroup = g;
        }
      }
      // frames past the end of the clip hold garbage (phase_a): keep them out of the maximum
      const bool va_ok = cur.t0 + 2 * c.p < n_frames, vb_ok = cur.t0 + 2 * c.p + 1 < n_frames;
      float tmin = INFINITY;
      if (EPI == EPI400_MFCC) {
        // 80 mels are exactly NR = 4 rounds in which every table row holds a mel (mfcc_fused_ok, checked at kernel entry):
        // no `round exists` / `row holds a mel` tests; the accumulators are sums of products (canonical), so the amin clamp
        // is a bare v_max_f32 (fmaxf puts a v_max x, x, x in front of each: the compiler cannot see through phase C's switch);
        // y = (multiplier log10 2) log2 x - db_sub in one fma (epi_db multiplies twice: <= 1 ulp of y apart, 8e-6 dB at 100 dB)
        const float db_c1 = epi.multiplier * 0.30102999566398120f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          acc_a[r] = __builtin_fmaf(__log2f(vmax_raw(acc_a[r], epi.amin)), db_c1, -epi.db_sub);
          acc_b[r] = __builtin_fmaf(__log2f(vmax_raw(acc_b[r], epi.amin)), db_c1, -epi.db_sub);
        }
        if (cur.t0 + kFramesPerWave <= n_frames) {       // interior tile (wave-uniform; 166 of a 10 s clip's 167): six live frames
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            wmax = vmax3_raw(wmax, acc_a[r], acc_b[r]);
            tmin = vmin3_raw(tmin, acc_a[r], acc_b[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            wmax = fmaxf(wmax, fmaxf(va_ok ? acc_a[r] : -INFINITY, vb_ok ? acc_b[r] : -INFINITY));
            tmin = fminf(tmin, fminf(va_ok ? acc_a[r] : INFINITY, vb_ok ? acc_b[r] : INFINITY));
          }
        }
        if (fix) {
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            acc_a[r] = vmax_raw(acc_a[r], fix_cut);
            acc_b[r] = vmax_raw(acc_b[r], fix_cut);
          }
        }
      } else {
        // MEL_DB: the same two savings (round 6) -- a bare v_max_f32 for the amin clamp (epi_db's arithmetic otherwise), and no
        // "frame exists" selects in interior tiles
        const bool interior = cur.t0 + kFramesPerWave <= n_frames;       // wave-uniform
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if (r < mt.n_rounds) {
            acc_a[r] = epi.multiplier * fast_log10(vmax_raw(acc_a[r], epi.amin)) - epi.db_sub;
            acc_b[r] = epi.multiplier * fast_log10(vmax_raw(acc_b[r], epi.amin)) - epi.db_sub;
            if (interior) wmax = vmax3_raw(wmax, acc_a[r], acc_b[r]);
            else wmax = fmaxf(wmax, fmaxf(va_ok ? acc_a[r] : -INFINITY, vb_ok ? acc_b[r] : -INFINITY));
          }
        }
      }
      if (EPI == EPI400_MFCC && !fix && !(elab & 4)) {
        // min-scan on DPP operands (VALU; a butterfly of __shfl_xor would be 6 LDS round trips): lane 63 ends with the minimum.
        // One v_min_f32_dpp per step: a lane without a source keeps its value (bound_ctrl off), and the values are results of
        // arithmetic (canonical), so none of the v_mov / v_max x, x, x the compiler wraps around fminf(update_dpp(..)) is needed;
        // s_nop 1 = the two wait states between a VALU write and a DPP read of the same register.
        asm volatile(
            "s_nop 1\n\t"
            "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
            "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
            : "+v"(tmin));
        if (lane == 63) epi.tile_min[blk_first + cur_idx] = tmin;
      }
    }
    if (EPI == EPI400_MFCC) {
      // the DCT product on the f16 matrix pipe: A fragments (hi / lo planes) from the cache-resident table, B from the
      // staged dB rows, which are written as two binary16 planes (hi, lo) of 6 x 80 halves each
      using f32x4 = __attribute__((ext_vector_type(4))) float;
      using h8 = __attribute__((ext_vector_type(8))) _Float16;
      using h4 = __attribute__((ext_vector_type(4))) _Float16;
      using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
      using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
      wave_lds_fence();
      {   // stage: v = y / 16 = hi + lo.  (Lanes 60 .. 63 shadow lanes 40 .. 43: the same halves to the same addresses.)
        uint16_t* hs = reinterpret_cast<uint16_t*>(lds);
        uint16_t* ls = hs + kMfccPlaneHalves;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int m = mh ? mh->mel(r) : mt.row_mel[r * kMelSlots + c.pi];
          const float va = acc_a[r] * kMfccYScale, vb = acc_b[r] * kMfccYScale;
          const _Float16 ha = (_Float16)va, hb = (_Float16)vb;
          const _Float16 la = (_Float16)(va - (float)ha), lb = (_Float16)(vb - (float)hb);
          hs[2 * c.p * kMfccMels + m] = __builtin_bit_cast(uint16_t, ha);
          ls[2 * c.p * kMfccMels + m] = __builtin_bit_cast(uint16_t, la);
          hs[(2 * c.p + 1) * kMfccMels + m] = __builtin_bit_cast(uint16_t, hb);
          ls[(2 * c.p + 1) * kMfccMels + m] = __builtin_bit_cast(uint16_t, lb);
        }
      }
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < kMfccMT; ++t) cf[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      // The product (round 6: 18 matrix instructions instead of 27).  The six live frames fill 6 of the 16 B columns, so the
      // hi plane goes to columns 0 .. 5 and the lo plane to columns 8 .. 13 of ONE operand: A_lo x B and A_hi x B then leave
      // (A_hi + A_lo) B_hi in lane j and (A_hi + A_lo) B_lo in lane j + 8 of every row of 16 lanes (all four terms; the
      // three-term form dropped lo x lo), and one DPP add per accumulator register joins them.  One B read per step instead of two.
      // The fragment table is read through a pointer of its own address space (kFragLds, per instantiation: a pointer selected
      // at run time made every one of the 18 reads a flat_load).
      auto product = [&](const u32x4* __restrict__ ftab) {
        const uint16_t* hs = reinterpret_cast<const uint16_t*>(lds);
        u32x4 a0[kMfccMT][2];
        auto frag_load = [&](int sidx) {      // step sidx: 3 coefficient tiles x (hi, lo), 16 B each
#pragma unroll
          for (int t = 0; t < kMfccMT; ++t)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
              a0[t][hl] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
              if (!(elab & 1)) a0[t][hl] = ftab[mfcc_frag_piece(t, sidx, hl, lane)];
            }
        };
        // consecutive MFMAs go to different accumulators (a dependent one would wait out the passes of its predecessor)
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx) {
          const h8 bb = __builtin_bit_cast(h8, *reinterpret_cast<const u32x4*>(hs + mfcc_b_index(lane, sidx)));
          frag_load(sidx);
#pragma unroll
          for (int t = 0; t < kMfccMT; ++t)
            cf[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a0[t][1]), bb, cf[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < kMfccMT; ++t)
            cf[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a0[t][0]), bb, cf[t], 0, 0, 0);
        }
        {   // mels 64 .. 79: K = 16 (4 halves per lane: the low 8 bytes of the fragment pieces)
          frag_load(2);
          const h4 bb = __builtin_bit_cast(h4, *reinterpret_cast<const u32x2*>(hs + mfcc_b_index(lane, 2)));
#define AAMD_A4(T, HL) __builtin_bit_cast(h4, u32x2{a0[T][HL][0], a0[T][HL][1]})
#pragma unroll
          for (int t = 0; t < kMfccMT; ++t) cf[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(AAMD_A4(t, 1), bb, cf[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < kMfccMT; ++t) cf[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(AAMD_A4(t, 0), bb, cf[t], 0, 0, 0);
#undef AAMD_A4
        }
      };
      if (!(elab & 2)) {
        if (kFragLds) product(reinterpret_cast<const u32x4*>(frag_lds));
        else product(reinterpret_cast<const u32x4*>(epi.dct_frag));
      }
      // (Round 6 also deferred this join + store behind the NEXT tile's claim and gather, so that the wave does not sit out its
      // dependent chain of matrix instructions: no gain, 179.8 against 177.3 us -- profiles/r06_v_mfcc_lab_defer.txt.)
      if (!(elab & 2)) mfcc_finish(cur.row, cur.t0);
      wave_lds_fence();
      cur = nxt;
      cur_idx = nxt_idx;
      continue;
    }
    if (EPI == EPI400_MEL_NORM) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r < mt.n_rounds) {
          acc_a[r] = (epi_plog(acc_a[r] * epi.gain) - nrm_mean[r]) * nrm_inv[r];
          acc_b[r] = (epi_plog(acc_b[r] * epi.gain) - nrm_mean[r]) * nrm_inv[r];
        }
      }
    }
    float* out_row = out + cur.row * (EPI == EPI400_MEL_NORM ? epi.out_frames : (int64_t)n_frames) * (int64_t)mb.n_mels;
    if (LAB & 4194304) out_row = out + (int64_t)(wave + kWavesPerBlock * (blockIdx.x & 63)) * 6 * mb.n_mels - cur.t0 * (int64_t)mb.n_mels;   // lab bit 22: every store of a wave goes to one cache-resident tile
    if (out_wide) {
      wave_lds_fence();
      store_stage<NR>(c, mt, acc_a, acc_b, lds, mh);
      wave_lds_fence();
      if (!(LAB & 2)) store_wide(lane, mt, lds, out_row, cur.t0, n_frames);
    } else {
      if (!(LAB & 2)) store_direct<NR, SIG>(c, mt, acc_a, acc_b, out_row, cur.t0, n_frames, mh);
    }
    wave_lds_fence();
    AAMD_M400_STAMP(5)     // stores
    cur = nxt;
    cur_idx = nxt_idx;
  }
#undef AAMD_M400_STAMP
  if ((LAB & 8388608) && lane == 0) {
    long long* rec = reinterpret_cast<long long*>(epi.fix_count) + 4 * 12 * 256 + 8 * (blockIdx.x * kWavesPerBlock + wave);
#pragma unroll
    for (int k = 0; k < 6; ++k) rec[k] = ph_acc[k];
    rec[6] = ph_tiles;
  }
  if (kDb && !fix && epi.group_max != nullptr) {
    // final flush through LDS: one atomic per workgroup and group instead of one per wave
    // (thousands of same-address atomics from every XCD serialise at one L2 channel)
    float m = wmax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) {
      reinterpret_cast<int*>(lds)[0] = (int)wgroup;
      lds[1] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int g_run = -1;
      float m_run = -INFINITY;
      for (int w = 0; w < kWavesPerBlock; ++w) {
        const int gw = reinterpret_cast<const int*>(smem400 + w * HC::lds_dwords)[0];
        const float mw = smem400[w * HC::lds_dwords + 1];
        if (gw != g_run) {
          if (g_run >= 0) atomic_max_f32(epi.group_max + g_run, m_run);
          g_run = gw;
          m_run = mw;
        } else {
          m_run = fmaxf(m_run, mw);
        }
      }
      if (g_run >= 0) atomic_max_f32(epi.group_max + g_run, m_run);
    }
  }
  if ((LAB & 1048576) && lane == 0) {  // lab (tools/mel400_lab.py): cycle counter and 100 MHz wall clock at the wave's exit
    long long* rec = reinterpret_cast<long long*>(epi.fix_count);
    const int w = blockIdx.x * kWavesPerBlock + wave;
    rec[4 * w + 2] = (long long)clock64();
    rec[4 * w + 3] = wall_clock64();
  }
  if ((LAB & 1024) && lane == 0) {   // lab census: per-wave entry / tables ready / done (100 MHz clock)
    long long* rec = reinterpret_cast<long long*>(const_cast<float*>(tw400) + 1024);
    const int w = blockIdx.x * kWavesPerBlock + wave;
    rec[3 * w] = lab_t0;
    rec[3 * w + 1] = lab_t1;
    rec[3 * w + 2] = wall_clock64();
  }
}
#endif  // __HIPCC__

}  // namespace m400
}  // namespace aamd

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
  int chunks_per_row;
  int64_t n_chunks;
  int chunks_per_block;
  int buf_floats;            // floats per LDS chunk buffer (multiple of 4)
  int vec_in, vec_out;       // 16-B global loads / stores are legal
  int tap_lo[kMaxPhaseTiles];
};

// k-steps needed for a band of `span` taps, from the supported set (all = 16 mod 32); 0 = too wide
AAMD_HD int pick_ks(int span) {
  const int need = (span + 3) / 4;
  if (need <= 16) return 16;
  if (need <= 48) return 48;
  if (need <= 80) return 80;
  if (need <= 112) return 112;
  return 0;
}
AAMD_HD int max_compute_waves(int ks) { return ks >= 80 ? 10 : 14; }

AAMD_HD int chunk_q(const Geom& g) { return kQPerGroup * g.qg; }

// floats one chunk buffer must hold: every index a compute wave can read, + 3 for the 16-B phase
AAMD_HD int buf_floats_needed(int qc, int orig, int taps, int max_tap_lo, int ks) {
  int reach = max_tap_lo + 4 * ks;
  if (reach < taps) reach = taps;
  return (((qc - 1) * orig + reach + 3) + 3) & ~3;
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
  const float c[4] = {c0, c1, c2, c3};
  for (int i = 0; i < 4; ++i)
    if (p0 + i < g.new_ && oi + i < g.out_len) out_row[oi + i] = c[i];
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
#endif  // __HIPCC__

}  // namespace rsm
}  // namespace aamd

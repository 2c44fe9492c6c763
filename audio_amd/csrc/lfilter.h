// IIR filtering (F.lfilter / F.biquad cascades) as a chunked linear-recurrence scan.
//
// Reference semantics (functional/filtering.py:1027-1099, libtorchaudio/lfilter.cpp:17-48):
//   b^ = b/a0, a^ = a/a0;  w[n] = sum_k b^[k] x[n-k];  y[n] = w[n] - sum_{k>=1} a^[k] y[n-k]
//   zero initial state; clamp(y,-1,1) AFTER the recursion (never fed back).
// The reference CUDA kernel (iir_cuda.cu:10-35) runs one serial thread per sequence.  Here a
// 256-thread workgroup owns one (batch, channel) sequence and walks it in blocks of
// 256*CH samples staged in LDS:
//   1. thread i filters chunk i (CH samples) from ZERO state, all in registers
//      (chunk 0 starts from the carried true state);
//   2. the chunk-end states are combined with a Hillis-Steele scan using the precomputed
//      powers M^(2^k) of the CH-step state-transition matrix (D x D, D = order);
//   3. each chunk adds the homogeneous response H[j][:] . S_{i-1} of the true state entering it;
//   4. clamp, write back in place -> the block is the next cascade stage's input.
// A cascade of n_stages filters therefore reads x once and writes y once.
#pragma once
#include "hd.h"

namespace aamd {

constexpr int kLfThreads = 256;
constexpr int kLfChunk = 32;                 // CH
constexpr int kLfChunkStride = kLfChunk + 1; // LDS padding -> conflict-free per-thread rows
constexpr int kLfBlock = kLfThreads * kLfChunk;
constexpr int kLfScanSteps = 8;              // log2(256)

template <int D>
struct LfThread {   // per-thread registers that live across phases
  float z[kLfChunk];
  float s[D];        // running scan value (state at end of chunk)
};

// LDS carve-up (floats)
template <int D>
struct LfLds {
  static constexpr int blk = 0;                                     // kLfThreads*kLfChunkStride
  static constexpr int scanA = blk + kLfThreads * kLfChunkStride;   // kLfThreads*D
  static constexpr int scanB = scanA + kLfThreads * D;
  static constexpr int H = scanB + kLfThreads * D;                  // kLfChunk*D
  static constexpr int Mp = H + kLfChunk * D;                       // kLfScanSteps*D*D
  static constexpr int ah = Mp + kLfScanSteps * D * D;              // D+1 (a^)
  static constexpr int bh = ah + (D + 1);                           // D+1 (b^)
  static constexpr int cx = bh + (D + 1);                           // carried inputs  [D]
  static constexpr int cy = cx + D;                                 // carried outputs [D]
  static constexpr int total = cy + D;
};

// ---- tables for one stage: run by ONE thread (tid 0), double precision -------------------
template <int D>
AAMD_HD void lf_build_tables(const float* a_row, const float* b_row, int n_order, float* lds) {
  using L = LfLds<D>;
  float* ah = lds + L::ah;
  float* bh = lds + L::bh;
  const float a0 = a_row[0];
  for (int k = 0; k <= D; ++k) {
    ah[k] = (k < n_order) ? a_row[k] / a0 : 0.0f;   // same fp32 division as the reference
    bh[k] = (k < n_order) ? b_row[k] / a0 : 0.0f;
  }
  double M[D][D];
  for (int d = 0; d < D; ++d) {
    double hist[D];  // hist[e] = y[j-1-e]
    for (int e = 0; e < D; ++e) hist[e] = (e == d) ? 1.0 : 0.0;
    double resp[kLfChunk];
    for (int j = 0; j < kLfChunk; ++j) {
      double y = 0.0;
      for (int k = D; k >= 1; --k) y -= (double)ah[k] * hist[k - 1];
      for (int e = D - 1; e > 0; --e) hist[e] = hist[e - 1];
      hist[0] = y;
      resp[j] = y;
      lds[L::H + j * D + d] = (float)y;
    }
    for (int e = 0; e < D; ++e) M[e][d] = resp[kLfChunk - 1 - e];
  }
  for (int k = 0; k < kLfScanSteps; ++k) {
    for (int e = 0; e < D; ++e)
      for (int d = 0; d < D; ++d) lds[L::Mp + (k * D + e) * D + d] = (float)M[e][d];
    double M2[D][D];
    for (int e = 0; e < D; ++e)
      for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int f = 0; f < D; ++f) acc += M[e][f] * M[f][d];
        M2[e][d] = acc;
      }
    for (int e = 0; e < D; ++e)
      for (int d = 0; d < D; ++d) M[e][d] = M2[e][d];
  }
}

// ---- phase 1: chunk pass from zero state (chunk 0: from the carried state) ------------------
template <int D>
AAMD_HD void lf_chunk_pass(int tid, float* lds, LfThread<D>& th) {
  using L = LfLds<D>;
  const float* ah = lds + L::ah;
  const float* bh = lds + L::bh;
  const float* blk = lds + L::blk;
  float hu[D], hz[D];  // hu[e] = u[j-1-e], hz[e] = z[j-1-e]
#pragma unroll
  for (int e = 0; e < D; ++e) {
    if (tid == 0) {
      hu[e] = lds[L::cx + e];
      hz[e] = lds[L::cy + e];
    } else {
      // the D samples before this chunk live at the tail of chunk tid-1 (D <= CH)
      hu[e] = blk[(tid - 1) * kLfChunkStride + (kLfChunk - 1 - e)];
      hz[e] = 0.0f;
    }
  }
  const float* mine = blk + tid * kLfChunkStride;
#pragma unroll
  for (int j = 0; j < kLfChunk; ++j) {
    const float u = mine[j];
    float w = 0.0f;
#pragma unroll
    for (int k = D; k >= 1; --k) w += bh[k] * hu[k - 1];   // oldest tap first
    w += bh[0] * u;
    float y = w;
#pragma unroll
    for (int k = D; k >= 1; --k) y -= ah[k] * hz[k - 1];
#pragma unroll
    for (int e = D - 1; e > 0; --e) { hu[e] = hu[e - 1]; hz[e] = hz[e - 1]; }
    hu[0] = u;
    hz[0] = y;
    th.z[j] = y;
  }
#pragma unroll
  for (int e = 0; e < D; ++e) {
    th.s[e] = hz[e];
    lds[L::scanA + tid * D + e] = hz[e];
  }
}

// save the stage input's last D samples (next block's FIR history); run by tid 0 AFTER the
// chunk pass read them and BEFORE the in-place write-back.
template <int D>
AAMD_HD void lf_save_input_carry(float* lds) {
  using L = LfLds<D>;
  const float* last = lds + L::blk + (kLfThreads - 1) * kLfChunkStride;
  for (int e = 0; e < D; ++e) lds[L::cx + e] = last[kLfChunk - 1 - e];
}

// ---- phase 2: one Hillis-Steele step, offset 2^k, src -> dst ---------------------------------
template <int D>
AAMD_HD void lf_scan_step(int tid, int k, float* lds, LfThread<D>& th, bool src_is_a) {
  using L = LfLds<D>;
  const int off = 1 << k;
  const float* src = lds + (src_is_a ? L::scanA : L::scanB);
  float* dst = lds + (src_is_a ? L::scanB : L::scanA);
  if (tid >= off) {
    const float* Mk = lds + L::Mp + k * D * D;
    const float* v = src + (tid - off) * D;
    float add[D];
#pragma unroll
    for (int e = 0; e < D; ++e) {
      float acc = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) acc += Mk[e * D + d] * v[d];
      add[e] = acc;
    }
#pragma unroll
    for (int e = 0; e < D; ++e) th.s[e] += add[e];
  }
#pragma unroll
  for (int e = 0; e < D; ++e) dst[tid * D + e] = th.s[e];
}

// ---- phase 3: add the homogeneous response of the true entering state, clamp, write back ----
//   `fin` = LDS scan buffer holding the final (true) end-of-chunk states.
template <int D>
AAMD_HD void lf_correct_store(int tid, float* lds, LfThread<D>& th, bool fin_is_a, int clamp) {
  using L = LfLds<D>;
  const float* fin = lds + (fin_is_a ? L::scanA : L::scanB);
  float* mine = lds + L::blk + tid * kLfChunkStride;
  float sp[D];
#pragma unroll
  for (int e = 0; e < D; ++e) sp[e] = (tid > 0) ? fin[(tid - 1) * D + e] : 0.0f;
  const float* H = lds + L::H;
#pragma unroll
  for (int j = 0; j < kLfChunk; ++j) {
    float y = th.z[j];
    if (tid > 0) {
#pragma unroll
      for (int d = 0; d < D; ++d) y += H[j * D + d] * sp[d];
    }
    if (clamp) y = fmin(fmax(y, -1.0f), 1.0f);
    mine[j] = y;
  }
}

// carry the true (unclamped) final state of the block; run by tid 0 after phase 3's barrier
template <int D>
AAMD_HD void lf_save_output_carry(float* lds, bool fin_is_a) {
  using L = LfLds<D>;
  const float* fin = lds + (fin_is_a ? L::scanA : L::scanB);
  for (int e = 0; e < D; ++e) lds[L::cy + e] = fin[(kLfThreads - 1) * D + e];
}

#if defined(__HIPCC__)
// The working slots H..cy hold the CURRENT stage's tables and carried state; a copy per
// stage is parked behind the working set in LDS and swapped in at each stage.
template <int D>
__global__ void __launch_bounds__(kLfThreads)
lfilter_kernel(const float* __restrict__ x, const float* __restrict__ a,
               const float* __restrict__ b, float* __restrict__ y, int64_t n_seq, int channels,
               int64_t length, int n_order, int n_coeff_rows, int n_stages, int clamp) {
  using L = LfLds<D>;
  extern __shared__ __attribute__((aligned(16))) float smem_lf[];
  float* lds = smem_lf;
  // per-stage tables + carries are stored after the working set: stage st at lds_st(st)
  const int stage_floats = L::total - L::H;   // H, Mp, ah, bh, cx, cy
  float* stage_store = lds + L::total;        // n_stages * stage_floats
  const int tid = threadIdx.x;
  LfThread<D> th;

  for (int64_t seq = blockIdx.x; seq < n_seq; seq += gridDim.x) {
    const int ch = (int)(seq % channels);
    const int crow = (n_coeff_rows == 1) ? 0 : ch;
    __syncthreads();
    // thread 0 builds every stage's tables (fp64) and zeroes the carried state
    if (tid == 0) {
      for (int st = 0; st < n_stages; ++st) {
        const int64_t coff = ((int64_t)st * n_coeff_rows + crow) * n_order;
        lf_build_tables<D>(a + coff, b + coff, n_order, lds);
        for (int e = 0; e < D; ++e) { lds[L::cx + e] = 0.0f; lds[L::cy + e] = 0.0f; }
        for (int i = 0; i < stage_floats; ++i) stage_store[st * stage_floats + i] = lds[L::H + i];
      }
    }
    __syncthreads();
    const float* xs = x + seq * length;
    float* ys = y + seq * length;
    for (int64_t n0 = 0; n0 < length; n0 += kLfBlock) {
      // stage the block (zero beyond the end)
      for (int i = tid; i < kLfBlock; i += kLfThreads) {
        const int64_t n = n0 + i;
        lds[L::blk + (i / kLfChunk) * kLfChunkStride + (i % kLfChunk)] = (n < length) ? xs[n] : 0.0f;
      }
      for (int st = 0; st < n_stages; ++st) {
        __syncthreads();
        for (int i = tid; i < stage_floats; i += kLfThreads)
          lds[L::H + i] = stage_store[st * stage_floats + i];
        __syncthreads();
        lf_chunk_pass<D>(tid, lds, th);
        __syncthreads();
        if (tid == 0) lf_save_input_carry<D>(lds);
        bool src_is_a = true;
        for (int k = 0; k < kLfScanSteps; ++k) {
          lf_scan_step<D>(tid, k, lds, th, src_is_a);
          __syncthreads();
          src_is_a = !src_is_a;
        }
        lf_correct_store<D>(tid, lds, th, src_is_a, clamp == 1 || (clamp == 2 && st == n_stages - 1));   // 2: last stage only
        if (tid == 0) lf_save_output_carry<D>(lds, src_is_a);
        __syncthreads();
        for (int i = tid; i < 2 * D; i += kLfThreads)   // persist the carries of this stage
          stage_store[st * stage_floats + (L::cx - L::H) + i] = lds[L::cx + i];
      }
      __syncthreads();
      for (int i = tid; i < kLfBlock; i += kLfThreads) {
        const int64_t n = n0 + i;
        if (n < length) ys[n] = lds[L::blk + (i / kLfChunk) * kLfChunkStride + (i % kLfChunk)];
      }
      __syncthreads();
    }
  }
}
#endif

}  // namespace aamd

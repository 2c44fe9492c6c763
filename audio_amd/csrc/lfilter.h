// IIR filtering (F.lfilter), general order: a chunked linear-recurrence scan carried in FLOAT64.
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
//
// Why float64 inside (round 3): the state of a direct form is D consecutive outputs, and for the clustered poles of any
// sharp design (Chebyshev / elliptic / Butterworth of order >= 6) the powers of the transition matrix have entries many
// orders of magnitude above the outputs they combine into -- in float32 the scan lost 2e-3 (Butterworth 6) to 8e-2
// (Chebyshev 6) of the peak, 20 - 800 x the 1e-4 bar, where the reference's serial float32 loop loses ~1e-4 .. 5e-3.
// Samples stay float32 in HBM and LDS; coefficients are normalised in float32 exactly as the reference does and then
// widened; every accumulation, the tables, the scan and the carried state are double (gfx950 issues a wave64 v_fma_f64
// at the fp32 rate), and the result is rounded to float32 once -- closer to the exact response of the float32 filter than
// the reference's own float32 recursion is.  Biquads and host-factored second-order sections never come here
// (lfilter_wave.h); this is the path of learnable coefficients, orders the host cannot factor safely, and > 8 sections.
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
  double z[kLfChunk];
  double s[D];       // running scan value (state at end of chunk)
};

// LDS carve-up.  The float block comes first, the double tables behind it (offsets in DOUBLES from `tab`).
template <int D>
struct LfLds {
  static constexpr int blk_floats = (kLfThreads * kLfChunkStride + 1) & ~1;   // even: the doubles behind stay 8-aligned
  // scan rows: D doubles per thread at a row stride of SD doubles with SD odd -- 2 SD dwords = 2 (mod 4), so the b64
  // accesses of 32 consecutive threads land on 32 distinct bank pairs (a stride of D doubles is an 8-way (D = 4) to 32-way
  // (D = 16) bank conflict on every read of a scan step)
  static constexpr int SD = D + 1 + (D & 1);
  static constexpr int scanA = 0;                                   // kLfThreads*SD
  static constexpr int scanB = scanA + kLfThreads * SD;
  static constexpr int H = scanB + kLfThreads * SD;                 // kLfChunk*D      | per-stage slots from here
  static constexpr int Mp = H + kLfChunk * D;                       // kLfScanSteps*D*D
  static constexpr int ah = Mp + kLfScanSteps * D * D;              // D+1 (a^)
  static constexpr int bh = ah + (D + 1);                           // D+1 (b^)
  static constexpr int cx = bh + (D + 1);                           // carried inputs  [D]
  static constexpr int cy = cx + D;                                 // carried outputs [D] (true, unclamped)
  static constexpr int neff = cy + D;                               // scan steps that matter (as a double), + kLfScanSteps flags
  static constexpr int total = neff + 1 + kLfScanSteps;
  static constexpr int stage_doubles = total - H;
  // bytes for a cascade of n_stages: one working copy + (n_stages > 1) a parked copy per stage
  static constexpr size_t bytes(int n_stages) {
    return (size_t)blk_floats * sizeof(float) +
           ((size_t)total + (n_stages > 1 ? (size_t)n_stages * stage_doubles : 0)) * sizeof(double);
  }
};

// ---- double-double helpers for the TABLES ------------------------------------------------------------------------------
// The powers M^(2^k) come from repeated squaring, and a perturbation d of M moves M^2 by M d + d M: with the transient
// growth of a sharp direct form (|M| ~ 7e3 for a 6th-order Chebyshev at 0.1) table errors compound as the PRODUCT of the
// norms of successive powers -- tables built in float64 left the scan 8e-5 of the peak off on that design although every
// runtime operation was float64 (numpy experiment: float64 tables 1.1e-5, 80-bit tables 8.6e-9, same runtime).  Runtime
// round-off does not compound that way (powers of one matrix commute: an error vector is amplified by ONE power, at most
// max |A^n|).  So the recursion that yields H and M and the squarings run in double-double (~106 bits) and the tables are
// rounded to float64 once.  A few thousand operations per (sequence, stage), outside the sample loop.
struct LfDD { double hi, lo; };
AAMD_HD LfDD lf_dd_quick(double a, double b) { const double s = a + b; return {s, b - (s - a)}; }
AAMD_HD LfDD lf_dd_add(LfDD a, LfDD b) {
  const double s = a.hi + b.hi;
  const double bb = s - a.hi;
  const double e = ((a.hi - (s - bb)) + (b.hi - bb)) + (a.lo + b.lo);
  return lf_dd_quick(s, e);
}
AAMD_HD LfDD lf_dd_mul(LfDD a, LfDD b) {
  const double p = a.hi * b.hi;
  double e = fma(a.hi, b.hi, -p);
  e = fma(a.hi, b.lo, e);
  e = fma(a.lo, b.hi, e);
  return lf_dd_quick(p, e);
}

// ---- tables for one stage, built by the first D*D threads in three kinds of phases ---------------------------------
// phase A (tid <= D): the normalised coefficients -- the SAME float32 division as the reference, then widened
template <int D>
AAMD_HD void lf_tables_coeffs(int tid, const float* a_row, const float* b_row, int n_order, double* tab) {
  using L = LfLds<D>;
  if (tid > D) return;
  const float a0 = a_row[0];
  tab[L::ah + tid] = (tid < n_order) ? (double)(a_row[tid] / a0) : 0.0;
  tab[L::bh + tid] = (tid < n_order) ? (double)(b_row[tid] / a0) : 0.0;
}
// phase B (tid < D): homogeneous response to the unit state e_tid over one chunk -> column tid of H and of M.
// The low words of M are parked in the (still unused) scan buffer A for the squarings.
template <int D>
AAMD_HD void lf_tables_response(int tid, double* tab) {
  using L = LfLds<D>;
  if (tid >= D) return;
  const double* ah = tab + L::ah;
  LfDD hist[D];  // hist[e] = y[j-1-e]
  for (int e = 0; e < D; ++e) hist[e] = {(e == tid) ? 1.0 : 0.0, 0.0};
  for (int j = 0; j < kLfChunk; ++j) {
    LfDD y = {0.0, 0.0};
    for (int k = D; k >= 1; --k) y = lf_dd_add(y, lf_dd_mul(LfDD{-ah[k], 0.0}, hist[k - 1]));
    for (int e = D - 1; e > 0; --e) hist[e] = hist[e - 1];
    hist[0] = y;
    tab[L::H + j * D + tid] = y.hi;
  }
  // M[e][tid] = y[CH-1-e] = state component e after the chunk
  for (int e = 0; e < D; ++e) {
    tab[L::Mp + e * D + tid] = hist[e].hi;
    tab[L::scanA + e * D + tid] = hist[e].lo;
  }
}
// phase C_k (tid < D*D), k = 1 .. kLfScanSteps-1: M^(2^k) = (M^(2^(k-1)))^2, one entry per thread; low words ping-pong
// between the scan buffers (k odd: read A, write B)
template <int D>
AAMD_HD void lf_tables_square(int tid, int k, double* tab) {
  using L = LfLds<D>;
  if (tid >= D * D) return;
  const int e = tid / D, d = tid % D;
  const double* Mk = tab + L::Mp + (k - 1) * D * D;
  const double* lo_in = tab + ((k & 1) ? L::scanA : L::scanB);
  double* lo_out = tab + ((k & 1) ? L::scanB : L::scanA);
  LfDD acc = {0.0, 0.0};
  for (int f = 0; f < D; ++f)
    acc = lf_dd_add(acc, lf_dd_mul(LfDD{Mk[e * D + f], lo_in[e * D + f]}, LfDD{Mk[f * D + d], lo_in[f * D + d]}));
  tab[L::Mp + k * D * D + tid] = acc.hi;
  lo_out[tid] = acc.lo;
}

// phase D (tid < kLfScanSteps, then tid 0): scan steps whose matrix is below 2^-70 everywhere add nothing a float32 output can
// see, even through a homogeneous response of 1e6 (poles at radius r: |M^(2^k)| ~ r^(32 2^k); r = 0.95 -> steps 5, 6, 7).
// Once a power is that small all higher ones are (they are its squares), so the scan simply stops early.
template <int D>
AAMD_HD void lf_tables_small(int tid, double* tab) {
  using L = LfLds<D>;
  if (tid >= kLfScanSteps) return;
  double m = 0.0;
  for (int i = 0; i < D * D; ++i) {
    const double v = tab[L::Mp + tid * D * D + i];
    m = fmax(m, v < 0.0 ? -v : v);
  }
  tab[L::neff + 1 + tid] = (m < 8.470329472543003e-22) ? 1.0 : 0.0;      // 2^-70
}
template <int D>
AAMD_HD void lf_tables_neff(double* tab) {
  using L = LfLds<D>;
  int n = kLfScanSteps;
  while (n > 0 && tab[L::neff + n] != 0.0) --n;        // flags at neff + 1 + k
  tab[L::neff] = (double)n;
}

// ---- phase 1: chunk pass from zero state (chunk 0: from the carried state) ------------------
template <int D>
AAMD_HD void lf_chunk_pass(int tid, const float* blk, double* tab, LfThread<D>& th) {
  using L = LfLds<D>;
  double ah[D + 1], bh[D + 1];
#pragma unroll
  for (int k = 0; k <= D; ++k) { ah[k] = tab[L::ah + k]; bh[k] = tab[L::bh + k]; }
  double hu[D], hz[D];  // hu[e] = u[j-1-e], hz[e] = z[j-1-e]
#pragma unroll
  for (int e = 0; e < D; ++e) {
    if (tid == 0) {
      hu[e] = tab[L::cx + e];
      hz[e] = tab[L::cy + e];
    } else {
      // the D samples before this chunk live at the tail of chunk tid-1 (D <= CH)
      hu[e] = (double)blk[(tid - 1) * kLfChunkStride + (kLfChunk - 1 - e)];
      hz[e] = 0.0;
    }
  }
  const float* mine = blk + tid * kLfChunkStride;
#pragma unroll
  for (int j = 0; j < kLfChunk; ++j) {
    const double u = (double)mine[j];
    double y = bh[0] * u;
#pragma unroll
    for (int k = D; k >= 1; --k) y += bh[k] * hu[k - 1];   // the FIR part does not depend on the recursion
#pragma unroll
    for (int k = D; k >= 1; --k) y -= ah[k] * hz[k - 1];   // oldest output first: only the last term waits for y[j-1]
#pragma unroll
    for (int e = D - 1; e > 0; --e) { hu[e] = hu[e - 1]; hz[e] = hz[e - 1]; }
    hu[0] = u;
    hz[0] = y;
    th.z[j] = y;
  }
#pragma unroll
  for (int e = 0; e < D; ++e) {
    th.s[e] = hz[e];
    tab[L::scanA + tid * L::SD + e] = hz[e];
  }
}

// save the stage input's last D samples (next block's FIR history); run by tid 0 AFTER the
// chunk pass read them and BEFORE the in-place write-back.
template <int D>
AAMD_HD void lf_save_input_carry(const float* blk, double* tab) {
  using L = LfLds<D>;
  const float* last = blk + (kLfThreads - 1) * kLfChunkStride;
  for (int e = 0; e < D; ++e) tab[L::cx + e] = (double)last[kLfChunk - 1 - e];
}

// ---- phase 2: one Hillis-Steele step, offset 2^k, src -> dst ---------------------------------
template <int D>
AAMD_HD void lf_scan_step(int tid, int k, double* tab, LfThread<D>& th, bool src_is_a) {
  using L = LfLds<D>;
  const int off = 1 << k;
  const double* src = tab + (src_is_a ? L::scanA : L::scanB);
  double* dst = tab + (src_is_a ? L::scanB : L::scanA);
  if (tid >= off) {
    const double* Mk = tab + L::Mp + k * D * D;
    const double* vp = src + (tid - off) * L::SD;
    double v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = vp[d];
#pragma unroll
    for (int e = 0; e < D; ++e) {
      double acc = 0.0;
#pragma unroll
      for (int d = 0; d < D; ++d) acc += Mk[e * D + d] * v[d];
      th.s[e] += acc;
    }
  }
#pragma unroll
  for (int e = 0; e < D; ++e) dst[tid * L::SD + e] = th.s[e];
}

// ---- phase 3: add the homogeneous response of the true entering state, clamp, write back ----
//   `fin` = LDS scan buffer holding the final (true) end-of-chunk states.
template <int D>
AAMD_HD void lf_correct_store(int tid, float* blk, const double* tab, LfThread<D>& th, bool fin_is_a, int clamp) {
  using L = LfLds<D>;
  const double* fin = tab + (fin_is_a ? L::scanA : L::scanB);
  float* mine = blk + tid * kLfChunkStride;
  double sp[D];
#pragma unroll
  for (int e = 0; e < D; ++e) sp[e] = (tid > 0) ? fin[(tid - 1) * L::SD + e] : 0.0;
  const double* H = tab + L::H;
#pragma unroll
  for (int j = 0; j < kLfChunk; ++j) {
    double y = th.z[j];
    if (tid > 0) {
#pragma unroll
      for (int d = 0; d < D; ++d) y += H[j * D + d] * sp[d];
    }
    float yf = (float)y;
    if (clamp) yf = fmin(fmax(yf, -1.0f), 1.0f);
    mine[j] = yf;
  }
}

// carry the true (unclamped) final state of the block; run by tid 0 after phase 3's barrier
template <int D>
AAMD_HD void lf_save_output_carry(double* tab, bool fin_is_a) {
  using L = LfLds<D>;
  const double* fin = tab + (fin_is_a ? L::scanA : L::scanB);
  for (int e = 0; e < D; ++e) tab[L::cy + e] = fin[(kLfThreads - 1) * L::SD + e];
}

#if defined(__HIPCC__)
// The working slots H..cy hold the CURRENT stage's tables and carried state; with more than one stage a copy per
// stage is parked behind the working set in LDS and swapped in at each stage.
template <int D>
__global__ void __launch_bounds__(kLfThreads)
lfilter_kernel(const float* __restrict__ x, const float* __restrict__ a,
               const float* __restrict__ b, float* __restrict__ y, int64_t n_seq, int channels,
               int64_t length, int n_order, int n_coeff_rows, int n_stages, int clamp) {
  using L = LfLds<D>;
  extern __shared__ __attribute__((aligned(16))) float smem_lf[];
  float* blk = smem_lf;
  double* tab = reinterpret_cast<double*>(smem_lf + L::blk_floats);
  double* stage_store = tab + L::total;        // n_stages * stage_doubles (n_stages > 1 only)
  const int tid = threadIdx.x;
  LfThread<D> th;

  for (int64_t seq = blockIdx.x; seq < n_seq; seq += gridDim.x) {
    const int ch = (int)(seq % channels);
    const int crow = (n_coeff_rows == 1) ? 0 : ch;
    for (int st = 0; st < n_stages; ++st) {
      const int64_t coff = ((int64_t)st * n_coeff_rows + crow) * n_order;
      __syncthreads();
      lf_tables_coeffs<D>(tid, a + coff, b + coff, n_order, tab);
      if (tid < 2 * D) tab[L::cx + tid] = 0.0;          // cx, cy: zero initial state
      __syncthreads();
      lf_tables_response<D>(tid, tab);
      for (int k = 1; k < kLfScanSteps; ++k) {
        __syncthreads();
        lf_tables_square<D>(tid, k, tab);
      }
      __syncthreads();
      lf_tables_small<D>(tid, tab);
      __syncthreads();
      if (tid == 0) lf_tables_neff<D>(tab);
      if (n_stages > 1) {
        __syncthreads();
        for (int i = tid; i < L::stage_doubles; i += kLfThreads) stage_store[st * L::stage_doubles + i] = tab[L::H + i];
      }
    }
    __syncthreads();
    const float* xs = x + seq * length;
    float* ys = y + seq * length;
    // The next block's samples travel while the current block is filtered: with one workgroup per sequence and (for the
    // 256 sequences of a cfg5a shard) one wave per SIMD, nothing else hides the ~2 us of an HBM round trip per block.
    // 32 registers per thread: orders <= 8 (the wider instantiations are at their register limit without it).
    constexpr bool kPrefetch = D <= 8;
    float pre[kPrefetch ? kLfChunk : 1];
    auto fetch = [&](int64_t n0) {
#pragma unroll
      for (int k = 0; k < kLfChunk; ++k) {
        const int64_t n = n0 + tid + (int64_t)k * kLfThreads;
        pre[kPrefetch ? k : 0] = (n < length) ? xs[n] : 0.0f;
      }
    };
    if (kPrefetch) fetch(0);
    for (int64_t n0 = 0; n0 < length; n0 += kLfBlock) {
      // stage the block (zero beyond the end)
      if (kPrefetch) {
#pragma unroll
        for (int k = 0; k < kLfChunk; ++k) {
          const int i = tid + k * kLfThreads;
          blk[(i / kLfChunk) * kLfChunkStride + (i % kLfChunk)] = pre[kPrefetch ? k : 0];
        }
        if (n0 + kLfBlock < length) fetch(n0 + kLfBlock);
      } else {
        for (int i = tid; i < kLfBlock; i += kLfThreads) {
          const int64_t n = n0 + i;
          blk[(i / kLfChunk) * kLfChunkStride + (i % kLfChunk)] = (n < length) ? xs[n] : 0.0f;
        }
      }
      for (int st = 0; st < n_stages; ++st) {
        __syncthreads();
        if (n_stages > 1) {
          for (int i = tid; i < L::stage_doubles; i += kLfThreads) tab[L::H + i] = stage_store[st * L::stage_doubles + i];
          __syncthreads();
        }
        lf_chunk_pass<D>(tid, blk, tab, th);
        __syncthreads();
        if (tid == 0) lf_save_input_carry<D>(blk, tab);
        bool src_is_a = true;
        const int n_eff = (int)tab[L::neff];             // (workgroup-uniform; written before the barrier above)
        // The carry save above READS the last chunk of the block (thread 0); lf_correct_store below WRITES it in place from
        // another wave.  The scan steps used to separate the two with their barriers, but the early-stopping scan runs none
        // of them for a stage whose memory dies within one chunk (an FIR through lfilter, poles with |r| < ~0.22): the
        // carried FIR history then raced with the filtered samples.
        if (n_eff == 0) __syncthreads();
        for (int k = 0; k < n_eff; ++k) {
          lf_scan_step<D>(tid, k, tab, th, src_is_a);
          __syncthreads();
          src_is_a = !src_is_a;
        }
        lf_correct_store<D>(tid, blk, tab, th, src_is_a, clamp == 1 || (clamp == 2 && st == n_stages - 1));   // 2: last stage only
        if (tid == 0) lf_save_output_carry<D>(tab, src_is_a);
        __syncthreads();
        if (n_stages > 1)
          for (int i = tid; i < 2 * D; i += kLfThreads)   // persist the carries of this stage
            stage_store[st * L::stage_doubles + (L::cx - L::H) + i] = tab[L::cx + i];
      }
      __syncthreads();
      for (int i = tid; i < kLfBlock; i += kLfThreads) {
        const int64_t n = n0 + i;
        if (n < length) ys[n] = blk[(i / kLfChunk) * kLfChunkStride + (i % kLfChunk)];
      }
      __syncthreads();
    }
  }
}
#endif

}  // namespace aamd

// Polyphase windowed-sinc resampling:  y[q*new + p] = sum_k h[p][k] * xpad[q*orig + k]
// (functional/functional.py:1405-1432: F.pad(width, width+orig) + conv1d(stride=orig) +
// phase interleave + crop to ceil(new*L/orig)).
//
// A workgroup owns QT consecutive q (QT*new contiguous outputs of one waveform).  The
// waveform halo (QT-1)*orig + taps is staged once in LDS with coalesced loads (zero padding
// resolved at staging time); each thread then accumulates outputs (q, p) over the taps,
// reading the waveform from LDS (broadcast across lanes that share q) and the tap from the
// phase-major kernel table.
#pragma once
#include "hd.h"

namespace aamd {

struct ResampleGeom {
  int64_t rows, length, row_stride, out_len;
  int orig, new_, width, taps;
  int qt;        // q values per workgroup
  int nq_tiles;  // workgroups per row
  int use_lds;
};

AAMD_HD float resample_x(const float* row, int64_t length, int64_t j /* padded index */, int width) {
  const int64_t i = j - width;
  return (i >= 0 && i < length) ? row[i] : 0.0f;
}

// stage the halo for q in [q0, q0+qt)
AAMD_HD void resample_stage(int tid, int nthr, const ResampleGeom& g, const float* row, int64_t q0,
                            float* xs) {
  const int n = (g.qt - 1) * g.orig + g.taps;
  const int64_t j0 = q0 * g.orig;
  for (int j = tid; j < n; j += nthr) xs[j] = resample_x(row, g.length, j0 + j, g.width);
}

AAMD_HD void resample_compute(int tid, int nthr, const ResampleGeom& g, const float* kern,
                              const float* row, const float* xs, int64_t q0, float* out_row) {
  const int n_out = g.qt * g.new_;
  for (int o = tid; o < n_out; o += nthr) {
    const int ql = o / g.new_;
    const int p = o - ql * g.new_;
    const int64_t oi = (q0 + ql) * g.new_ + p;
    if (oi >= g.out_len) continue;
    const float* h = kern + (int64_t)p * g.taps;
    float acc = 0.0f;
    if (g.use_lds) {
      const float* x = xs + ql * g.orig;
      for (int k = 0; k < g.taps; ++k) acc += h[k] * x[k];
    } else {
      const int64_t j0 = (q0 + ql) * g.orig;
      for (int k = 0; k < g.taps; ++k) acc += h[k] * resample_x(row, g.length, j0 + k, g.width);
    }
    out_row[oi] = acc;
  }
}

// Sparse evaluation for ratios whose reduced rates are huge (PitchShift: 20158 -> 16000 Hz = 10079 : 8000, a tap table of
// 8000 x 10095 of which ~36 taps per phase are non-zero; functional/functional.py:1790-1840 builds exactly that table).
// The host compacts the table once: hb[p][span] = kernel[p][lo[p] .. lo[p] + span), lo[p] = first non-negligible tap of
// phase p.  One thread per output sample:
//   y[q new + p] = sum_{j < span} hb[p][j] * xpad[q orig + lo[p] + j]
AAMD_HD float resample_sparse_one(const float* row, int64_t length, const float* hb, const int32_t* lo, int orig, int new_,
                                  int width, int span, int64_t i) {
  const int64_t q = i / new_;
  const int p = (int)(i - q * new_);
  const float* h = hb + (int64_t)p * span;
  const int64_t s0 = q * orig + lo[p] - width;      // signal index of the band's first tap
  float acc = 0.0f;
  for (int j = 0; j < span; ++j) {
    const int64_t s = s0 + j;
    if (s >= 0 && s < length) acc += h[j] * row[s];
  }
  return acc;
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(256)
resample_sparse_kernel(const float* __restrict__ wav, const float* __restrict__ hb, const int32_t* __restrict__ lo,
                       float* __restrict__ out, int64_t rows, int64_t length, int64_t row_stride, int orig, int new_,
                       int width, int span, int64_t out_len) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * out_len) return;
  const int64_t row = idx / out_len, i = idx - row * out_len;
  out[idx] = resample_sparse_one(wav + row * row_stride, length, hb, lo, orig, new_, width, span, i);
}

__global__ void __launch_bounds__(256)
resample_kernel(ResampleGeom g, const float* __restrict__ wav, const float* __restrict__ kern,
                float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem_rs[];
  const int64_t row = blockIdx.x / g.nq_tiles;
  const int64_t q0 = (int64_t)(blockIdx.x - row * g.nq_tiles) * g.qt;
  const float* wrow = wav + row * g.row_stride;
  if (g.use_lds) {
    resample_stage(threadIdx.x, blockDim.x, g, wrow, q0, smem_rs);
    __syncthreads();
  }
  resample_compute(threadIdx.x, blockDim.x, g, kern, wrow, smem_rs, q0, out + row * g.out_len);
}
#endif

}  // namespace aamd

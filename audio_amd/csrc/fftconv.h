// Linear convolution along the last dim (F.fftconvolve, functional/functional.py:2222-2258):
//   z[n] = sum_m x[m] y[n-m],  n in [start, start+out_len)  (mode crop :2207-2219).
// The contract is the linear-convolution RESULT, not the FFT length, so this
// first implementation evaluates it directly with LDS tiling: a workgroup owns TN outputs of
// one row, the shorter operand is streamed through LDS in chunks of TY taps together with the
// matching window of the longer operand.  Exact up to fp32 summation order.
// (Overlap-save on the LDS FFT for long impulse responses is the planned fast path.)
#pragma once
#include "hd.h"

namespace aamd {

constexpr int kFcThreads = 256;
constexpr int kFcOutPerThread = 4;
constexpr int kFcTN = kFcThreads * kFcOutPerThread;  // outputs per workgroup
constexpr int kFcTY = 1024;                          // taps per LDS chunk

struct FcGeom {
  int64_t rows, nx, ny, start, out_len;
  int n_tiles;  // per row
};

// stage taps ys[j] = y[j0 + j] and window xs[i] = x[xbase + i], i in [0, TN + TY - 1)
AAMD_HD void fc_stage(int tid, int nthr, const FcGeom& g, const float* xr, const float* yr,
                      int64_t j0, int64_t xbase, float* xs, float* ys) {
  for (int j = tid; j < kFcTY; j += nthr) {
    const int64_t jj = j0 + j;
    ys[j] = (jj < g.ny) ? yr[jj] : 0.0f;
  }
  for (int i = tid; i < kFcTN + kFcTY - 1; i += nthr) {
    const int64_t m = xbase + i;
    xs[i] = (m >= 0 && m < g.nx) ? xr[m] : 0.0f;
  }
}

// acc[i] += sum_j ys[j] * x[n_i - j0 - j],  n_i = n0 + tid + i*threads,
// x[n_i - j0 - j] = xs[(n_i - n0) + (TY - 1) - j]  with  xbase = n0 - j0 - (TY - 1)
AAMD_HD void fc_accumulate(int tid, const float* xs, const float* ys, float (&acc)[kFcOutPerThread]) {
  for (int j = 0; j < kFcTY; ++j) {
    const float h = ys[j];
#pragma unroll
    for (int i = 0; i < kFcOutPerThread; ++i)
      acc[i] += h * xs[tid + i * kFcThreads + (kFcTY - 1) - j];
  }
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(kFcThreads)
fftconv_direct_kernel(FcGeom g, const float* __restrict__ x, const float* __restrict__ y,
                      const int64_t* __restrict__ x_row_of, const int64_t* __restrict__ y_row_of,
                      float* __restrict__ out) {
  __shared__ float xs[kFcTN + kFcTY];
  __shared__ float ys[kFcTY];
  const int64_t row = blockIdx.x / g.n_tiles;
  const int tile = blockIdx.x - (int)row * g.n_tiles;
  const int64_t rx = x_row_of ? x_row_of[row] : row;
  const int64_t ry = y_row_of ? y_row_of[row] : row;
  const float* xr = x + rx * g.nx;
  const float* yr = y + ry * g.ny;
  const int64_t n0 = g.start + (int64_t)tile * kFcTN;
  float acc[kFcOutPerThread] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t j0 = 0; j0 < g.ny; j0 += kFcTY) {
    // skip chunks whose x window is entirely outside [0, nx)
    const int64_t xbase = n0 - j0 - (kFcTY - 1);
    if (xbase >= g.nx || xbase + kFcTN + kFcTY - 1 <= 0) continue;
    __syncthreads();
    fc_stage(threadIdx.x, kFcThreads, g, xr, yr, j0, xbase, xs, ys);
    __syncthreads();
    fc_accumulate(threadIdx.x, xs, ys, acc);
  }
#pragma unroll
  for (int i = 0; i < kFcOutPerThread; ++i) {
    const int64_t n = (int64_t)tile * kFcTN + threadIdx.x + i * kFcThreads;
    if (n < g.out_len) out[row * g.out_len + n] = acc[i];
  }
}
#endif

}  // namespace aamd

// float64 entry points (the reference guarantees gradcheck AND gradgradcheck of lfilter / biquad /
// Spectrogram / MelSpectrogram / MFCC / Resample / fftconvolve, and those run in float64:
// test/torchaudio_unittest/functional/autograd_impl.py:21-35, transforms/autograd_test_impl.py:30-45; the native IIR
// loop dispatches on double as well, libtorchaudio/lfilter.cpp:62-68, iir_cuda.cu:66-76).
//
// float64 is NOT the throughput path (BASELINE configs are fp32), so these kernels are the simple, obviously-correct
// forms: the STFT / inverse STFT are the generic LDS Stockham kernels of stft_generic.h / istft.h instantiated on double;
// lfilter runs one thread per sequence (the reference's own CUDA kernel does the same), resampling and convolution one
// thread per output sample.  Same semantics, same index maps as the fp32 kernels.
#pragma once
#include "hd.h"

namespace aamd {
namespace f64 {

// y[n] = (sum_k b[k] x[n-k] - sum_{k>=1} a[k] y[n-k]) / a[0], zero initial state (functional/filtering.py:1027-1099)
AAMD_HD void lfilter_seq(const double* x, const double* a, const double* b, double* y, int64_t length, int n_order, int clamp) {
  const double inv_a0 = 1.0 / a[0];
  for (int64_t n = 0; n < length; ++n) {
    double acc = 0.0;
    for (int k = 0; k < n_order; ++k)
      if (n - k >= 0) acc += b[k] * x[n - k];
    for (int k = 1; k < n_order; ++k)
      if (n - k >= 0) acc -= a[k] * y[n - k];      // y holds the UNCLAMPED recursion state until the end of the row
    y[n] = acc * inv_a0;
  }
  if (clamp)
    for (int64_t n = 0; n < length; ++n) y[n] = y[n] < -1.0 ? -1.0 : (y[n] > 1.0 ? 1.0 : y[n]);
}

// polyphase resampling (functional/functional.py:1421-1428): output i = q new + p
AAMD_HD double resample_one(const double* wav_row, const double* kern, int64_t length, int orig, int new_, int width,
                            int64_t i) {
  const int64_t q = i / new_;
  const int p = (int)(i - q * new_);
  const int taps = 2 * width + orig;
  const double* h = kern + (int64_t)p * taps;
  double acc = 0.0;
  for (int k = 0; k < taps; ++k) {
    const int64_t s = q * orig + k - width;       // xpad = [0]*width ++ x ++ [0]*(width + orig)
    if (s >= 0 && s < length) acc += h[k] * wav_row[s];
  }
  return acc;
}

// one sample of the full linear convolution (functional/functional.py:2252-2258)
AAMD_HD double conv_one(const double* x, int64_t nx, const double* y, int64_t ny, int64_t n) {
  int64_t m_lo = n - (ny - 1);
  if (m_lo < 0) m_lo = 0;
  int64_t m_hi = n < nx - 1 ? n : nx - 1;
  double acc = 0.0;
  for (int64_t m = m_lo; m <= m_hi; ++m) acc += x[m] * y[n - m];
  return acc;
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(64)
lfilter_kernel(const double* __restrict__ x, const double* __restrict__ a, const double* __restrict__ b,
               double* __restrict__ y, int64_t n_seq, int channels, int64_t length, int n_order, int n_rows, int clamp) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seq) return;
  const int c = (int)(s % channels);
  const int r = n_rows == 1 ? 0 : c;
  lfilter_seq(x + s * length, a + (int64_t)r * n_order, b + (int64_t)r * n_order, y + s * length, length, n_order, clamp);
}

__global__ void __launch_bounds__(256)
resample_kernel(const double* __restrict__ wav, const double* __restrict__ kern, double* __restrict__ out, int64_t rows,
                int64_t length, int64_t row_stride, int orig, int new_, int width, int64_t out_len) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * out_len) return;
  const int64_t row = idx / out_len, i = idx - row * out_len;
  out[idx] = resample_one(wav + row * row_stride, kern, length, orig, new_, width, i);
}

__global__ void __launch_bounds__(256)
conv_kernel(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out, int64_t rows, int64_t nx,
            int64_t ny, const int64_t* __restrict__ x_row_of, const int64_t* __restrict__ y_row_of, int64_t start,
            int64_t out_len) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * out_len) return;
  const int64_t row = idx / out_len, i = idx - row * out_len;
  const int64_t rx = x_row_of ? x_row_of[row] : row, ry = y_row_of ? y_row_of[row] : row;
  out[idx] = conv_one(x + rx * nx, nx, y + ry * ny, ny, start + i);
}
#endif

}  // namespace f64
}  // namespace aamd

// dB conversion, top_db clamp, MelScale-on-a-spectrogram and the MFCC tail (log/dB + DCT-II).
// Reference: functional/functional.py:356-404 (amplitude_to_DB), transforms/_transforms.py:403-415
// (MelScale.forward), :692-709 (MFCC.forward).
#pragma once
#include "hd.h"
#include "stft_generic.h"

namespace aamd {

AAMD_HD float to_db(float x, float multiplier, float amin, float db_multiplier) {
  return multiplier * log10(fmax(x, amin)) - multiplier * db_multiplier;
}

// y for one mel value under the three MFCC log modes (see audio_amd.h)
AAMD_HD float mfcc_log(float v, int log_mode, float cut) {
  if (log_mode == 1) return log(v + 1e-6f);
  float y = (log_mode == 0) ? to_db(v, 10.0f, 1e-10f, 0.0f) : v;
  return fmax(y, cut);
}

#if defined(__HIPCC__)

// float max via integer atomics; target must be initialised to -inf (or any float).
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ void __launch_bounds__(256)
amplitude_to_db_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                       float multiplier, float amin, float db_multiplier,
                       float* __restrict__ group_max, int64_t group_size) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < n; base += stride) {
    const int64_t i = base + threadIdx.x;
    const bool ok = i < n;
    float v = -INFINITY;
    if (ok) {
      v = to_db(x[i], multiplier, amin, db_multiplier);
      out[i] = v;
    }
    if (group_max != nullptr) {
      // one atomic per wave when the whole wave sits in one group, else per lane
      const int64_t g = ok ? i / group_size : -1;
      const int64_t g0 = __shfl(g, 0, 64);
      const bool uniform = __all(g == g0 || !ok) && g0 >= 0;
      if (uniform) {
        const float m = wave_max(v);
        if ((threadIdx.x & 63) == 0) atomic_max_float(group_max + g0, m);
      } else if (ok) {
        atomic_max_float(group_max + g, v);
      }
    }
  }
}

__global__ void __launch_bounds__(256)
db_clamp_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                const float* __restrict__ group_max, int64_t group_size, float top_db) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = fmaxf(x[i], group_max[i / group_size] - top_db);
}

// MelScale on a frame-major spectrogram: one thread per (vector, mel).
__global__ void __launch_bounds__(256)
mel_scale_kernel(const float* __restrict__ spec, MelBandsDev mb, float* __restrict__ out,
                 int64_t n_vec, int n_freq) {
  const int64_t total = n_vec * mb.n_mels;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int64_t v = o / mb.n_mels;
    const int m = (int)(o - v * mb.n_mels);
    const int lo = mb.lo[m], w = mb.width[m];
    const float* wt = mb.weights + (int64_t)m * mb.max_width;
    const float* P = spec + v * n_freq + lo;
    float acc = 0.0f;
    for (int i = 0; i < w; ++i) acc += wt[i] * P[i];
    out[o] = acc;
  }
}

// MFCC tail.  Block: VPB vectors at a time; dct staged in LDS once per block.
constexpr int kMfccVecPerBlock = 8;

__global__ void __launch_bounds__(256)
mfcc_dct_kernel(const float* __restrict__ mel, const float* __restrict__ dct,
                float* __restrict__ out, int64_t n_vec, int n_mels, int n_mfcc, int log_mode,
                const float* __restrict__ group_max, int64_t vec_per_group, float top_db) {
  extern __shared__ __attribute__((aligned(16))) float smem_mfcc[];
  float* sd = smem_mfcc;                       // n_mels * n_mfcc
  float* sy = sd + n_mels * n_mfcc;            // VPB * n_mels
  for (int i = threadIdx.x; i < n_mels * n_mfcc; i += blockDim.x) sd[i] = dct[i];
  const int64_t n_groups_of_vec = (n_vec + kMfccVecPerBlock - 1) / kMfccVecPerBlock;
  for (int64_t gv = blockIdx.x; gv < n_groups_of_vec; gv += gridDim.x) {
    const int64_t v0 = gv * kMfccVecPerBlock;
    __syncthreads();
    for (int i = threadIdx.x; i < kMfccVecPerBlock * n_mels; i += blockDim.x) {
      const int64_t v = v0 + i / n_mels;
      float y = 0.0f;
      if (v < n_vec) {
        float cut = -INFINITY;
        if (log_mode != 1 && top_db >= 0.0f && group_max != nullptr)
          cut = group_max[v / vec_per_group] - top_db;
        y = mfcc_log(mel[v0 * n_mels + i], log_mode, cut);
      }
      sy[i] = y;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < kMfccVecPerBlock * n_mfcc; o += blockDim.x) {
      const int vl = o / n_mfcc, k = o - vl * n_mfcc;
      if (v0 + vl >= n_vec) continue;
      const float* y = sy + vl * n_mels;
      float acc = 0.0f;
      for (int m = 0; m < n_mels; ++m) acc += y[m] * sd[m * n_mfcc + k];
      out[(v0 + vl) * n_mfcc + k] = acc;
    }
  }
}

#endif  // __HIPCC__
}  // namespace aamd

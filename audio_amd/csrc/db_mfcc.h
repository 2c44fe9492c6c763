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

// fragment geometry of the matrix-core DCT kernel below (shared with tests/cpu_sim)
constexpr int kDctFramesPerTile = 16;
constexpr int kDctMaxChunks = 8;        // n_mels <= 128 on the matrix-core path

AAMD_HD int dct_frag_floats(int n_mels, int n_mfcc) {
  const int kc = (n_mels + 15) / 16, nt = (n_mfcc + 15) / 16;
  return nt * kc * 4 * 64;
}

// fragment table value for (tile nt, chunk c, slot j, lane l): dct[16c + 4(l/16) + j][16nt + l%16]
AAMD_HD float dct_frag_value(const float* dct, int n_mels, int n_mfcc, int nt, int c, int j, int lane) {
  const int mel = 16 * c + 4 * (lane >> 4) + j, k = 16 * nt + (lane & 15);
  return (mel < n_mels && k < n_mfcc) ? dct[mel * n_mfcc + k] : 0.0f;
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
  // each thread keeps the running maximum of its current group and only touches memory when
  // the group changes; the final flush is one atomic per wave when the wave agrees on the group
  float run = -INFINITY;
  int64_t run_g = -1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = to_db(x[i], multiplier, amin, db_multiplier);
    out[i] = v;
    if (group_max != nullptr) {
      const int64_t g = i / group_size;
      if (g != run_g) {
        if (run_g >= 0) atomic_max_float(group_max + run_g, run);
        run_g = g;
        run = v;
      } else {
        run = fmaxf(run, v);
      }
    }
  }
  if (group_max != nullptr) {
    const int64_t g0 = __shfl(run_g, 0, 64);
    if (__all(run_g == g0 || run_g < 0)) {
      const float m = wave_max(run);
      if ((threadIdx.x & 63) == 0 && g0 >= 0) atomic_max_float(group_max + g0, m);
    } else if (run_g >= 0) {
      atomic_max_float(group_max + run_g, run);
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

// amplitude_to_DB, second version: one workgroup = one chunk of ONE cut-off group, so there is no per-element 64-bit
// division, the body moves 16 bytes per lane, and the group maximum costs one atomic per workgroup.
//   STORE  write the dB values;  REDUCE  max-reduce them into group_max[g];  CLAMP  y = max(y, group_max[g] - top_db)
// (REDUCE without STORE = the first pass of a top_db conversion: it reads x once and writes nothing; the second pass
// recomputes the logarithm, clamps and stores -- 3 instead of 4 sweeps over the tensor.)
constexpr int kDbChunk = 8192;
template <bool STORE, bool REDUCE, bool CLAMP>
__global__ void __launch_bounds__(256)
db_group_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n, float multiplier, float amin,
                float db_multiplier, float* __restrict__ group_max, int64_t group_size, int64_t chunks_per_group,
                float top_db) {
  __shared__ float wmax[4];
  const int64_t g = blockIdx.x / chunks_per_group;
  const int64_t c = blockIdx.x - g * chunks_per_group;
  const int64_t gend = (g + 1) * group_size < n ? (g + 1) * group_size : n;
  const int64_t lo = g * group_size + c * kDbChunk;
  const int64_t hi = lo + kDbChunk < gend ? lo + kDbChunk : gend;
  const float cut = CLAMP ? group_max[g] - top_db : -INFINITY;
  float run = -INFINITY;
  auto one = [&](int64_t i) {
    float v = to_db(x[i], multiplier, amin, db_multiplier);
    if (REDUCE) run = fmaxf(run, v);
    if (CLAMP) v = fmaxf(v, cut);
    if (STORE) out[i] = v;
  };
  // scalar head up to the next 16-byte boundary of x (out shares the index, hence the alignment, when both bases are
  // 16-byte aligned; otherwise everything goes through the scalar loop), 16-byte body, scalar tail.
  const bool vec_ok = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (!STORE || reinterpret_cast<uintptr_t>(out) % 16 == 0);
  int64_t head_end = hi, body_end = hi;
  if (vec_ok) {
    head_end = (lo + 3) & ~(int64_t)3;
    if (head_end > hi) head_end = hi;
    body_end = head_end + ((hi - head_end) & ~(int64_t)3);
  }
#pragma clang loop vectorize(disable) unroll(disable)
  for (int64_t i = lo + threadIdx.x; i < head_end; i += 256) one(i);
  const int64_t n_vec4 = (body_end - head_end) >> 2;
  const float4* xv4 = reinterpret_cast<const float4*>(x + head_end);
  float4* ov4 = reinterpret_cast<float4*>(out + head_end);
#pragma unroll 4
  for (int64_t j = threadIdx.x; j < n_vec4; j += 256) {
    const float4 xv = xv4[j];
    float4 y;
    y.x = to_db(xv.x, multiplier, amin, db_multiplier);
    y.y = to_db(xv.y, multiplier, amin, db_multiplier);
    y.z = to_db(xv.z, multiplier, amin, db_multiplier);
    y.w = to_db(xv.w, multiplier, amin, db_multiplier);
    if (REDUCE) run = fmaxf(run, fmaxf(fmaxf(y.x, y.y), fmaxf(y.z, y.w)));
    if (CLAMP) { y.x = fmaxf(y.x, cut); y.y = fmaxf(y.y, cut); y.z = fmaxf(y.z, cut); y.w = fmaxf(y.w, cut); }
    if (STORE) ov4[j] = y;
  }
#pragma clang loop vectorize(disable) unroll(disable)
  for (int64_t i = body_end + threadIdx.x; i < hi; i += 256) one(i);
  if (REDUCE) {
    const float m = wave_max(run);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomic_max_float(group_max + g, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
  }
}

// MelScale, second version: the band table and kMsVec spectrum rows live in LDS; workgroups are persistent
constexpr int kMsVec = 16;
AAMD_HD int ms_stride(int max_width) { return max_width | 1; }
AAMD_HD size_t ms_lds_floats(int n_mels, int max_width, int n_freq) {
  return (size_t)n_mels * ms_stride(max_width) + 2 * (size_t)n_mels + (size_t)kMsVec * (n_freq | 1);
}
__global__ void __launch_bounds__(256)
mel_scale_lds_kernel(const float* __restrict__ spec, MelBandsDev mb, float* __restrict__ out, int64_t n_vec, int n_freq) {
  extern __shared__ __attribute__((aligned(16))) float smem_ms[];
  const int ms = ms_stride(mb.max_width), fs = n_freq | 1;
  float* wt = smem_ms;
  int* lo = reinterpret_cast<int*>(wt + mb.n_mels * ms);
  int* wd = lo + mb.n_mels;
  float* S = reinterpret_cast<float*>(wd + mb.n_mels);
  for (int i = threadIdx.x; i < mb.n_mels * ms; i += 256) {
    const int m = i / ms, j = i - m * ms;
    wt[i] = j < mb.max_width ? mb.weights[(int64_t)m * mb.max_width + j] : 0.0f;
  }
  for (int m = threadIdx.x; m < mb.n_mels; m += 256) { lo[m] = mb.lo[m]; wd[m] = mb.width[m]; }
  const int64_t n_grp = (n_vec + kMsVec - 1) / kMsVec;
  for (int64_t grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
    const int64_t v0 = grp * kMsVec;
    const int nv = n_vec - v0 < kMsVec ? (int)(n_vec - v0) : kMsVec;
    __syncthreads();                                     // table ready / previous rows consumed
    const float* src = spec + v0 * n_freq;
    for (int i = threadIdx.x; i < nv * n_freq; i += 256) {
      const int v = i / n_freq, k = i - v * n_freq;
      S[v * fs + k] = src[i];
    }
    __syncthreads();
    float* dst = out + v0 * mb.n_mels;
    for (int o = threadIdx.x; o < nv * mb.n_mels; o += 256) {
      const int v = o / mb.n_mels, m = o - v * mb.n_mels;
      const float* w = wt + m * ms;
      const float* P = S + v * fs + lo[m];
      const int n = wd[m];
      float acc = 0.0f;
      for (int i = 0; i < n; ++i) acc += w[i] * P[i];
      dst[o] = acc;
    }
  }
}

// RNN-T feature post-processing on a frame-major mel buffer (the unfused form of EPI400_MEL_NORM)
__global__ void __launch_bounds__(256)
lognorm_kernel(float* __restrict__ x, int64_t n, int n_mels, float gain, const float* __restrict__ mean,
               const float* __restrict__ invstd) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int m = (int)(i % n_mels);
    const float y = x[i] * gain;
    const float t = y > 2.718281828459045f ? __log2f(y) * 0.69314718055994531f : y;   // see epi_plog (melspec400.h):
    const float l = t <= 2.718281828459045f ? t / 2.718281828459045f : t;             // the reference's second mask
    x[i] = (l - mean[m]) * invstd[m];
  }
}

// Backward of |X|^p: G = dP * p * |X|^(p-2) * X per bin (0 where X = 0 and p < 2) -- the spectrum-domain cotangent the
// STFT adjoint (aamd_istft_f32, adjoint = 1) consumes; X and G interleaved complex, dP real, all frame-major
__global__ void __launch_bounds__(256)
spec_grad_kernel(const float2* X, const float* __restrict__ dP, float2* G, int64_t n, float power) {   // G may alias X (in-place backward)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float2 x = X[i];
    float f = power * dP[i];
    if (power != 2.0f) {
      const float m2 = x.x * x.x + x.y * x.y;
      f = m2 > 0.0f ? f * powf(m2, 0.5f * power - 1.0f) : 0.0f;
    }
    G[i] = make_float2(f * x.x, f * x.y);
  }
}

// Backward of MelSpectrogram's two element-wise stages in one pass: dP[k] = sum_m fb[k][m] dY[m] (band table of fb^T)
// and G = dP p |X|^(p-2) X, written over X (`XG`)
__global__ void __launch_bounds__(256)
mel_grad_kernel(float2* __restrict__ XG, const float* __restrict__ dY, MelBandsDev bt /* bands of fb^T: one per bin */,
                int64_t n_vec, int n_mels, float power) {
  const int n_freq = bt.n_mels;                          // "mels" of the transposed table are the bins
  const int64_t total = n_vec * n_freq;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
    const int64_t v = o / n_freq;
    const int k = (int)(o - v * n_freq);
    const int lo = bt.lo[k], w = bt.width[k];
    const float* wt = bt.weights + (int64_t)k * bt.max_width;
    const float* row = dY + v * n_mels + lo;
    float dp = 0.0f;
    for (int i = 0; i < w; ++i) dp += wt[i] * row[i];
    const float2 x = XG[o];
    float f = power * dp;
    if (power != 2.0f) {
      const float m2 = x.x * x.x + x.y * x.y;
      f = m2 > 0.0f ? f * powf(m2, 0.5f * power - 1.0f) : 0.0f;
    }
    XG[o] = make_float2(f * x.x, f * x.y);
  }
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


// ---- MFCC tail on the matrix cores --------------------------------------------------------
//   out[v][k] = sum_m y[v][m] * dct[m][k] is a (n_vec x n_mels) x (n_mels x n_mfcc) product with
//   n_vec in the hundreds of thousands: HBM-bound if the contraction is cheap.  It runs on
//   v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, same rate as the vector ALU, which stays free
//   for the log/clamp prologue):  D^T tile (16 coefficients x 4 mels) x Y^T tile (4 mels x 16
//   frames), so a lane ends up with 4 CONSECUTIVE coefficients of one frame = one 16-B store.
//   The contraction order inside a chunk of 16 mels is permuted (k-slot (j, g) <-> mel
//   16 c + 4 g + j) so that every lane loads one contiguous float4 of its frame's mel row.
//   Needs n_mels % 4 == 0; the scalar kernel above covers the rest.
template <int NT>
__global__ void __launch_bounds__(256)
mfcc_dct_mfma_kernel(const float* __restrict__ mel, const float* __restrict__ dct,
                     float* __restrict__ out, int64_t n_vec, int n_mels, int n_mfcc, int log_mode,
                     const float* __restrict__ group_max, int64_t vec_per_group, float top_db) {
  using f32x4 = __attribute__((ext_vector_type(4))) float;
  extern __shared__ __attribute__((aligned(16))) float smem_dct[];
  const int kc = (n_mels + 15) / 16;
  for (int i = threadIdx.x; i < NT * kc * 4 * 64; i += blockDim.x) {
    const int lane = i & 63, slot = i >> 6;
    const int j = slot & 3, c = (slot >> 2) % kc, nt = (slot >> 2) / kc;
    smem_dct[i] = dct_frag_value(dct, n_mels, n_mfcc, nt, c, j, lane);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int f = lane & 15, g = lane >> 4;
  const int64_t n_tiles = (n_vec + kDctFramesPerTile - 1) / kDctFramesPerTile;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const bool clampy = log_mode != 1 && top_db >= 0.0f && group_max != nullptr;
  for (int64_t tile = wave0; tile < n_tiles; tile += n_waves) {
    const int64_t v = tile * kDctFramesPerTile + f;
    const bool vok = v < n_vec;
    float cut = -INFINITY;
    if (clampy && vok) cut = group_max[v / vec_per_group] - top_db;
    const float* row = mel + (vok ? v : 0) * (int64_t)n_mels + 4 * g;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // all of the frame's mel row first (kc <= kDctMaxChunks loads in flight per lane), then the math
    F4 m4[kDctMaxChunks];
#pragma unroll
    for (int c = 0; c < kDctMaxChunks; ++c) {
      m4[c] = F4{0.0f, 0.0f, 0.0f, 0.0f};
      if (c < kc && vok && 16 * c + 4 * g < n_mels) m4[c] = *reinterpret_cast<const F4*>(row + 16 * c);
    }
#pragma unroll
    for (int c = 0; c < kDctMaxChunks; ++c) {
      if (c < kc) {
        float y[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (vok && 16 * c + 4 * g < n_mels) {
          y[0] = mfcc_log(m4[c].x, log_mode, cut);
          y[1] = mfcc_log(m4[c].y, log_mode, cut);
          y[2] = mfcc_log(m4[c].z, log_mode, cut);
          y[3] = mfcc_log(m4[c].w, log_mode, cut);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float* af = smem_dct + ((nt * kc + c) * 4) * 64 + lane;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[64 * j], y[j], acc[nt], 0, 0, 0);
        }
      }
    }
    if (vok) {
      float* orow = out + v * (int64_t)n_mfcc;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int k0 = 16 * nt + 4 * g;
        if (k0 + 4 <= n_mfcc && (n_mfcc & 3) == 0) {
          *reinterpret_cast<F4*>(orow + k0) = F4{acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k0 + i < n_mfcc) orow[k0 + i] = acc[nt][i];
        }
      }
    }
  }
}

#endif  // __HIPCC__
}  // namespace aamd

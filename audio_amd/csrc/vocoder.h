// Spectrogram-domain helpers of the time-stretch / phase-recovery callers of the STFT kernels:
//   * phase_vocoder (functional/functional.py:732-803): one thread walks one (row, frequency) chain through
//     time -- magnitude interpolation, phase-difference unwrap and the running phase sum (accumulated in
//     double and rounded to float per step, exactly like aten's CPU cumsum), polar();
//   * the per-iteration phase update of Griffin-Lim (functional/functional.py:318-343), fused:
//       angles = rebuilt - momentum * tprev;  angles /= |angles| + 1e-16;  tprev = rebuilt;  next = mag * angles.
// Both are element streams (HBM-bound); layouts are given as strides so the kernels read the reference's
// (..., freq, time) tensors and this library's frame-major buffers alike.
#pragma once
#include "hd.h"

namespace aamd {

struct VocoderGeom {
  int64_t rows;
  int32_t n_freq, n_in, n_out;          // frames in / out
  int64_t in_row, in_f, in_t;           // input strides in complex elements
  int64_t out_row, out_f, out_t;        // output strides in complex elements
  double rate;
};

template <typename T>
AAMD_HD cplx<T> voc_at(const VocoderGeom& g, const cplx<T>* in, int64_t idx) {
  return idx < g.n_in ? in[idx * g.in_t] : cplx<T>{0, 0};           // F.pad(spec, [0, 2])
}

// chain = one (row, f); in / out already offset to the chain
AAMD_HD void vocoder_chain(const VocoderGeom& g, const cplx<float>* in, float phase_advance, cplx<float>* out) {
  const float two_pi = 6.283185307179586f;                           // float(2 * math.pi)
  double acc = 0.0;
  float prev_phase = 0.0f;
  for (int t = 0; t < g.n_out; ++t) {
    const float ts = (float)((double)t * g.rate);                    // torch.arange(0, T, rate, dtype=float32)
    const float alpha = ts - floorf(ts);                             // time_steps % 1.0
    const int64_t i0 = (int64_t)ts;
    const cplx<float> a = voc_at<float>(g, in, i0), b = voc_at<float>(g, in, i0 + 1);
    const float ang0 = atan2f(a.y, a.x), ang1 = atan2f(b.y, b.x);
    const float n0 = hypotf(a.x, a.y), n1 = hypotf(b.x, b.y);
    float ph = ang1 - ang0 - phase_advance;
    ph = ph - two_pi * rintf(ph / two_pi);                           // torch.round: half to even
    ph = ph + phase_advance;
    // phase = cat([phase_0, phase[:-1]]); cumsum
    const float term = (t == 0) ? ang0 : prev_phase;                 // ang0 at t = 0 is angle(spec[..., 0])
    prev_phase = ph;
    acc += (double)term;
    const float pa = (float)acc;
    const float mag = alpha * n1 + (1.0f - alpha) * n0;
    out[t * g.out_t] = cplx<float>{mag * cosf(pa), mag * sinf(pa)};
  }
}

AAMD_HD void griffinlim_update_elem(const cplx<float>& rebuilt, cplx<float>& tprev, float mag, float momentum,
                                    cplx<float>& next) {
  cplx<float> a = rebuilt;
  if (momentum != 0.0f) { a.x -= tprev.x * momentum; a.y -= tprev.y * momentum; }
  const float d = hypotf(a.x, a.y) + 1e-16f;
  a.x /= d; a.y /= d;
  tprev = rebuilt;
  next = cplx<float>{mag * a.x, mag * a.y};
}

#if defined(__HIPCC__)
__global__ void __launch_bounds__(256)
phase_vocoder_kernel(VocoderGeom g, const cplx<float>* __restrict__ in, const float* __restrict__ phase_advance,
                     cplx<float>* __restrict__ out) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= g.rows * g.n_freq) return;
  const int64_t row = chain / g.n_freq;
  const int f = (int)(chain - row * g.n_freq);
  vocoder_chain(g, in + row * g.in_row + f * g.in_f, phase_advance[f], out + row * g.out_row + f * g.out_f);
}

// tprev <- rebuilt happens in place: the caller passes the two buffers swapped on the next iteration instead
__global__ void __launch_bounds__(256)
griffinlim_update_kernel(const cplx<float>* __restrict__ rebuilt, cplx<float>* __restrict__ tprev,
                         const float* __restrict__ mag, cplx<float>* __restrict__ next, int64_t n, float momentum) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    cplx<float> tp = tprev[i], nx;
    griffinlim_update_elem(rebuilt[i], tp, mag[i], momentum, nx);
    tprev[i] = tp;
    next[i] = nx;
  }
}
#endif

}  // namespace aamd

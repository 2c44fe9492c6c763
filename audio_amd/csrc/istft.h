// Frames -> waveform: batched inverse real FFT of spectrogram frames, window, overlap-add.  One kernel
// serves two callers:
//   * InverseSpectrogram (functional/functional.py:148-225 -> torch.istft): Hermitian inverse FFT of
//     each frame, x window, overlap-add, divided by the window envelope sum_t w^2 (passed in as
//     `inv_env`), centre trimming = frames placed on the padded time axis and samples outside [0, L) dropped;
//   * the ADJOINT of the STFT used by autograd of Spectrogram / MelSpectrogram / MFCC:
//       dL/dx[i] += w[n] Re sum_{k=0}^{N/2} G[k, t] e^{+2 pi i nk/N}   for every (t, n) whose padded sample
//     maps to i -- the same index map as the forward's reflect / replicate / circular / constant padding
//     (pad_source_index), run backwards with atomic adds, so the fold of the reflected edges is automatic.
// Both are one inverse complex FFT per frame PAIR (z = a + i b for two Hermitian spectra), done as the
// forward Stockham stages of stft_generic.h on conj(Z):  IFFT(Z) = conj(FFT(conj Z)).
// The overlap-add inside a workgroup (2 PB consecutive frames) is a gather over LDS; between workgroups
// it is fp32 atomic adds into the pre-zeroed output.
#pragma once
#include "hd.h"
#include "stft_generic.h"

namespace aamd {

struct OlaGeom {
  StftGeom g;            // rows, length (= output samples per row), n_fft, hop, pad, center, pad_mode, n_frames, n_freq, radix plan
  float interior;        // weight of the bins 0 < k < N/2: 1 (Hermitian inverse FFT) or 0.5 (adjoint of a onesided STFT)
  float scale;           // multiplies every output sample (1/N and spectrum normalisation folded in)
  double scale_d;        // the same in float64 (the float64 instantiation: 1/N as a float would cost 3e-8)
};

// conj(Z[k]) for the pair of Hermitian spectra A (frame ta) and B (frame tb); S rows are interleaved complex
template <typename T>
AAMD_HD cplx<T> ola_conj_z(const OlaGeom& og, const T* Sa, const T* Sb, int k) {
  const int N = og.g.n_fft, half = N / 2;
  const int kk = k <= half ? k : N - k;               // onesided bin that defines Z[k]
  const bool edge = (kk == 0) || (2 * kk == N);
  const T wgt = edge ? (T)1 : (T)og.interior;
  T ar = Sa ? Sa[2 * kk] * wgt : (T)0, ai = Sa ? Sa[2 * kk + 1] * wgt : (T)0;
  T br = Sb ? Sb[2 * kk] * wgt : (T)0, bi = Sb ? Sb[2 * kk + 1] * wgt : (T)0;
  if (edge) { ai = 0; bi = 0; }                        // irfft ignores the imaginary part of DC / Nyquist
  if (k > half) { ai = -ai; bi = -bi; }                // Hermitian extension: conj
  // Z = (ar - bi) + i (ai + br);  conj(Z) = (ar - bi, -(ai + br))
  return {ar - bi, -(ai + br)};
}

template <typename T>
AAMD_HD void ola_load(int tid, int nthr, const OlaGeom& og, const T* spec_row /* frame 0 of this row */, int64_t t0,
                      int pb, cplx<T>* buf) {
  const int N = og.g.n_fft, SL = gen_seq_len(N);
  const int64_t fstride = 2 * (int64_t)og.g.n_freq;
  for (int pair = 0; pair < pb; ++pair) {
    const int64_t ta = t0 + 2 * pair, tb = ta + 1;
    const T* Sa = ta < og.g.n_frames ? spec_row + ta * fstride : nullptr;
    const T* Sb = tb < og.g.n_frames ? spec_row + tb * fstride : nullptr;
    cplx<T>* dst = buf + pair * SL;
    for (int k = tid; k < N; k += nthr) dst[gen_pad(k)] = ola_conj_z<T>(og, Sa, Sb, k);
  }
}

// padded-axis coordinate u (frame t covers u = t hop + n) -> output sample index, or -1
AAMD_HD int64_t ola_target(const StftGeom& g, int64_t u) {
  const int64_t L1 = g.length + 2 * (int64_t)g.pad;
  const int64_t i1 = u - (g.center ? g.n_fft / 2 : 0);
  const int64_t s1 = g.center ? pad_source_index(i1, L1, g.pad_mode) : i1;
  if (s1 < 0 || s1 >= L1) return -1;
  const int64_t s0 = s1 - g.pad;
  return (s0 >= 0 && s0 < g.length) ? s0 : -1;
}

// value of sample j of the workgroup's span: sum over its nf EXISTING frames f (0 .. nf - 1 <= 2 pb - 1) covering j.  (Until round 5
// the sum ran to 2 pb - 1: behind an odd number of frames the last pair's missing partner is an all-zero spectrum whose "samples" are
// the inverse transform's rounding cross-talk from the real frame, ~1e-7 of its peak, added at FULL window weight to the last hop of
// the row -- invisible unless the envelope there is tiny: a hann window at hop = n_fft / 2 divides the row's last samples by w^2 ~ 4e-5,
// and torch.istft was 100 x closer to the float64 result on those samples.  Found by the fuzz campaign, seed 311.)
template <typename T>
AAMD_HD T ola_gather(const OlaGeom& og, const cplx<T>* X, const T* window, int nf, int j) {
  const int N = og.g.n_fft, hop = og.g.hop, SL = gen_seq_len(N);
  int f_lo = (j - N + hop) / hop;          // ceil((j - N + 1) / hop) for j - N + 1 > 0
  if (j - N + 1 <= 0) f_lo = 0;
  int f_hi = j / hop;
  if (f_hi > nf - 1) f_hi = nf - 1;
  T acc = 0;
  for (int f = f_lo; f <= f_hi; ++f) {
    const int n = j - f * hop;
    const cplx<T> v = X[(f >> 1) * SL + gen_pad(n)];
    acc += ((f & 1) ? -v.y : v.x) * window[n];   // IFFT(Z) = conj(FFT(conj Z)): a = Re, b = -Im
  }
  return acc * (sizeof(T) == 8 ? (T)og.scale_d : (T)og.scale);
}

#if defined(__HIPCC__)
template <typename T, int LONG = 0>       // LONG = 1: twiddles from memory (stft_generic.h, gen_lds_floats_long)
__global__ void __launch_bounds__(kGenThreads)
ola_kernel(OlaGeom og, const T* __restrict__ spec, const T* __restrict__ window, const cplx<T>* __restrict__ tw,
           const T* __restrict__ inv_env, T* __restrict__ out, int pairs_per_block, int blocks_per_row) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ola[];
  const StftGeom& g = og.g;
  const int N = g.n_fft, pb = pairs_per_block, SL = gen_seq_len(N);
  cplx<T>* twl_lds = reinterpret_cast<cplx<T>*>(smem_ola);
  cplx<T>* bufA = LONG ? twl_lds : twl_lds + N;
  cplx<T>* bufB = bufA + pb * SL;
  const cplx<T>* twl = LONG ? tw : twl_lds;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int64_t row = blockIdx.x / blocks_per_row;
  const int chunk = blockIdx.x - (int)row * blocks_per_row;
  const int64_t t0 = (int64_t)chunk * 2 * pb;
  if (!LONG)
    for (int i = tid; i < N; i += nthr) twl_lds[i] = tw[i];
  ola_load<T>(tid, nthr, og, spec + row * g.n_frames * 2 * (int64_t)g.n_freq, t0, pb, bufA);
  __syncthreads();
  cplx<T>* x = bufA;
  cplx<T>* y = bufB;
  int s = 1;
  for (int st = 0; st < g.n_stages; ++st) {
    const int r = g.radix[st];
    gen_stage<T>(tid, nthr, N, r, s, pb, x, y, twl);
    __syncthreads();
    s *= r;
    cplx<T>* tmp = x; x = y; y = tmp;
  }
  int64_t nf = g.n_frames - t0;                    // frames of this workgroup
  if (nf > 2 * pb) nf = 2 * pb;
  const int span = (int)(nf - 1) * g.hop + N;
  T* out_row = out + row * g.length;
  for (int j = tid; j < span; j += nthr) {
    const int64_t i = ola_target(g, t0 * g.hop + j);
    if (i < 0) continue;
    T v = ola_gather<T>(og, x, window, (int)nf, j);
    if (inv_env != nullptr) v *= inv_env[i];
    atomicAdd(out_row + i, v);
  }
}
#endif  // __HIPCC__

}  // namespace aamd

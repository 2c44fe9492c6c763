// Inverse STFT / STFT adjoint for n_fft = 400 (hop 100 / 160 / 200) on the radix-20x20 register FFT of
// melspec400.h -- the default shape of torchaudio's InverseSpectrogram and GriffinLim (functional.py:148-225, 255-353)
// and of the backward pass of the headline MelSpectrogram.
//
//   IFFT(Z) = conj(FFT(conj Z)): lane (pair p, b) loads conj Z[b + 20 q] (q = 0..19) of z = A + i B, the Hermitian
//   extensions of the pair's two onesided spectra (coalesced 8-byte loads: 20 consecutive bins per 20-lane group, the
//   upper half mirrored), then the forward kernel's own machinery: DFT-20, W400 twiddles, the 20 x 20 transposition
//   through LDS, second DFT-20 -> time samples n = col + 20 d of frames a (real part) and b (minus imaginary part).
//   The 6 frames of the tile are overlap-added in a wave-private LDS buffer (1200 samples; four read-add-write phases
//   coloured so that no two lanes of a phase touch one word), multiplied by the window-envelope reciprocal and written:
//   the middle of an interior tile (no other tile reaches it) as plain stores, its two 240-sample halos and edge tiles
//   (index map of the forward's padding run backwards) as atomic adds into the zero-filled output.
#pragma once
#include "melspec400.h"
#include "istft.h"

namespace aamd {
namespace m400 {

template <int H>
struct Inv400 {
  static constexpr int hop = 20 * H;
  static constexpr int tile = (kFramesPerWave - 1) * hop + kN;          // samples one tile's frames span
  static constexpr int step = kFramesPerWave * hop;                      // samples between tiles
  static constexpr int halo = tile - step;                               // shared with the next (previous) tile
  // accumulation buffer index: 20 dwords of padding per 2 * hop samples keep the three pairs' lanes on disjoint banks
  static constexpr int buf = tile + 20 * ((tile - 1) / (2 * hop)) + 20;
  static_assert(buf <= kLdsDwordsPerWave || buf <= Hop<H>::lds_dwords, "overlap-add buffer must fit the wave's LDS");
};
template <int H>
AAMD_HD int ola_idx(int pos) { return pos + 20 * (pos / (2 * 20 * H)); }

struct Inv400Geom {
  StftGeom g;              // as OlaGeom::g (length = output samples per row)
  float interior;          // 1: irfft, 0.5: adjoint of the onesided STFT
};

// conj Z[b + 20 q] of the pair (frames ta, ta + 1) -> xr, xi
AAMD_HD void inv400_load(const LaneConst& c, const Inv400Geom& ig, const cplx<float>* row_spec /* frame 0 of the row */,
                         int64_t t0, float (&xr)[20], float (&xi)[20]) {
  const int64_t ta = t0 + 2 * c.p;
  const bool va = ta < ig.g.n_frames, vb = ta + 1 < ig.g.n_frames;
  // frames past the end read frame 0 of the row and are zeroed AFTER the load: a load under a per-lane predicate makes
  // the compiler wait for each one before the next (40 serialised round trips per tile, 20 us)
  const cplx<float>* Sa = row_spec + (va ? ta : 0) * (int64_t)kSpecBins;
  const cplx<float>* Sb = row_spec + (vb ? ta + 1 : 0) * (int64_t)kSpecBins;
  const float ma = va ? 1.0f : 0.0f, mb = vb ? 1.0f : 0.0f;
  // all 40 loads first (branch-free addresses), then the arithmetic with selects: any branch between two loads makes
  // the compiler drain the memory counter
  cplx<float> a[20], b[20];
#pragma unroll
  for (int q = 0; q < 20; ++q) {
    const int k = c.pi + 20 * q;
    const int kk = k < kN - k ? k : kN - k;             // min(k, 400 - k): the onesided bin that defines Z[k]
    a[q] = Sa[kk];
    b[q] = Sb[kk];
  }
#pragma unroll
  for (int q = 0; q < 20; ++q) {
    const int k = c.pi + 20 * q;
    const int kk = k < kN - k ? k : kN - k;
    const bool edge = (kk == 0) | (kk == 200);
    const float wgt = edge ? 1.0f : ig.interior;
    const float sgn = edge ? 0.0f : (k > 200 ? -1.0f : 1.0f);   // Im: dropped at DC / Nyquist, negated in the mirror half
    const float ar = a[q].x * (wgt * ma), ai = a[q].y * (wgt * ma * sgn);
    const float br = b[q].x * (wgt * mb), bi = b[q].y * (wgt * mb * sgn);
    xr[q] = ar - bi;                       // conj(A + i B) = (ar - bi) - i (ai + br)
    xi[q] = -(ai + br);
  }
}

// zero the accumulation buffer (it aliases the transposition rows, which every lane has read by now)
template <int H>
AAMD_HD void inv400_zero(int lane, float* buf) {
  for (int i = lane; i < Inv400<H>::buf; i += 64) buf[i] = 0.0f;
}

// one colour of the overlap-add: lanes of pairs with (p & 1) == parity add frame a (or b) of their pair.  All 20
// words are read before any is written (they are distinct): 20 independent LDS round trips instead of a chain
template <int H>
AAMD_HD void inv400_add(const LaneConst& c, const float* win_row /* window[col + 20 d] * scale */, const float (&zr)[20],
                        const float (&zi)[20], int parity, int frame_b, int n_valid, float* buf) {
  if (!c.active || (c.p & 1) != parity) return;
  const int f = 2 * c.p + frame_b;
  if (f >= n_valid) return;
  const int base = f * Inv400<H>::hop + c.col;
  float cur[20];
#pragma unroll
  for (int d = 0; d < 20; ++d) cur[d] = buf[ola_idx<H>(base + 20 * d)];
#pragma unroll
  for (int d = 0; d < 20; ++d) buf[ola_idx<H>(base + 20 * d)] = cur[d] + (frame_b ? -zi[d] : zr[d]) * win_row[d];
}

// write the tile's samples: u = t0 * hop + pos on the padded axis.  Straight-line code for interior tiles: every
// envelope load and LDS read is in flight before the first store (a rolled loop paid one memory round trip per 64 samples)
template <int H, typename AddFn, typename StoreFn>
AAMD_HD void inv400_flush(int lane, const Inv400Geom& ig, int64_t t0, int n_valid, const float* buf, const float* inv_env,
                          float* out_row, AddFn add, StoreFn store) {
  using I = Inv400<H>;
  constexpr int kIter = (I::tile + 63) / 64;
  const StftGeom& g = ig.g;
  const int64_t start = t0 * (int64_t)I::hop - kPad - g.pad;              // output index of pos 0 (centre)
  const int span = (n_valid - 1) * I::hop + kN;
  const bool interior = n_valid == kFramesPerWave && start >= 0 && start + I::tile <= g.length;
  if (interior) {
    float e[kIter], v[kIter];
    const float* env = inv_env ? inv_env + start : nullptr;
#pragma unroll
    for (int i = 0; i < kIter; ++i) {
      const int pos = lane + 64 * i;
      e[i] = (env && pos < I::tile) ? env[pos] : 1.0f;
    }
#pragma unroll
    for (int i = 0; i < kIter; ++i) {
      const int pos = lane + 64 * i;
      v[i] = pos < I::tile ? buf[ola_idx<H>(pos)] : 0.0f;
    }
    float* o = out_row + start;
#pragma unroll
    for (int i = 0; i < kIter; ++i) {
      const int pos = lane + 64 * i;
      if (pos >= I::tile) continue;
      const float r = v[i] * e[i];
      // plain store only where no other tile's frame reaches (the tile's middle) AND no folded contribution of an edge
      // tile can land: reflect / circular / replicate padding map the outermost n_fft / 2 (+ 1) samples of the row back
      // into [0, kPad] and [length - kPad - 1, length) -- with hop 200 and length = 0 (mod 1200) the last interior
      // tile's middle ends exactly there
      const int64_t sidx = start + pos;
      const bool exclusive = pos >= I::halo && pos < I::step && sidx > kPad && sidx < g.length - kPad - 1;
      if (exclusive) store(o + pos, r);
      else add(o + pos, r);
    }
    return;
  }
  for (int pos = lane; pos < span; pos += 64) {
    const float v0 = buf[ola_idx<H>(pos)];
    const int64_t o = ola_target(g, t0 * (int64_t)I::hop + pos);
    if (o >= 0) add(out_row + o, inv_env ? v0 * inv_env[o] : v0);
  }
}

// 8 waves per workgroup (2 per SIMD, 256-VGPR budget): with 12 the 40 spectrum loads in flight + the two 20-point
// register sets spilled 26 VGPRs, and every spill reload was a serialised memory round trip inside the tile loop
constexpr int kInvWaves = 8;

#if defined(__HIPCC__)
template <int H>
__global__ void __launch_bounds__(64 * kInvWaves, 2)
istft400_kernel(Inv400Geom ig, const cplx<float>* __restrict__ spec, const float* __restrict__ window,
                const float* __restrict__ tw400, const float* __restrict__ inv_env, float* __restrict__ out,
                float out_scale, int tiles_per_row, int64_t n_tiles) {
  extern __shared__ __attribute__((aligned(16))) float smem_i400[];
  using HC = Hop<H>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* lds = smem_i400 + wave * HC::lds_dwords;
  float* const_tab = smem_i400 + kInvWaves * HC::lds_dwords;
  const_tab_build(threadIdx.x, blockDim.x, window, tw400, 2.0f * out_scale, const_tab);   // window * out_scale
  __syncthreads();
  LaneConst c;
  lane_init(lane, const_tab, c);
  const float* win_row = const_tab + 20 * kTwRow + 20 * c.col;           // window[col + 20 d]
  auto add = [](float* p, float v) { atomicAdd(p, v); };
  auto store = [](float* p, float v) { *p = v; };
  const int64_t n_waves = (int64_t)gridDim.x * kInvWaves;
#pragma unroll 1
  for (int64_t tile = (int64_t)blockIdx.x * kInvWaves + wave; tile < n_tiles; tile += n_waves) {
    const int64_t row = tile / tiles_per_row;
    const int64_t t0 = (tile - row * tiles_per_row) * kFramesPerWave;
    const int64_t left = ig.g.n_frames - t0;
    const int n_valid = left < kFramesPerWave ? (int)left : kFramesPerWave;
    float xr[20], xi[20], vr[20], vi[20], zr[20], zi[20];
    inv400_load(c, ig, spec + row * ig.g.n_frames * (int64_t)kSpecBins, t0, xr, xi);
    wave_lds_fence();
    phase_a_core<0>(c, xr, xi, lds);
    wave_lds_fence();
    phase_b1_load(c, lds, vr, vi);
    wave_lds_fence();
    inv400_zero<H>(lane, lds);
    dft20(vr, vi, zr, zi);
    wave_lds_fence();
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      inv400_add<H>(c, win_row, zr, zi, ph & 1, ph >> 1, n_valid, lds);
      wave_lds_fence();
    }
    inv400_flush<H>(lane, ig, t0, n_valid, lds, inv_env, out + row * ig.g.length, add, store);
    wave_lds_fence();
  }
}
#endif  // __HIPCC__

}  // namespace m400
}  // namespace aamd

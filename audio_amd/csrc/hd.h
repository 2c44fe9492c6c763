// Portability shim: the per-thread "phase" functions of every kernel are plain inline
// functions so that tests/cpu_sim can compile the SAME source with g++ and replay a
// workgroup phase-by-phase on the CPU (there is no GPU in the build container).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AAMD_HD __host__ __device__ __forceinline__
#define AAMD_D __device__ __forceinline__
#else
#include <cmath>
#define AAMD_HD inline
#define AAMD_D inline
using std::sqrt;
using std::pow;
using std::log;
using std::log10;
using std::fmax;
using std::fmin;
using std::fma;
#endif

namespace aamd {

// 8- / 16-byte LDS accesses (ds_read_b64 / ds_read_b128 need their natural alignment)
#if defined(__HIPCC__)
using F2 = float2;
using F4 = float4;
#else
struct alignas(8) F2 { float x, y; };
struct alignas(16) F4 { float x, y, z, w; };
#endif

template <typename T>
struct cplx {
  T x, y;
};

template <typename T>
AAMD_HD cplx<T> cmul(cplx<T> a, cplx<T> b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T>
AAMD_HD cplx<T> cadd(cplx<T> a, cplx<T> b) {
  return {a.x + b.x, a.y + b.y};
}
template <typename T>
AAMD_HD cplx<T> csub(cplx<T> a, cplx<T> b) {
  return {a.x - b.x, a.y - b.y};
}

// Padded-signal index -> source index, or -1 for a constant-zero sample.
// Restates aten's reflect/replicate/circular/constant padding used by
// torch.stft(center=True) (torch/functional.py:675-680).
AAMD_HD int64_t pad_source_index(int64_t i, int64_t len, int mode) {
  if (i >= 0 && i < len) return i;
  switch (mode) {
    case 0:  // reflect (edge sample not repeated); requires pad < len
      if (i < 0) i = -i;
      if (i >= len) i = 2 * (len - 1) - i;
      return (i >= 0 && i < len) ? i : -1;
    case 2:  // replicate
      return i < 0 ? 0 : len - 1;
    case 3: {  // circular
      int64_t m = i % len;
      return m < 0 ? m + len : m;
    }
    default:
      return -1;
  }
}

}  // namespace aamd

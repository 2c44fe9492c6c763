// Lab build of the real-block fftconvolve kernel (tools only): audio_amd/csrc/fftconv_fdr.h compiled alone, one shared library per
// source variant (-D switches), so that an A/B of a kernel change builds in seconds and several variants run interleaved in
// one process (tools/fdr_lab.py).  The launch logic is the plan-3 branch of aamd_fftconvolve_staged_f32 (csrc/c_api.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../audio_amd/csrc/fftconv_fdr.h"

using namespace aamd;

// workspace: twiddles (fco::kN complex) | tap spectra (tap_rows * n_part * 8192 complex)
extern "C" int64_t lab_fdr_workspace(int64_t tap_rows, int64_t ny) {
  const int64_t np = (ny + fdr::kHop - 1) / fdr::kHop;
  return (int64_t)sizeof(fco::C32) * (fco::kN + tap_rows * np * fdr::kHPerPart);
}

extern "C" int lab_fdr(const float* x, const float* y, float* out, int64_t rows, int64_t tap_rows, int64_t nx, int64_t ny,
                       void* workspace, int stages, int cu_count, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  fdr::Geom g{};
  g.rows = rows; g.nx = nx; g.ny = ny; g.start = 0; g.out_len = nx + ny - 1;
  if (!fdr::plan(rows, ny, g.out_len, cu_count, g)) return -3;
  fco::C32* tw = reinterpret_cast<fco::C32*>(workspace);
  fco::C32* H = tw + fco::kN;
  const size_t lds_r = (size_t)fdr::kLdsComplex * sizeof(fco::C32);
  if (stages & 1) {
    hipLaunchKernelGGL(fco::twiddle_kernel, dim3(fco::kN / 256), dim3(256), 0, s, tw);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fdr::spectrum_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_r) != hipSuccess) return -2;
    hipLaunchKernelGGL(fdr::spectrum_kernel, dim3((unsigned)(tap_rows * g.n_part)), dim3(fdr::kThreads), lds_r, s, ny, g.n_part, y,
                       tw, H);
  }
  if (stages & 2) {
    int64_t blocks = cu_count;
    if (blocks > rows * g.segs) blocks = rows * g.segs;
    const int64_t* ymap = nullptr;
    // tap_rows == 1: every output row uses tap row 0 -> the kernel needs a row map; the lab passes identity taps per row or one
    // shared row through a device map built by the caller (out of scope here: tap_rows must be 1 or rows)
#define LAB_FDR(NP)                                                                                                       \
    do {                                                                                                                  \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(fdr::delay_line_kernel<NP>),                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r) != hipSuccess) return -2;           \
      hipLaunchKernelGGL(fdr::delay_line_kernel<NP>, dim3((unsigned)blocks), dim3(fdr::kThreads), lds_r, s, g, x, tw, H,   \
                         (const int64_t*)nullptr, ymap, out);                                                             \
    } while (0)
    if (tap_rows == 1) ymap = reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(workspace) + lab_fdr_workspace(1, ny));
    if (g.n_part == 1) LAB_FDR(1); else if (g.n_part == 2) LAB_FDR(2); else if (g.n_part == 3) LAB_FDR(3); else LAB_FDR(4);
#undef LAB_FDR
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

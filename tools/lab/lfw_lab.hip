// Lab build of the biquad-cascade kernel (tools only): the SAME audio_amd/csrc/lfilter_wave.h compiled alone, one shared
// library per source / compiler-flag variant (tools/lfw_ab.py), so that A/B runs do not need the whole product library.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/audio_amd.h"
#ifdef LAB_LFW_HEADER
#include LAB_LFW_HEADER
#else
#include "../../audio_amd/csrc/lfilter_wave.h"
#endif

using namespace aamd;

#ifndef LAB_BITS
#define LAB_BITS 0
#endif

extern "C" int lab_lfw_mover(const float* x, const float* a, const float* b, float* y, int64_t n_seq, int channels,
                             int64_t length, int n_order, int n_coeff_rows, int n_stages, int clamp, void* stream) {
#ifndef LAB_W
#define LAB_W 8
#endif
  const size_t plds = lfw::pipe_lds_bytes(8, n_stages) + 4096;
  auto kern = lfw::lfilter_wave_mover_kernel<LAB_BITS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds) != hipSuccess)
    return -2;
  int blocks = (int)(n_seq < 2048 ? n_seq : 2048);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * (LAB_W + lfw::kMovers)), plds, (hipStream_t)stream, x, a, b, y, n_seq, channels,
                     length, n_order, n_coeff_rows, n_stages, clamp);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

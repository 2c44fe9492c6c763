// Lab build of the headline kernel (tools only): the SAME audio_amd/csrc/melspec400.h compiled alone, one shared library
// per source variant (-D switches), so that A/B runs of kernel changes take 30 s to build instead of the whole product
// library, and so that profiling variants (LAB bits of melspec400_kernel) can be launched from tools/mel400_lab.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/audio_amd.h"
#include "../../audio_amd/csrc/melspec400.h"

using namespace aamd;

#ifndef LAB_BITS
#define LAB_BITS 0
#endif

static long long* g_dbg = nullptr;
extern "C" void lab_set_debug(void* p) { g_dbg = (long long*)p; }

template <int LAB>
static int launch(const float* wav, const float* window, const float* tw, const MelBandsDev& mb, float* out, int64_t rows,
                  int64_t length, int64_t row_stride, int n_frames, float scale, int out_wide, int blocks_override,
                  hipStream_t s) {
  const int tiles_per_row = (n_frames + m400::kFramesPerWave - 1) / m400::kFramesPerWave;
  const int64_t n_tiles = rows * tiles_per_row;
  const int wpb = m400::kWavesPerBlock;
  const size_t lds = m400::lds_bytes(mb.n_mels, mb.max_width, m400::Hop<8>::lds_dwords);
#ifndef LAB_SIG
#define LAB_SIG 0
#endif
  auto kern = m400::melspec400_kernel<LAB, m400::EPI400_MEL, 8, float, 4, LAB_SIG>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
  hipDeviceProp_t dp;
  int dev = 0;
  hipGetDevice(&dev);
  static int cus = 0;
  if (!cus) { hipGetDeviceProperties(&dp, dev); cus = dp.multiProcessorCount; }
  int64_t blocks = blocks_override > 0 ? blocks_override : cus;
  const int64_t need = (n_tiles + wpb - 1) / wpb;
  if (blocks > need) blocks = need;
  if (blocks >= 8) blocks -= blocks % 8;
  if (blocks < 1) blocks = 1;
  const int tiles_per_block = (int)((n_tiles + blocks - 1) / blocks);
  const int in_aligned = (reinterpret_cast<uintptr_t>(wav) % 16 == 0) && (row_stride % 4 == 0);
  m400::Epi400 epi{};
  epi.fix_count = reinterpret_cast<int*>(g_dbg);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wpb), lds, s, wav, window, tw, mb, out, rows, length, row_stride,
                     n_frames, scale, tiles_per_row, n_tiles, tiles_per_block, in_aligned, out_wide, epi);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int lab_mel400(const float* wav, const float* window, const float* tw, const aamd_mel_bands* b, float* out,
                          int64_t rows, int64_t length, int64_t row_stride, int n_frames, float scale, int out_wide,
                          int blocks_override, void* stream) {
  MelBandsDev mb{};
  mb.n_mels = b->n_mels; mb.max_width = b->max_width; mb.lo = b->lo; mb.width = b->width; mb.weights = b->weights;
  mb.order = b->lane_order; mb.table400 = b->table400; mb.table_sig = b->table_sig;
  return launch<LAB_BITS>(wav, window, tw, mb, out, rows, length, row_stride, n_frames, scale, out_wide, blocks_override,
                          (hipStream_t)stream);
}
extern "C" int lab_info(int* waves, int* lds_per_wave_dwords) {
  *waves = m400::kWavesPerBlock;
  *lds_per_wave_dwords = m400::Hop<8>::lds_dwords;
  return LAB_BITS;
}

// ---- the one-kernel MFCC (EPI400_MFCC), both passes: what aamd_mfcc_fused_f32 + launch_fft400_nr do for hop 160 / 80 mels --------
#ifndef LAB_MFCC_SIG
#define LAB_MFCC_SIG 0
#endif
#ifndef LAB_MFCC_BITS
#define LAB_MFCC_BITS 0
#endif
extern "C" int lab_mfcc_frag_floats(void) { return m400::kMfccFragFloats; }
extern "C" int lab_mfcc_frag_build(const float* dct, int n_mels, int n_mfcc, float* frag, void* stream) {
  hipLaunchKernelGGL(m400::mfcc_frag_build_kernel, dim3(15), dim3(256), 0, (hipStream_t)stream, dct, n_mels, n_mfcc, frag);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int lab_mfcc400(const float* wav, const float* window, const float* tw, const aamd_mel_bands* b, float* out,
                           int64_t rows, int64_t length, int64_t row_stride, int n_frames, float scale,
                           const aamd_mfcc_fused* f, int lab, void* stream) {
  MelBandsDev mb{};
  mb.n_mels = b->n_mels; mb.max_width = b->max_width; mb.lo = b->lo; mb.width = b->width; mb.weights = b->weights;
  mb.order = b->lane_order; mb.table400 = b->table400; mb.table_sig = b->table_sig;
  const int tiles_per_row = (n_frames + m400::kFramesPerWave - 1) / m400::kFramesPerWave;
  const int64_t n_tiles = rows * tiles_per_row;
  const int wpb = m400::kWavesPerBlock;
  const int wdw = m400::Hop<8>::lds_dwords;
  m400::Epi400 epi{};
  epi.multiplier = f->multiplier; epi.amin = f->amin; epi.db_sub = f->multiplier * f->db_multiplier;
  epi.group_max = f->group_max; epi.rows_per_group = f->rows_per_group;
  epi.dct_frag = f->dct_frag; epi.n_mfcc = f->n_mfcc; epi.top_db = f->top_db; epi.tile_min = f->tile_min;
  epi.fix_count = f->fix_count; epi.fixup = f->pass; epi.fix_list = f->tile_list;
  epi.lab = lab;
  size_t lds = m400::lds_bytes(mb.n_mels, mb.max_width, wdw, true);
  epi.frag_in_lds = 1;
  auto kern = m400::melspec400_kernel<LAB_MFCC_BITS, m400::EPI400_MFCC, 8, float, 4, LAB_MFCC_SIG>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
  static int cus = 0;
  if (!cus) { hipDeviceProp_t dp; int dev = 0; hipGetDevice(&dev); hipGetDeviceProperties(&dp, dev); cus = dp.multiProcessorCount; }
  int64_t blocks = cus;
  const int64_t need = (n_tiles + wpb - 1) / wpb;
  if (blocks > need) blocks = need;
  if (blocks >= 8) blocks -= blocks % 8;
  if (blocks < 1) blocks = 1;
  const int tiles_per_block = (int)((n_tiles + blocks - 1) / blocks);
  const int in_aligned = (reinterpret_cast<uintptr_t>(wav) % 16 == 0) && (row_stride % 4 == 0);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wpb), lds, (hipStream_t)stream, wav, window, tw, mb, out, rows, length,
                     row_stride, n_frames, scale, tiles_per_row, n_tiles, tiles_per_block, in_aligned, 0, epi);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- the Spectrogram epilogue (EPI400_SPEC, power 2): what launch_fft400_nr does for hop 160 ---------------------------------------
#ifndef LAB_SPEC_BITS
#define LAB_SPEC_BITS 0
#endif
extern "C" int lab_spec400(const float* wav, const float* window, const float* tw, float* out, int64_t rows, int64_t length,
                           int64_t row_stride, int n_frames, float scale, float power, void* stream) {
  MelBandsDev mb{};
  const int tiles_per_row = (n_frames + m400::kFramesPerWave - 1) / m400::kFramesPerWave;
  const int64_t n_tiles = rows * tiles_per_row;
  const int wpb = m400::kWavesPerBlock;
  const size_t lds = m400::lds_bytes(0, 1, m400::Hop<8>::lds_dwords);
  m400::Epi400 epi{};
  epi.power = power;
  auto kern = m400::melspec400_kernel<LAB_SPEC_BITS, m400::EPI400_SPEC, 8, float, m400::kMelMaxRounds, 0>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
  static int cus = 0;
  if (!cus) { hipDeviceProp_t dp; int dev = 0; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&dp, dev); cus = dp.multiProcessorCount; }
  int64_t blocks = cus;
  const int64_t need = (n_tiles + wpb - 1) / wpb;
  if (blocks > need) blocks = need;
  if (blocks >= 8) blocks -= blocks % 8;
  if (blocks < 1) blocks = 1;
  const int tiles_per_block = (int)((n_tiles + blocks - 1) / blocks);
  const int in_aligned = (reinterpret_cast<uintptr_t>(wav) % 16 == 0) && (row_stride % 4 == 0);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wpb), lds, (hipStream_t)stream, wav, window, tw, mb, out, rows, length,
                     row_stride, n_frames, scale, tiles_per_row, n_tiles, tiles_per_block, in_aligned, 1, epi);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

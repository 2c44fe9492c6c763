// Lab build of the binary16-split resampler (tools only): audio_amd/csrc/resample_mfma.h compiled alone, one shared library per
// source variant (-D switches), so that an A/B of a kernel change builds in seconds and several variants run interleaved in one
// process (tools/rsm_lab.py).  The launch logic is the f16 branch of aamd_resample_banded_f32 (csrc/c_api.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../audio_amd/csrc/resample_mfma.h"

using namespace aamd;

#ifndef LAB_RSM_RD
#define LAB_RSM_RD 0
#endif
#ifndef LAB_RSM_LABM              /* 2: the instantiation that honours the timing-only switches of Geom::lab (wrong results) */
#define LAB_RSM_LABM 0
#endif
#ifndef LAB_RSM_BITS              /* Geom::lab of this variant: 1 no conversion, 2 no MFMA, 4 no LDS operand reads, 8 no global loads, 16 no tap fragments, 32 no stores */
#define LAB_RSM_BITS 0
#endif

extern "C" int lab_rsm_rd() { return LAB_RSM_RD; }

extern "C" int lab_rsm(const float* wav, const float* kernel, float* out, int64_t rows, int64_t length, int64_t row_stride,
                       int orig, int new_, int width, int64_t out_len, const int32_t* tap_lo, int tap_span, int lab,
                       int cu_count, void* stream) {
  const int n_tiles = (new_ + 15) / 16;
#ifndef LAB_RSM_KS                 /* the k-step count of this variant: 112 (rounds 2-5) or 104 (round 6: the cfg3 band is 414 taps) */
#define LAB_RSM_KS 112
#endif
  const int ks = LAB_RSM_KS;
  if (rsm::pick_ks(tap_span, orig) == 0 || rsm::pick_ks(tap_span, orig) > ks) return -2;
  rsm::Geom g{};
  g.lab = LAB_RSM_BITS ? LAB_RSM_BITS : lab;
  g.rows = rows; g.length = length; g.row_stride = row_stride; g.out_len = out_len;
  g.orig = orig; g.new_ = new_; g.width = width; g.taps = 2 * width + orig;
  g.vec_in = (reinterpret_cast<uintptr_t>(wav) % 16 == 0) && (row_stride % 4 == 0);
  g.vec_out = (reinterpret_cast<uintptr_t>(out) % 16 == 0) && (out_len % 4 == 0) && (new_ % 4 == 0);
#if defined(LAB_RSM_FRAG)          /* prepared tap fragments (built once per process: the lab runs one filter) */
  {
    static uint32_t* frag = nullptr;
    if (frag == nullptr) {
      if (hipMalloc(reinterpret_cast<void**>(&frag), (size_t)rsm::frag_bytes(n_tiles, ks)) != hipSuccess) return -7;
      rsm::Geom b = g;
      for (int pt0 = 0; pt0 < n_tiles; pt0 += rsm::kMaxPhaseTiles) {
        b.pt0 = pt0;
        b.n_pt = n_tiles - pt0 < rsm::kMaxPhaseTiles ? n_tiles - pt0 : rsm::kMaxPhaseTiles;
        for (int t = 0; t < b.n_pt; ++t) b.tap_lo[t] = tap_lo[pt0 + t];
        const int n = b.n_pt * (ks / 8) * 64;
        hipLaunchKernelGGL(rsm::frag_build_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b, ks, kernel, frag);
      }
      if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -8;
    }
    g.frag = frag;
  }
#endif
  const int64_t nq = (out_len + new_ - 1) / new_;
  const int max_cw = rsm::max_compute_waves(ks);
  const size_t lds_cap = 160 * 1024;
  for (int pt0 = 0; pt0 < n_tiles; pt0 += max_cw) {
    g.pt0 = pt0;
    g.n_pt = n_tiles - pt0 < max_cw ? n_tiles - pt0 : max_cw;
    int max_lo = 0;
    for (int t = 0; t < g.n_pt; ++t) {
      g.tap_lo[t] = tap_lo[pt0 + t];
      if (g.tap_lo[t] > max_lo) max_lo = g.tap_lo[t];
    }
    if (!rsm::plan_chunk(g, ks, true, nq, max_lo, lds_cap)) return -3;
    const int qc = rsm::chunk_q(g);
    const size_t lds = 2 * (size_t)g.buf_floats * sizeof(float) + 48;
    g.chunks_per_row = (int)((nq + qc - 1) / qc);
    g.n_chunks = rows * g.chunks_per_row;
    const int wg_waves = g.n_pt * g.qg + g.n_loaders;
    int per_cu = ks >= 80 ? 1 : 16 / wg_waves;
    if (per_cu > (int)(lds_cap / lds)) per_cu = (int)(lds_cap / lds);
    if (per_cu < 1) per_cu = 1;
    int64_t blocks = (int64_t)cu_count * per_cu;
    if (blocks > g.n_chunks) blocks = g.n_chunks;
    g.chunks_per_block = (int)((g.n_chunks + blocks - 1) / blocks);
    blocks = (g.n_chunks + g.chunks_per_block - 1) / g.chunks_per_block;
    constexpr int RD = LAB_RSM_RD;
    if (RD && !rsm::b64_ok(ks, orig)) return -5;
    const bool full = RD && rsm::chunk_is_full(g, ks);
    auto kern = full ? rsm::resample_f16_kernel<LAB_RSM_KS, LAB_RSM_LABM, RD, RD ? 1 : 0> : rsm::resample_f16_kernel<LAB_RSM_KS, LAB_RSM_LABM, RD, 0>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return -6;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * wg_waves), lds, (hipStream_t)stream, g, wav, kernel, out);
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Design lab for the headline kernel (NOT product): ablation variants of melspec400_kernel<LAB, 0, 8, float, NR> built
// from the same phase functions, timed with HIP events on one MI355X, plus a float64 host check.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize mel400_lab.hip -o mel400_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../audio_amd/csrc/melspec400.h"

using namespace aamd;
using namespace aamd::m400;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e_), #x); exit(1); } } while (0)

// pure-VALU probe: a chain of in-register DFT-20s (no LDS, no memory)
__global__ void __launch_bounds__(384, 3) dft_probe(float* out, int reps, float a, long long* cyc) {
  const long long t0 = __builtin_readcyclecounter();
  float xr[20], xi[20], yr[20], yi[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) { xr[i] = threadIdx.x * a + i; xi[i] = threadIdx.x - i * a; }
  for (int r = 0; r < reps; ++r) {
    dft20(xr, xi, yr, yi);
#pragma unroll
    for (int i = 0; i < 20; ++i) { xr[i] = yr[i] * a; xi[i] = yi[i] * a; }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 20; ++i) s += xr[i] + xi[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = __builtin_readcyclecounter() - t0;
}

static void htk_bands(int n_mels, std::vector<int>& lo, std::vector<int>& width, std::vector<float>& w, int& maxw) {
  const int F = 201; const double fmax = 8000.0;
  auto mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
  auto imel = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
  std::vector<double> fp(n_mels + 2);
  for (int i = 0; i < n_mels + 2; ++i) fp[i] = imel(mel(0) + (mel(fmax) - mel(0)) * i / (n_mels + 1));
  std::vector<std::vector<float>> fb(n_mels, std::vector<float>(F, 0.f));
  for (int m = 0; m < n_mels; ++m)
    for (int k = 0; k < F; ++k) {
      double f = 8000.0 * k / (F - 1);
      double up = (f - fp[m]) / (fp[m + 1] - fp[m]), dn = (fp[m + 2] - f) / (fp[m + 2] - fp[m + 1]);
      double v = std::fmax(0.0, std::fmin(up, dn));
      fb[m][k] = (float)v;
    }
  lo.assign(n_mels, 0); width.assign(n_mels, 0); maxw = 1;
  for (int m = 0; m < n_mels; ++m) {
    int a = -1, b = -1;
    for (int k = 0; k < F; ++k) if (fb[m][k] != 0.f) { if (a < 0) a = k; b = k; }
    if (a >= 0) { lo[m] = a; width[m] = b - a + 1; if (width[m] > maxw) maxw = width[m]; }
  }
  w.assign((size_t)n_mels * maxw, 0.f);
  for (int m = 0; m < n_mels; ++m) for (int j = 0; j < width[m]; ++j) w[(size_t)m * maxw + j] = fb[m][lo[m] + j];
}

static int g_rotate = 1;

template <int LAB, int NR = kMelMaxRounds>
static float run(const char* name, int blocks, size_t lds, const float* wav, const float* win, const float* tw,
                 MelBandsDev mb, float* out, int64_t rows, int64_t L, int T, int iters, int in_aligned, int out_wide) {
  const int tiles_per_row = (T + kFramesPerWave - 1) / kFramesPerWave;
  const int64_t n_tiles = rows * tiles_per_row;
  const int tpw = (int)((n_tiles + blocks - 1) / blocks);   // tiles per block
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(melspec400_kernel<LAB, 0, 8, float, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  static unsigned* ctr = nullptr;          // one zeroed ticket counter per launch (chip-wide queue variants)
  if (!ctr) CK(hipMalloc(&ctr, 4096 * 64));
  if (LAB & 131072) CK(hipMemset(ctr, 0, 4096 * 64));
  int nl = 0;
  auto epi_for = [&]() { Epi400 e{}; if (LAB & 131072) e.group_max = reinterpret_cast<float*>(ctr + 16 * (nl++ % 4096)); return e; };
  const int wout = g_rotate ? 4 : 1;       // rotate over 4 input / output buffer pairs (working set > 256 MiB L3)
  for (int i = 0; i < (name[0] ? 10 : 3); ++i)
    hipLaunchKernelGGL((melspec400_kernel<LAB, 0, 8, float, NR>), dim3(blocks), dim3(64 * kWavesPerBlock), lds, 0, wav + (size_t)(i % wout) * rows * L, win, tw, mb, out + (size_t)(i % wout) * rows * T * 80, rows, L, L, T, 1.0f, tiles_per_row, n_tiles, tpw, in_aligned, out_wide, epi_for());
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((melspec400_kernel<LAB, 0, 8, float, NR>), dim3(blocks), dim3(64 * kWavesPerBlock), lds, 0, wav + (size_t)(i % wout) * rows * L, win, tw, mb, out + (size_t)(i % wout) * rows * T * 80, rows, L, L, T, 1.0f, tiles_per_row, n_tiles, tpw, in_aligned, out_wide, epi_for());
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  if (name[0]) printf("%-44s %8.1f us\n", name, ms * 1e3f / iters);
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  const int64_t rows = 256, L = 160000; const int T = 1001, M = 80;
  int occ_blocks = argc > 1 ? atoi(argv[1]) : 3;
  const size_t lds_pad = argc > 2 ? (size_t)atoi(argv[2]) : 0;
  std::vector<float> hx((size_t)rows * L), hwin(400), htw(800);
  srand(1);
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  for (int n = 0; n < 400; ++n) hwin[n] = 0.5f - 0.5f * (float)std::cos(2.0 * M_PI * n / 400.0);
  for (int n = 0; n < 400; ++n) { htw[2 * n] = (float)std::cos(2.0 * M_PI * n / 400.0); htw[2 * n + 1] = (float)-std::sin(2.0 * M_PI * n / 400.0); }
  std::vector<int> lo, wd; std::vector<float> ww; int maxw;
  htk_bands(M, lo, wd, ww, maxw);
  float *dx, *dwin, *dtw, *dw, *dout; int *dlo, *dwd;
  CK(hipMalloc(&dx, hx.size() * 4 * 4)); CK(hipMalloc(&dwin, 1600)); CK(hipMalloc(&dtw, 4096 + 24 * 4096));
  CK(hipMalloc(&dw, ww.size() * 4)); CK(hipMalloc(&dlo, M * 4)); CK(hipMalloc(&dwd, M * 4));
  CK(hipMalloc(&dout, (size_t)rows * T * M * 4 * 4));
  for (int r = 0; r < 4; ++r) CK(hipMemcpy(dx + (size_t)r * hx.size(), hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwin, hwin.data(), 1600, hipMemcpyHostToDevice)); CK(hipMemcpy(dtw, htw.data(), 3200, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, ww.data(), ww.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dlo, lo.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwd, wd.data(), M * 4, hipMemcpyHostToDevice));
  MelBandsDev mb{M, maxw, dlo, dwd, dw};
  const size_t lds = lds_bytes(M, maxw) + lds_pad;
  const int blocks = 256 * occ_blocks;
  int occ = 0;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(melspec400_kernel<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, melspec400_kernel<0, 0>, 64 * kWavesPerBlock, lds));
  printf("blocks/CU %d (occupancy API: %d), LDS/block %zu B, max band width %d\n", occ_blocks, occ, lds, maxw);
  const int it = 50;
  run<0>("product: staged in + wide out", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  // float64 check of a few frames
  {
    std::vector<float> ho((size_t)T * M);
    CK(hipMemcpy(ho.data(), dout + (size_t)3 * T * M, ho.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, peak = 0;
    for (int t : {0, 1, 2, 500, 999, 1000}) {
      std::vector<double> P(201);
      for (int k = 0; k <= 200; ++k) {
        double re = 0, im = 0;
        for (int n = 0; n < 400; ++n) {
          int64_t i = (int64_t)t * 160 - 200 + n; if (i < 0) i = -i; if (i >= L) i = 2 * (L - 1) - i;
          double v = (double)hx[3 * L + i] * hwin[n];
          re += v * std::cos(2 * M_PI * n * k / 400.0); im -= v * std::sin(2 * M_PI * n * k / 400.0);
        }
        P[k] = re * re + im * im;
      }
      for (int m = 0; m < M; ++m) {
        double acc = 0; for (int j = 0; j < wd[m]; ++j) acc += (double)ww[(size_t)m * maxw + j] * P[lo[m] + j];
        worst = std::fmax(worst, std::fabs(acc - ho[(size_t)t * M + m])); peak = std::fmax(peak, std::fabs(acc));
      }
    }
    printf("float64 check: peak-rel err %.3e\n", worst / peak);
  }
  run<0>("product: staged in + narrow out", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 0);
  if (getenv("LAB_R4")) {   // round-2, third batch: where do the LDS bank conflicts come from?  (PMC per variant: tools/pmc_lab.sh)
    MelBandsDev mbt = mb;
    float* dtab; CK(hipMalloc(&dtab, (size_t)mel_tab_dwords(M, maxw) * 4));
    CK(hipMemset(dtab, 0, (size_t)mel_tab_dwords(M, maxw) * 4));
    hipLaunchKernelGGL(mel_tab_build_kernel, dim3(1), dim3(256), 0, 0, mb, dtab);
    CK(hipDeviceSynchronize());
    mbt.table400 = dtab;
    for (int i = 0; i < 3; ++i) run<0, 4>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 120, 1, 0);
    run<0, 4>("R4 product (NR4, prebuilt table)", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0);
    run<11, 4>("R4 LAB11 no DMA / wait / stores", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0);
    run<15, 4>("R4 LAB15 ... and no phase C", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0);
    run<31, 4>("R4 LAB31 ... and no phase B", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0);
    fflush(stdout);
    if (getenv("LAB_R4_ONLY")) return 0;
  }
  if (getenv("LAB_R3")) {   // round-2, second batch: everything with the prebuilt table; NR = 4 = control words in registers
    MelBandsDev mbt = mb;
    float* dtab; CK(hipMalloc(&dtab, (size_t)mel_tab_dwords(M, maxw) * 4));
    CK(hipMemset(dtab, 0, (size_t)mel_tab_dwords(M, maxw) * 4));
    hipLaunchKernelGGL(mel_tab_build_kernel, dim3(1), dim3(256), 0, 0, mb, dtab);
    CK(hipDeviceSynchronize());
    mbt.table400 = dtab;
    // warm the clocks: ~60 ms of launches
    for (int i = 0; i < 6; ++i) run<0>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 120, 1, 0);
    const int NV = 8;
    std::vector<float> r[NV];
    const char* nm[NV] = {"NR8 window regs", "NR8 window in LDS (262144)", "NR4 window regs (candidate product)",
                          "NR4 window in LDS", "NR4 + no stage wait (LAB1)", "NR4 + no stores (LAB2)",
                          "NR4 no DMA/wait/stores (LAB11)", "NR4 WIDE stores"};
    for (int rep = 0; rep < 9; ++rep) {
      r[0].push_back(run<0>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0));
      r[1].push_back(run<262144>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0));
      r[2].push_back(run<0, 4>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0));
      r[3].push_back(run<262144, 4>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0));
      r[4].push_back(run<1, 4>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0));
      r[5].push_back(run<2, 4>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0));
      r[6].push_back(run<11, 4>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 0));
      r[7].push_back(run<0, 4>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 40, 1, 1));
    }
    for (int i = 0; i < NV; ++i) { std::sort(r[i].begin(), r[i].end()); printf("R3 A/B median %-40s %7.1f us (min %.1f max %.1f)\n", nm[i], r[i][4], r[i].front(), r[i].back()); }
    std::vector<float> ref((size_t)rows * T * M), got((size_t)rows * T * M);
    g_rotate = 0;
    run<262144>("", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, 1, 1, 0);
    CK(hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(dout, 0, ref.size() * 4)); run<0, 4>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 1, 1, 0);
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; double worst = 0; for (size_t i = 0; i < ref.size(); ++i) { bad += (ref[i] != got[i]); worst = std::fmax(worst, std::fabs((double)ref[i] - got[i])); }
    printf("R3 check NR4 + window regs + prebuilt table vs LDS-window in-kernel-table NR8: %zu of %zu values differ, max abs %.3e\n", bad, ref.size(), worst);
    g_rotate = 1;
    { // census of the candidate
      CK(hipMemset(dtw + 1024, 0, 24 * 4096));
      run<1024, 4>("census run (NR4)", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 1, 1, 0);
      const int nw = blocks * kWavesPerBlock;
      std::vector<long long> rec(3 * nw);
      CK(hipMemcpy(rec.data(), dtw + 1024, rec.size() * 8, hipMemcpyDeviceToHost));
      long long tmin = rec[0], tmax = rec[2];
      for (int w = 0; w < nw; ++w) if (rec[3 * w + 2] >= rec[0]) { tmin = std::min(tmin, rec[3 * w]); tmax = std::max(tmax, rec[3 * w + 2]); }
      double e = 0, rr = 0, d = 0, emax = 0, rmax = 0, dmin = 1e30; int used = 0;
      std::vector<double> blk_done(blocks, 0.0);
      for (int w = 0; w < nw; ++w) {
        if (rec[3 * w + 2] < rec[0]) continue;
        ++used;
        const double a = (rec[3 * w] - tmin) * 0.01, bb = (rec[3 * w + 1] - tmin) * 0.01, c = (rec[3 * w + 2] - tmin) * 0.01;
        e += a; rr += bb; d += c; emax = std::max(emax, a); rmax = std::max(rmax, bb); dmin = std::min(dmin, c);
        blk_done[w / kWavesPerBlock] = std::max(blk_done[w / kWavesPerBlock], c);
      }
      printf("R3 census (us from first wave entry): entry mean %.1f max %.1f | tables ready mean %.1f max %.1f | done mean %.1f min %.1f max %.1f\n",
             e / used, emax, rr / used, rmax, d / used, dmin, (tmax - tmin) * 0.01);
      // per-XCD finishing times (block b runs on XCD b % 8)
      for (int x = 0; x < 8; ++x) { double mx = 0, mn = 1e30, sm = 0; int n = 0; for (int b = x; b < blocks; b += 8) { mx = std::max(mx, blk_done[b]); mn = std::min(mn, blk_done[b]); sm += blk_done[b]; ++n; }
        printf("R3 census XCD %d: block done min %.1f mean %.1f max %.1f\n", x, mn, sm / n, mx); }
    }
    fflush(stdout);
    if (getenv("LAB_R3_ONLY")) return 0;
  }
  if (getenv("LAB_R2")) {   // round-2 experiments: interleaved A/B medians, buffers rotating over > L3
    MelBandsDev mbt = mb;
    float* dtab; CK(hipMalloc(&dtab, (size_t)mel_tab_dwords(M, maxw) * 4));
    CK(hipMemset(dtab, 0, (size_t)mel_tab_dwords(M, maxw) * 4));
    hipLaunchKernelGGL(mel_tab_build_kernel, dim3(1), dim3(256), 0, 0, mb, dtab);
    CK(hipDeviceSynchronize());
    mbt.table400 = dtab;
    const int NV = 10;
    std::vector<float> r[NV];
    const char* nm[NV] = {"product (in-kernel table build)", "prebuilt table", "prebuilt + direct global gather",
                          "prebuilt + chip-wide queue", "prebuilt + gather + queue", "prebuilt + window regs",
                          "prebuilt + gather + window regs", "prebuilt + gather + queue + window regs",
                          "prebuilt, no stage wait (LAB1)", "prebuilt + gather + queue, WIDE stores"};
    for (int rep = 0; rep < 7; ++rep) {
      r[0].push_back(run<0>("", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, 24, 1, 0));
      r[1].push_back(run<0>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 0));
      r[2].push_back(run<32768>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 0));
      r[3].push_back(run<131072>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 0));
      r[4].push_back(run<32768 + 131072>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 0));
      r[5].push_back(run<8192>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 0));
      r[6].push_back(run<32768 + 8192>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 0));
      r[7].push_back(run<32768 + 131072 + 8192>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 0));
      r[8].push_back(run<1>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 0));
      r[9].push_back(run<32768 + 131072>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 24, 1, 1));
    }
    for (int i = 0; i < NV; ++i) { std::sort(r[i].begin(), r[i].end()); printf("R2 A/B median %-44s %7.1f us (min %.1f max %.1f)\n", nm[i], r[i][3], r[i].front(), r[i].back()); }
    // correctness of the variants against the product output (bit-identical expected: same arithmetic, other data path)
    std::vector<float> ref((size_t)rows * T * M), got((size_t)rows * T * M);
    run<0>("", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, 1, 1, 0);
    CK(hipMemcpy(ref.data(), dout, ref.size() * 4, hipMemcpyDeviceToHost));
    auto cmp = [&](const char* what) {
      CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += (ref[i] != got[i]);
      printf("R2 check %-40s %zu of %zu values differ\n", what, bad, ref.size());
    };
    g_rotate = 0;
    CK(hipMemset(dout, 0, ref.size() * 4)); run<0>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 1, 1, 0); cmp("prebuilt table");
    CK(hipMemset(dout, 0, ref.size() * 4)); run<32768>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 1, 1, 0); cmp("direct global gather");
    CK(hipMemset(dout, 0, ref.size() * 4)); run<131072>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 1, 1, 0); cmp("chip-wide queue");
    CK(hipMemset(dout, 0, ref.size() * 4)); run<32768 + 131072 + 8192>("", blocks, lds, dx, dwin, dtw, mbt, dout, rows, L, T, 1, 1, 0); cmp("gather + queue + window regs");
    g_rotate = 1;
    fflush(stdout);
    if (getenv("LAB_R2_ONLY")) return 0;
  }
  {  // interleaved A/B (medians of 7 x 30 launches each): stagger on/off x lane order on/off, narrow stores
    MelBandsDev mbo = mb;
    std::vector<int> ord;
    if (FILE* f = fopen("tools/ubench/order80.txt", "r")) { int v; while (fscanf(f, "%d", &v) == 1) ord.push_back(v); fclose(f); }
    int* dord = nullptr;
    if ((int)ord.size() == 80) { CK(hipMalloc(&dord, 320)); CK(hipMemcpy(dord, ord.data(), 320, hipMemcpyHostToDevice)); mbo.order = dord; }
    {   // A/B: constants in registers (LAB 8192 window, 16384 twiddles) vs LDS tables
      std::vector<float> r[4];
      for (int rep = 0; rep < 9; ++rep) {
        r[0].push_back(run<0>("", blocks, lds, dx, dwin, dtw, mbo, dout, rows, L, T, 30, 1, 0));
        r[1].push_back(run<8192>("", blocks, lds, dx, dwin, dtw, mbo, dout, rows, L, T, 30, 1, 0));
        r[2].push_back(run<16384>("", blocks, lds, dx, dwin, dtw, mbo, dout, rows, L, T, 30, 1, 0));
        r[3].push_back(run<8192 + 16384>("", blocks, lds, dx, dwin, dtw, mbo, dout, rows, L, T, 30, 1, 0));
      }
      const char* nr[4] = {"LDS tables (product)", "window in registers", "twiddles in registers", "both in registers"};
      for (int i = 0; i < 4; ++i) { std::sort(r[i].begin(), r[i].end()); printf("A/B median %-28s %7.1f us (min %.1f max %.1f)\n", nr[i], r[i][4], r[i].front(), r[i].back()); }
      fflush(stdout);
      if (getenv("LAB_AB_ONLY")) return 0;
    }
    std::vector<float> t[4], tw2[2];
    for (int rep = 0; rep < 7; ++rep) {
      tw2[0].push_back(run<0>("", blocks, lds, dx, dwin, dtw, mbo, dout, rows, L, T, 30, 1, 1));   // wide stores
      tw2[1].push_back(run<2>("", blocks, lds, dx, dwin, dtw, mbo, dout, rows, L, T, 30, 1, 0));   // no stores at all
      t[0].push_back(run<0>("", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, 30, 1, 0));
      t[1].push_back(run<128>("", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, 30, 1, 0));
      if (dord) {
        t[2].push_back(run<0>("", blocks, lds, dx, dwin, dtw, mbo, dout, rows, L, T, 30, 1, 0));
        t[3].push_back(run<128>("", blocks, lds, dx, dwin, dtw, mbo, dout, rows, L, T, 30, 1, 0));
      }
    }
    const char* nm[4] = {"no stagger, identity order", "stagger, identity order", "no stagger, lane order", "stagger, lane order"};
    const char* nw[2] = {"lane order, WIDE (LDS-staged) stores", "lane order, NO global stores"};
    for (int i = 0; i < 2; ++i) { std::sort(tw2[i].begin(), tw2[i].end()); printf("A/B median %-36s %7.1f us (min %.1f max %.1f)\n", nw[i], tw2[i][3], tw2[i].front(), tw2[i].back()); }
    for (int i = 0; i < 4; ++i) if (!t[i].empty()) { std::sort(t[i].begin(), t[i].end()); printf("A/B median %-28s %7.1f us (min %.1f max %.1f)\n", nm[i], t[i][t[i].size() / 2], t[i].front(), t[i].back()); }
  }
  run<1>("LAB1 no stage wait", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<2>("LAB2 no global stores", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<3>("LAB3 no wait, no stores", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<3+32>("LAB35 no wait/stores, DMA same tile", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<3+64>("LAB67 no wait/stores, DMA 1 of 5", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<2048>("LAB2048 product, 12-phase stagger (SIMD-major)", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<4096>("LAB4096 product, thirds per SIMD + twelfths", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<4096>("LAB4096 + narrow out", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 0);
  run<128>("LAB128 stagger + narrow out", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 0);
  run<128>("LAB128 product with the wave stagger", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<8>("LAB8 no DMA issue", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<11>("LAB11 no DMA, no wait, no stores", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<15>("LAB15 ... and no phase C", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<31>("LAB31 ... and no phase B", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long long* dcyc; CK(hipMalloc(&dcyc, 8));
    const int reps = 2800;   // 100x the DFT-20s per wave of the real kernel (14 tiles x 2 passes)
    hipLaunchKernelGGL(dft_probe, dim3(512), dim3(384), 0, 0, dout, reps, 0.05f, dcyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(dft_probe, dim3(512), dim3(384), 0, 0, dout, reps, 0.05f, dcyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long hc; CK(hipMemcpy(&hc, dcyc, 8, hipMemcpyDeviceToHost));
    printf("pure-VALU probe: %d x (dft20 + 40 mul) per wave, 12 waves/CU: %.1f us per launch; wave 0 ran %lld s_memtime ticks -> %.2f GHz; %.0f ticks per dft20-group per SIMD\n",
           reps, ms * 100.f, hc, hc / (ms * 1e-4) * 1e-9, (double)hc / reps / 3.0 * 3.0 / 3.0);
  }
  {
    CK(hipMemset(dtw + 1024, 0, 24 * 4096));
    run<1024>("census run", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, 1, 1, 1);
    const int nw = blocks * kWavesPerBlock;
    std::vector<long long> rec(3 * nw);
    CK(hipMemcpy(rec.data(), dtw + 1024, rec.size() * 8, hipMemcpyDeviceToHost));
    long long tmin = rec[0], tmax = rec[2];
    int used = 0;
    for (int w = 0; w < nw; ++w) if (rec[3 * w + 2] >= rec[0]) { tmin = std::min(tmin, rec[3 * w]); tmax = std::max(tmax, rec[3 * w + 2]); }
    double e = 0, r = 0, d = 0, emax = 0, rmax = 0, dmin = 1e30;
    for (int w = 0; w < nw; ++w) {
      if (rec[3 * w + 2] < rec[0]) continue;
      ++used;
      const double a = (rec[3 * w] - tmin) * 0.01, bb = (rec[3 * w + 1] - tmin) * 0.01, c = (rec[3 * w + 2] - tmin) * 0.01;
      e += a; r += bb; d += c; emax = std::max(emax, a); rmax = std::max(rmax, bb); dmin = std::min(dmin, c);
    }
    printf("census (us from first wave entry): entry mean %.1f max %.1f | tables ready mean %.1f max %.1f | done mean %.1f min %.1f max %.1f\n",
           e / used, emax, r / used, rmax, d / used, dmin, (tmax - tmin) * 0.01);
  }
  run<0>("product again: staged in + wide out", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  run<0>("product again: staged in + narrow out", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 0);
  run<31+256>("LAB287 phase A, no S gather", blocks, lds, dx, dwin, dtw, mb, dout, rows, L, T, it, 1, 1);
  return 0;
}

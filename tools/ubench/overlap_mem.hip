// Does global-memory traffic of a CU overlap with the VALU work of its other waves?  (design probe)
// Each wave loops { 800 v_fma_f32 ; NL x global_load_dwordx4 (1 KiB per wave-instr, streaming) ;
// NS x global_store_dwordx4 }.  Loads are consumed one iteration later (prefetch distance 1).
//   hipcc --offload-arch=gfx950 -O3 overlap_mem.hip -o overlap_mem
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NV, int NL, int NS, int MODE>   // MODE 0: loads to VGPR, 1: LDS-DMA
__global__ void __launch_bounds__(256) k(const float4* __restrict__ in, float4* __restrict__ out, int reps, float a, float b,
                                         size_t wave_stride) {
  __shared__ __attribute__((aligned(16))) float sm[4 * 64 * 32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const float4* src = in + gw * wave_stride + lane;
  float4* dst = out + gw * wave_stride + lane;
  float x0 = lane, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float4 buf[NL > 0 ? NL : 1];
  for (int j = 0; j < NL; ++j) buf[j] = float4{0, 0, 0, 0};
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(sm + wave * 64 * 32));
  for (int it = 0; it < reps; ++it) {
    // consume the previous iteration's loads
    if (MODE == 0) { for (int j = 0; j < NL; ++j) x0 += buf[j].x + buf[j].w; }
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); x0 += sm[wave * 64 * 32 + lane]; }
    // issue this iteration's memory operations
    for (int j = 0; j < NL; ++j) {
      if (MODE == 0) buf[j] = src[(size_t)(it * NL + j) * 64];
      else {
        unsigned keep;
        const float4* g = src + (size_t)(it * NL + j) * 64;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(lds_addr + 1024 * (j & 7)) : "memory");
      }
    }
    for (int j = 0; j < NS; ++j) dst[(size_t)(it * NS + j) * 64] = float4{x0, x1, x2, x3};
#pragma unroll
    for (int i = 0; i < NV / 8; ++i) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                   "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
  }
  if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 12345.678f) out[0] = float4{x0, x1, x2, x3};
}

template <int NV, int NL, int NS, int MODE>
static void run(const char* name, int blocks, const float4* in, float4* out, int reps, size_t wave_stride) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, NL, NS, MODE><<<blocks, 256>>>(in, out, reps, 1.0001f, 0.5f, wave_stride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<NV, NL, NS, MODE><<<blocks, 256>>>(in, out, reps, 1.0001f, 0.5f, wave_stride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)blocks * 4 * reps * (NL + NS) * 1024.0;
  printf("%-40s %8.1f us   %7.1f GB/s   %.2f cyc/fma/SIMD\n", name, ms * 1e3, bytes / ms / 1e6,
         NV ? ms * 1e-3 * 2.4e9 / (reps * (double)NV * blocks * 4 / 1024.0) : 0.0);
}

int main() {
  const int reps = 14, bpc = 3, blocks = 256 * bpc;
  const size_t wave_stride = (size_t)reps * 8 * 64;           // float4 elements per wave
  const size_t n = wave_stride * blocks * 4;
  float4 *in, *out;
  hipMalloc(&in, n * sizeof(float4)); hipMalloc(&out, n * sizeof(float4));
  hipMemset(in, 0, n * sizeof(float4));
  printf("12 waves/CU, %d iterations of 800 fma per wave, buffers %.0f MB each\n", reps, n * 16 / 1e6);
  run<800, 0, 0, 0>("VALU only", blocks, in, out, reps, wave_stride);
  run<0, 5, 2, 0>("memory only: 5 ld + 2 st", blocks, in, out, reps, wave_stride);
  run<800, 5, 0, 0>("VALU + 5 loads (VGPR)", blocks, in, out, reps, wave_stride);
  run<800, 5, 0, 1>("VALU + 5 loads (LDS-DMA)", blocks, in, out, reps, wave_stride);
  run<800, 0, 2, 0>("VALU + 2 stores", blocks, in, out, reps, wave_stride);
  run<800, 5, 2, 0>("VALU + 5 loads (VGPR) + 2 stores", blocks, in, out, reps, wave_stride);
  run<800, 5, 2, 1>("VALU + 5 loads (LDS-DMA) + 2 stores", blocks, in, out, reps, wave_stride);
  return 0;
}

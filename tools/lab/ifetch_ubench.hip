// Does straight-line code issue as fast as a small loop?  (tools only)  Each wave runs N VALU instructions in 8 independent
// chains, either as a 16-instruction loop body or as a 1024-instruction straight-line body, in the 4-byte (VOP2 v_fmac_f32)
// or the 8-byte (VOP3 v_fma_f32) encoding.  Prints shader cycles per instruction per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R64(X) R4(R16(X))
#define OPS2(a, b) \
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[0]) : "v"(a), "v"(b)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[1]) : "v"(a), "v"(b)); \
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[2]) : "v"(a), "v"(b)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[3]) : "v"(a), "v"(b)); \
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[4]) : "v"(a), "v"(b)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[5]) : "v"(a), "v"(b)); \
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[6]) : "v"(a), "v"(b)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[7]) : "v"(a), "v"(b));
#define OPS3(a, b) \
  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[1]) : "v"(a), "v"(b)); \
  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[2]) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(a), "v"(b)); \
  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[4]) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[5]) : "v"(a), "v"(b)); \
  asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[6]) : "v"(a), "v"(b)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[7]) : "v"(a), "v"(b));
// KIND 0: loop of 16 VOP2, 1: loop of 16 VOP3, 2: straight line 1024 VOP2, 3: straight line 1024 VOP3, 4: straight line 4096 VOP3
template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, float a, float b, int iters) {
  const int lane = threadIdx.x & 63;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { OPS2(a, b) OPS2(a, b) }
    if (KIND == 1) { OPS3(a, b) OPS3(a, b) }
    if (KIND == 2) { R64(OPS2(a, b) OPS2(a, b)) }
    if (KIND == 3) { R64(OPS3(a, b) OPS3(a, b)) }
    if (KIND == 4) { R64(OPS3(a, b) OPS3(a, b)) R64(OPS3(a, b) OPS3(a, b)) R64(OPS3(a, b) OPS3(a, b)) R64(OPS3(a, b) OPS3(a, b)) }
  }
  long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, float* out, long long* cyc, int waves, int per_iter) {
  const int blocks = 256, total = 65536, iters = total / per_iter;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, 1.0001f, 0.5f, iters);
  hipDeviceSynchronize();
  long long* h = (long long*)malloc(blocks * 16 * 8);
  hipMemcpy(h, cyc, blocks * 16 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) s += h[b * 16 + w];
  const double per_wave = s / (blocks * waves) / (double)total;
  printf("%-34s waves/SIMD %d : %.2f cycles per instruction per wave, %.2f per SIMD\n", name, waves / 4, per_wave, per_wave / (waves / 4));
  free(h);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  const int ws[4] = {4, 8, 12, 16};
  for (int wi = 0; wi < 4; ++wi) {
    run<0>("loop of 16, 4-byte v_fmac", out, cyc, ws[wi], 16);
    run<1>("loop of 16, 8-byte v_fma", out, cyc, ws[wi], 16);
    run<2>("straight line 1024, 4-byte", out, cyc, ws[wi], 1024);
    run<3>("straight line 1024, 8-byte", out, cyc, ws[wi], 1024);
    run<4>("straight line 4096, 8-byte", out, cyc, ws[wi], 4096);
  }
  return 0;
}

// VALU issue rate on gfx950 as a function of instruction-level parallelism (tools only): each wave issues N back-to-back
// v_fma_f32 / v_fmac_f32 arranged in C independent dependency chains (C = 1: every instruction reads the result of the
// one before it), W waves per SIMD.  Prints shader cycles per instruction per SIMD (= elapsed / (N x W)) and per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define N_ITER 512
template <int CHAINS, int KIND>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, float a, float b) {
  const int lane = threadIdx.x & 63;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = lane * 0.001f + i;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float& x = v[u % CHAINS];
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));          // dependence through src0
      if (KIND == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));             // dependence through the accumulator
      if (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
      if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(a), "v"(b));          // one scalar operand
      if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(a));
    }
  }
  long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int CHAINS, int KIND> void run(const char* name, float* out, long long* cyc, int waves) {
  const int blocks = 256;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  long long* h = (long long*)malloc(blocks * 16 * 8);
  hipMemcpy(h, cyc, blocks * 16 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) s += h[b * 16 + w];
  const double per_wave = s / (blocks * waves) / (N_ITER * 16.0);
  printf("%-28s chains %d  waves/SIMD %d : %.2f cycles per instruction per wave, %.2f per SIMD\n", name, CHAINS, waves / 4,
         per_wave, per_wave / (waves / 4));
  free(h);
}
template <int KIND> void all(const char* name, float* out, long long* cyc) {
  const int ws[4] = {4, 8, 12, 16};
  for (int wi = 0; wi < 4; ++wi) {
    run<1, KIND>(name, out, cyc, ws[wi]); run<2, KIND>(name, out, cyc, ws[wi]); run<3, KIND>(name, out, cyc, ws[wi]);
    run<4, KIND>(name, out, cyc, ws[wi]); run<8, KIND>(name, out, cyc, ws[wi]);
  }
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  all<0>("v_fma_f32 (dep. src0)", out, cyc);
  all<1>("v_fmac_f32 (dep. acc)", out, cyc);
  all<2>("v_mul_f32", out, cyc);
  all<3>("v_fma_f32 (sgpr operand)", out, cyc);
  all<4>("v_add_f32", out, cyc);
  return 0;
}

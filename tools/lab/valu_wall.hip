// VALU issue rate by the WALL clock (tools only): does a wave64 fp32 instruction take 2 or 4 cycles of a SIMD?  256 workgroups of
// W waves per SIMD, C independent chains; prints ns per instruction per SIMD from hipEvents, the clock64() ticks per ns (the rate of
// the counter the round-3 table was measured in), and the same for v_pk_fma_f32 (two FMAs per lane and instruction).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_ITER 8192
template <int CHAINS, int KIND>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc, float a, float b) {
  const int lane = threadIdx.x & 63;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = lane * 0.001f + i;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 av = {a, a}, bv = {b, b};
  long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) { float& x = v[u % CHAINS]; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b)); }
      if (KIND == 1) { f2& x = reinterpret_cast<f2*>(v)[u % CHAINS]; asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(av), "v"(bv)); }
    }
  }
  long long t1 = clock64();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int CHAINS, int KIND> void run(const char* name, float* out, long long* cyc, int waves) {
  const int blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double instr_per_simd = (double)N_ITER * 16 * (waves / 4);
  printf("%-14s chains %d waves/SIMD %d : %.3f ns per instruction per SIMD (wall), clock64 %.3f ticks/ns, %.2f ticks per instr per SIMD\n",
         name, CHAINS, waves / 4, ms * 1e6 / instr_per_simd, h / (ms * 1e6), h / instr_per_simd);
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 64);
  for (int w : {4, 8, 12, 16}) { run<1, 0>("v_fma_f32", out, cyc, w); run<4, 0>("v_fma_f32", out, cyc, w); run<8, 0>("v_fma_f32", out, cyc, w); }
  for (int w : {4, 8, 16}) { run<1, 1>("v_pk_fma_f32", out, cyc, w); run<4, 1>("v_pk_fma_f32", out, cyc, w); run<8, 1>("v_pk_fma_f32", out, cyc, w); }
  return 0;
}

// LDS instruction throughput per CU on gfx950 (tools only): 12 waves per CU, each issues N back-to-back instructions of
// one kind at lane-contiguous (conflict-free) addresses; cycles per wave-instruction per CU = elapsed cycles / (12 N).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
#define N_ITER 256
template <int KIND>
__global__ void __launch_bounds__(768) k(float* out, long long* cyc) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = lds + wave * 2048;
  float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
  unsigned a32 = (unsigned)(uintptr_t)(base + lane);
  unsigned a64 = (unsigned)(uintptr_t)(base + 2 * lane);
  unsigned a128 = (unsigned)(uintptr_t)(base + 4 * lane);
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(a32), "v"(v0), "i"(u * 256) : "memory");
      if (KIND == 1) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(a64), "v"(*(double*)&v0), "i"(u * 512) : "memory");
      if (KIND == 2) asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(a32), "v"(v0), "v"(v1), "i"(u * 2), "i"(u * 2 + 64) : "memory");
      if (KIND == 3) { f4 q = {v0, v1, v2, v3}; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a128), "v"(q), "i"(u * 1024) : "memory"); }
      if (KIND == 4) { float r; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(a32), "i"(u * 256) : "memory"); v0 += r; }
      if (KIND == 5) { double r; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(a64), "i"(u * 512) : "memory"); v0 += (float)r; }
      if (KIND == 6) { f4 r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a128), "i"(u * 1024) : "memory"); v0 += r.x; }
      if (KIND == 7) asm volatile("ds_write_addtid_b32 %0 offset:%1" :: "v"(v0), "i"(u * 256) : "memory");
      if (KIND == 8) { f2v q = {v0, v1}; asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(a64), "v"(q), "i"(u * 512) : "memory"); }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 12 + wave] = t1 - t0;
  out[blockIdx.x * 768 + threadIdx.x] = v0;
}
template <int KIND> void run(const char* name, float* out, long long* cyc, int blocks) {
  hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 2048 * 4);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(768), 12 * 2048 * 4, 0, out, cyc);
  hipDeviceSynchronize();
  long long* h = (long long*)malloc(blocks * 12 * 8);
  hipMemcpy(h, cyc, blocks * 12 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks * 12; ++i) s += h[i];
  printf("%-22s %.2f cycles per wave-instruction per CU (12 waves)\n", name, s / (blocks * 12) / (N_ITER * 8 * 12.0));
  free(h);
}
int main() {
  float* out; long long* cyc; int blocks = 256;
  hipMalloc(&out, blocks * 768 * 4); hipMalloc(&cyc, blocks * 12 * 8);
  run<0>("ds_write_b32", out, cyc, blocks); run<1>("ds_write_b64", out, cyc, blocks); run<8>("ds_write_b64(f2)", out, cyc, blocks);
  run<2>("ds_write2_b32", out, cyc, blocks); run<3>("ds_write_b128", out, cyc, blocks); run<7>("ds_write_addtid_b32", out, cyc, blocks);
  run<4>("ds_read_b32", out, cyc, blocks); run<5>("ds_read_b64", out, cyc, blocks); run<6>("ds_read_b128", out, cyc, blocks);
  return 0;
}

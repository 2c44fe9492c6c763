// Which SIMD does each wave of a workgroup land on?  (tools only)  hipcc --offload-arch=gfx950 -O2 simd_map.hip -o simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, size_t lds_dummy) {
  extern __shared__ float sm[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = hw;
  if (lds_dummy == 12345) sm[threadIdx.x] = 1.0f;
}
int main() {
  for (int waves : {8, 12, 16}) {
    for (size_t lds : {(size_t)0, (size_t)140 * 1024}) {
      const int blocks = 512;
      unsigned* d;
      hipMalloc(&d, blocks * waves * 4);
      hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(64 * waves), lds, 0, d, lds);
      std::vector<unsigned> h(blocks * waves);
      hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
      printf("waves %d lds %zu:\n", waves, lds);
      for (int b : {0, 1, 300}) {
        printf("  block %3d simd of wave 0..: ", b);
        for (int w = 0; w < waves; ++w) printf("%u ", (h[b * waves + w] >> 4) & 3);
        printf(" | wave slot: ");
        for (int w = 0; w < waves; ++w) printf("%u ", h[b * waves + w] & 15);
        printf(" | cu %u se %u\n", (h[b * waves] >> 8) & 15, (h[b * waves] >> 13) & 7);
      }
      hipFree(d);
    }
  }
  return 0;
}

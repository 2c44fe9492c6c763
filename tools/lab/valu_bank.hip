// VGPR bank conflicts of fp32 VALU instructions on gfx950 (tools only): v_fma_f32 with its three sources in one bank (register number
// mod 4), in two, in three banks; 8 independent chains, W waves per SIMD; wall-clock ns and clock64 ticks per instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N_ITER 4096
#define REP8(x0, x1, x2, x3, x4, x5, x6, x7) x0 "\n" x1 "\n" x2 "\n" x3 "\n" x4 "\n" x5 "\n" x6 "\n" x7 "\n"
template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, long long* cyc) {
  long long t0 = clock64();
  for (int it = 0; it < N_ITER; ++it) {
    // chains in v40..v47 (fma: d = d * a + b or fmac d += a * b); operands from v48..v63
    if (KIND == 0)   // all three sources in the bank of the destination: d = v40 (bank 0), a = v48 (0), b = v52 (0)
      asm volatile(REP8("v_fma_f32 v40, v40, v48, v52", "v_fma_f32 v41, v41, v49, v53", "v_fma_f32 v42, v42, v50, v54", "v_fma_f32 v43, v43, v51, v55",
                        "v_fma_f32 v44, v44, v48, v52", "v_fma_f32 v45, v45, v49, v53", "v_fma_f32 v46, v46, v50, v54", "v_fma_f32 v47, v47, v51, v55")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    if (KIND == 1)   // three different banks: d bank i, a bank i + 1, b bank i + 2
      asm volatile(REP8("v_fma_f32 v40, v40, v49, v54", "v_fma_f32 v41, v41, v50, v55", "v_fma_f32 v42, v42, v51, v52", "v_fma_f32 v43, v43, v48, v53",
                        "v_fma_f32 v44, v44, v49, v54", "v_fma_f32 v45, v45, v50, v55", "v_fma_f32 v46, v46, v51, v52", "v_fma_f32 v47, v47, v48, v53")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    if (KIND == 2)   // a and b the same register
      asm volatile(REP8("v_fma_f32 v40, v40, v49, v49", "v_fma_f32 v41, v41, v50, v50", "v_fma_f32 v42, v42, v51, v51", "v_fma_f32 v43, v43, v48, v48",
                        "v_fma_f32 v44, v44, v49, v49", "v_fma_f32 v45, v45, v50, v50", "v_fma_f32 v46, v46, v51, v51", "v_fma_f32 v47, v47, v48, v48")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    if (KIND == 3)   // two-source instruction, both other banks: v_mul d = a * b with d also a source of nothing (no chain at all)
      asm volatile(REP8("v_mul_f32 v40, v49, v54", "v_mul_f32 v41, v50, v55", "v_mul_f32 v42, v51, v52", "v_mul_f32 v43, v48, v53",
                        "v_mul_f32 v44, v49, v54", "v_mul_f32 v45, v50, v55", "v_mul_f32 v46, v51, v52", "v_mul_f32 v47, v48, v53")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    if (KIND == 4)   // v_fmac (VOP2, 4-byte): d += a * b, three banks
      asm volatile(REP8("v_fmac_f32 v40, v49, v54", "v_fmac_f32 v41, v50, v55", "v_fmac_f32 v42, v51, v52", "v_fmac_f32 v43, v48, v53",
                        "v_fmac_f32 v44, v49, v54", "v_fmac_f32 v45, v50, v55", "v_fmac_f32 v46, v51, v52", "v_fmac_f32 v47, v48, v53")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    if (KIND == 5)   // v_fmac, a and b in the destination's bank
      asm volatile(REP8("v_fmac_f32 v40, v48, v52", "v_fmac_f32 v41, v49, v53", "v_fmac_f32 v42, v50, v54", "v_fmac_f32 v43, v51, v55",
                        "v_fmac_f32 v44, v48, v52", "v_fmac_f32 v45, v49, v53", "v_fmac_f32 v46, v50, v54", "v_fmac_f32 v47, v51, v55")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    if (KIND == 6)   // v_add with one source (two-operand, d = d + a)
      asm volatile(REP8("v_add_f32 v40, v40, v49", "v_add_f32 v41, v41, v50", "v_add_f32 v42, v42, v51", "v_add_f32 v43, v43, v48",
                        "v_add_f32 v44, v44, v49", "v_add_f32 v45, v45, v50", "v_add_f32 v46, v46, v51", "v_add_f32 v47, v47, v48")
                   ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
  }
  long long t1 = clock64();
  float s;
  asm volatile("v_add_f32 %0, v40, v41\n v_add_f32 %0, %0, v42\n v_add_f32 %0, %0, v43\n v_add_f32 %0, %0, v44" : "=v"(s));
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, float* out, long long* cyc, int waves) {
  const int blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(64 * waves), 0, 0, out, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(64 * waves), 0, 0, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)N_ITER * 8 * (waves / 4);
  printf("%-44s waves/SIMD %d : %.3f ns, %.2f ticks per instruction per SIMD (clock %.2f GHz)\n", name, waves / 4, ms * 1e6 / n, h / n, h / (ms * 1e6));
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 64);
  for (int w : {8, 12, 16}) {
    run<0>("v_fma d,d,a,b  one bank", out, cyc, w);
    run<1>("v_fma d,d,a,b  three banks", out, cyc, w);
    run<2>("v_fma d,d,a,a  two banks", out, cyc, w);
    run<3>("v_mul d,a,b    no chain, three banks", out, cyc, w);
    run<4>("v_fmac d,a,b   three banks", out, cyc, w);
    run<5>("v_fmac d,a,b   one bank", out, cyc, w);
    run<6>("v_add d,d,a    two banks", out, cyc, w);
  }
  return 0;
}

// Do VALU and LDS work of DIFFERENT waves of one CU overlap on gfx950?  (design probe, not product)
// Each wave loops { NV dependent-free v_fma_f32 ; NR ds_read_b128 + NW ds_write_b64 }; odd waves start
// with the LDS half so that phases are staggered.  Compare VALU-only, LDS-only and both.
//   hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NV, int NR, int NW>
__global__ void __launch_bounds__(256) k(float* out, int reps, float a, float b) {
  __shared__ __attribute__((aligned(16))) float sm[4 * 64 * 48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = sm + wave * 64 * 48;
  for (int i = lane; i < 64 * 48; i += 64) base[i] = i;
  __syncthreads();
  float x0 = lane, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const unsigned raddr = (unsigned)(uintptr_t)(base) + 16 * lane;        // conflict-free b128 rows
  const unsigned waddr = (unsigned)(uintptr_t)(base) + 8 * lane;         // conflict-free b64
  float4 r0, r1, r2, r3;
  float2 wv = {1.f, 2.f};
  for (int it = 0; it < reps; ++it) {
    const bool lds_first = ((wave + blockIdx.x) & 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((half == 0) != lds_first) {
        if (NV > 0) {
#pragma unroll
          for (int i = 0; i < NV / 8; ++i) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                         "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
          }
        }
      } else {
        if (NW > 0) {
#pragma unroll
          for (int i = 0; i < NW; ++i)
            asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(waddr), "v"(wv), "i"(512 * (i % 8) + 16384) : "memory");
        }
        if (NR > 0) {
#pragma unroll
          for (int i = 0; i < NR / 4; ++i) {
            asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                         "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(raddr) : "memory");
            x0 += r0.x + r1.y + r2.z + r3.w;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int NV, int NR, int NW>
static float run(const char* name, int blocks, float* out, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV, NR, NW><<<blocks, 256>>>(out, reps, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<NV, NR, NW><<<blocks, 256>>>(out, reps, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double waves_per_simd = blocks * 4 / 1024.0, waves_per_cu = blocks * 4 / 256.0;
  const double cyc = ms * 1e-3 * 2.4e9;
  printf("%-28s %8.1f us", name, ms * 1e3);
  if (NV) printf("  VALU %5.2f cyc/instr/SIMD", cyc / (reps * (double)NV * waves_per_simd));
  if (NR) printf("  LDS  %5.2f cyc/(op)/CU over %d rd128 + %d wr64", cyc / (reps * (double)(NR + NW) * waves_per_cu), NR, NW);
  printf("\n");
  return ms;
}

int main(int argc, char** argv) {
  float* out; hipMalloc(&out, 256 * 16 * 256 * sizeof(float));
  const int reps = 200;
  for (int bpc : {1, 2, 3, 4}) {
    const int blocks = 256 * bpc;
    printf("--- %d waves per CU\n", 4 * bpc);
    run<400, 0, 0>("VALU only (400 fma)", blocks, out, reps);
    run<0, 40, 20>("LDS only (40 rd128+20 wr64)", blocks, out, reps);
    run<400, 40, 20>("both", blocks, out, reps);
    run<800, 40, 20>("both, 800 fma", blocks, out, reps);
  }
  return 0;
}

// Instruction-throughput probes for gfx950 (design aid, not product): packed vs scalar f32 VALU
// rate and LDS instruction costs.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;

__global__ void k_fma(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITERS; ++i) {
    x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
    x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_pkfma(float* out, float a, float b) {
  float2_ A = {a, a}, B = {b, b};
  float2_ x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  for (int i = 0; i < ITERS; ++i) {
    x0 = __builtin_elementwise_fma(x0, A, B); x1 = __builtin_elementwise_fma(x1, A, B);
    x2 = __builtin_elementwise_fma(x2, A, B); x3 = __builtin_elementwise_fma(x3, A, B);
    x4 = __builtin_elementwise_fma(x4, A, B); x5 = __builtin_elementwise_fma(x5, A, B);
    x6 = __builtin_elementwise_fma(x6, A, B); x7 = __builtin_elementwise_fma(x7, A, B);
  }
  float2_ s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_pkadd(float* out, float a, float b) {
  float2_ A = {a, b};
  float2_ x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  for (int i = 0; i < ITERS; ++i) {
    x0 += A; x1 += A; x2 += A; x3 += A; x4 += A; x5 += A; x6 += A; x7 += A;
  }
  float2_ s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_add(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITERS; ++i) {
    x0 += a; x1 += b; x2 += a; x3 += b; x4 += a; x5 += b; x6 += a; x7 += b;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
__global__ void k_lds(float* out) {
  __shared__ __attribute__((aligned(16))) float sm[64 * 44 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = sm + wave * 64 * 44;
  float acc = 0;
  for (int i = 0; i < 64 * 44; i += 64) base[i + lane] = lane;
  __syncthreads();
  for (int it = 0; it < ITERS / 16; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (MODE == 0) {            // ds_read_b32 consecutive lanes
        acc += base[lane + 64 * j + (it & 3)];
      } else if (MODE == 1) {     // ds_read_b64 stride 42 dwords per lane (transposition row read)
        float2_ v = *reinterpret_cast<float2_*>(base + 42 * lane + 2 * j);
        acc += v.x + v.y;
      } else if (MODE == 2) {     // ds_write_b64, lane-contiguous complex (column write), row stride 42
        float2_ v = {acc, (float)j};
        *reinterpret_cast<float2_*>(base + 42 * ((j + it) & 31) + 2 * lane) = v;
      } else if (MODE == 3) {     // ds_write_b32 consecutive
        base[lane + 64 * j] = acc + j;
      } else if (MODE == 4) {     // ds_bpermute
        acc += __shfl(acc + j, (lane * 7 + j) & 63, 64);
      } else if (MODE == 5) {     // ds_read_b128 lane-contiguous 16B
        typedef float float4_ __attribute__((ext_vector_type(4)));
        float4_ v = *reinterpret_cast<float4_*>(base + 4 * lane + 256 * (j & 7));
        acc += v.x + v.y + v.z + v.w;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + base[lane];
}

template <typename F>
float run(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  float* out; hipMalloc(&out, 256 * 256 * 16 * sizeof(float));
  const int blocks = 256 * 8, threads = 256;   // 8 WGs per CU, 32 waves/CU
  const double waves_per_simd = blocks * (threads / 64) / (256.0 * 4);
  auto report = [&](const char* name, float ms, double instr_per_wave) {
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    double cyc = ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
    printf("%-34s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz)\n", name, ms, cyc);
  };
  for (int wps : {1, 2, 3, 4, 8}) {   // VALU issue rate vs waves per SIMD (256-thread WGs: 1 wave per SIMD each)
    const int nb = 256 * wps;
    auto rep = [&](const char* name, float ms) {
      printf("  %d waves/SIMD %-14s %6.2f cycles/wave-instr/SIMD (at 2.4 GHz nominal)\n", wps, name,
             ms * 1e-3 * 2.4e9 / (8.0 * ITERS * wps));
    };
    rep("v_fma_f32", run([&] { k_fma<<<nb, threads>>>(out, 1.0001f, 0.5f); }));
    rep("v_add_f32", run([&] { k_add<<<nb, threads>>>(out, 1.0001f, 0.5f); }));
    rep("v_pk_fma_f32", run([&] { k_pkfma<<<nb, threads>>>(out, 1.0001f, 0.5f); }));
  }
  report("v_fma_f32", run([&] { k_fma<<<blocks, threads>>>(out, 1.0001f, 0.5f); }), 8.0 * ITERS);
  report("v_pk_fma_f32", run([&] { k_pkfma<<<blocks, threads>>>(out, 1.0001f, 0.5f); }), 8.0 * ITERS);
  report("v_add_f32", run([&] { k_add<<<blocks, threads>>>(out, 1.0001f, 0.5f); }), 8.0 * ITERS);
  report("v_pk_add_f32", run([&] { k_pkadd<<<blocks, threads>>>(out, 1.0001f, 0.5f); }), 8.0 * ITERS);
  const double wpc = blocks * (threads / 64) / 256.0;  // waves per CU
  auto rep_lds = [&](const char* name, float ms) {
    double cyc = ms * 1e-3 * 2.4e9 / ((double)ITERS * wpc);
    printf("%-34s %8.3f ms  %6.2f cycles/wave-instr/CU\n", name, ms, cyc);
  };
  rep_lds("ds_read_b32 (lane-contiguous)", run([&] { k_lds<0><<<blocks, threads>>>(out); }));
  rep_lds("ds_read_b64 (stride 42 dw)", run([&] { k_lds<1><<<blocks, threads>>>(out); }));
  rep_lds("ds_write_b64 (lane-contig)", run([&] { k_lds<2><<<blocks, threads>>>(out); }));
  rep_lds("ds_write_b32 (lane-contig)", run([&] { k_lds<3><<<blocks, threads>>>(out); }));
  rep_lds("ds_bpermute_b32", run([&] { k_lds<4><<<blocks, threads>>>(out); }));
  rep_lds("ds_read_b128 (lane-contig)", run([&] { k_lds<5><<<blocks, threads>>>(out); }));
  return 0;
}

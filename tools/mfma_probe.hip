// Developer probe (not part of the product): issue rate and dependent latency of v_mfma_f64_16x16x4_f64 on gfx950.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f64x4 = __attribute__((ext_vector_type(4))) double;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, long long* cyc) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
  // operands with busy mantissas (a power-friendly constant operand would flatter the clock)
  unsigned long long h = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 977 * blockIdx.x);
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  double a = 0.5 + (double)(h & 0xFFFFFFFFFFFFFull) * 0x1p-53, b = 1.5 - (double)((h >> 7) & 0xFFFFFFFFFFFFFull) * 0x1p-53;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(const char* name, int grid, int iters) {
  double* out;
  long long* cyc;
  CK(hipMalloc(&out, (size_t)grid * 256 * 8));
  CK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, 16, cyc);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_mfma<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
  hipEventRecord(e1, 0);
  CK(hipDeviceSynchronize());
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long hc = 0;
  CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
  const double nm = (double)iters * NACC;  // MFMAs per wave
  const double flops = nm * 2048.0 * 4 * grid;
  printf("%-34s grid %5d: %.3f ms, %.1f TFLOP/s, %.1f counter ticks per MFMA per wave (s_memtime)\n", name, grid, ms,
         flops / ms / 1e9, hc / nm);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<1>("1 accumulator (dependent chain)", 256, 20000);
  run<2>("2 accumulators", 256, 10000);
  run<4>("4 accumulators", 256, 5000);
  run<8>("8 accumulators", 256, 2500);
  run<8>("8 accumulators, 2 workgroups per CU", 512, 2500);
  run<8>("8 accumulators, 4 workgroups per CU", 1024, 2500);
  run<4>("4 accumulators, 8 workgroups per CU", 2048, 2500);
  run<2>("2 accumulators, 8 workgroups per CU", 2048, 5000);
  run<8>("8 accumulators, one workgroup", 1, 2500);
  return 0;
}

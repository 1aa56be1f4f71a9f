// Developer probe (not part of the product): issue cost of the instruction kinds of the 16 x 16 factor routine, one wave
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/issue_probe tools/issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ double rdlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void k(double* out, long long* t, double seed) {
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = seed + threadIdx.x * 0.001 + c;
  double l = seed * 0.5 + threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    if (MODE == 0) {  // 15 x (2 readlane + fma), independent accumulators
#pragma unroll
      for (int c = 1; c < 16; ++c) a[c] -= l * rdlane(l, c);
    } else if (MODE == 1) {  // 15 fma only (vector operand)
#pragma unroll
      for (int c = 1; c < 16; ++c) a[c] -= l * a[0];
    } else if (MODE == 2) {  // 15 dependent fma
#pragma unroll
      for (int c = 1; c < 16; ++c) a[1] = fma(-l, a[1], a[0]);
    } else if (MODE == 3) {  // readlanes first, then fmas
      double s[16];
#pragma unroll
      for (int c = 1; c < 16; ++c) s[c] = rdlane(l, c);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 1; c < 16; ++c) a[c] -= l * s[c];
    } else if (MODE == 4) {  // rsqrt chain
      l = rsqrt(l + 1.5);
    } else if (MODE == 5) {  // b64 dpp row broadcast + fma
#pragma unroll
      for (int c = 1; c < 16; ++c) {
        double b;
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(l), "n"(1));
        a[c] -= l * b;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    l += a[15] * 1e-300;
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = l;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += a[c];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) t[MODE] = t1 - t0;
}
int main() {
  double* out; long long* t;
  hipMalloc(&out, 64 * 8); hipMalloc(&t, 8 * 8); hipMemset(t, 0, 64);
  for (int r = 0; r < 2; ++r) {
    k<0><<<1, 64>>>(out, t, 1.25); k<1><<<1, 64>>>(out, t, 1.25); k<2><<<1, 64>>>(out, t, 1.25);
    k<3><<<1, 64>>>(out, t, 1.25); k<4><<<1, 64>>>(out, t, 1.25); k<5><<<1, 64>>>(out, t, 1.25);
    hipDeviceSynchronize();
  }
  long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
  const char* nm[] = {"15 x (2 readlane + fma)", "15 fma (independent)", "15 fma (dependent)", "30 readlane then 15 fma", "rsqrt(double) chain", "15 x (mov_b64 dpp bcast + fma)"};
  for (int i = 0; i < 6; ++i) printf("%-34s %8.1f cycles per iteration\n", nm[i], h[i] / 64.0);
  return 0;
}

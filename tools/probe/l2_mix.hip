// Developer probe (not product code): do lines written with write-through (agent-scope) stores in one kernel, then with plain stores by
// ANOTHER XCD in the next kernel, read back fresh in a third kernel on the first XCD -- with plain loads and with agent-scope loads?
// build: hipcc -O2 --offload-arch=gfx950 -o l2_mix l2_mix.hip ; run: ./l2_mix [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kLines = 4096;  // 128-byte lines of doubles (16 per line)
__device__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
// (workgroup b of a launch runs on XCD b % 8: the loops below share the entries among the gridDim.x / 8 workgroups of one XCD)
// phase 1: workgroups on XCD `xa` write value a with agent-scope stores (and read it back at agent scope: the line is in that L2)
__global__ void k_wt(double* buf, double a, unsigned xa, int* ran) {
  if (xcc_id() != xa) return;
  if (threadIdx.x == 0) atomicAdd(ran, 1);
  for (int i = (blockIdx.x / 8) * blockDim.x + threadIdx.x; i < kLines * 16; i += (gridDim.x / 8) * blockDim.x) {
    __hip_atomic_store(buf + i, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ void k_touch(const double* buf, double* sink, unsigned xa, int mode) {
  if (xcc_id() != xa) return;
  double s = 0;
  for (int i = (blockIdx.x / 8) * blockDim.x + threadIdx.x; i < kLines * 16; i += (gridDim.x / 8) * blockDim.x)
    s += mode ? __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : buf[i];
  if (s == -1.0) *sink = s;
}
// phase 2: workgroups on XCD `xb` write value b with plain stores
__global__ void k_plain(double* buf, double b, unsigned xb) {
  if (xcc_id() != xb) return;
  for (int i = (blockIdx.x / 8) * blockDim.x + threadIdx.x; i < kLines * 16; i += (gridDim.x / 8) * blockDim.x) buf[i] = b;
}
// phase 3: workgroups on XCD `xa` read: plain (mode 0) or agent scope (mode 1); count entries that are not b
__global__ void k_read(const double* buf, double b, unsigned xa, int mode, int* bad) {
  if (xcc_id() != xa) return;
  int n = 0;
  for (int i = (blockIdx.x / 8) * blockDim.x + threadIdx.x; i < kLines * 16; i += (gridDim.x / 8) * blockDim.x) {
    const double v = mode ? __hip_atomic_load(buf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : buf[i];
    n += v != b;
  }
  if (n) atomicAdd(bad, n);
}
int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200;
  double *buf, *sink;
  int *bad, *ran;
  CHK(hipMalloc(&buf, kLines * 128));
  CHK(hipMalloc(&sink, 8));
  CHK(hipMalloc(&bad, 4 * 8));
  CHK(hipMalloc(&ran, 4));
  CHK(hipMemset(bad, 0, 32));
  CHK(hipMemset(ran, 0, 4));
  hipStream_t s;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // variants: [first writer: 0 = agent-scope stores, 1 = plain stores + agent-scope touch, 2 = plain stores + plain touch] x [reader: plain, agent]
  // variant 3 (three XCDs): XCD xa writes a at agent scope, XCD xc READS it at agent scope (a clean copy in that L2), XCD xb writes b with
  // plain stores in the next kernel, XCD xc reads again in a third kernel
  for (int mode = 0; mode < 2; ++mode)
    for (int r = 0; r < rounds; ++r) {
      const double a = 1.0 + r, b = -1.0 - r;
      const unsigned xa = r % 8, xb = (r + 1 + r / 8 % 7) % 8;
      unsigned xc = (xa + 3) % 8;
      if (xc == xb) xc = (xc + 1) % 8;
      if (xc == xa) xc = (xc + 1) % 8;
      hipLaunchKernelGGL(k_wt, dim3(512), dim3(256), 0, s, buf, a, xa, ran);
      hipLaunchKernelGGL(k_touch, dim3(512), dim3(256), 0, s, buf, sink, xc, 1);
      hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, s, buf, b, xb);
      hipLaunchKernelGGL(k_read, dim3(512), dim3(256), 0, s, buf, b, xc, mode, bad + 6 + mode);
    }
  for (int var = 0; var < 3; ++var)
    for (int mode = 0; mode < 2; ++mode)
      for (int r = 0; r < rounds; ++r) {
        const double a = 1.0 + r, b = -1.0 - r;
        const unsigned xa = r % 8, xb = (r + 1 + r / 8 % 7) % 8;
        if (var == 0) hipLaunchKernelGGL(k_wt, dim3(512), dim3(256), 0, s, buf, a, xa, ran);
        else {
          hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, s, buf, a, xa);
          hipLaunchKernelGGL(k_touch, dim3(512), dim3(256), 0, s, buf, sink, xa, var == 1 ? 1 : 0);
        }
        hipLaunchKernelGGL(k_plain, dim3(512), dim3(256), 0, s, buf, b, xb);
        hipLaunchKernelGGL(k_read, dim3(512), dim3(256), 0, s, buf, b, xa, mode, bad + var * 2 + mode);
      }
  CHK(hipStreamSynchronize(s));
  int hb[8], hr;
  CHK(hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost));
  CHK(hipMemcpy(&hr, ran, 4, hipMemcpyDeviceToHost));
  printf("workgroups that found themselves on the chosen XCD in phase 1 (variant 0): %d over %d rounds\n", hr, 2 * rounds);
  const char* vn[3] = {"agent-scope stores", "plain stores + agent-scope reads", "plain stores + plain reads"};
  for (int var = 0; var < 3; ++var)
    printf("first XCD: %-34s | other XCD: plain stores | first XCD reads plain: %d stale entries, at agent scope: %d stale entries\n", vn[var],
           hb[var * 2], hb[var * 2 + 1]);
  printf("three XCDs: first agent-scope stores, third reads at agent scope | second: plain stores | third reads plain: %d stale entries, at agent scope: %d stale entries\n",
         hb[6], hb[7]);
  return 0;
}

// Developer probe (not part of the product): what does "the last workgroup to finish reduces the partial sums" cost
// on an 8-XCD part, compared with a separate 1-workgroup reduction kernel after the kernel boundary?
//   hipcc -O3 --offload-arch=gfx950 -o tools/lastblock_probe tools/lastblock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int kB = 256;

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

// streaming body: every workgroup reads `per` doubles and writes one partial sum
template <int MODE>  // 0: partial only; 1: + fence + ticket, last block reduces; 2: ticket without fence (wrong, for cost)
__global__ __launch_bounds__(kB) void k_body(const double* __restrict__ x, double* y, int per, double* partials,
                                              unsigned* counter, unsigned target, double* result) {
  __shared__ double red[16];
  __shared__ int last;
  const size_t base = (size_t)blockIdx.x * per;
  double s = 0;
  for (int i = threadIdx.x; i < per; i += kB) {
    const double v = x[base + i];
    y[base + i] = v * 1.0001;
    s += v;
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
  if (MODE == 0) return;
  if (threadIdx.x == 0) {
    if (MODE == 1) __threadfence();
    const unsigned k = atomicAdd(counter, 1u);
    last = (k == target - 1);
  }
  __syncthreads();
  if (!last) return;
  if (MODE == 1) __threadfence();
  double a = 0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kB) a += __builtin_nontemporal_load(partials + i);
  const double r = block_sum(a, red);
  if (threadIdx.x == 0) {
    result[0] = r;
    *counter = 0;
  }
}

// hierarchical tickets, no fences: the partial is published with an agent-scope atomic store (goes to the coherence
// point, no L2 write-back), s_waitcnt orders it before the ticket; the last workgroup reads with agent-scope loads.
template <int NC>
__global__ __launch_bounds__(kB) void k_body_h(const double* __restrict__ x, double* y, int per, double* partials,
                                               unsigned* counters /* (NC + 1) * 32 */, double* result) {
  __shared__ double red[16];
  __shared__ int last;
  const size_t base = (size_t)blockIdx.x * per;
  double s = 0;
  for (int i = threadIdx.x; i < per; i += kB) {
    const double v = x[base + i];
    y[base + i] = v * 1.0001;
    s += v;
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) {
    __hip_atomic_store(partials + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the store is acknowledged
    const int j = blockIdx.x % NC;
    const unsigned pop = gridDim.x / NC + (j < (int)(gridDim.x % NC) ? 1 : 0);
    unsigned* c = counters + (size_t)(j + 1) * 32;
    int l = 0;
    if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pop - 1) {
      __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned groups = gridDim.x < NC ? gridDim.x : NC;
      if (__hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1) {
        __hip_atomic_store(counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        l = 1;
      }
    }
    last = l;
  }
  __syncthreads();
  if (!last) return;
  double a = 0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kB)
    a += __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double r = block_sum(a, red);
  if (threadIdx.x == 0) result[0] = r;
}

__global__ __launch_bounds__(1024) void k_reduce(const double* partials, int n, double* result) {
  __shared__ double red[16];
  double a = 0;
  for (int i = threadIdx.x; i < n; i += 1024) a += partials[i];
  const double r = block_sum(a, red);
  if (threadIdx.x == 0) result[0] = r;
}

int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 4883;
  const int per = argc > 2 ? atoi(argv[2]) : 4096;  // doubles per workgroup (32 KB read + 32 KB write)
  const int reps = 200;
  double *x, *y, *partials, *result;
  unsigned* counter;
  CK(hipMalloc(&x, (size_t)nblk * per * 8));
  CK(hipMalloc(&y, (size_t)nblk * per * 8));
  CK(hipMalloc(&partials, nblk * 8));
  CK(hipMalloc(&result, 64));
  CK(hipMalloc(&counter, 65536));
  CK(hipMemset(counter, 0, 65536));
  std::vector<double> hx((size_t)nblk * per, 1.0);
  CK(hipMemcpy(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](int mode, const char* name) {
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) {
        if (mode == 0) {
          hipLaunchKernelGGL(k_body<0>, dim3(nblk), dim3(kB), 0, s, x, y, per, partials, counter, (unsigned)nblk, result);
          hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, s, partials, nblk, result);
        } else if (mode == 1) {
          hipLaunchKernelGGL(k_body<1>, dim3(nblk), dim3(kB), 0, s, x, y, per, partials, counter, (unsigned)nblk, result);
        } else if (mode == 2) {
          hipLaunchKernelGGL(k_body<2>, dim3(nblk), dim3(kB), 0, s, x, y, per, partials, counter, (unsigned)nblk, result);
        } else if (mode == 4) {
          hipLaunchKernelGGL(k_body_h<64>, dim3(nblk), dim3(kB), 0, s, x, y, per, partials, counter, result);
        } else if (mode == 5) {
          hipLaunchKernelGGL(k_body_h<256>, dim3(nblk), dim3(kB), 0, s, x, y, per, partials, counter, result);
        } else {
          hipLaunchKernelGGL(k_body<0>, dim3(nblk), dim3(kB), 0, s, x, y, per, partials, counter, (unsigned)nblk, result);
        }
      }
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double r;
    CK(hipMemcpy(&r, result, 8, hipMemcpyDeviceToHost));
    printf("%-44s %8.2f us/iter   result %.1f (expect %.1f)\n", name, ms * 1e3 / reps, r, (double)nblk * per);
  };
  run(3, "body only");
  run(0, "body + separate reduce kernel");
  run(1, "body with fence + ticket + last-block reduce");
  run(2, "body with ticket, no fence (cost only)");
  run(4, "hierarchical tickets (64), atomic partials");
  run(5, "hierarchical tickets (256), atomic partials");
  return 0;
}

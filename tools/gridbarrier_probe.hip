// Developer probe (not part of the product): cost of a grid-wide barrier among G co-resident workgroups on the
// 8-XCD MI355X -- what a single-launch persistent Krylov kernel would pay per reduction instead of a kernel boundary.
// Each round: every workgroup publishes one partial (agent-scope store), takes a ticket, waits for the round's
// release flag, then reads ALL partials (the redundant reduction a persistent kernel would do).  Bounded spins: a
// workgroup that waits more than ~50 ms gives up and raises `fail` (no hang whatever happens).
//   hipcc -O3 --offload-arch=gfx950 -o tools/gridbarrier_probe tools/gridbarrier_probe.hip && tools/gridbarrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// MODE 0: flat (one counter, one flag); MODE 1: two-level (8 group counters by blockIdx & 7 -- the XCD a workgroup is
// dispatched to -- then one global counter of group leaders)
template <int MODE>
__global__ __launch_bounds__(256) void k_rounds(int rounds, unsigned* cnt, unsigned* gcnt, volatile unsigned* flag,
                                                double* partials, double* out, int* fail, int work) {
  __shared__ double sred[4];
  const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  double acc = 0.0;
  for (int r = 1; r <= rounds; ++r) {
    // some local work standing for the product tile
    double v = (double)(b + t + r);
    for (int k = 0; k < work; ++k) v = v * 1.0000001 + 1e-9;
    double s = v;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((t & 63) == 0) sred[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
      __hip_atomic_store(&partials[b], sred[0] + sred[1] + sred[2] + sred[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool last = false;
      if (MODE == 0) {
        last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(r * G - 1);
      } else {
        const int g = b & 7, gsize = (G + 7 - g) / 8;
        if (__hip_atomic_fetch_add(&gcnt[g * 32], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(r * gsize - 1))
          last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(r * 8 - 1);
      }
      if (last) __hip_atomic_store((unsigned*)flag, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      while (__hip_atomic_load((unsigned*)flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {
        if (++spins > 20000000L) {
          *fail = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (*fail) return;
    // redundant reduction of all partials by every workgroup (fixed order)
    double a = 0.0;
    for (int i = t; i < G; i += 256) a += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((t & 63) == 0) sred[t >> 6] = a;
    __syncthreads();
    acc += sred[0] + sred[1] + sred[2] + sred[3];
    __syncthreads();
  }
  if (t == 0) out[b] = acc;
}

// MODE 2: flag array -- no read-modify-write on a shared address at all.  Workgroup b stores {partial, round} into its
// own slot; every workgroup then polls ALL slots (thread t watches slots t, t + 256, ...) until each carries the round
// number, and reduces the values it just read: one store-to-visibility plus one load round trip after the last arrival.
__global__ __launch_bounds__(256) void k_rounds_flags(int rounds, unsigned long long* vals, unsigned* tags, double* out,
                                                      int* fail, int work) {
  __shared__ double sred[4];
  __shared__ int sfail;
  const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  double acc = 0.0;
  if (t == 0) sfail = 0;
  for (int r = 1; r <= rounds; ++r) {
    double v = (double)(b + t + r);
    for (int k = 0; k < work; ++k) v = v * 1.0000001 + 1e-9;
    double s = v;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((t & 63) == 0) sred[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
      const double mine = sred[0] + sred[1] + sred[2] + sred[3];
      // double buffer by round parity: a fast workgroup may publish round r + 1 while a slow one still reads round r
      __hip_atomic_store(&vals[(r & 1) * 4096 + b], (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&tags[(r & 1) * 4096 + b], (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    double a = 0.0;
    for (int i = t; i < G; i += 256) {
      long spins = 0;
      while (__hip_atomic_load(&tags[(r & 1) * 4096 + i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {
        if (++spins > 20000000L) {
          sfail = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      a += __longlong_as_double((long long)__hip_atomic_load(&vals[(r & 1) * 4096 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    __syncthreads();
    if ((t & 63) == 0) sred[t >> 6] = a;
    __syncthreads();
    if (sfail) {
      *fail = 1;
      return;
    }
    acc += sred[0] + sred[1] + sred[2] + sred[3];
    __syncthreads();
  }
  if (t == 0) out[b] = acc;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  unsigned *cnt, *gcnt, *flag;
  double *partials, *out;
  int* fail;
  CK(hipMalloc(&cnt, 4));
  CK(hipMalloc(&gcnt, 8 * 32 * 4));
  CK(hipMalloc(&flag, 4));
  CK(hipMalloc(&partials, 4096 * 8));
  CK(hipMalloc(&out, 4096 * 8));
  CK(hipMalloc(&fail, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int rounds = 2000;
  printf("%s, %d CUs: microseconds per round (barrier + redundant reduction of G partials), %d rounds\n", p.name,
         p.multiProcessorCount, rounds);
  for (int mode = 0; mode < 2; ++mode)
    for (int G : {64, 128, 256, 512, 1024})
      for (int work : {0, 2000}) {
        if (G > p.multiProcessorCount * 4) continue;  // must be co-resident (4 workgroups of 256 threads per CU at most here)
        CK(hipMemset(cnt, 0, 4));
        CK(hipMemset(gcnt, 0, 8 * 32 * 4));
        CK(hipMemset(flag, 0, 4));
        CK(hipMemset(fail, 0, 4));
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(k_rounds<0>, dim3(G), dim3(256), 0, 0, rounds, cnt, gcnt, flag, partials, out, fail, work);
        else hipLaunchKernelGGL(k_rounds<1>, dim3(G), dim3(256), 0, 0, rounds, cnt, gcnt, flag, partials, out, fail, work);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int f = 0;
        CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("mode %s  G %4d  work %4d : %7.2f us/round%s\n", mode ? "two-level" : "flat     ", G, work, 1e3 * ms / rounds,
               f ? "  (GAVE UP: not co-resident?)" : "");
      }
  unsigned long long* vals;
  unsigned* tags;
  CK(hipMalloc(&vals, 2 * 4096 * 8));
  CK(hipMalloc(&tags, 2 * 4096 * 4));
  for (int G : {64, 128, 256, 512, 1024})
    for (int work : {0, 2000}) {
      if (G > p.multiProcessorCount * 4) continue;
      CK(hipMemset(tags, 0, 2 * 4096 * 4));
      CK(hipMemset(fail, 0, 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rounds_flags, dim3(G), dim3(256), 0, 0, rounds, vals, tags, out, fail, work);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      int f = 0;
      CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
      printf("mode flag-array G %4d  work %4d : %7.2f us/round%s\n", G, work, 1e3 * ms / rounds,
             f ? "  (GAVE UP: not co-resident?)" : "");
    }
  // reference point: the same number of rounds as kernel boundaries (an empty 1-workgroup kernel after a G-workgroup one)
  return 0;
}

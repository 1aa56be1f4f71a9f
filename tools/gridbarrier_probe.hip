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

// MODE 3: the "barrier-xcd" row of /opt/skills/guides/MI355X_MICROARCH.md (persistent-kernel price list: 4.1 us at 256
// workgroups with nothing published, 4.7 re-reading a 128-B record per workgroup), built as that row describes it:
//   * one arrival counter PER XCC (its own 128-byte line), bumped by thread 0 of a workgroup with a RELAXED agent-scope
//     add after the workgroup's own stores have left the CU (s_waitcnt vmcnt(0): the vector L1 is write-through, so they
//     are in the XCC's L2);
//   * the LAST arriver of an XCC is its leader: one release fence (writes the XCC's L2 back -- the only release in the
//     XCC), a relaxed add on the top counter, a relaxed sc1-load poll (+ s_sleep) until all XCC leaders have arrived, one
//     acquire fence, then the XCC's generation word;
//   * everybody else polls its XCC's generation word with relaxed sc1 loads (+ s_sleep) and takes one acquire fence.
// The XCC of a workgroup is read from the hardware (s_getreg XCC_ID); the per-XCC head counts come from a census at the
// start of the kernel (one slow flat barrier, once).
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }

struct XcdBarrier {
  unsigned xcnt[8 * 32];   // per-XCC arrival counters, 128 bytes apart
  unsigned xgen[8 * 32];   // per-XCC generation words
  unsigned top[32];
  unsigned census[8 * 32];
  unsigned flat[32], flatflag[32];
};

__global__ __launch_bounds__(256) void k_rounds_xcd(int rounds, XcdBarrier* B, double* partials, double* out, int* fail,
                                                    int work, int readback) {
  __shared__ double sred[4];
  __shared__ int s_nx, s_nxcc, s_fail;
  const int G = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  const int x = xcc_id();
  if (t == 0) {
    s_fail = 0;
    // census + one flat barrier
    __hip_atomic_fetch_add(&B->census[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(&B->flat[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(G - 1))
      __hip_atomic_store(&B->flatflag[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(&B->flatflag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      if (++spins > 20000000L) {
        s_fail = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    int nxcc = 0;
    for (int k = 0; k < 8; ++k) nxcc += __hip_atomic_load(&B->census[k * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    s_nxcc = nxcc;
    s_nx = (int)__hip_atomic_load(&B->census[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_fail) {
    if (t == 0) *fail = 1;
    return;
  }
  const unsigned nx = (unsigned)s_nx, nxcc = (unsigned)s_nxcc;
  double acc = 0.0;
  for (int r = 1; r <= rounds; ++r) {
    double v = (double)(b + t + r);
    for (int k = 0; k < work; ++k) v = v * 1.0000001 + 1e-9;
    double s = v;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((t & 63) == 0) sred[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
      partials[(size_t)(r & 1) * 4096 + b] = sred[0] + sred[1] + sred[2] + sred[3];  // plain store (double-buffered by parity)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 // ... it has left the CU
      const unsigned prev = __hip_atomic_fetch_add(&B->xcnt[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      long spins = 0;
      if (prev == (unsigned)r * nx - 1u) {  // the XCC's leader
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&B->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&B->top[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * nxcc) {
          if (++spins > 20000000L) {
            s_fail = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(&B->xgen[x * 32], (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(&B->xgen[x * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) {
          if (++spins > 20000000L) {
            s_fail = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    if (s_fail) {
      if (t == 0) *fail = 1;
      return;
    }
    // what every workgroup reads back after the barrier: nothing / one 128-byte record / ALL G partials (the redundant
    // reduction a persistent Krylov kernel would do)
    double a = 0.0;
    if (readback == 1) {
      if (t < 16) a = partials[(size_t)(r & 1) * 4096 + ((b + 1) % G & ~15) + t];
    } else if (readback == 2) {
      for (int i = t; i < G; i += 256) a += partials[(size_t)(r & 1) * 4096 + i];
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if ((t & 63) == 0) sred[t >> 6] = a;
    __syncthreads();
    acc += sred[0] + sred[1] + sred[2] + sred[3];
    __syncthreads();
  }
  if (t == 0) out[b] = acc;
}

// reference point: the same "phases" as kernel boundaries -- a G-workgroup kernel publishing a partial, then a ONE-workgroup
// kernel reducing the G partials (what the product + k_step pair of libfpsq does today)
__global__ __launch_bounds__(256) void k_phase_publish(double* partials, int r, int work) {
  __shared__ double sred[4];
  const int b = blockIdx.x, t = threadIdx.x;
  double v = (double)(b + t + r);
  for (int k = 0; k < work; ++k) v = v * 1.0000001 + 1e-9;
  double s = v;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if ((t & 63) == 0) sred[t >> 6] = s;
  __syncthreads();
  if (t == 0) partials[b] = sred[0] + sred[1] + sred[2] + sred[3];
}
__global__ __launch_bounds__(256) void k_phase_reduce(const double* partials, int G, double* out) {
  __shared__ double sred[4];
  const int t = threadIdx.x;
  double a = 0.0;
  for (int i = t; i < G; i += 256) a += partials[i];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if ((t & 63) == 0) sred[t >> 6] = a;
  __syncthreads();
  if (t == 0) out[0] += sred[0] + sred[1] + sred[2] + sred[3];
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
  // ---- barrier-xcd (the guide's row), with nothing / a 128-byte record / all G partials read back after each barrier
  XcdBarrier* XB;
  CK(hipMalloc(&XB, sizeof(XcdBarrier)));
  double* part2;
  CK(hipMalloc(&part2, 2 * 4096 * 8));
  for (int G : {64, 128, 256, 512, 1024})
    for (int readback : {0, 1, 2})
      for (int work : {0, 2000}) {
        if (G > p.multiProcessorCount * 4) continue;
        CK(hipMemset(XB, 0, sizeof(XcdBarrier)));
        CK(hipMemset(part2, 0, 2 * 4096 * 8));
        CK(hipMemset(fail, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rounds_xcd, dim3(G), dim3(256), 0, 0, rounds, XB, part2, out, fail, work, readback);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int f = 0;
        CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("mode barrier-xcd G %4d  readback %s  work %4d : %7.2f us/round%s\n", G,
               readback == 0 ? "none   " : readback == 1 ? "128 B  " : "all G  ", work, 1e3 * ms / rounds,
               f ? "  (GAVE UP: not co-resident?)" : "");
      }
  // ---- reference point: the same rounds as KERNEL BOUNDARIES (a G-workgroup kernel, then a one-workgroup reduction kernel:
  // two launches per round, what the product + scalar-step pair costs today)
  for (int G : {64, 128, 256, 512, 1024})
    for (int work : {0, 2000}) {
      CK(hipMemset(out, 0, 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 1; r <= rounds; ++r) {
        hipLaunchKernelGGL(k_phase_publish, dim3(G), dim3(256), 0, 0, part2, r, work);
        hipLaunchKernelGGL(k_phase_reduce, dim3(1), dim3(256), 0, 0, part2, G, out);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode two launches G %4d  work %4d : %7.2f us/round (publish kernel + 1-workgroup reduce kernel)\n", G, work,
             1e3 * ms / rounds);
    }
  return 0;
}

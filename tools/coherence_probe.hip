// Developer probe (round 4): what an agent-scope (sc1) load returns when the XCD's own L2 holds an OLDER copy of the line.
// The fused launch (k_iter_fused) never has such a copy: nobody reads a line before it is final.  A launch covering several
// iterations would (the previous iteration's value of the same address), so whether the gathers may stay L2-hits there
// depends on this.  One reader workgroup on XCC 0, one writer on XCC 1, a ping-pong of N rounds on one 128-byte line:
//   reader: read the line (plain or sc1: mode bit 0) -> it is now in L2_0;  tell the writer
//   writer: write round number r into the line with an sc1 (written-through) store, vmcnt(0), raise its flag
//   reader: see the flag, read the line again (plain or sc1: mode bit 1) and compare with r
// prints how many of the N re-reads returned the OLD value, for the four combinations.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/coherence_probe tools/coherence_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7; }
__device__ __forceinline__ unsigned long long ld_ag(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a plain, cacheable load (a volatile C++ load would be emitted with scope bits)
__device__ __forceinline__ unsigned long long ld_plain(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_ag(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_probe(unsigned long long* line /* 16 words */, unsigned long long* flag_r, unsigned long long* flag_w,
                        unsigned int* claim /* [2] */, unsigned long long* result /* [4]: stale, rounds, xcc_r, xcc_w */, int rounds,
                        int mode) {
  if (threadIdx.x != 0) return;
  const int x = xcc_id();
  int role = -1;
  if (x == 0 && atomicCAS(&claim[0], 0u, 1u) == 0u) role = 0;
  else if (x == 1 && atomicCAS(&claim[1], 0u, 1u) == 0u) role = 1;
  if (role < 0) return;
  if (role == 0) {
    unsigned long long stale = 0;
    for (int r = 1; r <= rounds; ++r) {
      unsigned long long first = (mode & 1) ? ld_ag(line + 3) : ld_plain(line + 3);
      (void)first;
      st_ag(flag_r, (unsigned long long)r);
      long spins = 0;
      while (ld_ag(flag_w) < (unsigned long long)r && ++spins < 100000000L) __builtin_amdgcn_s_sleep(1);
      const unsigned long long again = (mode & 2) ? ld_ag(line + 3) : ld_plain(line + 3);
      if (again != (unsigned long long)r) ++stale;
    }
    result[0] = stale;
    result[1] = rounds;
    result[2] = x;
  } else {
    for (int r = 1; r <= rounds; ++r) {
      long spins = 0;
      while (ld_ag(flag_r) < (unsigned long long)r && ++spins < 100000000L) __builtin_amdgcn_s_sleep(1);
      st_ag(line + 3, (unsigned long long)r);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st_ag(flag_w, (unsigned long long)r);
    }
    result[3] = x;
  }
}

// the same XCD: a plain read (the line is in this XCD's L2), then an sc1 store to it from the same workgroup, then a plain read
// again -- does the written-through store leave the older copy in the L2?
__global__ void k_self(unsigned long long* line, unsigned long long* result, int rounds) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long stale = 0;
  for (int r = 1; r <= rounds; ++r) {
    (void)ld_plain(line + 5);
    st_ag(line + 5, (unsigned long long)r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ld_plain(line + 5) != (unsigned long long)r) ++stale;
  }
  result[0] = stale;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned long long *line, *fr, *fw, *res;
  unsigned int* claim;
  if (hipMalloc(&line, 4096) != hipSuccess || hipMalloc(&fr, 256) != hipSuccess || hipMalloc(&fw, 256) != hipSuccess ||
      hipMalloc(&res, 64) != hipSuccess || hipMalloc(&claim, 8) != hipSuccess)
    return 2;
  const char* names[4] = {"first read plain, re-read plain", "first read sc1,   re-read plain", "first read plain, re-read sc1  ",
                          "first read sc1,   re-read sc1  "};
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(line, 0, 4096);
    hipMemset(fr, 0, 256);
    hipMemset(fw, 0, 256);
    hipMemset(res, 0xff, 64);
    hipMemset(claim, 0, 8);
    hipLaunchKernelGGL(k_probe, dim3(64), dim3(64), 0, 0, line, fr, fw, claim, res, rounds, mode);
    if (hipDeviceSynchronize() != hipSuccess) {
      printf("mode %d: kernel failed\n", mode);
      return 1;
    }
    unsigned long long h[8];
    hipMemcpy(h, res, 64, hipMemcpyDeviceToHost);
    printf("%s: %llu of %llu re-reads returned the old value (reader on XCC %lld, writer on XCC %lld)\n", names[mode], h[0], h[1],
           (long long)h[2], (long long)h[3]);
  }
  hipMemset(line, 0, 4096);
  hipLaunchKernelGGL(k_self, dim3(1), dim3(64), 0, 0, line, res, rounds);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  unsigned long long h0 = 0;
  hipMemcpy(&h0, res, 8, hipMemcpyDeviceToHost);
  printf("same XCD: plain read, sc1 store, plain read: %llu of %d re-reads returned the old value\n", h0, rounds);
  return 0;
}

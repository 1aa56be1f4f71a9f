// stream_probe.hip -- developer probe: how fast do ONE-TILE workgroups (256 threads, 2048 entries of 8-byte value + 4-byte
// index, the shape of the product kernels' matrix stream) read HBM, as a function of the load shape and of the workgroups
// resident per CU?  Nothing but the stream: loads, a sum, one store per workgroup.
//   hipcc -O3 --offload-arch=gfx950 -o stream_probe tools/stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <utility>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kTile = 2048, kBlock = 256;

// V0: entry order, 8 x (8-byte + 4-byte) loads per thread
template <int LDS_KB, int TILES>
__global__ __launch_bounds__(kBlock) void k_v0(const double* __restrict__ vals, const unsigned* __restrict__ idx, double* out, int ntile) {
  __shared__ double pad[LDS_KB > 0 ? LDS_KB * 128 : 256];
  double acc = 0.0;
  unsigned ia = 0;
  for (int t = 0; t < TILES; ++t) {
    const int L = blockIdx.x * TILES + t;
    if (L >= ntile) break;
    double v[8];
    unsigned c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t i = (size_t)L * kTile + threadIdx.x + k * kBlock;
      v[k] = vals[i];
      c[k] = idx[i];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc += v[k];
      ia += c[k];
    }
  }
  if (LDS_KB > 0) pad[threadIdx.x] = acc;
  if (acc == 1.2345e300 || ia == 0xdeadbeefu) out[blockIdx.x] = acc + ia + (LDS_KB > 0 ? pad[(threadIdx.x + 1) & 255] : 0.0);
}

// V1: per-thread contiguous, 16-byte loads: 4 for the values (pairs), 2 for the indices
template <int LDS_KB, int TILES>
__global__ __launch_bounds__(kBlock) void k_v1(const double* __restrict__ vals, const unsigned* __restrict__ idx, double* out, int ntile) {
  __shared__ double pad[LDS_KB > 0 ? LDS_KB * 128 : 256];
  double acc = 0.0;
  unsigned ia = 0;
  for (int t = 0; t < TILES; ++t) {
    const int L = blockIdx.x * TILES + t;
    if (L >= ntile) break;
    double2 v[4];
    uint4 c[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const double2*>(vals + (size_t)L * kTile + 2 * threadIdx.x + 512 * j);
#pragma unroll
    for (int j = 0; j < 2; ++j) c[j] = *reinterpret_cast<const uint4*>(idx + (size_t)L * kTile + 4 * threadIdx.x + 1024 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc += v[j].x + v[j].y;
#pragma unroll
    for (int j = 0; j < 2; ++j) ia += c[j].x + c[j].y + c[j].z + c[j].w;
  }
  if (LDS_KB > 0) pad[threadIdx.x] = acc;
  if (acc == 1.2345e300 || ia == 0xdeadbeefu) out[blockIdx.x] = acc + ia + (LDS_KB > 0 ? pad[(threadIdx.x + 1) & 255] : 0.0);
}


// residency: entry / exit stamps (s_memrealtime) of every workgroup of a streaming launch holding LDS_KB of LDS
template <int LDS_KB>
__global__ __launch_bounds__(kBlock) void k_res(const double* __restrict__ vals, long long* stamps, double* out, int ntile) {
  __shared__ double pad[LDS_KB * 128];
  const long long t0 = wall_clock64();
  const int L = blockIdx.x;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) acc += vals[(size_t)L * kTile + threadIdx.x + k * kBlock];
  pad[threadIdx.x] = acc;
  __syncthreads();
  if (acc == 1.2345e300) out[blockIdx.x] = pad[(threadIdx.x + 1) & 255];
  if (threadIdx.x == 0) {
    stamps[2 * L] = t0;
    stamps[2 * L + 1] = wall_clock64();
  }
}

template <class K>
void residency(const char* name, K kern, const double* vals, double* out, int ntile) {
  long long* st;
  CHECK(hipMalloc(&st, (size_t)ntile * 16));
  hipLaunchKernelGGL(kern, dim3(ntile), dim3(kBlock), 0, 0, vals, st, out, ntile);
  hipLaunchKernelGGL(kern, dim3(ntile), dim3(kBlock), 0, 0, vals, st, out, ntile);
  CHECK(hipDeviceSynchronize());
  std::vector<long long> h((size_t)ntile * 2);
  CHECK(hipMemcpy(h.data(), st, (size_t)ntile * 16, hipMemcpyDeviceToHost));
  std::vector<std::pair<long long, int>> ev;
  for (int i = 0; i < ntile; ++i) {
    ev.push_back({h[2 * i], +1});
    ev.push_back({h[2 * i + 1], -1});
  }
  std::sort(ev.begin(), ev.end(), [](const std::pair<long long, int>& a, const std::pair<long long, int>& b) { return a.first < b.first || (a.first == b.first && a.second < b.second); });
  int cur = 0, mx = 0;
  for (auto& e : ev) {
    cur += e.second;
    mx = std::max(mx, cur);
  }
  printf("%-34s most workgroups resident at once: %d (%.2f per CU)\n", name, mx, mx / 256.0);
  CHECK(hipFree(st));
}

template <class K>
void run(const char* name, K kern, int grid, const double* vals, const unsigned* idx, double* out, int ntile, double bytes) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, 0, vals, idx, out, ntile);
  CHECK(hipDeviceSynchronize());
  const int reps = 50;
  CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), 0, 0, vals, idx, out, ntile);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("%-46s grid %6d  %7.2f us  %6.2f TB/s\n", name, grid, 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e12);
}

int main(int argc, char** argv) {
  const int ntile = argc > 1 ? atoi(argv[1]) : 4883 * 4;  // x 24 KB: 480 MB by default (beyond the 256 MB Infinity Cache)
  const size_t n = (size_t)ntile * kTile;
  double* vals;
  unsigned* idx;
  double* out;
  CHECK(hipMalloc(&vals, n * 8));
  CHECK(hipMalloc(&idx, n * 4));
  CHECK(hipMalloc(&out, (size_t)ntile * 8));
  CHECK(hipMemset(vals, 0, n * 8));
  CHECK(hipMemset(idx, 0, n * 4));
  const double bytes = (double)n * 12;
  printf("%d tiles of %d entries (8 + 4 bytes): %.0f MB per launch\n", ntile, kTile, bytes / 1e6);
  run("V0 entry order 8x(8B+4B), 2 KB LDS, 1 tile/WG", k_v0<0, 1>, ntile, vals, idx, out, ntile, bytes);
  run("V0 entry order, 32 KB LDS (5 WG/CU), 1 tile/WG", k_v0<32, 1>, ntile, vals, idx, out, ntile, bytes);
  run("V0 entry order, 32 KB LDS, 4 tiles/WG", k_v0<32, 4>, (ntile + 3) / 4, vals, idx, out, ntile, bytes);
  run("V1 16-byte loads, 2 KB LDS, 1 tile/WG", k_v1<0, 1>, ntile, vals, idx, out, ntile, bytes);
  run("V1 16-byte loads, 32 KB LDS (5 WG/CU), 1 tile/WG", k_v1<32, 1>, ntile, vals, idx, out, ntile, bytes);
  run("V1 16-byte loads, 32 KB LDS, 4 tiles/WG", k_v1<32, 4>, (ntile + 3) / 4, vals, idx, out, ntile, bytes);
  run("V1 16-byte loads, 16 KB LDS (10 WG/CU), 1 tile/WG", k_v1<16, 1>, ntile, vals, idx, out, ntile, bytes);
  run("V0 entry order, 16 KB LDS (10 WG/CU), 1 tile/WG", k_v0<16, 1>, ntile, vals, idx, out, ntile, bytes);
  residency("residency, 2 KB LDS", k_res<2>, vals, out, ntile);
  residency("residency, 16 KB LDS", k_res<16>, vals, out, ntile);
  residency("residency, 24 KB LDS", k_res<24>, vals, out, ntile);
  residency("residency, 30 KB LDS", k_res<30>, vals, out, ntile);
  residency("residency, 32 KB LDS", k_res<32>, vals, out, ntile);
  residency("residency, 40 KB LDS", k_res<40>, vals, out, ntile);
  return 0;
}

// Developer probe (not part of the product): phase timing inside the 128 x 128 diagonal-block kernel of the direct
// back-ends (k_potrf_inv128m), via s_memtime stamps of thread 0.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFPSQ_POTRF_TIMING -I fletcherpenaltysolver.jl_amd/csrc -o tools/potrf_probe tools/potrf_probe.hip
#include "fpsq_dense.hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
using namespace fpsq;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
  const int gen = argc > 1 ? atoi(argv[1]) : 5;
  const int n = 128;
  std::vector<double> A(n * n), M(n * n);
  srand(1);
  for (auto& v : A) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = i == j ? 1.0 : 0.0;
      for (int k = 0; k < n; ++k) s += A[i * n + k] * A[j * n + k];
      M[i * n + j] = s;
    }
  double *dM, *dinv, *dinvT;
  int* info;
  long long* stamps;
  CK(hipMalloc(&dM, n * n * 8)); CK(hipMalloc(&dinv, n * n * 8)); CK(hipMalloc(&dinvT, n * n * 8));
  CK(hipMalloc(&info, 16)); CK(hipMalloc(&stamps, 64 * 8));
  CK(hipMemset(dinv, 0, n * n * 8)); CK(hipMemset(dinvT, 0, n * n * 8));
  CK(hipFuncSetAttribute((const void*)k_potrf_inv128m, hipFuncAttributeMaxDynamicSharedMemorySize, kPotrfLds5));
  std::vector<long long> hs(64);
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(dM, M.data(), n * n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(info, 0, 16)); CK(hipMemset(stamps, 0, 64 * 8));
    (void)gen;  // (the earlier generations were removed from the source in round 3)
    hipLaunchKernelGGL(k_potrf_inv128m, dim3(1), dim3(kPotrfThreads5), kPotrfLds5, 0, dM, n, dinv, dinvT, 0, info, 0.0, 0.0, stamps);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hs.data(), stamps, 64 * 8, hipMemcpyDeviceToHost));
  }
  // s_memtime ticks at 100 MHz on gfx9-family parts (10 ns)
  const char* names[] = {"start", "loaded"};
  (void)names;
  printf("phase stamps (us since start, 100 MHz counter assumed):\n");
  for (int i = 1; i < 64 && hs[i]; ++i) printf("  %2d: %8.0f ticks  (+%.0f)\n", i, (double)(hs[i] - hs[0]), (double)(hs[i] - hs[i - 1]));
  std::vector<double> L(n * n), X(n * n), XT(n * n);
  CK(hipMemcpy(L.data(), dM, n * n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(X.data(), dinv, n * n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(XT.data(), dinvT, n * n * 8, hipMemcpyDeviceToHost));
  double e3 = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) e3 = fmax(e3, fabs(X[i * n + j] - XT[j * n + i]));
  printf("generation %d: max |X - (X')'| = %.2e\n", gen, e3);
  double e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0, t = 0;
      for (int k = 0; k <= j; ++k) s += L[i * n + k] * L[j * n + k];
      for (int k = j; k <= i; ++k) t += L[i * n + k] * X[k * n + j];
      e1 = fmax(e1, fabs(s - M[i * n + j]));
      e2 = fmax(e2, fabs(t - (i == j ? 1.0 : 0.0)));
    }
  printf("max |LL' - M| = %.2e, max |L X - I| = %.2e\n", e1, e2);
  return 0;
}

// fpsq_dense.hip.h -- dense-block Jacobian variant: normal equations M = A A' + delta I on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64), blocked Cholesky, blocked triangular solves.  This is the direct back-end of the seam for
// small / dense problems (reference: LDLtSolver path, src/solve_linear_system.jl:206-252 and the dense
// A A' + tau I contraction of src/model-Fletcherpenaltynlp.jl:478-484); MFMA is used only here.
//
// All matrices are row-major fp64, padded with zeros to multiples of kDB = 128 (rows of A, order of M) and 16 (columns
// of A); the padded diagonal of M is set to 1 so the factorisation is unaffected.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpsq {

constexpr int kDB = 128;      // block size of the Cholesky / GEMM tiles
constexpr int kDK = 16;       // k-depth of one LDS stage
constexpr int kDLd = 144;     // LDS leading dimension (doubles) of a [k][row] tile: 128 + 16 so that the four k-planes a
                              // wave reads with one ds_read_b64 fall in disjoint bank halves
using f64x4 = __attribute__((ext_vector_type(4))) double;

// C (M x N, ldc) = alpha * A (M x K, lda) * B (N x K, ldb)' + beta * C.   M, N multiples of 128, K multiple of 16.
// One workgroup = one 128 x 128 tile of C, 4 waves in a 2 x 2 grid, each wave 64 x 64 = 4 x 4 MFMA tiles of 16 x 16.
// LOWER: only tiles with blockIdx.y >= blockIdx.x are computed (symmetric rank-k update of the lower triangle).
// Fragment maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3): lane l holds A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15]; D register r of lane l is D[row = (l >> 4) + 4 r][col = l & 15].
template <bool LOWER>
__global__ __launch_bounds__(256) void k_gemm_nt_f64(double* C, int ldc, const double* __restrict__ A, int lda,
                                                     const double* __restrict__ B, int ldb, int K, double alpha,
                                                     double beta) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (LOWER && bi < bj) return;
  __shared__ double sA[2][kDK * kDLd];
  __shared__ double sB[2][kDK * kDLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;  // this wave's 64 x 64 sub-tile
  const double* Ab = A + (size_t)bi * kDB * lda;
  const double* Bb = B + (size_t)bj * kDB * ldb;
  // staging: thread t copies 8 consecutive k of row (t >> 1) for both operands
  const int srow = tid >> 1, sk = (tid & 1) * 8;
  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  double ra[8], rb[8];
  auto gload = [&](int k0) {
    const double2* pa = reinterpret_cast<const double2*>(Ab + (size_t)srow * lda + k0 + sk);
    const double2* pb = reinterpret_cast<const double2*>(Bb + (size_t)srow * ldb + k0 + sk);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 a = pa[q], b = pb[q];
      ra[2 * q] = a.x;
      ra[2 * q + 1] = a.y;
      rb[2 * q] = b.x;
      rb[2 * q + 1] = b.y;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      sA[buf][(sk + q) * kDLd + srow] = ra[q];
      sB[buf][(sk + q) * kDLd + srow] = rb[q];
    }
  };
  gload(0);
  lstore(0);
  __syncthreads();
  const int fr = lane & 15, fk = lane >> 4;
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += kDK) {
    const bool more = k0 + kDK < K;
    if (more) gload(k0 + kDK);
#pragma unroll
    for (int ks = 0; ks < kDK / 4; ++ks) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sA[buf][(ks * 4 + fk) * kDLd + wr + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = sB[buf][(ks * 4 + fk) * kDLd + wc + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  double* Cb = C + (size_t)(bi * kDB + wr) * ldc + bj * kDB + wc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* p = Cb + (size_t)(i * 16 + fk + 4 * r) * ldc + j * 16 + fr;
        const double v = alpha * acc[i][j][r];
        *p = (beta != 0.0) ? v + beta * *p : v;
      }
}

// M[i][i] += delta for i < m; M[i][i] = 1 on the padding
__global__ void k_dense_diag(double* M, int ld, int m, int mpad, double delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mpad) M[(size_t)i * ld + i] = i < m ? M[(size_t)i * ld + i] + delta : 1.0;
}

// Cholesky of ONE 128 x 128 diagonal block (lower, in place in global memory) AND the inverse of its factor
// (`inv`, 128 x 128 row-major lower).  One workgroup of 256 threads; the block lives in LDS (dynamic: 128 x 129 + 128
// doubles).  After L has been written back, it is inverted IN PLACE in LDS (unblocked lower inversion, column by
// column from the right: X[j+1:, j] = -X[j+1:, j+1:] L[j+1:, j] / L[j][j]).
// info[0] = first non-positive pivot (1-based global row) or stays 0.
constexpr int kPotrfThreads = 1024;  // 8 threads per row: the loops are LDS-latency bound, so spread each row thin
__global__ __launch_bounds__(kPotrfThreads) void k_potrf_inv128(double* Mkk, int ld, double* inv, int row0, int* info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* L = sm;
  constexpr int LD = kDB + 1, NT = kPotrfThreads, TPR = NT / kDB;
  double* col = sm + kDB * LD;  // 128 doubles
  const int tid = threadIdx.x;
  const int i = tid / TPR, h = tid % TPR;
  for (int e = tid; e < kDB * kDB; e += NT) {
    const int r = e >> 7, c = e & 127;
    L[r * LD + c] = (c <= r) ? Mkk[(size_t)r * ld + c] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < kDB; ++j) {
    const double d = L[j * LD + j];
    const bool bad = !(d > 0.0);
    if (bad && tid == 0) atomicCAS(info, 0, row0 + j + 1);
    const double piv = bad ? 1.0 : sqrt(d);  // a unit pivot keeps the kernel finite; the caller reports `info`
    const double lij = (i > j) ? L[i * LD + j] / piv : 0.0;  // every thread of row i computes the same scaled entry
    __syncthreads();
    if (h == 0) {
      if (i > j) L[i * LD + j] = lij;
      if (i == j) L[j * LD + j] = piv;
    }
    // the scaled column j is needed by all rows: stage it
    if (h == 1 && i > j) col[i] = lij;
    __syncthreads();
    // trailing update of the lower triangle: L[i][c] -= L[i][j] L[c][j], j < c <= i
    if (i > j) {
      int c = j + 1 + h;
      for (; c + 3 * TPR <= i; c += 4 * TPR) {  // four independent read-modify-writes in flight
        const double a0 = col[c], a1 = col[c + TPR], a2 = col[c + 2 * TPR], a3 = col[c + 3 * TPR];
        const double b0 = L[i * LD + c], b1 = L[i * LD + c + TPR], b2 = L[i * LD + c + 2 * TPR], b3 = L[i * LD + c + 3 * TPR];
        L[i * LD + c] = b0 - lij * a0;
        L[i * LD + c + TPR] = b1 - lij * a1;
        L[i * LD + c + 2 * TPR] = b2 - lij * a2;
        L[i * LD + c + 3 * TPR] = b3 - lij * a3;
      }
      for (; c <= i; c += TPR) L[i * LD + c] -= lij * col[c];
    }
    __syncthreads();
  }
  for (int e = tid; e < kDB * kDB; e += NT) {
    const int r = e >> 7, c = e & 127;
    if (c <= r) Mkk[(size_t)r * ld + c] = L[r * LD + c];
  }
  __syncthreads();
  // in-place inverse of the lower-triangular L, column by column from the right
  for (int j = kDB - 1; j >= 0; --j) {
    const double djj = 1.0 / L[j * LD + j];
    if (tid < kDB) col[tid] = (tid > j) ? L[tid * LD + j] : 0.0;  // column j of L below the diagonal
    __syncthreads();
    if (tid == 0) L[j * LD + j] = djj;
    {
      double s0 = 0.0, s1 = 0.0;
      if (i > j) {
        int p = j + 1 + h;
        for (; p + TPR <= i; p += 2 * TPR) {
          s0 += L[i * LD + p] * col[p];
          s1 += L[i * LD + p + TPR] * col[p + TPR];
        }
        for (; p <= i; p += TPR) s0 += L[i * LD + p] * col[p];
      }
      double s = s0 + s1;
#pragma unroll
      for (int off = TPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (i > j && h == 0) L[i * LD + j] = -s * djj;
    }
    __syncthreads();
  }
  for (int e = tid; e < kDB * kDB; e += NT) {
    const int r = e >> 7, c = e & 127;
    inv[(size_t)r * kDB + c] = (c <= r) ? L[r * LD + c] : 0.0;
  }
}

// ---- the same diagonal-block job, second generation: 64 x 64 sub-blocks factored and inverted by ONE WAVE with the
// rows in registers (lane i = row i; broadcasts are v_readlane, no LDS traffic and no workgroup barrier in the 64
// column steps), glued by 4 x 4 register-tiled 64^3 products on the LDS copy:
//   A = [A11 0; A21 A22]:  L11 = chol(A11), X11 = L11^-1, L21 = A21 X11', A22 -= L21 L21', L22 = chol(A22), X22 = L22^-1,
//   X21 = -X22 (L21 X11).
// The unblocked kernel above costs ~1.07 us per column step (three 16-wave barriers each): 274 us per block.
__device__ __forceinline__ double rdlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Cholesky of the 64 x 64 block at (o, o) of the LDS matrix, by the calling wave; the factor replaces the block's lower
// triangle in LDS.  a[] (the lane's row of the factor) stays live for the caller.
__device__ __forceinline__ void wave_potrf64(double* L, int LD, int o, int row0, int* info, double* a) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 64; ++c) a[c] = L[(o + lane) * LD + o + c];
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const double d = rdlane(a[j], j);
    const bool bad = !(d > 0.0);
    if (bad && lane == 0) atomicCAS(info, 0, row0 + o + j + 1);
    const double piv = bad ? 1.0 : sqrt(d);  // a unit pivot keeps the kernel finite; the caller reports `info`
    const double rp = 1.0 / piv;              // (uniform: one division per column, not one per row)
    const double l = lane > j ? a[j] * rp : (lane == j ? piv : 0.0);
    a[j] = l;
#pragma unroll
    for (int c = j + 1; c < 64; ++c) a[c] -= l * rdlane(l, c);  // (meaningful for lanes >= c; the rest is never read)
  }
#pragma unroll
  for (int c = 0; c < 64; ++c) L[(o + lane) * LD + o + c] = c <= lane ? a[c] : 0.0;
}

// X = L^-1 for the 64 x 64 lower-triangular block whose rows are in a[] (wave_potrf64); X replaces the block in LDS.
// acc_i = e_i - sum_{k<i} L[i][k] X[k][.] is built by sweeping k; row k of X = acc_k / L[k][k] is broadcast on the fly.
__device__ __forceinline__ void wave_trtri64(double* L, int LD, int o, const double* a) {
  const int lane = threadIdx.x & 63;
  double acc[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) acc[c] = c == lane ? 1.0 : 0.0;
  double dsel = 1.0;  // this lane's diagonal entry (a[lane]), selected without dynamic register indexing
#pragma unroll
  for (int c = 0; c < 64; ++c) dsel = c == lane ? a[c] : dsel;
  const double dinv = 1.0 / dsel;
#pragma unroll
  for (int k = 0; k < 63; ++k) {
    const double dk = rdlane(dinv, k);
    const double lik = lane > k ? a[k] : 0.0;
#pragma unroll
    for (int c = 0; c <= k; ++c) acc[c] -= lik * (rdlane(acc[c], k) * dk);
  }
#pragma unroll
  for (int c = 0; c < 64; ++c) L[(o + lane) * LD + o + c] = c <= lane ? acc[c] * dinv : 0.0;
}

// C (64 x 64 at LDS (cr, cc)) = alpha * A B^T or alpha * A B (+ C) with A at (ar, ac), B at (br, bc); 256 threads, 4 x 4
// outputs each.  TB: 0 -> sum_k A[i][k] B[j][k], 1 -> sum_k A[i][k] B[k][j].  The result is RETURNED in registers: the caller
// synchronises before storing (outputs may overlay inputs).
template <int TB>
__device__ __forceinline__ void tile64(const double* L, int LD, int ar, int ac, int br, int bc, double out[4][4]) {
  const int ti = (threadIdx.x >> 4) * 4, tj = (threadIdx.x & 15) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) out[r][c] = 0.0;
  for (int k = 0; k < 64; ++k) {
    double av[4], bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) av[r] = L[(ar + ti + r) * LD + ac + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[c] = TB == 0 ? L[(br + tj + c) * LD + bc + k] : L[(br + k) * LD + bc + tj + c];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) out[r][c] += av[r] * bv[c];
  }
}

__global__ __launch_bounds__(256) void k_potrf_inv128w(double* Mkk, int ld, double* inv, int row0, int* info) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* L = sm;
  constexpr int LD = kDB + 1;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int ti = (tid >> 4) * 4, tj = (tid & 15) * 4;
  for (int e = tid; e < kDB * kDB; e += 256) {
    const int r = e >> 7, c = e & 127;
    L[r * LD + c] = (c <= r) ? Mkk[(size_t)r * ld + c] : 0.0;
  }
  __syncthreads();
  double t[4][4];
  // L11, then X11 in its place (L11 goes to global first)
  if (wave == 0) {
    double a[64];
    wave_potrf64(L, LD, 0, row0, info, a);
    const int lane = tid;
    for (int c = 0; c <= lane; ++c) Mkk[(size_t)lane * ld + c] = L[lane * LD + c];
    wave_trtri64(L, LD, 0, a);
  }
  __syncthreads();
  // L21 = A21 X11'   (X11 lower: terms with k > column vanish because X11's upper part is stored as zeros)
  tile64<0>(L, LD, 64, 0, 0, 0, t);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      L[(64 + ti + r) * LD + tj + c] = t[r][c];
      Mkk[(size_t)(64 + ti + r) * ld + tj + c] = t[r][c];
    }
  __syncthreads();
  // A22 -= L21 L21'
  tile64<0>(L, LD, 64, 0, 64, 0, t);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(64 + ti + r) * LD + 64 + tj + c] -= t[r][c];
  __syncthreads();
  // L22, X22
  if (wave == 0) {
    double a[64];
    wave_potrf64(L, LD, 64, row0, info, a);
    const int lane = tid;
    for (int c = 0; c <= lane; ++c) Mkk[(size_t)(64 + lane) * ld + 64 + c] = L[(64 + lane) * LD + 64 + c];
    wave_trtri64(L, LD, 64, a);
  }
  __syncthreads();
  // X21 = -X22 (L21 X11)
  tile64<1>(L, LD, 64, 0, 0, 0, t);   // T = L21 X11
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(64 + ti + r) * LD + tj + c] = t[r][c];
  __syncthreads();
  tile64<1>(L, LD, 64, 64, 64, 0, t);  // X22 T
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) L[(64 + ti + r) * LD + tj + c] = -t[r][c];
  __syncthreads();
  for (int e = tid; e < kDB * kDB; e += 256) {
    const int r = e >> 7, c = e & 127;
    inv[(size_t)r * kDB + c] = (c <= r) ? L[r * LD + c] : 0.0;
  }
}

// y (len rows) = A (rows x cols, lda) x, for NR right-hand sides interleaved [..][NR]; one wave per row.
template <int NR>
__global__ __launch_bounds__(256) void k_dense_gemv(const double* __restrict__ A, int lda, int rows, int cols,
                                                    const double* __restrict__ x, double alpha, const double* yin,
                                                    double beta, double* y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  double acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = 0.0;
  const double* a = A + (size_t)row * lda;
  for (int c = lane; c < cols; c += 64) {
    const double v = a[c];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] += v * x[(size_t)c * NR + r];
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[r] += __shfl_down(acc[r], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
      y[(size_t)row * NR + r] = alpha * acc[r] + (beta != 0.0 ? beta * yin[(size_t)row * NR + r] : 0.0);
  }
}

// part[chunk][c][NR] = sum over the chunk's rows of A[i][c] x[i][NR]   (A' x in two deterministic stages: thread per
// column, coalesced across columns; blockIdx.y splits the rows so that the whole chip streams A)
template <int NR>
__global__ __launch_bounds__(256) void k_dense_gemvt_part(const double* __restrict__ A, int lda, int rows, int cols,
                                                          const double* __restrict__ x, double* part, int rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int i0 = blockIdx.y * rows_per_chunk, i1 = min(rows, i0 + rows_per_chunk);
  double acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = 0.0;
  for (int i = i0; i < i1; ++i) {
    const double v = A[(size_t)i * lda + c];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] += v * x[(size_t)i * NR + r];
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) part[((size_t)blockIdx.y * cols + c) * NR + r] = acc[r];
}

// out0[c] = a0[c] - sum_chunks part[.][c][0];  out1[c] = (a1 ? a1[c] : 0) - sum_chunks part[.][c][1]
__global__ __launch_bounds__(256) void k_dense_finish_p(const double* __restrict__ part, int nchunk, int cols, int n,
                                                        const double* a0, const double* a1, double* out0, double* out1) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double s0 = 0.0, s1 = 0.0;
  for (int k = 0; k < nchunk; ++k) {
    s0 += part[((size_t)k * cols + c) * 2];
    s1 += part[((size_t)k * cols + c) * 2 + 1];
  }
  out0[c] = a0[c] - s0;
  out1[c] = (a1 ? a1[c] : 0.0) - s1;
}

// out[i][0] = sa * a[i], out[i][1] = sb * b[i] for i < len, zero on the padding
__global__ __launch_bounds__(256) void k_dense_pack2(const double* a, double sa, const double* b, double sb, double* out,
                                                     int len, int lenpad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= lenpad) return;
  out[(size_t)i * 2] = (i < len && a) ? sa * a[i] : 0.0;
  out[(size_t)i * 2 + 1] = (i < len && b) ? sb * b[i] : 0.0;
}

// out0[i] = in[i][0], out1[i] = in[i][1]
__global__ __launch_bounds__(256) void k_dense_unpack2(const double* in, double* out0, double* out1, int len) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  out0[i] = in[(size_t)i * 2];
  out1[i] = in[(size_t)i * 2 + 1];
}

// Blocked triangular solves with the Cholesky factor (2 interleaved right-hand sides).
// forward step k:  y_k = Linv_kk r_k ;  r_i -= L_ik y_k (i > k).     backward step k:  q_k = Linv_kk' y_k ; y_i -= L_ki' q_k (i < k)
// One launch per step, one workgroup per 128-row block still to be updated plus one that stores the solved block.
// Every workgroup first recomputes the (tiny) diagonal solve of block k redundantly into LDS -- block k of `r` is
// only READ in this launch (the solved values go to `out`), so there is no race.
template <bool FORWARD>
__global__ __launch_bounds__(256) void k_trsv_step(const double* __restrict__ Lm, int ld, const double* __restrict__ invs,
                                                   double* r, double* out, int k) {
  __shared__ double yk[kDB * 2];
  const int tid = threadIdx.x;
  const int blk = FORWARD ? k + (int)blockIdx.x : (int)blockIdx.x;  // forward: blocks k..nb-1, backward: blocks 0..k
  const double* inv = invs + (size_t)k * kDB * kDB;
  const int i = tid >> 1, rr = tid & 1;
  {
    double s = 0.0;
    if (FORWARD) {
      for (int p = 0; p <= i; ++p) s += inv[(size_t)i * kDB + p] * r[(size_t)(k * kDB + p) * 2 + rr];
    } else {
      for (int p = i; p < kDB; ++p) s += inv[(size_t)p * kDB + i] * r[(size_t)(k * kDB + p) * 2 + rr];
    }
    yk[i * 2 + rr] = s;
  }
  __syncthreads();
  if (blk == k) {
    out[(size_t)(k * kDB) * 2 + tid] = yk[tid];
    return;
  }
  double s = 0.0;
  if (FORWARD) {
    const double* Lb = Lm + (size_t)(blk * kDB + i) * ld + k * kDB;
    for (int p = 0; p < kDB; ++p) s += Lb[p] * yk[p * 2 + rr];
  } else {
    const double* Lb = Lm + (size_t)(k * kDB) * ld + blk * kDB + i;
    for (int p = 0; p < kDB; ++p) s += Lb[(size_t)p * ld] * yk[p * 2 + rr];
  }
  r[(size_t)(blk * kDB + i) * 2 + rr] -= s;
}

}  // namespace fpsq

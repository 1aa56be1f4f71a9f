// fpsq_dense.hip.h -- dense-block Jacobian variant: normal equations M = A A' + delta I on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64), blocked Cholesky, blocked triangular solves.  This is the direct back-end of the seam for
// small / dense problems (reference: LDLtSolver path, src/solve_linear_system.jl:206-252 and the dense
// A A' + tau I contraction of src/model-Fletcherpenaltynlp.jl:478-484); MFMA is used only here.
//
// All matrices are row-major fp64, padded with zeros to multiples of kDB = 128 (rows of A, order of M) and 32 (columns
// of A: whole k-stages of the Gram product); the padded diagonal of M is set to 1 so the factorisation is unaffected.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpsq {

constexpr int kDB = 128;      // block size of the Cholesky / GEMM tiles
using f64x4 = __attribute__((ext_vector_type(4))) double;
using f64x2 = __attribute__((ext_vector_type(2))) double;  // (HIP's double2 is a struct: arrays of it stay in scratch)

// Tile addressing of k_gemm128_lds on a BLOCK-BANDED matrix (fpsq_band): 128 x 128 blocks stored contiguously (row
// stride 128), tile (bi, bj) of C at C + bi * ci + bj * cj, tile bi of A at A + bi * a, tile bj of B at B + bj * b.
struct BlockStrides {
  int on = 0;
  size_t a = 0, b = 0, ci = 0, cj = 0;
};

// C (M x N, ldc) = alpha * A (M x K, lda) * B (N x K, ldb)' + beta * C on the fp64 matrix cores.  M, N multiples of 128, K (or
// the k-chunk of a slice, when gridDim.z > 1: slice z writes the plane C + z * zstride) a multiple of 32.
// One workgroup = one 128 x 128 tile of C on SIXTEEN waves, each a 32 x 32 sub-tile (2 x 2 MFMA tiles).  LOWER: only tiles with
// blockIdx.y >= blockIdx.x are computed (symmetric rank-k update of the lower triangle: the Gram product M = A A').
// Fragment maps of v_mfma_f64_16x16x4_f64 (cdna_hip_programming.md section 3): lane l holds A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15]; D register r of lane l is D[row = (l >> 4) + 4 r][col = l & 15].
// Why sixteen waves (tools/mfma_probe.hip): one wave issues a v_mfma_f64_16x16x4_f64 only every ~140 cycles (196 when it
// depends on the previous one), whatever the number of independent accumulators; a SIMD reaches its rate only with several
// waves resident (2 per SIMD: 99 cycles per MFMA).  The four-wave kernel of rounds 1-2 (one wave per SIMD, 64 x 64 per wave,
// removed in round 3) took 0.77 ms for the m = 2048 Gram matrix where this one takes 0.55.
// LDS tiles are [row][k] with leading dimension KD + 1 doubles: the banks are 4 bytes wide and a ds_read_b64 is served 16
// lanes at a time, so the 16 rows of an MFMA operand must start 2 banks apart (leading dimension = 1 mod 16) to cover the
// 32 banks once -- KD + 2, two 8-byte banks apart, measured 50 % conflict cycles.  KD = 32 per stage halves the barriers
// of sixteen waves.
constexpr int kW16Kd = 32, kW16Ld = kW16Kd + 1;
constexpr int kW16Lds = 2 * 2 * kDB * kW16Ld * 8;
template <bool LOWER>
__global__ __launch_bounds__(1024) void k_gemm_nt_f64_w16(double* C, int ldc, const double* __restrict__ A, int lda,
                                                          const double* __restrict__ B, int ldb, int K, double alpha,
                                                          double beta, int kchunk = 0, size_t zstride = 0) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (LOWER && bi < bj) return;
  if (kchunk > 0) {
    const int kbeg = (int)blockIdx.z * kchunk;
    K = min(K, kbeg + kchunk) - kbeg;
    if (K < 0) K = 0;
    C += (size_t)blockIdx.z * zstride;
    A += kbeg;
    B += kbeg;
  }
  extern __shared__ __attribute__((aligned(16))) double gsm[];
  constexpr int KD = kW16Kd, LD = kW16Ld;
  double* sA = gsm;                  // [2][128 * LD]
  double* sB = gsm + 2 * kDB * LD;   // [2][128 * LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 2) * 32, wc = (wave & 3) * 32;
  const double* Ab = A + (size_t)bi * kDB * lda;
  const double* Bb = B + (size_t)bj * kDB * ldb;
  const int srow = tid >> 3, sk = (tid & 7) * 4;  // staging: four consecutive k of row (t >> 3) for both operands
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  f64x2 ra0, ra1, rb0, rb1;
  auto gload = [&](int k0) {
    const f64x2* pa = reinterpret_cast<const f64x2*>(Ab + (size_t)srow * lda + k0 + sk);
    const f64x2* pb = reinterpret_cast<const f64x2*>(Bb + (size_t)srow * ldb + k0 + sk);
    ra0 = pa[0];
    ra1 = pa[1];
    rb0 = pb[0];
    rb1 = pb[1];
  };
  auto lstore = [&](int buf) {
    double* qa = sA + buf * kDB * LD + srow * LD + sk;
    double* qb = sB + buf * kDB * LD + srow * LD + sk;
    qa[0] = ra0[0];
    qa[1] = ra0[1];
    qa[2] = ra1[0];
    qa[3] = ra1[1];
    qb[0] = rb0[0];
    qb[1] = rb0[1];
    qb[2] = rb1[0];
    qb[3] = rb1[1];
  };
  const int fr = lane & 15, fk = lane >> 4;
  int buf = 0;
  if (K > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += KD) {
    const bool more = k0 + KD < K;
    if (more) gload(k0 + KD);
    const double* pa = sA + buf * kDB * LD + (wr + fr) * LD + fk;
    const double* pb = sB + buf * kDB * LD + (wc + fr) * LD + fk;
#pragma unroll
    for (int ks = 0; ks < KD / 4; ++ks) {
      const double a0 = pa[4 * ks], a1 = pa[16 * LD + 4 * ks], b0 = pb[4 * ks], b1 = pb[16 * LD + 4 * ks];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  double* Cb = C + (size_t)(bi * kDB + wr) * ldc + bj * kDB + wc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* p = Cb + (size_t)(i * 16 + fk + 4 * r) * ldc + j * 16 + fr;
        const double v = alpha * acc[i][j][r];
        *p = (beta != 0.0) ? v + beta * *p : v;
      }
}

// The K = 128 products of a factorisation step in ONE memory round trip (the default for the panel and the trailing
// update).  A step of the blocked Cholesky is a link of a dependent chain (nb of them dense, m / 128 banded), and the
// staged kernels above pay eight load -> LDS -> barrier round trips for a 128-deep product (27 us for the panel, ~16 us
// for the update, against 2-3 us of matrix-core time per workgroup).  Here a workgroup issues every global load of its
// operand tiles at once -- whole 1 KB rows per wave instruction, 16 bytes per lane --, parks the tiles in LDS row-major
// with leading dimension = 1 mod 16 doubles (129 / 65: the 16 rows of an MFMA operand start two 4-byte banks apart) and
// runs the k-steps from there, SIXTEEN waves of one 16 x 16 tile each (one wave issues an fp64 MFMA only every
// ~140-196 cycles, tools/mfma_probe.hip: four waves of 2 x 2 tiles took 12.5 / 8.7 us per launch).
// (Loading the MFMA fragments straight from global memory, 8 bytes per lane in 32-byte runs, was measured first:
// 18 / 22 us per launch -- the address unit serialises such loads.)
//   MODE 0: trailing update, tile (bi, bj) of 64 x 64, bi >= bj:  C -= A_bi B_bj'      grid (2 rem, 2 rem), waves 4 x 4
//   MODE 1: panel IN PLACE, rows [32 bi, 32 bi + 32):  P <- P X'  (B = X = the 128 x 128 inverse block, lower
//           triangular: a wave sums only the k <= column part); waves 2 (row tiles) x 8 (column tiles); X goes through
//           LDS in two k-halves (the second only for columns >= 64).
constexpr int kG128Ld = 129, kG128LdX = 65;
constexpr int kG128Lds0 = 2 * 64 * kG128Ld * 8;
constexpr int kG128Lds1 = (32 * kG128Ld + 128 * kG128LdX) * 8;
template <int MODE>
__global__ __launch_bounds__(1024) void k_gemm128_lds(double* C, int ldc, const double* A, int lda, const double* B,
                                                      int ldb, BlockStrides bs) {
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (MODE == 0 && bi < bj) return;
  extern __shared__ __attribute__((aligned(16))) double gsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  constexpr int LD = kG128Ld;
  if (MODE == 0) {
    double* sA = gsm;
    double* sB = gsm + 64 * LD;
    const int wr = (wave >> 2) * 16, wc = (wave & 3) * 16;
    const int Ib = bi >> 1, Jb = bj >> 1, si = (bi & 1) * 64, sj = (bj & 1) * 64;
    const double* Ab = A + (bs.on ? (size_t)Ib * bs.a : (size_t)Ib * kDB * lda) + (size_t)si * lda;
    const double* Bb = B + (bs.on ? (size_t)Jb * bs.b : (size_t)Jb * kDB * ldb) + (size_t)sj * ldb;
    double* Cb = C + (bs.on ? (size_t)Ib * bs.ci + (size_t)Jb * bs.cj : (size_t)Ib * kDB * ldc + (size_t)Jb * kDB) +
                 (size_t)(si + wr) * ldc + sj + wc;
    f64x2 va[4], vb[4];
    double cold[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      va[u] = *reinterpret_cast<const f64x2*>(Ab + (size_t)(u * 16 + wave) * lda + 2 * lane);
      vb[u] = *reinterpret_cast<const f64x2*>(Bb + (size_t)(u * 16 + wave) * ldb + 2 * lane);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) cold[r] = Cb[(size_t)(fk + 4 * r) * ldc + fr];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double* qa = sA + (u * 16 + wave) * LD + 2 * lane;
      double* qb = sB + (u * 16 + wave) * LD + 2 * lane;
      qa[0] = va[u][0];
      qa[1] = va[u][1];
      qb[0] = vb[u][0];
      qb[1] = vb[u][1];
    }
    __syncthreads();
    f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const double* ar = sA + (wr + fr) * LD + fk;
    const double* br = sB + (wc + fr) * LD + fk;
#pragma unroll 4
    for (int s = 0; s < 32; s += 2) {
      const double a0 = ar[4 * s], b0 = br[4 * s], a1 = ar[4 * s + 4], b1 = br[4 * s + 4];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Cb[(size_t)(fk + 4 * r) * ldc + fr] = cold[r] - (acc0[r] + acc1[r]);
  } else {
    constexpr int LX = kG128LdX;
    double* sA = gsm;
    double* sX = gsm + 32 * LD;
    const int Ib = bi >> 2, si = (bi & 3) * 32;
    const double* Ab = A + (bs.on ? (size_t)Ib * bs.a : (size_t)Ib * kDB * lda) + (size_t)si * lda;
    double* Cb = C + (bs.on ? (size_t)Ib * bs.ci : (size_t)Ib * kDB * ldc) + (size_t)si * ldc;
    f64x2 va[2], vx[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) va[u] = *reinterpret_cast<const f64x2*>(Ab + (size_t)(u * 16 + wave) * lda + 2 * lane);
#pragma unroll
    for (int u = 0; u < 8; ++u) vx[u] = *reinterpret_cast<const f64x2*>(B + (size_t)(u * 16 + wave) * ldb + 2 * lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      double* qa = sA + (u * 16 + wave) * LD + 2 * lane;
      qa[0] = va[u][0];
      qa[1] = va[u][1];
    }
    if (lane < 32) {  // first k-half of X, all rows
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        double* qx = sX + (u * 16 + wave) * LX + 2 * lane;
        qx[0] = vx[u][0];
        qx[1] = vx[u][1];
      }
    }
    __syncthreads();  // (every wave has read its rows of the panel: the stores below cannot overtake a load)
    const int ri = wave & 1, c = wave >> 1;  // row tile, column tile
    f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const double* ar = sA + (16 * ri + fr) * LD + fk;
    {
      const double* xr = sX + (16 * c + fr) * LX + fk;
      const int lim = min(16, 4 * (c + 1));  // (a multiple of 4) X[j][k] = 0 for k > j
      for (int s = 0; s < lim; s += 2) {
        const double a0 = ar[4 * s], b0 = xr[4 * s], a1 = ar[4 * s + 4], b1 = xr[4 * s + 4];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
      }
    }
    __syncthreads();
    if (lane >= 32) {  // second k-half of X, rows j >= 64 only, row j - 64 of the buffer
#pragma unroll
      for (int u = 4; u < 8; ++u) {
        double* qx = sX + (u * 16 + wave - 64) * LX + 2 * lane - 64;
        qx[0] = vx[u][0];
        qx[1] = vx[u][1];
      }
    }
    __syncthreads();
    if (c >= 4) {
      const double* xr = sX + (16 * c - 64 + fr) * LX + fk;
      const int lim = 4 * (c + 1) - 16;
      for (int s = 0; s < lim; s += 2) {
        const double a0 = ar[64 + 4 * s], b0 = xr[4 * s], a1 = ar[64 + 4 * s + 4], b1 = xr[4 * s + 4];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Cb[(size_t)(16 * ri + fk + 4 * r) * ldc + 16 * c + fr] = acc0[r] + acc1[r];
  }
}

// M[i][i] += delta for i < m; M[i][i] = 1 on the padding
__global__ void k_dense_diag(double* M, int ld, int m, int mpad, double delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < mpad) M[(size_t)i * ld + i] = i < m ? M[(size_t)i * ld + i] + delta : 1.0;
}

// ---- Cholesky of ONE 128 x 128 diagonal block AND the inverse of its factor (one workgroup; the serial heart of the blocked
// factorisation).  One generation is left in the source, the fifth (k_potrf_inv128m, 44 us per block); the others are in
// the git history of rounds 1-2: (1) unblocked in LDS, three 16-wave barriers per column, 274 us; (2, 3) 64 x 64 / 32 x 32
// sub-blocks factored by ONE wave with the rows in registers, ~220 us whatever their arithmetic -- thousands of straight-line
// instructions executed once per call; (4) a ROLLED loop over eight 16-column panels whose only unrolled part is a 16 x 16
// factor routine on v_readlane broadcasts, left-looking panel updates and row substitutions on the LDS copy, X = L^-1 by
// doubling: 104 us.
__device__ __forceinline__ double rdlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

#ifdef FPSQ_POTRF_TIMING  // tools/potrf_probe.hip: s_memtime stamps of thread 0 after every phase
#define POTRF_STAMP() do { if (threadIdx.x == 0 && stamps) stamps[nst++] = (long long)__builtin_readcyclecounter(); } while (0)
#define POTRF_TIMING_ARG , long long* stamps
#else
#define POTRF_STAMP() do {} while (0)
#define POTRF_TIMING_ARG
#endif
// ---- the fifth generation: generation 4's scheme with its two GEMM-shaped parts on the matrix cores.  The phase
// probe of generation 4 (profiles/r02_potrf_phase_probe.txt, cycles of 276k): left-looking panel updates 55k (LDS
// bandwidth: 6 reads per 8 FMAs), the 16 x 16 factor routine 8 x 8.1k, row substitutions 8 x 2.8k, doubling inverse 84k,
// load / stores 44k.  Here
//   (a) the panel update is v_mfma_f64_16x16x4_f64 on 16 x 16 tiles read straight from the LDS copy (leading dimension
//       130: the 16 rows x 4 k of an operand fall in distinct banks); wave 0 updates the diagonal tile and goes on to
//       factor it while waves 1-3 update the tiles below -- their work hides behind the serial 16 x 16 routine;
//   (d) the doubling steps T = L21 X11 and X21 = -X22 T are MFMA tile products too (T kept transposed, 32 columns at a
//       time, so both operands of both products are read k-contiguous); entries of the triangular 16 x 16 diagonal
//       sub-blocks of X are selected on load (strictly lower from the transposed store, diagonal from `dinv`, else 0).
// Step (b) also yields the 16 x 16 inverse (see wave_diag16), which turns (c) into an MFMA product as well.
#ifndef FPSQ_POTRF_LD5
#define FPSQ_POTRF_LD5 (kDB + 2)
#endif
constexpr int kPotrfLd5 = FPSQ_POTRF_LD5;
constexpr int kPotrfTld5 = 66;
constexpr int kPotrfLds5 = (kDB * kPotrfLd5 + 32 * kPotrfTld5 + kDB) * 8;

// the 16 x 16 factor routine (one wave, the serial heart of the kernel: 8 x 7.1k of its ~100k cycles).  Lane r < 16 holds
// row r of the tile in registers and column j is eliminated with v_readlane broadcasts of L[c][j].  Lanes 16 .. 31
// compute X16 = L16^-1 ALONGSIDE, for free: lane 16 + c carries column c of X through the same instruction stream (its
// a[r] starts as e_c; at step j its a[j] * rp is X[j][c], and `a[r] -= X[j][c] * L[r][j]` is the same fused multiply-add
// with the same broadcast L[r][j] the factor lanes use).  X16 goes, transposed, to the upper triangle of the tile (where
// the doubling steps expect it) and lets step (c) be a matrix-core product.
// The routine is ISSUE bound, not latency bound (tools/issue_probe.hip, one wave, counter units: an fp64 FMA 6.4, a
// v_readlane_b32 4 in a batch but 8 when the FMA that consumes it follows at once, rsqrt(double) ~100 for ten dependent
// instructions; eliminating TWO columns per link of the dependent chain -- 1 / l22 = rsqrt(a c - b^2) l11, two independent
// reciprocal square roots -- was built and measured: 8.1k per tile against 7.7k).  So it carries few instructions:
//   * no row selects: the registers of a factor lane above its diagonal hold values nobody reads (lane c is read only
//     for columns < c, and only the lower triangle is stored), and the diagonal lane's own a[j] * rp IS the pivot;
//   * a vanishing pivot is a (uniform, rare) branch instead of selects on every column;
//   * the reciprocal square root is v_rsq_f64 + one third-order correction without the special-value tests (d > tol >= 0
//     is finite here);  1 / L[j][j] is the diagonal of the 16 x 16 inverse the lanes 16 .. 31 carry;
//   * the readlanes of a column's updates are issued as a batch ahead of its FMAs, behind the update of column j + 1 and
//     the next pivot's broadcast.
// 7.7k -> 7.1k per tile against the select-based form of round 2 (~900 instructions -> ~760).  Also tried, slower: the tile
// spread over all 64 lanes, 4 columns each, with ds_bpermute fetches (7.0k against the 6.6k of its time); an unnormalised
// elimination (reciprocal square roots at the end: 9.4k); one LDS store of rp by all lanes (same address: 7.7k).
__device__ __forceinline__ double rsqrt_pos(double d) {
  const double y0 = __builtin_amdgcn_rsq(d);  // ~2^-23 relative
  const double e = fma(-(d * y0), y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);  // y0 (1 + e / 2 + 3 e^2 / 8): ~e^3
}
__device__ __forceinline__ void wave_diag16(double* L, int LD, int o, int row0, int* info, double tol, double reg,
                                             double* dinv) {
  const int lane = threadIdx.x & 63;
  const int rl = lane & 15;
  const bool inv = (lane >> 4) == 1;
  double a[16];
  {
    const f64x2* row = reinterpret_cast<const f64x2*>(L + (o + rl) * LD + o);  // (16-byte aligned: LD and o are even)
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      const f64x2 v = row[c / 2];
      a[c] = inv ? (c == rl ? 1.0 : 0.0) : v[0];
      a[c + 1] = inv ? (c + 1 == rl ? 1.0 : 0.0) : v[1];
    }
  }
  const bool dyn = reg > 0.0;
  const double thr = dyn ? tol : 0.0, sub = dyn ? reg : 1.0;  // (a unit pivot keeps the kernel finite when none is set)
  int nbad = 0, first = 0;
  double d = rdlane(a[0], 0);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (__builtin_expect(!(d > thr), 0)) {
      first = nbad == 0 ? j + 1 : first;
      ++nbad;
      d = sub;
      if (!inv && rl == j) a[j] = sub;
    }
    const double rp = rsqrt_pos(d);
    const double l = a[j] * rp;
    a[j] = l;
    if (j < 15) {
      a[j + 1] = fma(-l, rdlane(l, j + 1), a[j + 1]);
      d = rdlane(a[j + 1], j + 1);
      double sc[16];
#pragma unroll
      for (int c = j + 2; c < 16; ++c) sc[c] = rdlane(l, c);
#pragma unroll
      for (int c = j + 2; c < 16; ++c) a[c] = fma(-l, sc[c], a[c]);
    }
  }
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c <= lane) L[(o + lane) * LD + o + c] = a[c];  // L16, lower
  } else if (inv) {
    // X16(r, rl), r > rl, transposed into the upper triangle; X16(rl, rl) = 1 * rp_rl, bit for bit, is 1 / L[rl][rl]
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (r >= rl) *(r == rl ? dinv + o + rl : L + (o + rl) * LD + o + r) = a[r];
  }
  if (lane == 0 && nbad) {
    if (dyn)
      atomicAdd(info + 1, nbad);
    else
      atomicCAS(info, 0, row0 + o + first);
  }
}

// inv / invT are written in their non-zero triangles only: the caller zero-fills both buffers ONCE (at allocation).
// 512 threads: a wave issues an fp64 MFMA only every ~140-196 cycles (tools/mfma_probe.hip), so the MFMA phases want
// more than one wave per SIMD.
constexpr int kPotrfThreads5 = 512;
__global__ __launch_bounds__(kPotrfThreads5) void k_potrf_inv128m(double* Mkk, int ld, double* inv, double* invT, int row0,
                                                                  int* info, double tol, double reg POTRF_TIMING_ARG) {
#ifdef FPSQ_POTRF_TIMING
  int nst = 0;
#endif
  POTRF_STAMP();
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* L = sm;
  constexpr int LD = kPotrfLd5;
  constexpr int NW = kPotrfThreads5 / 64;
  double* Tt = sm + kDB * LD;  // Tt[c][row]: 32 columns x 64 rows of the doubling steps' T, transposed
  constexpr int TLD = kPotrfTld5;
  double* dinv = Tt + 32 * TLD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fk = lane >> 4;
  const int tr = (tid & 255) >> 4, tc = tid & 15, th = tid >> 8;  // element of a 16 x 16 tile; tiles 2 u + th
  // the 36 lower tiles, every load in flight at once (one HBM round trip for the block)
  {
    double v[18];
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      int t0 = 2 * u, ti0 = 0;
      while ((ti0 + 1) * (ti0 + 2) / 2 <= t0) ++ti0;
      const int tj0 = t0 - ti0 * (ti0 + 1) / 2;
      int t1 = 2 * u + 1, ti1 = 0;
      while ((ti1 + 1) * (ti1 + 2) / 2 <= t1) ++ti1;
      const int tj1 = t1 - ti1 * (ti1 + 1) / 2;
      const int ti = th ? ti1 : ti0, tj = th ? tj1 : tj0;
      v[u] = Mkk[(size_t)(16 * ti + tr) * ld + 16 * tj + tc];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      int t0 = 2 * u, ti0 = 0;
      while ((ti0 + 1) * (ti0 + 2) / 2 <= t0) ++ti0;
      const int tj0 = t0 - ti0 * (ti0 + 1) / 2;
      int t1 = 2 * u + 1, ti1 = 0;
      while ((ti1 + 1) * (ti1 + 2) / 2 <= t1) ++ti1;
      const int tj1 = t1 - ti1 * (ti1 + 1) / 2;
      const int ti = th ? ti1 : ti0, tj = th ? tj1 : tj0;
      L[(16 * ti + tr) * LD + 16 * tj + tc] = v[u];
    }
  }
  __syncthreads();
  POTRF_STAMP();
  // C(rows of tile t, columns cb) -= L[rows, k0 .. k1) L[cb rows, k0 .. k1)'
  auto tile_update = [&](int t, int cb, int k0, int k1) {
    f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    const double* ar = L + (16 * t + fr) * LD + fk;   // A[i = fr][k = fk]
    const double* br = L + (16 * cb + fr) * LD + fk;  // B[k = fk][j = fr] = L[16 cb + j][k]
    for (int k = k0; k < k1; k += 8) {
      const double a0 = ar[k], b0 = br[k], a1 = ar[k + 4], b1 = br[k + 4];
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) L[(16 * t + fk + 4 * r) * LD + 16 * cb + fr] -= acc0[r] + acc1[r];
  };
#pragma unroll 1
  for (int pb = 0; pb < 8; ++pb) {
    const int o = pb * 16;
    if (pb > 0) {  // (a)
      // wave 0 runs the serial part; wave 4 shares its SIMD and stays out of its way (with it busy the 16 x 16 routine
      // took 8.0k instead of 6.6k cycles); the other six waves update the tiles below
      if (wave == 0) {
        tile_update(pb, pb, o - 16, o);  // the diagonal tile: earlier panels were applied one iteration ago (below)
      } else if (wave != 4) {
        const int wi = wave < 4 ? wave - 1 : wave - 2;  // 0 .. 5
        if (wi == 5 && pb < 7) tile_update(pb + 1, pb + 1, 0, o);  // next diagonal tile, the panels before this one
        for (int t = pb + 1 + wi; t < 8; t += 6) tile_update(t, pb, 0, o);
      }
    }
    if (wave == 0) wave_diag16(L, LD, o, row0, info, tol, reg, dinv);  // (b)
    __syncthreads();
    POTRF_STAMP();
    // (c) tiles below: P <- P X16' on the matrix cores.  B[k][j] = X16(j, k): strictly lower entries from the transposed
    // store, the diagonal from dinv, zero above
    for (int t = pb + 1 + wave; t < 8; t += NW) {
      double av[4], bv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 4 * q + fk;
        av[q] = L[(16 * t + fr) * LD + o + k];
        const double xv = L[(o + k) * LD + o + fr];
        bv[q] = fr > k ? xv : (fr == k ? dinv[o + fr] : 0.0);
      }
      f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) L[(16 * t + fk + 4 * r) * LD + o + fr] = acc0[r] + acc1[r];
    }
    __syncthreads();
    POTRF_STAMP();
  }
#pragma unroll
  for (int u = 0; u < 18; ++u) {
    int t0 = 2 * u, ti0 = 0;
    while ((ti0 + 1) * (ti0 + 2) / 2 <= t0) ++ti0;
    const int tj0 = t0 - ti0 * (ti0 + 1) / 2;
    int t1 = 2 * u + 1, ti1 = 0;
    while ((ti1 + 1) * (ti1 + 2) / 2 <= t1) ++ti1;
    const int tj1 = t1 - ti1 * (ti1 + 1) / 2;
    const int ti = th ? ti1 : ti0, tj = th ? tj1 : tj0;
    if (ti != tj || tc <= tr) Mkk[(size_t)(16 * ti + tr) * ld + 16 * tj + tc] = L[(16 * ti + tr) * LD + 16 * tj + tc];
  }
  POTRF_STAMP();
  // (d) X = L^-1 by doubling; X(r, c), r > c, lives at L[c * LD + r] (the 16 x 16 diagonal inverses are there already)
#pragma unroll 1
  for (int h = 16; h < kDB; h *= 2) {
    const int w = h < 32 ? h : 32;
    const int ntile = 4 * (w / 16);  // 64 rows (all pairs of the level) x w columns of T in 16 x 16 tiles
#pragma unroll 1
    for (int cc = 0; cc < h; cc += w) {
      for (int tl = wave; tl < ntile; tl += NW) {  // T[q h + r][c] = sum_{p >= c} L21[r][p] X11(p, c)
        const int gr0 = (tl & 3) * 16, ct = tl >> 2;
        const int q = gr0 / h, r0 = gr0 % h, b0 = q * 2 * h, c0 = cc + ct * 16;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        const double* arow = L + (b0 + h + r0 + fr) * LD + b0;  // A[i][p] = L21[r0 + i][p]
        const double* bcol = L + (b0 + c0 + fr) * LD + b0;      // B[p][j] = X11(p, c0 + j)
        const double dj = dinv[b0 + c0 + fr];
#pragma unroll
        for (int p0 = 0; p0 < 16; p0 += 4) {
          const int p = c0 + p0 + fk, c = c0 + fr;
          const double a = arow[p];
          const double xv = bcol[p];
          const double b = p > c ? xv : (p == c ? dj : 0.0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
        f64x4 acc2 = {0.0, 0.0, 0.0, 0.0};
        for (int p0 = c0 + 16; p0 < h; p0 += 16) {  // (h - c0 is a multiple of 16) operands of four k-steps, then the MFMAs
          double av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            av[u] = arow[p0 + 4 * u + fk];
            bv[u] = bcol[p0 + 4 * u + fk];
          }
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Tt[(ct * 16 + fr) * TLD + gr0 + fk + 4 * r] = acc[r] + acc2[r];
      }
      __syncthreads();
      POTRF_STAMP();
      for (int tl = wave; tl < ntile; tl += NW) {  // X21[r][c] = - sum_{p <= r} X22(r, p) T[p][c]
        const int gr0 = (tl & 3) * 16, ct = tl >> 2;
        const int q = gr0 / h, r0 = gr0 % h, b0 = q * 2 * h;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        const double* xcol = L + (size_t)(b0 + h) * LD + b0 + h + r0 + fr;  // A[i][p] = X22(r0 + i, p) = xcol[p * LD], p < r0 + i
        const double* tb = Tt + (ct * 16 + fr) * TLD + q * h;               // B[p][j] = T[q h + p][ct 16 + j]
        f64x4 acc2 = {0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < r0; p0 += 16) {
          double av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            av[u] = xcol[(p0 + 4 * u + fk) * LD];
            bv[u] = tb[p0 + 4 * u + fk];
          }
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
        }
#pragma unroll
        for (int p0 = 0; p0 < 16; p0 += 4) {
          const int p = r0 + p0 + fk, ri = r0 + fr;
          const double xv = xcol[p * LD];
          const double a = ri > p ? xv : (ri == p ? dinv[b0 + h + p] : 0.0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tb[p], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) L[(b0 + cc + ct * 16 + fr) * LD + b0 + h + r0 + fk + 4 * r] = -(acc[r] + acc2[r]);
      }
      __syncthreads();
      POTRF_STAMP();
    }
  }
  // inv = X (lower), invT = X' (upper): tile (ti, tj), tj <= ti, of inv and its mirror image (tj, ti) of invT
#pragma unroll
  for (int u = 0; u < 18; ++u) {
    int t0 = 2 * u, ti0 = 0;
    while ((ti0 + 1) * (ti0 + 2) / 2 <= t0) ++ti0;
    const int tj0 = t0 - ti0 * (ti0 + 1) / 2;
    int t1 = 2 * u + 1, ti1 = 0;
    while ((ti1 + 1) * (ti1 + 2) / 2 <= t1) ++ti1;
    const int tj1 = t1 - ti1 * (ti1 + 1) / 2;
    const int ti = th ? ti1 : ti0, tj = th ? tj1 : tj0;
    const int r = 16 * ti + tr, c = 16 * tj + tc;  // element (r, c) of inv, r >= c except above a diagonal tile's diagonal
    const double xl = L[c * LD + r];               // X(r, c) for r > c
    const int r2 = 16 * tj + tr, c2 = 16 * ti + tc;  // element (r2, c2) of invT, c2 >= r2 except below the diagonal
    const double xu = L[r2 * LD + c2];               // X(c2, r2) for c2 > r2
    if (ti != tj) {
      inv[(size_t)r * kDB + c] = xl;
      invT[(size_t)r2 * kDB + c2] = xu;
    } else {
      inv[(size_t)r * kDB + c] = c < r ? xl : (c == r ? dinv[r] : 0.0);
      invT[(size_t)r2 * kDB + c2] = c2 > r2 ? xu : (c2 == r2 ? dinv[r2] : 0.0);
    }
  }
  POTRF_STAMP();
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
// the same with a row permutation: out{0,1}[perm[i]] = in[i][{0,1}]
__global__ __launch_bounds__(256) void k_unpack2_scatter(const double* __restrict__ in, const int32_t* __restrict__ perm,
                                                         double* out0, double* out1, int len) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  out0[perm[i]] = in[(size_t)i * 2];
  out1[perm[i]] = in[(size_t)i * 2 + 1];
}

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
// The same step organised for LATENCY (the default): a step is a chain link of the triangular solve -- nb (dense) or
// 2 m / 128 (band) of them run back to back, each with a handful of workgroups -- so what counts is the number of
// dependent memory round trips inside it.  Here every global load of the step (the 128 x 128 inverse block AND the
// workgroup's own off-diagonal block, 64 + 64 values per thread) is issued before the first use: one round trip.  The
// triangular half of the inverse that is identically zero is skipped by whole waves.  Forward updates reduce their 64
// (row, right-hand side) partial products per wave with a transposing butterfly (63 shuffles instead of 384; lane l ends
// with the total of value l, stored coalesced); backward updates read the block by columns and need none.
template <bool FORWARD>
__global__ __launch_bounds__(256) void k_trsv_step3(const double* __restrict__ Lm, int ld, const double* __restrict__ inv,
                                                    const double* __restrict__ invT, double* r, double* out, int k,
                                                    int band_w = 0, int bstride = 1) {
  __shared__ double rk[kDB * 2];
  __shared__ double part[2][kDB * 2];
  __shared__ double yk[kDB * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (bstride = 2: the blocks of k's own elimination chain, fpsq_band_create)
  const int blk = FORWARD ? k + bstride * (int)blockIdx.x
                          : (band_w > 0 ? k - bstride * (int)blockIdx.x : (int)blockIdx.x);
  const int i = tid & 127, hf = tid >> 7;
  // forward: y_i = sum_{p <= i} X'[p][i] r_p;   backward: q_i = sum_{p >= i} X[p][i] y_p   (p in this thread's half)
  const double* Xc = (FORWARD ? invT : inv) + (size_t)k * kDB * kDB + (size_t)(hf * 64) * kDB + i;
  const bool xskip = FORWARD ? (hf == 1 && i < 64) : (hf == 0 && i >= 64);  // (wave-uniform) all-zero part of the triangle
  double xs[64], lb[64];
  if (!xskip) {
#pragma unroll
    for (int q = 0; q < 64; ++q) xs[q] = Xc[(size_t)q * kDB];
  }
  const size_t lds = band_w > 0 ? (size_t)kDB : (size_t)ld;
  if (blk != k) {
    if (FORWARD) {  // block (blk, k), rows 32 wave .. + 31, lanes along the columns
      const double* Lb = (band_w > 0 ? Lm + ((size_t)blk * band_w + (k - blk + band_w - 1)) * kDB * kDB
                                     : Lm + (size_t)(blk * kDB) * ld + k * kDB) + (size_t)(wave * 32) * lds + lane;
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        lb[2 * u] = Lb[(size_t)u * lds];
        lb[2 * u + 1] = Lb[(size_t)u * lds + 64];
      }
    } else {  // block (k, blk) read by columns: column i, rows of this thread's half
      const double* Lb = (band_w > 0 ? Lm + ((size_t)k * band_w + (blk - k + band_w - 1)) * kDB * kDB
                                     : Lm + (size_t)(k * kDB) * ld + blk * kDB) + (size_t)(hf * 64) * lds + i;
#pragma unroll
      for (int q = 0; q < 64; ++q) lb[q] = Lb[(size_t)q * lds];
    }
  }
  rk[tid] = r[(size_t)(k * kDB) * 2 + tid];
  __syncthreads();
  {
    double s0 = 0.0, s1 = 0.0;
    if (!xskip) {
#pragma unroll
      for (int q = 0; q < 64; ++q) {
        s0 += xs[q] * rk[(hf * 64 + q) * 2];
        s1 += xs[q] * rk[(hf * 64 + q) * 2 + 1];
      }
    }
    part[hf][i * 2] = s0;
    part[hf][i * 2 + 1] = s1;
  }
  __syncthreads();
  yk[tid] = part[0][tid] + part[1][tid];
  __syncthreads();
  if (blk == k) {
    out[(size_t)(k * kDB) * 2 + tid] = yk[tid];
    return;
  }
  if (FORWARD) {
    const double y00 = yk[lane * 2], y01 = yk[lane * 2 + 1], y10 = yk[(lane + 64) * 2], y11 = yk[(lane + 64) * 2 + 1];
    double v[64];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      v[2 * u] = lb[2 * u] * y00 + lb[2 * u + 1] * y10;
      v[2 * u + 1] = lb[2 * u] * y01 + lb[2 * u + 1] * y11;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const bool hi = (lane & off) != 0;
#pragma unroll
      for (int idx = 0; idx < off; ++idx) {
        const double send = hi ? v[idx] : v[idx + off];
        const double keep = hi ? v[idx + off] : v[idx];
        v[idx] = keep + __shfl_xor(send, off, 64);
      }
    }
    r[(size_t)(blk * kDB + wave * 32) * 2 + lane] -= v[0];
  } else {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < 64; ++q) {
      s0 += lb[q] * yk[(hf * 64 + q) * 2];
      s1 += lb[q] * yk[(hf * 64 + q) * 2 + 1];
    }
    __syncthreads();
    part[hf][i * 2] = s0;
    part[hf][i * 2 + 1] = s1;
    __syncthreads();
    r[(size_t)(blk * kDB) * 2 + tid] -= part[0][tid] + part[1][tid];
  }
}

// ---- the whole sweep in ONE launch (the default; FPSQ_TRSV_CHAIN=0 selects the step kernels above).  A sweep is a chain of
// nb links and a launch per link costs ~3.5 us of dispatch before its single memory round trip starts (8.5 / 6.3 us per
// forward / backward step).  Here workgroup w owns block b (forward: b = w, backward: b = nb - 1 - w) and PULLS: for every
// coupled block j eliminated before b it takes the solved y_j from the publication buffer, subtracts L_bj y_j (forward) or
// L_jb' q_j (backward) from its own right-hand side -- kept in registers, thread t <-> entry t of the [128][2] block --, then
// solves with the diagonal inverse and publishes.  Publication as in the product kernels' leader records (fpsq_spmv.hip.h):
// every 8-byte word carries half a double and the launch number `seq`, written through and read with agent-scope atomics,
// so a reader that sees the number sees the payload: no flag, no fence, one round trip per look, and the look IS the
// fetch.  WHICH block a workgroup owns is decided by a TICKET it draws when it starts (one agent-scope atomic add; round 4),
// not by its index in the grid: dependencies point to lower tickets only, and a lower ticket is held by a workgroup that
// is already RUNNING -- whatever else shares the device.  (By grid index -- round 3 -- that only holds inside one kernel:
// workgroup i is dispatched by XCD i mod 8, in order within that XCD, so with a second sweep on the device -- another
// handle, stream or process -- XCD a can be full of kernel Y's waiting workgroups while X's lowest unfinished block is not
// yet dispatched there, and vice versa: the circular wait across kernels that the riding leaders of the product kernels
// ran into, fpsq_spmv.hip.h "WHO LEADS".  The ticket's round trip, ~1.5 us, is paid once per workgroup at its start, long
// before its turn in a 50-70 us sweep.)  Every wait is bounded all the same (kChainPolls looks, then the error word is
// raised and the workgroup goes on publishing, so nobody behind it waits in turn; an abort word behind the buffer, set with
// it and looked at before and during every wait, keeps the waits that are still to come short: a failed sweep ends after ONE
// waiting time, not one per link; the call fails with FPSQ_ERR_TIMEOUT).
// coupled(b, j) for the banded factor with two elimination chains (fpsq_band_create): inside the chain region (both < 2 cs)
// only blocks of the same parity within 2 cb; otherwise the plain band |b - j| <= w.  Dense: w = nb, cs = 0.
// The off-diagonal block of a link is requested BEFORE the look at y_j: it is in flight while the workgroup waits.
constexpr int kChainPolls = 1 << 20;
struct ChainArgs {
  unsigned long long* pub;  // [nb][512]: block j's 256 doubles as (high half | seq), (low half << 32 | seq); [nb * 512]: abort
  unsigned int seq;
  unsigned int pubseq;      // what a workgroup publishes: `seq` (anything else only in the test of the bounded wait)
  int nb, band_w, cs, cb;
  unsigned long long* err;  // host-mapped
  unsigned long long* ticket;      // monotone counter (never reset): this launch's workgroups draw ticket_base .. + nb - 1
  unsigned long long ticket_base;
};
__device__ __forceinline__ bool chain_coupled(const ChainArgs& c, int b, int j) {
  const int w = c.band_w > 0 ? c.band_w - 1 : c.nb;
  const int d = b > j ? b - j : j - b;
  if (b < 2 * c.cs && j < 2 * c.cs) return (d & 1) == 0 && d <= 2 * c.cb;
  return d <= w;
}
// this thread's entry of block j's published vector (bounded wait)
__device__ __forceinline__ double chain_take(const ChainArgs& c, int j) {
  const unsigned long long* p = c.pub + (size_t)j * 512 + 2 * threadIdx.x;
  unsigned long long* ab = c.pub + (size_t)c.nb * 512;
  unsigned long long w0, w1;
  int n = __hip_atomic_load(ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.seq ? kChainPolls : 0;
  for (;;) {
    w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (((unsigned int)w0 == c.seq && (unsigned int)w1 == c.seq) || ++n >= kChainPolls) break;
    if ((n & 1023) == 0 && __hip_atomic_load(ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == c.seq) n = kChainPolls - 1;
    __builtin_amdgcn_s_sleep(2);
  }
  if (n >= kChainPolls) {
    __hip_atomic_store(ab, (unsigned long long)c.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(c.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return __longlong_as_double((long long)((w0 & 0xffffffff00000000ull) | (w1 >> 32)));
}
template <bool FORWARD>
__global__ __launch_bounds__(256) void k_trsv_chain(const double* __restrict__ Lm, int ld, const double* __restrict__ inv,
                                                    const double* __restrict__ invT, const double* __restrict__ r, double* out,
                                                    ChainArgs c) {
  __shared__ double rk[kDB * 2];
  __shared__ double part[2][kDB * 2];
  __shared__ double yk[kDB * 2];
  __shared__ int ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0)
    ticket = (int)(__hip_atomic_fetch_add(c.ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c.ticket_base);
  __syncthreads();
  const int b = FORWARD ? ticket : c.nb - 1 - ticket;
  const int i = tid & 127, hf = tid >> 7;
  const double* Xc = (FORWARD ? invT : inv) + (size_t)b * kDB * kDB + (size_t)(hf * 64) * kDB + i;
  const bool xskip = FORWARD ? (hf == 1 && i < 64) : (hf == 0 && i >= 64);  // (wave-uniform) all-zero part of the triangle
  double xs[64];
  if (!xskip) {
#pragma unroll
    for (int q = 0; q < 64; ++q) xs[q] = Xc[(size_t)q * kDB];
  }
  double racc = r[(size_t)b * (kDB * 2) + tid];
  const int band_w = c.band_w;
  const size_t lds = band_w > 0 ? (size_t)kDB : (size_t)ld;
  const int w = band_w > 0 ? band_w - 1 : c.nb;
  if (FORWARD) {
    for (int j = max(0, b - w); j < b; ++j) {
      if (!chain_coupled(c, b, j)) continue;
      // block (b, j), rows 32 wave .. + 31, lanes along the columns
      const double* Lb = (band_w > 0 ? Lm + ((size_t)b * band_w + (j - b + band_w - 1)) * kDB * kDB
                                     : Lm + (size_t)(b * kDB) * ld + j * kDB) + (size_t)(wave * 32) * lds + lane;
      double lb[64];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        lb[2 * u] = Lb[(size_t)u * lds];
        lb[2 * u + 1] = Lb[(size_t)u * lds + 64];
      }
      yk[tid] = chain_take(c, j);
      __syncthreads();
      const double y00 = yk[lane * 2], y01 = yk[lane * 2 + 1], y10 = yk[(lane + 64) * 2], y11 = yk[(lane + 64) * 2 + 1];
      double v[64];
#pragma unroll
      for (int u = 0; u < 32; ++u) {
        v[2 * u] = lb[2 * u] * y00 + lb[2 * u + 1] * y10;
        v[2 * u + 1] = lb[2 * u] * y01 + lb[2 * u + 1] * y11;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int idx = 0; idx < off; ++idx) {
          const double send = hi ? v[idx] : v[idx + off];
          const double keep = hi ? v[idx + off] : v[idx];
          v[idx] = keep + __shfl_xor(send, off, 64);
        }
      }
      racc -= v[0];
      __syncthreads();  // (yk is overwritten by the next link)
    }
  } else {
    for (int j = min(c.nb - 1, b + w); j > b; --j) {
      if (!chain_coupled(c, b, j)) continue;
      // block (j, b) read by columns: column i, rows of this thread's half
      const double* Lb = (band_w > 0 ? Lm + ((size_t)j * band_w + (b - j + band_w - 1)) * kDB * kDB
                                     : Lm + (size_t)(j * kDB) * ld + b * kDB) + (size_t)(hf * 64) * lds + i;
      double lb[64];
#pragma unroll
      for (int q = 0; q < 64; ++q) lb[q] = Lb[(size_t)q * lds];
      yk[tid] = chain_take(c, j);
      __syncthreads();
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int q = 0; q < 64; ++q) {
        s0 += lb[q] * yk[(hf * 64 + q) * 2];
        s1 += lb[q] * yk[(hf * 64 + q) * 2 + 1];
      }
      part[hf][i * 2] = s0;
      part[hf][i * 2 + 1] = s1;
      __syncthreads();
      racc -= part[0][tid] + part[1][tid];
      __syncthreads();  // (part and yk are overwritten by the next link)
    }
  }
  // forward: y_i = sum_{p <= i} X'[p][i] r_p;   backward: q_i = sum_{p >= i} X[p][i] y_p   (p in this thread's half)
  rk[tid] = racc;
  __syncthreads();
  {
    double s0 = 0.0, s1 = 0.0;
    if (!xskip) {
#pragma unroll
      for (int q = 0; q < 64; ++q) {
        s0 += xs[q] * rk[(hf * 64 + q) * 2];
        s1 += xs[q] * rk[(hf * 64 + q) * 2 + 1];
      }
    }
    part[hf][i * 2] = s0;
    part[hf][i * 2 + 1] = s1;
  }
  __syncthreads();
  const double y = part[0][tid] + part[1][tid];
  const unsigned long long bits = (unsigned long long)__double_as_longlong(y);
  unsigned long long* p = c.pub + (size_t)b * 512 + 2 * tid;
  __hip_atomic_store(p, (bits & 0xffffffff00000000ull) | c.pubseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(p + 1, (bits << 32) | c.pubseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  out[(size_t)b * (kDB * 2) + tid] = y;
}

// ---- sparse direct path (fpsq_band): M = A A' + delta I of a BANDED sparse Jacobian as a block band
// One workgroup per 128-row block I.  For each of its rows i in turn: scatter the row into a dense LDS window over its
// column span, then every thread takes rows j <= i of the band (blocks I - bw .. I) and gathers its dot product with
// row i from the window (columns outside the window contribute nothing); M(i, j) goes to block (I, j / 128).
// (A wave per band row with unit-stride loads was measured slower -- 114 against 87 ms at the headline size: the loop
// over the band rows then is a chain of dependent loads, whereas 256 threads walking 256 rows keep 256 streams in flight.)
// Deterministic (fixed summation order, no atomics).  rowspan[i] = {first column, last column} of row i.
__global__ __launch_bounds__(256) void k_band_form(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                   const double* __restrict__ vals, const int2* __restrict__ rowspan,
                                                   int m, int mpad, int band_w, double delta, double* Mb, int span) {
  extern __shared__ __attribute__((aligned(16))) double win[];  // two windows of `span` doubles: rows i and i + 1
  const int I = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < 2 * span; k += 256) win[k] = 0.0;
  __syncthreads();
  const int i0 = I * kDB;
  const int bw = band_w - 1;
  const int jlo = max(0, I - bw) * kDB;
  // TWO rows of the block per pass: every entry of a band row that is loaded serves both dot products (the band rows
  // are re-read from L2 once per pass: 64 instead of 128 times)
  for (int ii = 0; ii < kDB; ii += 2) {
    const int i = i0 + ii;
    double* Mrow0 = Mb + ((size_t)I * band_w) * kDB * kDB + (size_t)ii * kDB;  // row ii of block (I, I - bw)
    double* Mrow1 = Mrow0 + kDB;
    if (i >= m) {  // padding: identity
      if (tid == 0 && i < mpad) Mrow0[(size_t)bw * kDB * kDB + ii] = 1.0;
      if (tid == 0 && i + 1 < mpad) Mrow1[(size_t)bw * kDB * kDB + ii + 1] = 1.0;
      continue;
    }
    const bool two = i + 1 < m;
    const int s0 = rowptr[i], e0 = rowptr[i + 1], e1 = two ? rowptr[i + 2] : e0;
    const int2 sp0 = rowspan[i];
    const int2 sp1 = two ? rowspan[i + 1] : int2{1, 0};  // (an empty span: nothing matches)
    for (int k = s0 + tid; k < e0; k += 256) win[colind[k] - sp0.x] = vals[k];
    for (int k = e0 + tid; k < e1; k += 256) win[span + colind[k] - sp1.x] = vals[k];
    __syncthreads();
    for (int j = jlo + tid; j <= i + 1 && j < m; j += 256) {
      const int js = rowptr[j], je = rowptr[j + 1];
      double a0 = 0.0, a1 = 0.0;
      for (int k = js; k < je; ++k) {
        const int c = colind[k];
        const double v = vals[k];
        if (c >= sp0.x && c <= sp0.y) a0 += v * win[c - sp0.x];
        if (c >= sp1.x && c <= sp1.y) a1 += v * win[span + c - sp1.x];
      }
      const size_t off = (size_t)((j >> 7) - I + bw) * kDB * kDB + (j & 127);
      if (j <= i) Mrow0[off] = j == i ? a0 + delta : a0;
      if (two) Mrow1[off] = j == i + 1 ? a1 + delta : a1;  // (j <= i + 1 by the loop bound)
    }
    if (!two && i + 1 < mpad && tid == 0) Mrow1[(size_t)bw * kDB * kDB + ii + 1] = 1.0;  // first padding row
    __syncthreads();
    for (int k = s0 + tid; k < e0; k += 256) win[colind[k] - sp0.x] = 0.0;
    for (int k = e0 + tid; k < e1; k += 256) win[span + colind[k] - sp1.x] = 0.0;
    __syncthreads();
  }
}

// The same band by COLUMNS of A (the default when A has no duplicate entries): M(i, :) = sum over the entries (i, k) of
// row i of a_ik * A(:, k), the column read from the transposed structure.  Only structurally non-zero products are
// formed -- nnz(A) * (entries per column) of them, 1e8 at the headline size against the 1.6e10 gather-FMAs of the
// row-pair scheme above (61 ms there).  One workgroup per 128-row block, R rows of it per pass, one group of G = 256 / R
// lanes per row with a dense accumulator row of W * 128 doubles in LDS (columns (I - bw) * 128 ...): the lanes of a
// group take the entries of ONE column of A (distinct rows j: no two lanes touch the same accumulator), the entries
// (i, k) of the row are taken in CSR order, four columns' loads in flight -- so every M(i, j) is summed in a fixed order,
// no atomics.  The accumulator rows are then written out whole (zeros included) and cleared.
__global__ __launch_bounds__(256) void k_band_form_t(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                     const double* __restrict__ vals, const int32_t* __restrict__ t_rowptr,
                                                     const int32_t* __restrict__ t_rowind, const double* __restrict__ t_vals,
                                                     int m, int mpad, int band_w, double delta, double* Mb, int R) {
  extern __shared__ __attribute__((aligned(16))) double win[];  // R accumulator rows of band_w * 128
  const int I = blockIdx.x, tid = threadIdx.x;
  const int G = 256 / R, g = tid / G, gl = tid % G;
  const int roww = band_w * kDB;
  const int base = (I - (band_w - 1)) * kDB;  // global column of accumulator entry 0 (may be negative: never touched)
  for (int k = tid; k < R * roww; k += 256) win[k] = 0.0;
  __syncthreads();
  double* acc = win + (size_t)g * roww - base;  // acc[j], j a global row index of A = column of M
  for (int pass = 0; pass < kDB; pass += R) {
    const int i = I * kDB + pass + g;
    if (i < m) {
      const int s = rowptr[i], e = rowptr[i + 1];
      for (int t = s; t < e; t += 4) {
        int us[4], ue[4], j[4];
        double a[4], v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool ok = t + q < e;
          const int k = colind[ok ? t + q : s];
          a[q] = ok ? vals[t + q] : 0.0;
          us[q] = t_rowptr[k];
          ue[q] = ok ? t_rowptr[k + 1] : us[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int u = us[q] + gl;
          const bool ok = u < ue[q];
          j[q] = ok ? t_rowind[u] : INT32_MAX;
          v[q] = ok ? t_vals[u] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (j[q] <= i) acc[j[q]] += a[q] * v[q];
          for (int u = us[q] + gl + G; u < ue[q]; u += G) {  // columns longer than the group
            const int jj = t_rowind[u];
            if (jj <= i) acc[jj] += a[q] * t_vals[u];
          }
        }
      }
      if (gl == 0) acc[i] += delta;
    } else if (i < mpad && gl == 0) {
      acc[i] = 1.0;  // padding: identity
    }
    __syncthreads();
    for (int idx = tid; idx < R * roww; idx += 256) {
      const int r = idx / roww, e = idx - r * roww;
      Mb[((size_t)I * band_w + (e >> 7)) * kDB * kDB + (size_t)(pass + r) * kDB + (e & 127)] = win[idx];
      win[idx] = 0.0;
    }
    __syncthreads();
  }
}

// y[r][0..1] = sum_k vals[k] x[colind[k]][0..1] over row r of a CSR matrix (two interleaved right-hand sides); one
// thread per row
__global__ __launch_bounds__(256) void k_csr_mv2(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colind,
                                                 const double* __restrict__ vals, const double* __restrict__ x, double* y,
                                                 int rows) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  double a0 = 0.0, a1 = 0.0;
  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) {
    const double v = vals[k];
    const double2 xv = *reinterpret_cast<const double2*>(x + (size_t)colind[k] * 2);
    a0 += v * xv.x;
    a1 += v * xv.y;
  }
  y[(size_t)r * 2] = a0;
  y[(size_t)r * 2 + 1] = a1;
}

// jac_coord! hand-over on the device (src/solve_linear_system.jl:223-233: `jac_coord!` then `sparse(rows, cols, vals)`):
// slot i of the back-end's own storage = the sum of the caller's COO entries perm[slotptr[i] .. slotptr[i + 1]) in that
// (sorted, fixed) order -- duplicates are summed like SparseArrays.sparse does, deterministically; slotptr == null: one
// entry per slot.  target != null: the slot lives at out[target[i]] (dense row-major storage), else at out[i].
__global__ __launch_bounds__(256) void k_coo_to_slots(const double* __restrict__ coo, const int32_t* __restrict__ perm,
                                                      const int32_t* __restrict__ slotptr, const int64_t* __restrict__ target,
                                                      double* __restrict__ out, int64_t nslots) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nslots; i += (int64_t)gridDim.x * 256) {
    double v;
    if (slotptr) {
      v = 0.0;
      for (int k = slotptr[i]; k < slotptr[i + 1]; ++k) v += coo[perm[k]];
    } else {
      v = coo[perm[i]];
    }
    out[target ? target[i] : i] = v;
  }
}

__global__ __launch_bounds__(256) void k_gather_d(const double* __restrict__ in, const int32_t* __restrict__ perm,
                                                  double* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = in[perm[i]];
}

// r[i] = {ag[i][0], sb * b[i]} on rows < m, zero on the padding (the right-hand sides of the two M-solves)
__global__ __launch_bounds__(256) void k_band_rhs(const double* __restrict__ ag, int col, const double* b, double sb,
                                                  double* r, int m, int mpad, int both) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= mpad) return;
  const bool in = i < m;
  r[(size_t)i * 2] = in ? ag[(size_t)i * 2] : 0.0;
  r[(size_t)i * 2 + 1] = in ? (both ? ag[(size_t)i * 2 + 1] : sb * b[i]) : 0.0;
  (void)col;
}

// p1 = a0 - atq[.][0];  p2 = (a1 ? a1 : 0) - atq[.][1]
__global__ __launch_bounds__(256) void k_band_finish(const double* __restrict__ atq, const double* a0, const double* a1,
                                                     double* p1, double* p2, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  p1[i] = a0[i] - atq[(size_t)i * 2];
  p2[i] = (a1 ? a1[i] : 0.0) - atq[(size_t)i * 2 + 1];
}

}  // namespace fpsq

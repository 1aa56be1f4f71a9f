// spmv_bench.hip -- kernel-variant micro-benchmark on the headline sparsity (developer tool, not shipped).
// Alternates A and A' products like the Krylov loops do (so neither matrix stays in the 256 MiB Infinity Cache
// on its own) and times every launch with HIP events.
//   hipcc -O3 --offload-arch=gfx950 -o spmv_bench spmv_bench.hip && ./spmv_bench [n m per_row window reps]
#include "../fletcherpenaltysolver.jl_amd/csrc/fpsq_kernels.hip.h"
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include <climits>
using namespace fpsq;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

static uint64_t sm64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

struct Host { int64_t nr, nc; std::vector<int32_t> rp, ci; std::vector<double> v; };
struct Dev { CsrView view; int32_t *rp, *ci, *rb; double* v; int nblk; int64_t nr, nc, nnz; };

static std::vector<int32_t> rowblocks(const std::vector<int32_t>& rp, int64_t nr) {
  std::vector<int32_t> rb{0};
  int64_t r = 0;
  while (r < nr) { int64_t r1 = r, nz = 0; while (r1 < nr && r1 - r < kMaxRowsPerBlk) { int64_t len = rp[r1+1]-rp[r1]; if (nz + len > kSpmvNnz) break; nz += len; ++r1; } if (r1 == r) r1 = r+1; rb.push_back((int32_t)r1); r = r1; }
  return rb;
}
static Dev upload(const Host& H) {
  Dev D; D.nr = H.nr; D.nc = H.nc; D.nnz = H.ci.size();
  auto rb = rowblocks(H.rp, H.nr); D.nblk = rb.size() - 1;
  CK(hipMalloc(&D.rp, H.rp.size()*4)); CK(hipMalloc(&D.ci, H.ci.size()*4)); CK(hipMalloc(&D.v, H.v.size()*8)); CK(hipMalloc(&D.rb, rb.size()*4));
  CK(hipMemcpy(D.rp, H.rp.data(), H.rp.size()*4, hipMemcpyHostToDevice)); CK(hipMemcpy(D.ci, H.ci.data(), H.ci.size()*4, hipMemcpyHostToDevice));
  CK(hipMemcpy(D.v, H.v.data(), H.v.size()*8, hipMemcpyHostToDevice)); CK(hipMemcpy(D.rb, rb.data(), rb.size()*4, hipMemcpyHostToDevice));
  std::vector<int4> bd(D.nblk); for (int b = 0; b < D.nblk; ++b) bd[b] = int4{rb[b], rb[b+1]-rb[b], H.rp[rb[b]], H.rp[rb[b+1]]};
  int4* dbd; CK(hipMalloc(&dbd, bd.size()*16)); CK(hipMemcpy(dbd, bd.data(), bd.size()*16, hipMemcpyHostToDevice));
  D.view = CsrView{D.rp, D.ci, D.v, D.rb, D.nblk, (int32_t)D.nr, nullptr, nullptr, dbd};
  return D;
}
static Host transpose(const Host& A) {
  Host T; T.nr = A.nc; T.nc = A.nr; T.rp.assign(T.nr + 1, 0); size_t nnz = A.ci.size();
  for (size_t k = 0; k < nnz; ++k) T.rp[A.ci[k] + 1]++;
  for (int64_t j = 0; j < T.nr; ++j) T.rp[j+1] += T.rp[j];
  T.ci.resize(nnz); T.v.resize(nnz); std::vector<int32_t> nx(T.rp.begin(), T.rp.end()-1);
  for (int64_t i = 0; i < A.nr; ++i) for (int k = A.rp[i]; k < A.rp[i+1]; ++k) { int t = nx[A.ci[k]]++; T.ci[t] = i; T.v[t] = A.v[k]; }
  return T;
}

// ---------------------------------------------------------------- candidate: "vector" kernel, G lanes per row, no LDS
template <int NL, int G, int UNROLL>
__global__ __launch_bounds__(kBlock) void k_spmv_vec(CsrView A, const double* __restrict__ x, const double* yin, double* yout,
                                                     const LaneCtl* ctl0, const LaneCtl* ctl1, double* partials, int blk_per_xcd, int nblk) {
  const int L = (blockIdx.x & 7) * blk_per_xcd + (blockIdx.x >> 3);
  if (L >= nblk) return;
  const LaneCtl* c[2] = {ctl0, ctl1};
  double ca[NL], cb[NL]; bool act[NL]; bool any = false;
#pragma unroll
  for (int l = 0; l < NL; ++l) { act[l] = !(c[l]->done | c[l]->skip); ca[l] = c[l]->ca; cb[l] = c[l]->cb; any |= act[l]; }
  if (!any) return;
  constexpr int RPB = kBlock / G;
  const int tid = threadIdx.x, gl = tid % G;
  const int row = L * RPB + tid / G;
  double acc[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) acc[l] = 0.0;
  const bool valid = row < A.nrows;
  if (valid) {
    const int a = A.rowptr[row], b = A.rowptr[row + 1];
    int j = a + gl;
    for (; j + (UNROLL - 1) * G < b; j += UNROLL * G) {
      int cc[UNROLL]; double vv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { cc[u] = A.colind[j + u * G]; vv[u] = A.vals[j + u * G]; }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (NL == 1) acc[0] += vv[u] * x[cc[u]];
        else { const double2 xv = *reinterpret_cast<const double2*>(x + (size_t)cc[u] * 2); acc[0] += vv[u] * xv.x; acc[NL-1] += vv[u] * xv.y; }
      }
    }
    for (; j < b; j += G) {
      const int cidx = A.colind[j]; const double v = A.vals[j];
      if (NL == 1) acc[0] += v * x[cidx];
      else { const double2 xv = *reinterpret_cast<const double2*>(x + (size_t)cidx * 2); acc[0] += v * xv.x; acc[NL-1] += v * xv.y; }
    }
  }
#pragma unroll
  for (int off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] += __shfl_down(acc[l], off, 64);
  }
  double sq[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = 0.0;
  if (valid && gl == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) if (act[l]) {
      const double o = ca[l] * acc[l] + (cb[l] != 0.0 ? cb[l] * yin[(size_t)row * NL + l] : 0.0);
      yout[(size_t)row * NL + l] = o; sq[l] = o * o;
    }
  }
  // per-wave partials: no workgroup barrier at all
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const double t = wave_sum(sq[l]);
    if ((tid & 63) == 0) partials[((size_t)l * nblk + L) * 4 + (tid >> 6)] = t;
  }
}


// ---------------------------------------------------------------- diagnostic: stream kernel with the gather made trivially L1-resident
template <int NL>
__global__ __launch_bounds__(kBlock) void k_spmv_nogather(CsrView A, const double* __restrict__ x, const double* yin, double* yout,
                                                          const LaneCtl* ctl0, double* partials, int blk_per_xcd) {
  const int L = (blockIdx.x & 7) * blk_per_xcd + (blockIdx.x >> 3);
  if (L >= A.nblk) return;
  __shared__ double prod[kSpmvNnz * NL];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int r0 = A.rowblk[L], r1 = A.rowblk[L + 1];
  const int s = A.rowptr[r0], e = A.rowptr[r1];
  const int nr = r1 - r0;
  constexpr int kPer = kSpmvNnz / kBlock;
  int cidx[kPer]; double v[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) { const int i = s + tid + k * kBlock; const bool ok = i < e; cidx[k] = ok ? A.colind[i] : -1; v[k] = ok ? A.vals[i] : 0.0; }
#pragma unroll
  for (int k = 0; k < kPer; ++k) if (cidx[k] >= 0) { const int j = tid + k * kBlock;
      for (int l = 0; l < NL; ++l) prod[j * NL + l] = v[k] * x[(size_t)((cidx[k] & 15) + (tid & 48)) * NL + l]; }
  __syncthreads();
  int G = 1; while (G < 64 && G * 2 * nr <= kBlock) G <<= 1;
  const int rows_per_pass = kBlock / G; const int g = tid / G, gl = tid % G;
  double sq = 0;
  for (int base = 0; base < nr; base += rows_per_pass) {
    const int rr = base + g; const bool valid = rr < nr; double acc[NL]; for (int l = 0; l < NL; ++l) acc[l] = 0;
    if (valid) { const int a = A.rowptr[r0 + rr] - s, b = A.rowptr[r0 + rr + 1] - s; for (int j = a + gl; j < b; j += G) for (int l = 0; l < NL; ++l) acc[l] += prod[j * NL + l]; }
    for (int off = G >> 1; off > 0; off >>= 1) for (int l = 0; l < NL; ++l) acc[l] += __shfl_down(acc[l], off, 64);
    if (valid && gl == 0) for (int l = 0; l < NL; ++l) { const double o = acc[l] + 0.5 * yin[(size_t)(r0 + rr) * NL + l]; yout[(size_t)(r0 + rr) * NL + l] = o; sq += o * o; }
  }
  const double t = block_sum(sq, red); if (tid == 0) partials[L] = t;
}

// ---------------------------------------------------------------- candidate: CSR-stream with the x window staged in LDS
// A workgroup owns S consecutive fine row-blocks (each <= kSpmvNnz nonzeros); all their columns lie in
// [cmin, cmin + span) with span <= WCAP, so x[cmin .. cmin+span) is copied to LDS once (coalesced) and every
// gather becomes an LDS read.
template <int NL, int T, int WCAP>
__global__ __launch_bounds__(T) void k_spmv_win(CsrView A, int S, int ngroups, const int32_t* __restrict__ gcmin, const int32_t* __restrict__ gspan,
                                               const double* __restrict__ x, const double* yin, double* yout,
                                               const LaneCtl* ctl0, const LaneCtl* ctl1, double* partials, int grp_per_xcd) {
  const int Lg = (blockIdx.x & 7) * grp_per_xcd + (blockIdx.x >> 3);
  if (Lg >= ngroups) return;
  const LaneCtl* c[2] = {ctl0, ctl1};
  double ca[NL], cb[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) { ca[l] = c[l]->ca; cb[l] = c[l]->cb; }
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xw = smem;                   // WCAP * NL
  double* prod = smem + WCAP * NL;     // kSpmvNnz * NL
  double* red = prod + kSpmvNnz * NL;  // T/64
  const int tid = threadIdx.x;
  const int cmin = gcmin[Lg], span = gspan[Lg];
  for (int j = tid; j < span * NL; j += T) xw[j] = x[(size_t)cmin * NL + j];
  constexpr int kPer = kSpmvNnz / T;
  const int fb0 = Lg * S, fb1 = min(fb0 + S, A.nblk);
  int cidx[kPer]; double v[kPer];
  // prefetch first chunk
  {
    const int s = A.rowptr[A.rowblk[fb0]], e = A.rowptr[A.rowblk[fb0 + 1]];
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const int i = s + tid + k * T; const bool ok = i < e; cidx[k] = ok ? A.colind[i] - cmin : -1; v[k] = ok ? A.vals[i] : 0.0; }
  }
  double sq[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = 0.0;
  __syncthreads();
  for (int fb = fb0; fb < fb1; ++fb) {
    const int r0 = A.rowblk[fb], r1 = A.rowblk[fb + 1];
    const int s = A.rowptr[r0];
    const int nr = r1 - r0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) if (cidx[k] >= 0) {
      const int j = tid + k * T;
      if (NL == 1) prod[j] = v[k] * xw[cidx[k]];
      else { const double2 xv = *reinterpret_cast<const double2*>(xw + 2 * cidx[k]); *reinterpret_cast<double2*>(prod + 2 * j) = make_double2(v[k] * xv.x, v[k] * xv.y); }
    }
    // prefetch the next chunk while this one is reduced
    if (fb + 1 < fb1) {
      const int s2 = A.rowptr[r1], e2 = A.rowptr[A.rowblk[fb + 2]];
#pragma unroll
      for (int k = 0; k < kPer; ++k) { const int i = s2 + tid + k * T; const bool ok = i < e2; cidx[k] = ok ? A.colind[i] - cmin : -1; v[k] = ok ? A.vals[i] : 0.0; }
    }
    __syncthreads();
    int G = 1; while (G < 64 && G * 2 * nr <= T) G <<= 1;
    const int rows_per_pass = T / G; const int g = tid / G, gl = tid % G;
    for (int base = 0; base < nr; base += rows_per_pass) {
      const int rr = base + g; const bool valid = rr < nr;
      double acc[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] = 0.0;
      if (valid) {
        const int a = A.rowptr[r0 + rr] - s, b = A.rowptr[r0 + rr + 1] - s;
        for (int j = a + gl; j < b; j += G) {
          if (NL == 1) acc[0] += prod[j];
          else { const double2 pv = *reinterpret_cast<const double2*>(prod + 2 * j); acc[0] += pv.x; acc[NL - 1] += pv.y; }
        }
      }
      for (int off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int l = 0; l < NL; ++l) acc[l] += __shfl_down(acc[l], off, 64);
      }
      if (valid && gl == 0) {
        const size_t row = (size_t)(r0 + rr);
#pragma unroll
        for (int l = 0; l < NL; ++l) { const double o = ca[l] * acc[l] + (cb[l] != 0.0 ? cb[l] * yin[row * NL + l] : 0.0); yout[row * NL + l] = o; sq[l] += o * o; }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    double t = wave_sum(sq[l]);
    if ((tid & 63) == 0) red[tid >> 6] = t;
    __syncthreads();
    if (tid == 0) { double a = 0; for (int w = 0; w < T / 64; ++w) a += red[w]; partials[(size_t)l * ngroups + Lg] = a; }
    __syncthreads();
  }
}

struct Groups { int S, ng; int32_t *cmin, *span; int maxspan; };
static Groups make_groups(const Host& H, int S) {
  auto rb = rowblocks(H.rp, H.nr); int nblk = rb.size() - 1; Groups g; g.S = S; g.ng = (nblk + S - 1) / S; g.maxspan = 0;
  std::vector<int32_t> cmin(g.ng), span(g.ng);
  for (int q = 0; q < g.ng; ++q) {
    int r0 = rb[q * S], r1 = rb[std::min(nblk, (q + 1) * S)]; int lo = INT32_MAX, hi = -1;
    for (int k = H.rp[r0]; k < H.rp[r1]; ++k) { lo = std::min(lo, H.ci[k]); hi = std::max(hi, H.ci[k]); }
    if (hi < 0) { lo = 0; hi = 0; }
    cmin[q] = lo; span[q] = hi - lo + 1; g.maxspan = std::max(g.maxspan, span[q]);
  }
  CK(hipMalloc(&g.cmin, g.ng * 4)); CK(hipMalloc(&g.span, g.ng * 4));
  CK(hipMemcpy(g.cmin, cmin.data(), g.ng * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g.span, span.data(), g.ng * 4, hipMemcpyHostToDevice));
  return g;
}

// ---------------------------------------------------------------- ceilings: pure streaming of (colind, vals), narrow vs wide loads
__global__ __launch_bounds__(kBlock) void k_stream_narrow(const int32_t* __restrict__ ci, const double* __restrict__ v, size_t nnz, double* out) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (size_t)gridDim.x * kBlock) acc += v[i] * (double)ci[i];
  if (acc == 1.2345e-300) out[0] = acc;
}
__global__ __launch_bounds__(kBlock) void k_stream_wide(const int4* __restrict__ ci, const double2* __restrict__ v, size_t nquad, double* out) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < nquad; i += (size_t)gridDim.x * kBlock) {
    const int4 c = ci[i]; const double2 a = v[2 * i], b = v[2 * i + 1];
    acc += a.x * c.x + a.y * c.y + b.x * c.z + b.y * c.w;
  }
  if (acc == 1.2345e-300) out[0] = acc;
}
// stream kernel, gather with a non-temporal hint
template <int NL>
__global__ __launch_bounds__(kBlock) void k_spmv_ntgather(CsrView A, const double* __restrict__ x, const double* yin, double* yout,
                                                          const LaneCtl* ctl0, double* partials, int blk_per_xcd) {
  const int L = (blockIdx.x & 7) * blk_per_xcd + (blockIdx.x >> 3);
  if (L >= A.nblk) return;
  __shared__ double prod[kSpmvNnz * NL];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int r0 = A.rowblk[L], r1 = A.rowblk[L + 1];
  const int s = A.rowptr[r0], e = A.rowptr[r1];
  const int nr = r1 - r0;
  constexpr int kPer = kSpmvNnz / kBlock;
  int cidx[kPer]; double v[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) { const int i = s + tid + k * kBlock; const bool ok = i < e; cidx[k] = ok ? __builtin_nontemporal_load(A.colind + i) : -1; v[k] = ok ? __builtin_nontemporal_load(A.vals + i) : 0.0; }
#pragma unroll
  for (int k = 0; k < kPer; ++k) if (cidx[k] >= 0) { const int j = tid + k * kBlock;
      for (int l = 0; l < NL; ++l) prod[j * NL + l] = v[k] * x[(size_t)cidx[k] * NL + l]; }
  __syncthreads();
  int G = 1; while (G < 64 && G * 2 * nr <= kBlock) G <<= 1;
  const int rows_per_pass = kBlock / G; const int g = tid / G, gl = tid % G;
  double sq = 0;
  for (int base = 0; base < nr; base += rows_per_pass) {
    const int rr = base + g; const bool valid = rr < nr; double acc[NL]; for (int l = 0; l < NL; ++l) acc[l] = 0;
    if (valid) { const int a = A.rowptr[r0 + rr] - s, b = A.rowptr[r0 + rr + 1] - s; for (int j = a + gl; j < b; j += G) for (int l = 0; l < NL; ++l) acc[l] += prod[j * NL + l]; }
    for (int off = G >> 1; off > 0; off >>= 1) for (int l = 0; l < NL; ++l) acc[l] += __shfl_down(acc[l], off, 64);
    if (valid && gl == 0) for (int l = 0; l < NL; ++l) { const double o = acc[l] + 0.5 * yin[(size_t)(r0 + rr) * NL + l]; yout[(size_t)(r0 + rr) * NL + l] = o; sq += o * o; }
  }
  const double t = block_sum(sq, red); if (tid == 0) partials[L] = t;
}

// ---------------------------------------------------------------- candidate: RGCS = row groups, column-sorted tiles
// Entries of a row group are stored sorted by COLUMN (so the 64 gathers of a wave instruction fall in a handful
// of cache lines) and carry, packed with the group-relative column, their slot in the tile's ROW-major order, so
// the products land in LDS grouped by row and are reduced exactly like in the CSR-stream kernel.
__device__ __forceinline__ void p_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int kTile = 2048;
constexpr int kColBits = 21;
constexpr int kMaxPass = 4;
struct PRgcsView {
  const uint32_t* pidx; const double* vals;
  const int32_t* grow; const int32_t* gent; const int32_t* gcmin; const int32_t* gtp; const uint16_t* tptr;
  int ng; int nrows;
};
template <int NL>
struct TileRegs {   // everything one tile needs from global memory, kept raw until consumed
  uint32_t pk[kTile / kBlock];
  double v[kTile / kBlock];
  uint32_t traw[kMaxPass];
};

template <int NL, int ABL = 0>
__global__ __launch_bounds__(kBlock) void k_spmv_rgcs_proto(PRgcsView M, const double* __restrict__ x, const double* yin, double* yout,
                                                      const LaneCtl* ctl0, const LaneCtl* ctl1, double* partials, int grp_per_xcd) {
  const int g = (blockIdx.x & 7) * grp_per_xcd + (blockIdx.x >> 3);
  if (g >= M.ng) return;
  const LaneCtl* c[2] = {ctl0, ctl1};
  double ca[NL], cb[NL]; bool act[NL]; bool any = false;
#pragma unroll
  for (int l = 0; l < NL; ++l) { act[l] = !(c[l]->done | c[l]->skip); ca[l] = c[l]->ca; cb[l] = c[l]->cb; any |= act[l]; }
  if (!any) return;
  __shared__ double prod[kTile * NL];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int r0 = M.grow[g], R = M.grow[g + 1] - r0;
  const int e0 = M.gent[g], e1 = M.gent[g + 1];
  const int cmin = M.gcmin[g];
  const uint16_t* tp = M.tptr + M.gtp[g];
  int G = 1; while (G < 64 && G * 2 * R <= kBlock) G <<= 1;
  const int rpp = kBlock / G, gid = tid / G, gl = tid % G;
  double acc[kMaxPass][NL];
#pragma unroll
  for (int p = 0; p < kMaxPass; ++p)
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[p][l] = 0.0;
  constexpr int kPer = kTile / kBlock;
  // Software pipeline, two register sets: while tile j is gathered / multiplied / reduced, the loads of tile j+1 are
  // already in flight.  Issue order inside an iteration is  gathers(j) -> loads(j+1)  so that waiting for the
  // gathers (s_waitcnt vmcnt(#loads of j+1)) leaves the younger streaming loads outstanding.
  auto fetch = [&](TileRegs<NL>& T, int base, int tile) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = base + tid + k * kBlock; const int ii = i < e1 ? i : e0;
      T.pk[k] = M.pidx[ii]; T.v[k] = M.vals[ii];
    }
    const uint16_t* tpt = tp + (size_t)tile * (R + 1);
#pragma unroll
    for (int p = 0; p < kMaxPass; ++p) {
      const int rr = p * rpp + gid; const int rq = rr < R ? rr : 0;
      __builtin_memcpy(&T.traw[p], tpt + rq, 4);
    }
  };
  auto process = [&](TileRegs<NL>& C, TileRegs<NL>& N, int base, int tile) {
    int sa[kMaxPass], sb[kMaxPass];
    double2 xv[kPer]; uint32_t pq[kPer]; double vq[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const bool ok = base + tid + k * kBlock < e1;
      pq[k] = ok ? C.pk[k] : ((uint32_t)(tid + k * kBlock) << kColBits); vq[k] = ok ? C.v[k] : 0.0;
      const int col = (ABL & 1) ? (int)(pq[k] & 15u) : cmin + (int)(pq[k] & ((1u << kColBits) - 1));
      if (NL == 1) xv[k].x = x[col]; else xv[k] = *reinterpret_cast<const double2*>(x + (size_t)col * 2);
    }
#pragma unroll
    for (int p = 0; p < kMaxPass; ++p) { const bool valid = p * rpp + gid < R; sa[p] = (int)(C.traw[p] & 0xffffu); sb[p] = valid ? (int)(C.traw[p] >> 16) : sa[p]; }
    __builtin_amdgcn_sched_barrier(0);   // gathers(j) must be OLDER than loads(j+1): vmcnt retires in order
    // unconditional (addresses are clamped inside): a branch here would make the waitcnt pass assume the loads might
    // not have been issued and wait for vmcnt(0)
    fetch(N, base + kTile, base + kTile < e1 ? tile + 1 : 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int slot = (ABL & 2) ? tid + k * kBlock : (int)(pq[k] >> kColBits);
      if (NL == 1) prod[slot] = vq[k] * xv[k].x;
      else *reinterpret_cast<double2*>(prod + 2 * slot) = make_double2(vq[k] * xv[k].x, vq[k] * xv[k].y);
    }
    p_lds_barrier();
#pragma unroll
    for (int p = 0; p < kMaxPass; ++p) {
      const int a = (ABL & 4) ? tid : sa[p], b = (ABL & 4) ? tid + (p == 0) : sb[p];
      row_segment_sum<NL>(prod, a + gl, b, G, acc[p]);
    }
    p_lds_barrier();
  };
  TileRegs<NL> RA, RB;
  fetch(RA, e0, 0);
  int tile = 0;
  for (int base = e0; base < e1; base += 2 * kTile, tile += 2) {
    process(RA, RB, base, tile);
    if (base + kTile < e1) process(RB, RA, base + kTile, tile + 1);
  }
  double sq[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = 0.0;
#pragma unroll
  for (int p = 0; p < kMaxPass; ++p) {
    for (int off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[p][l] += __shfl_down(acc[p][l], off, 64);
    }
    const int rr = p * rpp + gid;
    if (rr < R && gl == 0) {
      const size_t row = (size_t)(r0 + rr);
#pragma unroll
      for (int l = 0; l < NL; ++l) if (act[l]) { if (ABL & 32) { sq[l] += acc[p][l]; } else { const double o = ca[l] * acc[p][l] + (cb[l] != 0.0 ? cb[l] * yin[row * NL + l] : 0.0); yout[row * NL + l] = o; sq[l] += o * o; } }
    }
  }
#pragma unroll
  for (int l = 0; l < NL; ++l) { const double t = block_sum(sq[l], red); if (tid == 0) partials[(size_t)l * M.ng + g] = t; }
}

struct PRgcsDev { PRgcsView view; };
static PRgcsDev build_rgcs(const Host& H, int group_nnz, int max_rows) {
  std::vector<int32_t> grow{0}, gent{0}, gcmin, gtp{0};
  std::vector<uint32_t> pidx(H.ci.size()); std::vector<double> vals(H.ci.size()); std::vector<uint16_t> tptr;
  int64_t r = 0;
  std::vector<int32_t> ord, rank;
  while (r < H.nr) {
    int64_t r1 = r, nz = 0;
    while (r1 < H.nr && r1 - r < max_rows) { int64_t len = H.rp[r1 + 1] - H.rp[r1]; if (nz + len > group_nnz && r1 > r) break; nz += len; ++r1; if (nz >= group_nnz) break; }
    const int R = r1 - r; const int e0 = H.rp[r], e1 = H.rp[r1]; const int cnt = e1 - e0;
    int cmin = INT32_MAX; for (int k = e0; k < e1; ++k) cmin = std::min(cmin, H.ci[k]); if (cnt == 0) cmin = 0;
    // local row of each entry
    std::vector<int32_t> lrow(cnt); for (int rr = 0; rr < R; ++rr) for (int k = H.rp[r + rr]; k < H.rp[r + rr + 1]; ++k) lrow[k - e0] = rr;
    ord.resize(cnt); for (int k = 0; k < cnt; ++k) ord[k] = k;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return H.ci[e0 + a] < H.ci[e0 + b]; });
    const int ntile = (cnt + kTile - 1) / kTile;
    for (int t = 0; t < ntile; ++t) {
      const int a = t * kTile, b = std::min(cnt, a + kTile);
      // slot = rank of the entry within the tile in (row, col) order: counting by row (entries of a row keep column order)
      std::vector<int32_t> cntr(R + 1, 0);
      for (int k = a; k < b; ++k) cntr[lrow[ord[k]] + 1]++;
      for (int rr = 0; rr < R; ++rr) cntr[rr + 1] += cntr[rr];
      for (int rr = 0; rr <= R; ++rr) tptr.push_back((uint16_t)cntr[rr]);
      std::vector<int32_t> nxt(cntr.begin(), cntr.end() - 1);
      for (int k = a; k < b; ++k) {
        const int src = ord[k]; const int slot = nxt[lrow[src]]++;
        const uint32_t crel = (uint32_t)(H.ci[e0 + src] - cmin);
        if (crel >= (1u << kColBits)) { printf("column span too large for RGCS\n"); exit(1); }
        pidx[e0 + k] = ((uint32_t)slot << kColBits) | crel; vals[e0 + k] = H.v[e0 + src];
      }
    }
    grow.push_back((int32_t)r1); gent.push_back(e1); gcmin.push_back(cmin); gtp.push_back((int32_t)tptr.size());
    r = r1;
  }
  PRgcsDev D; PRgcsView& V = D.view; V.ng = gcmin.size(); V.nrows = H.nr;
  uint32_t* dp; double* dv; int32_t *d1, *d2, *d3, *d4; uint16_t* d5;
  CK(hipMalloc(&dp, pidx.size() * 4)); CK(hipMalloc(&dv, vals.size() * 8)); CK(hipMalloc(&d1, grow.size() * 4)); CK(hipMalloc(&d2, gent.size() * 4));
  CK(hipMalloc(&d3, gcmin.size() * 4)); CK(hipMalloc(&d4, gtp.size() * 4)); CK(hipMalloc(&d5, tptr.size() * 2 + 16));
  CK(hipMemcpy(dp, pidx.data(), pidx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dv, vals.data(), vals.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d1, grow.data(), grow.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d2, gent.data(), gent.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d3, gcmin.data(), gcmin.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d4, gtp.data(), gtp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d5, tptr.data(), tptr.size() * 2, hipMemcpyHostToDevice));
  V.pidx = dp; V.vals = dv; V.grow = d1; V.gent = d2; V.gcmin = d3; V.gtp = d4; V.tptr = d5;
  printf("RGCS: %d groups (group_nnz=%d max_rows=%d), tptr %.2f MB\n", V.ng, group_nnz, max_rows, tptr.size() * 2 / 1e6);
  return D;
}

// ---------------------------------------------------------------- structure probes: what costs the 10 us between pure streaming and a tile kernel?
// FEAT bit0: LDS write of the products + 2 barriers per tile; bit1: XCD swizzle; bit2: single tile per block (grid = ntiles)
template <int FEAT>
__global__ __launch_bounds__(kBlock) void k_probe(const uint32_t* __restrict__ pidx, const double* __restrict__ vals, int ntiles, int tiles_per_blk,
                                                  int blk_per_xcd, double* out) {
  __shared__ double prod[kTile * 2];
  int b = blockIdx.x;
  if (FEAT & 2) b = (blockIdx.x & 7) * blk_per_xcd + (blockIdx.x >> 3);
  const int t0 = b * tiles_per_blk, t1 = min(ntiles, t0 + tiles_per_blk);
  const int tid = threadIdx.x;
  double acc = 0;
  constexpr int kPer = kTile / kBlock;
  for (int t = t0; t < t1; ++t) {
    uint32_t pk[kPer]; double v[kPer];
    const size_t base = (size_t)t * kTile;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { pk[k] = pidx[base + tid + k * kBlock]; v[k] = vals[base + tid + k * kBlock]; }
    if (FEAT & 1) {
#pragma unroll
      for (int k = 0; k < kPer; ++k) *reinterpret_cast<double2*>(prod + 2 * (tid + k * kBlock)) = make_double2(v[k] * (double)pk[k], v[k]);
      p_lds_barrier();
      acc += prod[2 * ((tid * 7) & (kTile - 1))];
      p_lds_barrier();
    } else {
#pragma unroll
      for (int k = 0; k < kPer; ++k) acc += v[k] * (double)pk[k];
    }
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

__global__ void k_ctl(LaneCtl* c, double ca, double cb) { c->ca = ca; c->cb = cb; c->done = 0; c->skip = 0; c->upd_iter = -1; }

int main(int argc, char** argv) {
  int64_t n = argc > 1 ? atoll(argv[1]) : 1000000, m = argc > 2 ? atoll(argv[2]) : 100000;
  int per = argc > 3 ? atoi(argv[3]) : 100, window = argc > 4 ? atoi(argv[4]) : 8192, reps = argc > 5 ? atoi(argv[5]) : 20;
  Host A; A.nr = m; A.nc = n; A.rp.resize(m + 1); A.ci.resize((size_t)m * per); A.v.resize((size_t)m * per);
  for (int64_t i = 0; i <= m; ++i) A.rp[i] = i * per;
  for (int64_t i = 0; i < m; ++i) {
    int64_t center = i * n / m, start = std::min<int64_t>(std::max<int64_t>(center - window / 2, 0), n - window);
    for (int k = 0; k < per; ++k) {
      int64_t lo = (int64_t)k * window / per, hi = (int64_t)(k + 1) * window / per; uint64_t r = sm64(i * 1315423911ull + k);
      A.ci[i * per + k] = start + lo + r % (hi - lo); A.v[i * per + k] = (double)(sm64(r) >> 11) / 9007199254740992.0 * 2 - 1;
    }
  }
  Host T = transpose(A);
  Dev dA = upload(A), dT = upload(T);
  const size_t nnz = A.ci.size();
  printf("n=%lld m=%lld nnz=%zu  A blocks=%d  AT blocks=%d\n", (long long)n, (long long)m, nnz, dA.nblk, dT.nblk);
  double *xn, *xm, *yn, *ym, *part; LaneCtl* ctl;
  CK(hipMalloc(&xn, n * 16)); CK(hipMalloc(&xm, m * 16)); CK(hipMalloc(&yn, n * 16)); CK(hipMalloc(&ym, m * 16));
  CK(hipMalloc(&part, 8 * 8 * (size_t)std::max<int64_t>(n, m))); CK(hipMalloc(&ctl, sizeof(LaneCtl)));
  std::vector<double> hx(2 * n); for (auto& v : hx) v = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(xn, hx.data(), n * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(xm, hx.data(), m * 16, hipMemcpyHostToDevice));
  CK(hipMemset(yn, 0, n * 16)); CK(hipMemset(ym, 0, m * 16));
  hipLaunchKernelGGL(k_ctl, dim3(1), dim3(1), 0, 0, ctl, 1.0, 0.5);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);

  struct Variant { std::string name; int nl; std::function<void()> launchA, launchT; };
  std::vector<Variant> vs;
  auto add_stream = [&](int nl) {
    Variant v; v.name = std::string("stream2048 NL=") + std::to_string(nl); v.nl = nl;
    int pa = (dA.nblk + 7) / 8, pt = (dT.nblk + 7) / 8;
    if (nl == 1) {
      v.launchA = [=]() { hipLaunchKernelGGL((k_spmv<1, 0>), dim3(pa * 8), dim3(kBlock), 0, 0, dA.view, xn, ym, ym, ctl, ctl, part, pa); };
      v.launchT = [=]() { hipLaunchKernelGGL((k_spmv<1, 1>), dim3(pt * 8), dim3(kBlock), 0, 0, dT.view, xm, yn, yn, ctl, ctl, part, pt); };
    } else {
      v.launchA = [=]() { hipLaunchKernelGGL((k_spmv<2, 0>), dim3(pa * 8), dim3(kBlock), 0, 0, dA.view, xn, ym, ym, ctl, ctl, part, pa); };
      v.launchT = [=]() { hipLaunchKernelGGL((k_spmv<2, 1>), dim3(pt * 8), dim3(kBlock), 0, 0, dT.view, xm, yn, yn, ctl, ctl, part, pt); };
    }
    vs.push_back(v);
  };
  add_stream(1); add_stream(2);
#define ADD_VEC(NLv, GA, GT, U)                                                                                          \
  { Variant v; v.name = "vec NL=" #NLv " GA=" #GA " GT=" #GT " U=" #U; v.nl = NLv;                                        \
    int nba = (m + kBlock / GA - 1) / (kBlock / GA), nbt = (n + kBlock / GT - 1) / (kBlock / GT);                          \
    int pa = (nba + 7) / 8, pt = (nbt + 7) / 8;                                                                            \
    v.launchA = [=]() { hipLaunchKernelGGL((k_spmv_vec<NLv, GA, U>), dim3(pa * 8), dim3(kBlock), 0, 0, dA.view, xn, ym, ym, ctl, ctl, part, pa, nba); }; \
    v.launchT = [=]() { hipLaunchKernelGGL((k_spmv_vec<NLv, GT, U>), dim3(pt * 8), dim3(kBlock), 0, 0, dT.view, xm, yn, yn, ctl, ctl, part, pt, nbt); }; \
    vs.push_back(v); }
  ADD_VEC(1, 16, 4, 2) ADD_VEC(1, 16, 4, 4) ADD_VEC(1, 32, 8, 2) ADD_VEC(1, 8, 2, 4) ADD_VEC(1, 8, 4, 4) ADD_VEC(1, 16, 2, 4)
  ADD_VEC(2, 16, 4, 2) ADD_VEC(2, 16, 4, 4) ADD_VEC(2, 8, 4, 4) ADD_VEC(2, 32, 4, 2)


  { Variant v; v.name = "stream NOGATHER NL=1 (diagnostic)"; v.nl = 1; int pa = (dA.nblk + 7) / 8, pt = (dT.nblk + 7) / 8;
    v.launchA = [=]() { hipLaunchKernelGGL((k_spmv_nogather<1>), dim3(pa * 8), dim3(kBlock), 0, 0, dA.view, xn, ym, ym, ctl, part, pa); };
    v.launchT = [=]() { hipLaunchKernelGGL((k_spmv_nogather<1>), dim3(pt * 8), dim3(kBlock), 0, 0, dT.view, xm, yn, yn, ctl, part, pt); };
    vs.push_back(v); }
#define ADD_WIN(NLv, TT, SA, ST)                                                                                          \
  { Groups ga = make_groups(A, SA), gt = make_groups(T, ST);                                                              \
    constexpr int WCAPA = 12288 / NLv, WCAPT = 4096;                                                                      \
    if (ga.maxspan <= WCAPA && gt.maxspan <= WCAPT) {                                                                     \
    Variant v; v.name = "win NL=" #NLv " T=" #TT " SA=" #SA " ST=" #ST; v.nl = NLv;                                       \
    int pa = (ga.ng + 7) / 8, pt = (gt.ng + 7) / 8;                                                                       \
    size_t sha = (size_t)(WCAPA * NLv + kSpmvNnz * NLv + 16) * 8, sht = (size_t)(WCAPT * NLv + kSpmvNnz * NLv + 16) * 8;  \
    CK(hipFuncSetAttribute((const void*)k_spmv_win<NLv, TT, WCAPA>, hipFuncAttributeMaxDynamicSharedMemorySize, sha));    \
    CK(hipFuncSetAttribute((const void*)k_spmv_win<NLv, TT, WCAPT>, hipFuncAttributeMaxDynamicSharedMemorySize, sht));    \
    v.launchA = [=]() { hipLaunchKernelGGL((k_spmv_win<NLv, TT, WCAPA>), dim3(pa * 8), dim3(TT), sha, 0, dA.view, SA, ga.ng, ga.cmin, ga.span, xn, ym, ym, ctl, ctl, part, pa); }; \
    v.launchT = [=]() { hipLaunchKernelGGL((k_spmv_win<NLv, TT, WCAPT>), dim3(pt * 8), dim3(TT), sht, 0, dT.view, ST, gt.ng, gt.cmin, gt.span, xm, yn, yn, ctl, ctl, part, pt); }; \
    vs.push_back(v); } else printf("win NL=%d SA=%d ST=%d: span too large (%d, %d)\n", NLv, SA, ST, ga.maxspan, gt.maxspan); }
  ADD_WIN(1, 256, 4, 4) ADD_WIN(1, 256, 8, 8) ADD_WIN(1, 512, 8, 8) ADD_WIN(1, 512, 16, 8) ADD_WIN(1, 1024, 16, 16)
  ADD_WIN(2, 256, 1, 4) ADD_WIN(2, 512, 1, 8) ADD_WIN(2, 512, 1, 16)

  { Variant v; v.name = "PURE STREAM narrow (cols+vals)"; v.nl = 0;
    v.launchA = [=]() { hipLaunchKernelGGL(k_stream_narrow, dim3(2048), dim3(kBlock), 0, 0, dA.ci, dA.v, nnz, part); };
    v.launchT = [=]() { hipLaunchKernelGGL(k_stream_narrow, dim3(2048), dim3(kBlock), 0, 0, dT.ci, dT.v, nnz, part); };
    vs.push_back(v); }
  { Variant v; v.name = "PURE STREAM wide 16B"; v.nl = 0;
    v.launchA = [=]() { hipLaunchKernelGGL(k_stream_wide, dim3(2048), dim3(kBlock), 0, 0, (const int4*)dA.ci, (const double2*)dA.v, nnz / 4, part); };
    v.launchT = [=]() { hipLaunchKernelGGL(k_stream_wide, dim3(2048), dim3(kBlock), 0, 0, (const int4*)dT.ci, (const double2*)dT.v, nnz / 4, part); };
    vs.push_back(v); }
  { Variant v; v.name = "PURE STREAM wide 16B grid 8192"; v.nl = 0;
    v.launchA = [=]() { hipLaunchKernelGGL(k_stream_wide, dim3(8192), dim3(kBlock), 0, 0, (const int4*)dA.ci, (const double2*)dA.v, nnz / 4, part); };
    v.launchT = [=]() { hipLaunchKernelGGL(k_stream_wide, dim3(8192), dim3(kBlock), 0, 0, (const int4*)dT.ci, (const double2*)dT.v, nnz / 4, part); };
    vs.push_back(v); }
  { Variant v; v.name = "stream NT loads NL=1"; v.nl = 1; int pa = (dA.nblk + 7) / 8, pt = (dT.nblk + 7) / 8;
    v.launchA = [=]() { hipLaunchKernelGGL((k_spmv_ntgather<1>), dim3(pa * 8), dim3(kBlock), 0, 0, dA.view, xn, ym, ym, ctl, part, pa); };
    v.launchT = [=]() { hipLaunchKernelGGL((k_spmv_ntgather<1>), dim3(pt * 8), dim3(kBlock), 0, 0, dT.view, xm, yn, yn, ctl, part, pt); };
    vs.push_back(v); }
  { // rocSPARSE reference points (calibration only; never linked into the product)
    rocsparse_handle rh; rocsparse_create_handle(&rh);
    rocsparse_spmat_descr mA, mT; rocsparse_dnvec_descr vxn, vxm, vyn, vym;
    rocsparse_create_csr_descr(&mA, m, n, nnz, dA.rp, dA.ci, dA.v, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r);
    rocsparse_create_csr_descr(&mT, n, m, nnz, dT.rp, dT.ci, dT.v, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f64_r);
    rocsparse_create_dnvec_descr(&vxn, n, xn, rocsparse_datatype_f64_r); rocsparse_create_dnvec_descr(&vxm, m, xm, rocsparse_datatype_f64_r);
    rocsparse_create_dnvec_descr(&vyn, n, yn, rocsparse_datatype_f64_r); rocsparse_create_dnvec_descr(&vym, m, ym, rocsparse_datatype_f64_r);
    static double one = 1.0, half = 0.5;
    for (int alg_i = 0; alg_i < 2; ++alg_i) {
      rocsparse_spmv_alg alg = alg_i == 0 ? rocsparse_spmv_alg_csr_adaptive : rocsparse_spmv_alg_csr_rowsplit;
      size_t bsA = 0, bsT = 0; void *bufA = nullptr, *bufT = nullptr;
      rocsparse_spmv(rh, rocsparse_operation_none, &one, mA, vxn, &half, vym, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_buffer_size, &bsA, nullptr);
      rocsparse_spmv(rh, rocsparse_operation_none, &one, mT, vxm, &half, vyn, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_buffer_size, &bsT, nullptr);
      CK(hipMalloc(&bufA, bsA + 16)); CK(hipMalloc(&bufT, bsT + 16));
      rocsparse_spmv(rh, rocsparse_operation_none, &one, mA, vxn, &half, vym, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_preprocess, &bsA, bufA);
      rocsparse_spmv(rh, rocsparse_operation_none, &one, mT, vxm, &half, vyn, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_preprocess, &bsT, bufT);
      Variant v; v.name = alg_i == 0 ? "rocSPARSE csr_adaptive NL=1" : "rocSPARSE csr_rowsplit NL=1"; v.nl = 1;
      v.launchA = [=]() { size_t b = bsA; rocsparse_spmv(rh, rocsparse_operation_none, &one, mA, vxn, &half, vym, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_compute, &b, bufA); };
      v.launchT = [=]() { size_t b = bsT; rocsparse_spmv(rh, rocsparse_operation_none, &one, mT, vxm, &half, vyn, rocsparse_datatype_f64_r, alg, rocsparse_spmv_stage_compute, &b, bufT); };
      vs.push_back(v);
    }
  }

#define ADD_RGCS(NLv, GNA, RA, GNT, RT)                                                                                   \
  { PRgcsDev ra = build_rgcs(A, GNA, RA), rt = build_rgcs(T, GNT, RT);                                                     \
    Variant v; v.name = "rgcs NL=" #NLv " A(" #GNA "," #RA ") AT(" #GNT "," #RT ")"; v.nl = NLv;                         \
    int pa = (ra.view.ng + 7) / 8, pt = (rt.view.ng + 7) / 8;                                                             \
    v.launchA = [=]() { hipLaunchKernelGGL((k_spmv_rgcs_proto<NLv>), dim3(pa * 8), dim3(kBlock), 0, 0, ra.view, xn, ym, ym, ctl, ctl, part, pa); }; \
    v.launchT = [=]() { hipLaunchKernelGGL((k_spmv_rgcs_proto<NLv>), dim3(pt * 8), dim3(kBlock), 0, 0, rt.view, xm, yn, yn, ctl, ctl, part, pt); }; \
    vs.push_back(v); }
  ADD_RGCS(1, 6400, 64, 8192, 1024) ADD_RGCS(2, 6400, 64, 8192, 1024) ADD_RGCS(2, 12800, 128, 4096, 512) ADD_RGCS(2, 3200, 32, 2048, 256)
  ADD_RGCS(2, 25600, 256, 16384, 1024)

#define ADD_RGCS_ABL(NLv, ABLv, GNA, RA)                                                                                  \
  { PRgcsDev ra = build_rgcs(A, GNA, RA);                                                                                  \
    Variant v; v.name = "rgcs ABL=" #ABLv " NL=" #NLv " A(" #GNA "," #RA ")"; v.nl = NLv;                                 \
    int pa = (ra.view.ng + 7) / 8, pt = (dT.nblk + 7) / 8;                                                                \
    v.launchA = [=]() { hipLaunchKernelGGL((k_spmv_rgcs_proto<NLv, ABLv>), dim3(pa * 8), dim3(kBlock), 0, 0, ra.view, xn, ym, ym, ctl, ctl, part, pa); }; \
    v.launchT = [=]() { hipLaunchKernelGGL((k_spmv<NLv, 1>), dim3(pt * 8), dim3(kBlock), 0, 0, dT.view, xm, yn, yn, ctl, ctl, part, pt); }; \
    vs.push_back(v); }
  ADD_RGCS_ABL(2, 31, 12800, 128) ADD_RGCS_ABL(2, 63, 12800, 128) ADD_RGCS_ABL(2, 47, 12800, 128) ADD_RGCS_ABL(2, 63, 3200, 32) ADD_RGCS_ABL(2, 63, 51200, 512)
  ADD_RGCS_ABL(2, 0, 12800, 128) ADD_RGCS_ABL(2, 8, 12800, 128) ADD_RGCS_ABL(2, 15, 12800, 128) ADD_RGCS_ABL(2, 8, 6400, 64) ADD_RGCS_ABL(2, 8, 25600, 256)
  ADD_RGCS_ABL(1, 8, 12800, 128) ADD_RGCS_ABL(1, 15, 12800, 128)

#define ADD_PROBE(F, TPB)                                                                                                 \
  { Variant v; v.name = "probe FEAT=" #F " tiles/blk=" #TPB; v.nl = 0; int nt = nnz / kTile; int nb = (nt + TPB - 1) / TPB; int px = (nb + 7) / 8; \
    int grid = (F & 2) ? px * 8 : nb;                                                                                     \
    v.launchA = [=]() { hipLaunchKernelGGL((k_probe<F>), dim3(grid), dim3(kBlock), 0, 0, (const uint32_t*)dA.ci, dA.v, nt, TPB, px, part); }; \
    v.launchT = [=]() { hipLaunchKernelGGL((k_probe<F>), dim3(grid), dim3(kBlock), 0, 0, (const uint32_t*)dT.ci, dT.v, nt, TPB, px, part); }; \
    vs.push_back(v); }
  ADD_PROBE(0, 1) ADD_PROBE(0, 4) ADD_PROBE(1, 1) ADD_PROBE(1, 4) ADD_PROBE(2, 1) ADD_PROBE(2, 4) ADD_PROBE(3, 4) ADD_PROBE(3, 1) ADD_PROBE(0, 19) ADD_PROBE(1, 19)

#define ADD_RGCS_ABL_T(NLv, ABLv, GNT, RT)                                                                                \
  { PRgcsDev rt = build_rgcs(T, GNT, RT);                                                                                  \
    Variant v; v.name = "rgcsT ABL=" #ABLv " NL=" #NLv " AT(" #GNT "," #RT ")"; v.nl = NLv;                               \
    int pa = (dA.nblk + 7) / 8, pt = (rt.view.ng + 7) / 8;                                                                \
    v.launchA = [=]() { hipLaunchKernelGGL((k_spmv<NLv, 0>), dim3(pa * 8), dim3(kBlock), 0, 0, dA.view, xn, ym, ym, ctl, ctl, part, pa); }; \
    v.launchT = [=]() { hipLaunchKernelGGL((k_spmv_rgcs_proto<NLv, ABLv>), dim3(pt * 8), dim3(kBlock), 0, 0, rt.view, xm, yn, yn, ctl, ctl, part, pt); }; \
    vs.push_back(v); }
  ADD_RGCS_ABL_T(2, 8, 2048, 256) ADD_RGCS_ABL_T(2, 9, 2048, 256) ADD_RGCS_ABL_T(2, 10, 2048, 256) ADD_RGCS_ABL_T(2, 12, 2048, 256) ADD_RGCS_ABL_T(2, 15, 2048, 256)
  ADD_RGCS_ABL_T(2, 8, 8192, 1024) ADD_RGCS_ABL_T(2, 12, 8192, 1024) ADD_RGCS_ABL_T(2, 15, 8192, 1024) ADD_RGCS_ABL_T(1, 8, 8192, 1024)

  const char* filt = argc > 6 ? argv[6] : nullptr;
  for (auto& v : vs) {
    if (filt && v.name.find(filt) == std::string::npos) continue;
    for (int w = 0; w < 3; ++w) { v.launchA(); v.launchT(); }
    CK(hipDeviceSynchronize());
    double ta = 0, tt = 0;
    for (int r = 0; r < reps; ++r) {
      hipEventRecord(e0, 0); v.launchA(); hipEventRecord(e1, 0); v.launchT(); hipEventRecord(e2, 0);
      CK(hipEventSynchronize(e2));
      float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); ta += a; tt += b;
    }
    ta /= reps; tt /= reps;
    const double ba = 12.0 * nnz + 4.0 * (m + 1) + 8.0 * v.nl * (n + 2 * m), bt = 12.0 * nnz + 4.0 * (n + 1) + 8.0 * v.nl * (m + 2 * n);
    printf("%-32s  A: %7.1f us %6.0f GB/s   AT: %7.1f us %6.0f GB/s   pair: %7.1f us %6.0f GB/s\n", v.name.c_str(), ta * 1e3,
           ba / ta / 1e6, tt * 1e3, bt / tt / 1e6, (ta + tt) * 1e3, (ba + bt) / (ta + tt) / 1e6);
  }
  // cross-check: vec vs stream results agree
  CK(hipMemset(ym, 0, m * 16)); vs[0].launchA(); std::vector<double> r0(m), r1(m);
  CK(hipMemcpy(r0.data(), ym, m * 8, hipMemcpyDeviceToHost)); CK(hipMemset(ym, 0, m * 16)); vs[2].launchA();
  CK(hipMemcpy(r1.data(), ym, m * 8, hipMemcpyDeviceToHost));
  double md = 0; for (int64_t i = 0; i < m; ++i) md = std::max(md, std::fabs(r0[i] - r1[i]));
  printf("max |stream - vec| on A product: %.3e\n", md);
  for (auto& v : vs) if (v.name.rfind("rgcs NL=1", 0) == 0) {
    CK(hipMemset(ym, 0, m * 16)); v.launchA(); CK(hipMemcpy(r1.data(), ym, m * 8, hipMemcpyDeviceToHost));
    double md2 = 0; for (int64_t i = 0; i < m; ++i) md2 = std::max(md2, std::fabs(r0[i] - r1[i]));
    printf("max |stream - %s| on A product: %.3e\n", v.name.c_str(), md2);
    std::vector<double> t0(n), t1(n);
    CK(hipMemset(yn, 0, n * 16)); vs[0].launchT(); CK(hipMemcpy(t0.data(), yn, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemset(yn, 0, n * 16)); v.launchT(); CK(hipMemcpy(t1.data(), yn, n * 8, hipMemcpyDeviceToHost));
    md2 = 0; for (int64_t i = 0; i < n; ++i) md2 = std::max(md2, std::fabs(t0[i] - t1[i]));
    printf("max |stream - %s| on AT product: %.3e\n", v.name.c_str(), md2);
  }
  return 0;
}

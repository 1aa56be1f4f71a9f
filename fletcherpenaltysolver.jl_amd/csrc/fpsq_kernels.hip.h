// fpsq_kernels.hip.h -- gfx950 device code for libfpsq: CSR-stream SpMV/SpMM with fused axpby + norm partials,
// vector update kernels, deterministic reductions.  fp64 values, int32 indices, wave64.
//
// Layout of every Krylov vector: [len][NL] interleaved "lanes" (NL = 1: a plain vector; NL = 2: the two
// recurrences of a fused solve_two_* call side by side, so one 16-byte gather feeds both).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpsq {

constexpr int kBlock = 256;        // threads per workgroup (4 waves)
static_assert(kBlock == 256, "block_sum / block_sum_lanes and the tile-per-thread constants assume 4 waves");
#ifndef FPSQ_SPMV_NNZ
#define FPSQ_SPMV_NNZ 2048
#endif
constexpr int kSpmvNnz = FPSQ_SPMV_NNZ;  // nonzeros staged through LDS per workgroup
constexpr int kMaxRowsPerBlk = 1024;
constexpr int kEwBlocksMax = 1024; // grid cap for element-wise kernels (grid-stride beyond)

// Per-recurrence control block read by the generic kernels; written only by that recurrence's scalar kernels.
struct LaneCtl {
  double ca, cb;   // next product kernel: out = ca * (Mat x) + cb * yin
  double e[8];     // coefficients of the next element-wise update kernel
  int32_t done;    // recurrence finished: its kernels become no-ops
  int32_t upd_iter;// iteration whose update kernel still has to run after `done` was raised
  int32_t skip;    // skip the next product (its input vector is exactly zero)
  int32_t pad;
};

struct CsrView {
  const int32_t* rowptr;
  const int32_t* colind;
  const double* vals;
  const int32_t* rowblk;  // nblk + 1 row boundaries; a block has <= kSpmvNnz nonzeros unless it is ONE long row
  int32_t nblk;
  int32_t nrows;
  // optional compressed column indices: col = colbase[block] + col16[k] (built when every row block spans < 65536
  // columns -- always the case for A' of a banded Jacobian); halves the index stream of the product
  const uint16_t* col16;
  const int32_t* colbase;
  // per row block {first row, #rows, first nonzero, end nonzero}: ONE 16-byte load instead of the dependent chain
  // rowblk[L] -> rowptr[r0] at the head of every workgroup
  const int4* blkdesc;
  // column-sorted padded blocks (k_spmv<.., CSORT>): per stored entry its row-major slot and block-relative column,
  // cs16 = slot | (col & 31) << 11, cs8 = col >> 5
  const uint16_t* cs16;
  const uint8_t* cs8;
  // shared values (fpsq.hip pad_blocks): non-null = the blocks hold no values; `vals` is the row-group array of A and
  // segdesc[32 L + s] locates the 64 entries of segment s of block L in it (base = blkdesc[L].w)
  const uint4* segdesc;
  int32_t zero_pos;  // an entry of that array that always holds 0.0 (the lanes past a segment's valid entries read it)
  // the few blocks that cannot be described by segment descriptors keep 2048 values of their own here, block i at
  // [2048 i, ...), i = -128 - blkdesc[L].w (w <= -128 marks such a block)
  const double* vals_own;
};

// ------------------------------------------------------------------------------------------------ reductions

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over the workgroup; result valid in thread 0.  Fixed order => bitwise reproducible.
__device__ __forceinline__ double block_sum(double v, double* red /* >= 4 doubles of LDS */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would
// hold every wave until its outstanding global loads/stores (prefetched tiles, the epilogue's stores) are acknowledged.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Workgroup sums of NL per-thread values with ONE LDS-only barrier; totals valid in thread 0 (fixed order).
// `red` holds 4 * NL doubles and must not be in use by other waves when this is entered.
template <int NL>
__device__ __forceinline__ void block_sum_lanes(double* v, double* red) {
#pragma unroll
  for (int l = 0; l < NL; ++l) v[l] = wave_sum(v[l]);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) red[w * NL + l] = v[l];
  }
  lds_barrier();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) v[l] = (red[l] + red[NL + l]) + (red[2 * NL + l] + red[3 * NL + l]);
  }
}

// Deterministic sum of a partials array by ONE workgroup (used by the scalar kernels). Valid in thread 0.
__device__ __forceinline__ double reduce_partials(const double* p, int n, double* red) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) s += p[i];
  return block_sum(s, red);
}

// Sum of LDS products prod[j], j = first, first + stride, ... < end (NL interleaved values each), accumulated into
// acc.  Four independent LDS reads are kept in flight per step: a one-read-per-iteration loop exposes the full LDS
// latency on every element and dominated the A' product (rows of ~10 nonzeros, one lane per row).
template <int NL>
__device__ __forceinline__ void row_segment_sum(const double* prod, int first, int end, int stride, double* acc) {
  int j = first;
  for (; j + 3 * stride < end; j += 4 * stride) {
    if (NL == 1) {
      const double d0 = prod[j], d1 = prod[j + stride], d2 = prod[j + 2 * stride], d3 = prod[j + 3 * stride];
      acc[0] += (d0 + d1) + (d2 + d3);
    } else {
      const double2 d0 = *reinterpret_cast<const double2*>(prod + 2 * j);
      const double2 d1 = *reinterpret_cast<const double2*>(prod + 2 * (j + stride));
      const double2 d2 = *reinterpret_cast<const double2*>(prod + 2 * (j + 2 * stride));
      const double2 d3 = *reinterpret_cast<const double2*>(prod + 2 * (j + 3 * stride));
      acc[0] += (d0.x + d1.x) + (d2.x + d3.x);
      acc[NL - 1] += (d0.y + d1.y) + (d2.y + d3.y);
    }
  }
  // tail: up to 3 elements, still issued together
  {
    const bool k0 = j < end, k1 = j + stride < end, k2 = j + 2 * stride < end;
    const int j0 = k0 ? j : first, j1 = k1 ? j + stride : first, j2 = k2 ? j + 2 * stride : first;
    if (first < end) {
      if (NL == 1) {
        const double d0 = prod[j0], d1 = prod[j1], d2 = prod[j2];
        acc[0] += ((k0 ? d0 : 0.0) + (k1 ? d1 : 0.0)) + (k2 ? d2 : 0.0);
      } else {
        const double2 d0 = *reinterpret_cast<const double2*>(prod + 2 * j0);
        const double2 d1 = *reinterpret_cast<const double2*>(prod + 2 * j1);
        const double2 d2 = *reinterpret_cast<const double2*>(prod + 2 * j2);
        acc[0] += ((k0 ? d0.x : 0.0) + (k1 ? d1.x : 0.0)) + (k2 ? d2.x : 0.0);
        acc[NL - 1] += ((k0 ? d0.y : 0.0) + (k1 ? d1.y : 0.0)) + (k2 ? d2.y : 0.0);
      }
    }
  }
}

// Pairs of an interleaved [len][2] vector through buffer instructions with the agent-scope bit (sc1): a load that is coherent
// across the XCDs' L2s, a store that is written through to its coherence point -- what a workgroup needs to hand rows to a
// workgroup on another XCD INSIDE a launch (k_iter_fused; the fences of the memory model cost an L2 write-back / invalidate
// per workgroup there: profiles/r04_coherence_whatif.txt).  Offsets are 32-bit: callers guarantee 16 len < 2^31.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pair_rsrc(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0x7fffffff, 0x00027000);
}
__device__ __forceinline__ double2 ld_pair_ag(const double* base, int row) {
  const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(pair_rsrc(base), row * 16, 0, /*sc1*/ 16);
  double2 d;
  __builtin_memcpy(&d, &w, 16);
  return d;
}
__device__ __forceinline__ void st_pair_wt(double* base, int row, double a, double b) {
  const double2 ov = make_double2(a, b);
  u32x4_t w;
  __builtin_memcpy(&w, &ov, 16);
  __builtin_amdgcn_raw_buffer_store_b128(w, pair_rsrc(base), row * 16, 0, /*sc1*/ 16);
}

// Row epilogue shared by the product kernels: out = ca * acc + cb * yin for the active lanes, squared-norm
// accumulation.  The yin values of both lanes are fetched with one (16-byte when NL = 2) load before any arithmetic
// and written back with one store when both lanes are active -- a per-lane load/use/store chain costs two dependent
// memory round trips per row.
// AG (several iterations per launch): yin was written by ANOTHER workgroup earlier in the same launch -- read it at agent scope
// (a plain load may hit a line this CU's L1 has kept from two iterations ago; the L1s are only invalidated between kernels)
template <int NL, bool WT = false, bool AG = false>
__device__ __forceinline__ void row_epilogue(size_t row, const double* acc, const double* ca, const double* cb,
                                             const bool* act, const double* yin, double* yout, double* sq,
                                             const double* ypre = nullptr) {
  double yv[NL];
  bool need = false;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    yv[l] = 0.0;
    need |= act[l] && cb[l] != 0.0;
  }
  if (ypre) {  // fetched at the head of the workgroup, one memory round trip ahead of this point
#pragma unroll
    for (int l = 0; l < NL; ++l) yv[l] = ypre[l];
  } else if (need) {
    if (NL == 2) {
      double2 t;
      if (AG) t = ld_pair_ag(yin, (int)row);
      else t = *reinterpret_cast<const double2*>(yin + row * 2);
      yv[0] = t.x;
      yv[NL - 1] = t.y;
    } else {
      yv[0] = yin[row];
    }
  }
  double o[NL];
  bool all = true;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    o[l] = ca[l] * acc[l] + (cb[l] != 0.0 ? cb[l] * yv[l] : 0.0);
    all &= act[l];
    if (act[l]) sq[l] += o[l] * o[l];
  }
  if (NL == 2 && all) {
    if (WT) {  // written through (one 16-byte buffer store, agent scope): see k_iter_fused
      st_pair_wt(yout, (int)row, o[0], o[NL - 1]);
    } else {
      *reinterpret_cast<double2*>(yout + row * 2) = make_double2(o[0], o[NL - 1]);
    }
  } else {
#pragma unroll
    for (int l = 0; l < NL; ++l)
      if (act[l]) {
        if (WT) __hip_atomic_store(yout + row * NL + l, o[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else yout[row * NL + l] = o[l];
      }
  }
}

// ---- Krylov vector updates.  Several independent updates (the LSQR x/w update of the previous iteration, the CRAIG
// long and short updates of this one) are merged into ONE launch: consecutive workgroup ranges run different bodies.

enum UpdKind : int32_t {
  UPD_NONE = 0, UPD_LSQR, UPD_LSQR_WINIT, UPD_CRAIG_LONG_REG, UPD_CRAIG_LONG, UPD_CRAIG_SHORT,
  UPD_NEG_COPY,  // a[i] = -src[i][lane]: keeps c = -(CRAIG's right-hand side) when the start-up product formed it
  UPD_MINRES_E1, UPD_MINRES_E2, UPD_MINRES_E3,  // the three element-wise stages of a MINRES iteration (upd_minres)
  UPD_LNLQ_LONG, UPD_LNLQ_SHORT                 // LNLQ: x and (y, wbar) updates (upd_lnlq_*)
};

struct UpdSeg {
  int32_t kind;
  int32_t it;        // iteration this update belongs to (see LaneCtl::upd_iter)
  const LaneCtl* ctl;
  const double* src; // interleaved Golub-Kahan vector [len][NL]
  int32_t lane;      // which interleaved lane of src
  int32_t nblk;      // workgroups assigned to this segment
  int32_t pad_[2];
  // Speculatively enqueued final flush: runs only if this recurrence AND the one behind `gate` have ended (null:
  // ungated).  While any lane of the call still iterates the loop goes on and the next product launch carries this
  // update -- it must not be applied twice.
  const LaneCtl* gate;
  double* a;         // LSQR: x      CRAIG long: xs      CRAIG short: w      MINRES: r1 (E1), r2 (E2), w (E3)
  double* b;         // LSQR: w      CRAIG long: w2s     CRAIG short: y      MINRES: r2 (E1), r_new (E2), x (E3)
  double* c;         // MINRES E2: w2 (read)
  double* d;         // MINRES E2: w1 (read, overwritten by the new w~)
  int64_t len;
  double* partials;  // [nblk] partial sums of ||w_new||^2 (LSQR, CRAIG short)
};

// ... and, in that mode, the vectors an update rewrites in place (x, w, y): the workgroup of the same index of the PREVIOUS iteration
// wrote them, on whatever CU / XCD it ran -- loads at agent scope, stores written through
template <bool AG>
__device__ __forceinline__ double upd_ld(const UpdSeg& s, const double* p, int64_t idx) {
  if constexpr (AG)
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p + idx), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
  return p[idx];
}
template <bool AG>
__device__ __forceinline__ void upd_st(const UpdSeg& s, double* p, int64_t idx, double v) {
  if constexpr (AG) __hip_atomic_store(reinterpret_cast<unsigned long long*>(p + idx), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
  else p[idx] = v;
}
template <bool AG>
__device__ __forceinline__ double upd_src(const UpdSeg& s, int64_t idx) {
  if constexpr (AG)
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(s.src + idx), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT));
  return s.src[idx];
}

// LSQR (Krylov.jl lsqr!): x += (phi/rho) w; w = v - (theta/rho) w, with v = vt / alpha deferred.
//   e[0] = phi/rho, e[1] = theta/rho, e[2] = 1/alpha.   WINIT: w = vt / alpha only (w_1 = v_1).
// AG (several iterations per launch, k_iter_multi): the operands were written by OTHER workgroups earlier in the same launch -- loads
// at agent scope, in-place stores written through (see upd_ld).  A template parameter: as a run-time switch in these inner loops it
// cost every one-launch iteration ~2 us (measured against round 4's library on one box: 978 -> 945 evals/s).
template <int NL, bool WINIT, bool AG = false>
__device__ __forceinline__ void upd_lsqr(const UpdSeg& s, int blk, double* red, const LaneCtl* ctl) {
  if (WINIT && ctl->done) {  // the recurrence ended at start-up (b = 0 or B'b = 0): the solution is x = 0
    for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < s.len; i += (int64_t)s.nblk * kBlock) s.a[i] = 0.0;
    return;
  }
  if (!WINIT && ctl->done && ctl->upd_iter != s.it) return;
  if (!WINIT && s.gate != nullptr && !(ctl->done && s.gate->done)) return;  // the call still iterates
  const double sg = ctl->e[0], tr = ctl->e[1], ia = ctl->e[2];
  double sq = 0.0;
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < s.len; i += (int64_t)s.nblk * kBlock) {
    double wn;
    if (WINIT) {
      wn = upd_src<AG>(s, i * NL + s.lane) * ia;
      s.a[i] = 0.0;  // x_0 = 0
    } else {
      const double wi = upd_ld<AG>(s, s.b, i);
      upd_st<AG>(s, s.a, i, upd_ld<AG>(s, s.a, i) + sg * wi);
      wn = upd_src<AG>(s, i * NL + s.lane) * ia - tr * wi;
    }
    upd_st<AG>(s, s.b, i, wn);
    sq += wn * wn;
  }
  const double t = block_sum(sq, red);
  if (threadIdx.x == 0) s.partials[blk] = t;
}

// CRAIG (Krylov.jl craig!), long (n) part.  `xs` accumulates sgn * x (sgn = -1 gives p2 = -x directly,
// src/solve_linear_system.jl:133).  With v = vt / alpha and the true w2 = omega * w2s (the scaling by s2 is deferred):
//   lambda > 0:  xs += e0 * vt + e1 * w2s;   w2s = e2 * vt + e3 * w2s
//       e0 = sgn xi c1 / alpha, e1 = sgn xi s1 omega, e2 = s1 / alpha, e3 = -c1 omega
//   lambda = 0:  xs += e0 * vt                      (e0 = sgn xi / alpha)
template <int NL, bool REG, bool AG = false>
__device__ __forceinline__ void upd_craig_long(const UpdSeg& s, int blk, const LaneCtl* ctl) {
  if (ctl->done && ctl->upd_iter != s.it) return;
  const double e0 = ctl->e[0], e1 = ctl->e[1], e2 = ctl->e[2], e3 = ctl->e[3];
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < s.len; i += (int64_t)s.nblk * kBlock) {
    const double v = upd_src<AG>(s, i * NL + s.lane);
    if (REG) {
      const double w2 = upd_ld<AG>(s, s.b, i);
      upd_st<AG>(s, s.a, i, upd_ld<AG>(s, s.a, i) + (e0 * v + e1 * w2));
      upd_st<AG>(s, s.b, i, e2 * v + e3 * w2);
    } else {
      upd_st<AG>(s, s.a, i, upd_ld<AG>(s, s.a, i) + e0 * v);
    }
  }
}

// CRAIG short (m) part:  w = u - (theta/rho_prev) w with u = e4 * mut (= mu Mu~ / beta);  y += e6 * w.
//   e4 = mu/beta, e5 = theta/rho_prev, e6 = xi/rho
template <int NL, bool AG = false>
__device__ __forceinline__ void upd_craig_short(const UpdSeg& s, int blk, double* red, const LaneCtl* ctl) {
  if (ctl->done && ctl->upd_iter != s.it) return;
  const double e4 = ctl->e[4], e5 = ctl->e[5], e6 = ctl->e[6];
  double sq = 0.0;
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < s.len; i += (int64_t)s.nblk * kBlock) {
    const double wn = e4 * upd_src<AG>(s, i * NL + s.lane) - e5 * upd_ld<AG>(s, s.a, i);
    upd_st<AG>(s, s.a, i, wn);
    upd_st<AG>(s, s.b, i, upd_ld<AG>(s, s.b, i) + e6 * wn);
    sq += wn * wn;
  }
  const double t = block_sum(sq, red);
  if (threadIdx.x == 0) s.partials[blk] = t;
}

// MINRES (Krylov.jl minres!, M = I) on (A A' + lambda I): the three element-wise stages of an iteration on m-vectors.
// The Lanczos vector under construction lives in lane `lane` of the short pair `src` (q after the A product).
//   E1: y0 = q - (beta/oldbeta) r1                              partial <r2, y0>   -> alpha          (e0)
//   E2: y = y0 - (alpha/beta) r2;  r_new = y (also back into the pair);
//       w~ = r2/beta - delta w2 - eps w1  (stored over w1)      partial ||y||^2    -> beta_new       (e1..e4)
//   E3: w = w~ / gamma; x += phi w                              partial ||x||^2    -> stopping tests (e5, e6)
template <int NL, int STAGE>
__device__ __forceinline__ void upd_minres(const UpdSeg& s, int blk, double* red, const LaneCtl* ctl) {
  if (STAGE == 3) {
    if (ctl->done && ctl->upd_iter != s.it) return;
    if (ctl->upd_iter != s.it) return;  // (stage B of this iteration did not run: nothing to apply)
  } else if (ctl->done) {
    return;
  }
  double* sp = const_cast<double*>(s.src);
  const double e0 = ctl->e[0], e1 = ctl->e[1], e2 = ctl->e[2], e3 = ctl->e[3], e4 = ctl->e[4], e5 = ctl->e[5],
               e6 = ctl->e[6];
  double acc = 0.0;
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < s.len; i += (int64_t)s.nblk * kBlock) {
    if (STAGE == 1) {
      const double y0 = sp[i * NL + s.lane] - (e0 != 0.0 ? e0 * s.a[i] : 0.0);
      sp[i * NL + s.lane] = y0;
      acc += s.b[i] * y0;
    } else if (STAGE == 2) {
      const double r = s.a[i];
      const double y = sp[i * NL + s.lane] - e1 * r;
      sp[i * NL + s.lane] = y;
      s.b[i] = y;
      s.d[i] = e2 * r - e3 * s.c[i] - e4 * s.d[i];
      acc += y * y;
    } else {
      const double w = s.a[i] * e5;
      s.a[i] = w;
      const double xn = s.b[i] + e6 * w;
      s.b[i] = xn;
      acc += xn * xn;
    }
  }
  const double t = block_sum(acc, red);
  if (threadIdx.x == 0) s.partials[blk] = t;
}

// LNLQ (Krylov.jl lnlq!, lambda = 0), long (n) part: xs += e0 * vt  with vt = alpha v the stored Golub-Kahan vector
//   e0 = sgn tau_k / alpha_k while iterating; at the end the transfer step sgn tau / alpha (CRAIG point) or
//   sgn eta zeta / alpha (LQ point).
template <int NL>
__device__ __forceinline__ void upd_lnlq_long(const UpdSeg& s, int blk, const LaneCtl* ctl) {
  if (ctl->done && ctl->upd_iter != s.it) return;
  const double e0 = ctl->e[0];
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < s.len; i += (int64_t)s.nblk * kBlock)
    s.a[i] += e0 * s.src[i * NL + s.lane];
}

// LNLQ short (m) part, with u = e4 * mut (= mu Mu~ / beta):
//   y += e5 * wbar + e6 * u            (= zeta_k w_k,  w_k = c wbar + s u)
//   wbar <- e2 * wbar + e3 * u         (= s wbar - c u)
//   y += e7 * wbar_new                 (transfer to the CRAIG point when the recurrence ends there, else e7 = 0)
template <int NL>
__device__ __forceinline__ void upd_lnlq_short(const UpdSeg& s, int blk, const LaneCtl* ctl) {
  if (ctl->done && ctl->upd_iter != s.it) return;
  const double e2 = ctl->e[2], e3 = ctl->e[3], e4 = ctl->e[4], e5 = ctl->e[5], e6 = ctl->e[6], e7 = ctl->e[7];
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < s.len; i += (int64_t)s.nblk * kBlock) {
    const double u = e4 * s.src[i * NL + s.lane];
    const double wb = s.a[i];
    const double wn = e2 * wb + e3 * u;
    s.a[i] = wn;
    s.b[i] += (e5 * wb + e6 * u) + (e7 != 0.0 ? e7 * wn : 0.0);
  }
}

// phi = f - c'ys + rho/2 c'c + eta/2 ||x - xk||^2 from the partial sums of the evaluation
// (src/model-Fletcherpenaltynlp.jl:419-433).  out = {phi, f, c'c, seq}: `seq` is stored LAST with system-scope release
// semantics, so a host that polls it (stream-ordered outputs, see fpsq_set_output_ordering) finds the three values there.
// stride: distance (doubles) between consecutive entries of every array (4 when the arrays are the ranks' gathered
// quadruples of a sharded run: entry r of array k at [4 r + k]; summed in rank order)
// xt (row-sharded runs whose sums over the ranks are formed inside the launch, fpsq_krylov.hip.h xch_sum): the arrays are this
// rank's local partials; the four local sums travel as exchange `xseq` and are added in rank order.  red16: 16 + 36 doubles then.
struct XchTable;
template <int NV>
__device__ __forceinline__ void xch_sum(const XchTable* xt, unsigned int seq, int half, double (&v)[NV], double* red);
__device__ __forceinline__ void qp_fx_core(const double* pf, const double* pdx, const double* pcy, const double* pcc, int np_n,
                                           int np_m, double rho, double eta, double* out, double seq, double* red16,
                                           int stride = 1, const XchTable* xt = nullptr, unsigned int xseq = 0) {
  // All four arrays in ONE batch of loads (<= 4 entries per thread and array: the grids of the kernels that wrote them
  // are capped at kEwBlocksMax = 4 x kBlock), unconditional with clamped indices like k_step's partial_batch; a longer
  // array falls back to the strided loop.  Fixed summation order.
  const double* arr[4] = {pf, pdx, pcy, pcc};
  const int cnt[4] = {np_n, np_n, np_m, np_m};
  const int t = threadIdx.x;
  double v[4];
  if (np_n <= 4 * kBlock && np_m <= 4 * kBlock) {
    double x[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = u * kBlock + t;
        x[k][u] = cnt[k] > 0 ? arr[k][(size_t)(i < cnt[k] ? i : cnt[k] - 1) * stride] : 0.0;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = 0.0;
#pragma unroll
      for (int u = 0; u < 4; ++u) v[k] += (u * kBlock + t < cnt[k]) ? x[k][u] : 0.0;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = 0.0;
      for (int i = t; i < cnt[k]; i += kBlock) v[k] += arr[k][(size_t)i * stride];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
  const int lane = t & 63, w = t >> 6;
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) red16[w * 4 + k] = v[k];
  }
  __syncthreads();
  double q4[4] = {0.0, 0.0, 0.0, 0.0};
  if (t == 0) {
    q4[0] = (red16[0] + red16[4]) + (red16[8] + red16[12]);
    q4[1] = (red16[1] + red16[5]) + (red16[9] + red16[13]);
    q4[2] = (red16[2] + red16[6]) + (red16[10] + red16[14]);
    q4[3] = (red16[3] + red16[7]) + (red16[11] + red16[15]);
  }
  if (xt != nullptr) xch_sum<4>(xt, xseq, 0, q4, red16 + 16);  // (uniform)
  if (t == 0) {
    const double f = q4[0], dx = q4[1], cy = q4[2], cc = q4[3];
    double fx = f - cy;
    if (rho > 0.0) fx += rho / 2 * cc;
    if (eta > 0.0) fx += eta / 2 * dx;
    out[0] = fx;
    out[1] = f;
    out[2] = cc;
    __hip_atomic_store(out + 3, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ctl: the recurrence's control block the update reads its coefficients from -- s.ctl in global memory, or (product kernels
// with riding leaders) the copy the workgroup has taken from the leaders' record into LDS
// the segments of a k_iter_multi launch (LSQR / CRAIG lanes only), operands at agent scope
template <int NL>
__device__ __forceinline__ void upd_run_ag(const UpdSeg& s, int blk, double* red, const LaneCtl* ctl) {
  switch (s.kind) {
    case UPD_LSQR: upd_lsqr<NL, false, true>(s, blk, red, ctl); break;
    case UPD_CRAIG_LONG_REG: upd_craig_long<NL, true, true>(s, blk, ctl); break;
    case UPD_CRAIG_LONG: upd_craig_long<NL, false, true>(s, blk, ctl); break;
    case UPD_CRAIG_SHORT: upd_craig_short<NL, true>(s, blk, red, ctl); break;
    default: break;
  }
}
template <int NL>
__device__ __forceinline__ void upd_run(const UpdSeg& s, int blk, double* red, const LaneCtl* ctl) {
  switch (s.kind) {
    case UPD_LNLQ_LONG: upd_lnlq_long<NL>(s, blk, ctl); break;
    case UPD_LNLQ_SHORT: upd_lnlq_short<NL>(s, blk, ctl); break;
    case UPD_MINRES_E1: upd_minres<NL, 1>(s, blk, red, ctl); break;
    case UPD_MINRES_E2: upd_minres<NL, 2>(s, blk, red, ctl); break;
    case UPD_MINRES_E3: upd_minres<NL, 3>(s, blk, red, ctl); break;
    case UPD_LSQR: upd_lsqr<NL, false>(s, blk, red, ctl); break;
    case UPD_LSQR_WINIT: upd_lsqr<NL, true>(s, blk, red, ctl); break;
    case UPD_CRAIG_LONG_REG: upd_craig_long<NL, true>(s, blk, ctl); break;
    case UPD_CRAIG_LONG: upd_craig_long<NL, false>(s, blk, ctl); break;
    case UPD_CRAIG_SHORT: upd_craig_short<NL>(s, blk, red, ctl); break;
    case UPD_NEG_COPY:
      for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < s.len; i += (int64_t)s.nblk * kBlock)
        s.a[i] = -s.src[i * NL + s.lane];
      break;
    default: break;
  }
}
template <int NL>
__device__ __forceinline__ void upd_run(const UpdSeg& s, int blk, double* red) { upd_run<NL>(s, blk, red, s.ctl); }

// Update segments riding in a product launch ("horizontal fusion"): the workgroups past the product's own grid
// (they are dispatched last and fill the product's tail) run vector updates that only READ what the product reads, so one launch replaces two and the streaming updates
// overlap the gather-bound product.  Returns true when this workgroup was an update workgroup.
// lds0 / lds1 (riding leaders): the lanes' control blocks as taken from the leaders' record (null: the segments' own)
template <int NL>
__device__ __forceinline__ bool run_fused_updates(const UpdSeg& u0, const UpdSeg& u1, int nprod, double* red,
                                                  const LaneCtl* lds0 = nullptr, const LaneCtl* lds1 = nullptr) {
  const int blk = (int)blockIdx.x - nprod;
  if (blk < 0) return false;
  const UpdSeg& u = blk < u0.nblk ? u0 : u1;
  const LaneCtl* ctl = lds0 ? (u.lane == 0 ? lds0 : lds1) : u.ctl;
  upd_run<NL>(u, blk < u0.nblk ? blk : blk - u0.nblk, red, ctl);
  return true;
}

// up to four vectors zeroed by the start-up launch (k_startup)
struct ZeroArgs {
  double* p[4];
  int64_t n[4];
};
// vals_out[t] = vals_in[perm[t]]  (refresh of the A' copy when the Jacobian values change)
__global__ __launch_bounds__(kBlock) void k_gather(const double* __restrict__ in, const int32_t* __restrict__ perm,
                                                   double* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int32_t p = perm[i];
    out[i] = p >= 0 ? in[p] : 0.0;  // (< 0: padding slot of a padded block layout)
  }
}

// ONE-PASS JACOBIAN REFRESH (fpsq_set_jacobian_values; the reference refreshes its operator at every new x:
// src/solve_linear_system.jl:118-122, :223-228).  Every stored copy of the values -- the column-sorted row groups of the A
// product, the (padded, column-sorted) row blocks of A', the CSR array when the CSR fallback serves a product -- is a
// segment of ONE launch: out[i] = in[perm[i]] with perm composed at structure time down to the CALLER's array (COO order or
// CSR order), -1 = a padding slot (0.0), perm == null = the identity.  A workgroup fills one chunk of 2048 consecutive
// destination slots: eight coalesced index loads per thread, then its eight gathers back to back, then eight coalesced
// stores (a load-gather-store loop per element is one dependent round trip each: the three grid-stride passes this
// replaces took ~95 us apiece at the headline size, latency-bound at ~1.3 TB/s).  Chunks are dealt to the XCDs in
// contiguous eighths: consecutive chunks of A' gather from the same few hundred rows of the caller's array, which that
// XCD's L2 then holds.  Pure copies: bitwise the three-pass result (FPSQ_JAC_REFRESH=3 keeps it, under test).
// (Measured and dropped: the same refresh in SOURCE order -- stream the caller's array, scatter through inverse
// permutations: 123 us against 81-87 us for this kernel at the headline size.)
struct RefreshSeg {
  double* out;
  const int32_t* perm;
  int64_t n;
  int32_t nchunk, pad_;
};
constexpr int kRefreshChunk = 2048;
__global__ __launch_bounds__(kBlock) void k_refresh(const double* __restrict__ in, const RefreshSeg s0, const RefreshSeg s1,
                                                    const RefreshSeg s2, int per_xcd) {
  int c = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
  // (the segments are kernel arguments: selected by branch, never through a pointer)
  double* out;
  const int32_t* perm;
  int64_t n;
  if (c < s0.nchunk) {
    out = s0.out, perm = s0.perm, n = s0.n;
  } else if (c < s0.nchunk + s1.nchunk) {
    c -= s0.nchunk;
    out = s1.out, perm = s1.perm, n = s1.n;
  } else if (c < s0.nchunk + s1.nchunk + s2.nchunk) {
    c -= s0.nchunk + s1.nchunk;
    out = s2.out, perm = s2.perm, n = s2.n;
  } else {
    return;
  }
  constexpr int kPer = kRefreshChunk / kBlock;
  const int64_t base = (int64_t)c * kRefreshChunk + threadIdx.x;
  int32_t p[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int64_t i = base + j * kBlock;
    const int64_t ii = i < n ? i : n - 1;
    p[j] = perm ? perm[ii] : (int32_t)ii;
  }
  double v[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) v[j] = in[p[j] >= 0 ? p[j] : 0];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int64_t i = base + j * kBlock;
    if (i < n) out[i] = p[j] >= 0 ? v[j] : 0.0;
  }
}

// perm[i] <- inner[perm[i]] (structure time: a value permutation composed with the COO -> CSR order; -1 stays -1)
__global__ __launch_bounds__(kBlock) void k_compose_perm(int32_t* perm, const int32_t* __restrict__ inner, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int32_t q = perm[i];
    if (q >= 0) perm[i] = inner[q];
  }
}

// CSR slot value = sum of the COO entries that map to it, in sorted (fixed) order: duplicates are summed like
// SparseArrays.sparse does, deterministically.
__global__ __launch_bounds__(kBlock) void k_gather_sum(const double* __restrict__ coo, const int32_t* __restrict__ perm,
                                                       const int32_t* __restrict__ slotptr, double* __restrict__ out,
                                                       int64_t nslots) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nslots; i += (int64_t)gridDim.x * kBlock) {
    double s = 0.0;
    for (int k = slotptr[i]; k < slotptr[i + 1]; ++k) s += coo[perm[k]];
    out[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------ vector kernels

// (the segments are kernel arguments: select by branch, never through a pointer, or they are spilled to scratch)
template <int NL>
__global__ __launch_bounds__(kBlock) void k_updates(const UpdSeg s0, const UpdSeg s1, const UpdSeg s2) {
  __shared__ double red[4];
  const int blk = blockIdx.x;
  if (blk < s0.nblk) {
    upd_run<NL>(s0, blk, red);
  } else if (blk < s0.nblk + s1.nblk) {
    upd_run<NL>(s1, blk - s0.nblk, red);
  } else {
    upd_run<NL>(s2, blk - s0.nblk - s1.nblk, red);
  }
}

// ---- multi-GPU (row-sharded A) helpers

// After the all-reduce of the raw partial products P = sum_r A_r' x_r:  y = ca * P + cb * y with the squared-norm
// partials -- the epilogue the single-GPU product kernel fuses.  Runs on replicated n-vectors.
// sum_len <= len: the norm is taken over the rank's OWNED prefix only (halo mode: the tail of the window is owned by the
// right neighbour and counted there).
template <int NL>
__global__ __launch_bounds__(kBlock) void k_axpby_norm(const double* __restrict__ P, double* y, const LaneCtl* ctl0,
                                                       const LaneCtl* ctl1, int64_t len, int64_t sum_len,
                                                       double* partials) {
  const LaneCtl* c[2] = {ctl0, ctl1};
  bool act[NL];
  double ca[NL], cb[NL];
  bool any = false;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    act[l] = !(c[l]->done | c[l]->skip);
    ca[l] = c[l]->ca;
    cb[l] = c[l]->cb;
    any |= act[l];
  }
  if (!any) return;
  __shared__ double red[4];
  double sq[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      if (act[l]) {
        const double o = ca[l] * P[i * NL + l] + (cb[l] != 0.0 ? cb[l] * y[i * NL + l] : 0.0);
        y[i * NL + l] = o;
        if (i < sum_len) sq[l] += o * o;
      }
    }
  }
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const double t = block_sum(sq[l], red);
    if (threadIdx.x == 0) partials[(size_t)l * gridDim.x + blockIdx.x] = t;
  }
}

// Halo mode, second half of the row-sharded A' product.  k_spmv<.., HALO> finalised the interior rows of the rank's column
// window and left the raw sums of its two overlap regions in `raw` ([tl + tr][NL], head region first); `recv` holds the
// neighbours' raw sums of the same global columns (same layout).  Here, for the overlap rows,
//   yout[row] = ca * (raw + recv) + cb * yin[row]      (a + b on one rank, b + a on the other: bitwise identical)
// with the squared-norm partials of the HEAD region only (owned by this rank; the tail region is owned -- and counted --
// by the right neighbour).  partials[l * pstride + blockIdx.x], may be null.
template <int NL>
__global__ __launch_bounds__(kBlock) void k_halo_finish(const double* __restrict__ raw, const double* __restrict__ recv,
                                                        int64_t tl, int64_t tr, int64_t tail0, const double* yin,
                                                        double* yout, const LaneCtl* ctl0, const LaneCtl* ctl1,
                                                        double* partials, int pstride, const LaneCtl* gate0,
                                                        const LaneCtl* gate1) {
  if (gate0 != nullptr && !(gate0->done && gate1->done)) return;
  const LaneCtl* c[2] = {ctl0, ctl1};
  bool act[NL];
  double ca[NL], cb[NL];
  bool any = false;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    act[l] = !(c[l]->done | c[l]->skip);
    ca[l] = c[l]->ca;
    cb[l] = c[l]->cb;
    any |= act[l];
  }
  if (!any) return;
  __shared__ double red[4];
  double sq[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = 0.0;
  double sq_tail[NL];  // (the tail region is owned -- and counted -- by the right neighbour)
#pragma unroll
  for (int l = 0; l < NL; ++l) sq_tail[l] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < tl + tr; i += (int64_t)gridDim.x * kBlock) {
    const int64_t row = i < tl ? i : tail0 + (i - tl);
    double acc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] = raw[i * NL + l] + recv[i * NL + l];
    // (ONE row routine for the three forms of the finish -- this kernel, k_p2p_halo_finish, the halo workgroups of k_iter_fused:
    // the same instructions, the same bits)
    row_epilogue<NL>((size_t)row, acc, ca, cb, act, yin, yout, i < tl ? sq : sq_tail);
  }
  if (partials != nullptr) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const double t = block_sum(sq[l], red);
      if (threadIdx.x == 0) partials[(size_t)l * pstride + blockIdx.x] = t;
    }
  }
}

// recv[r * count + i] = src_r[i]: the in-process all-gather over the logical shards of one GPU (LocalComm)
struct GatherSrc {
  const double* s[8];
  int32_t n;
};
__global__ __launch_bounds__(kBlock) void k_local_allgather(GatherSrc S, double* recv, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < (int64_t)S.n * count; i += (int64_t)gridDim.x * kBlock) {
    const int r = (int)(i / count);
    recv[i] = S.s[r][i - (int64_t)r * count];
  }
}

// ---- peer-to-peer route of the halo-sharded loop (fpsq.hip: P2PRoute; P2PLocalComm = the shards of ONE process on one GPU,
// IpcComm = the ranks of a node with hipIpc-mapped pointers, the stores then travel over xGMI).  No collective library
// call: a rank WRITES its record straight into its peers' receive areas, then a sequence number into their flag words, and
// waits -- in the same small kernel -- until its own flag words carry that number (the guide's "handoff-flag": plain
// payload -> release fence -> flag; bounded spins: a missing peer ends in an error flag, never in a hang).
// max_spins: the bound of every wait (~1-2 us per poll); *fail != 0 on entry (an earlier exchange of the call gave up): leave
// at once -- the call is lost anyway and must not pay one waiting time per exchange still enqueued.
struct P2PPeers {
  double* rx[8];                // peer p's receive area (this parity): [sender][count]
  unsigned long long* flag[8];  // peer p's flag words (one per sender)
  int32_t n;
};
// Workgroup barrier behind which EVERY wave's stores have been acknowledged.  __syncthreads() is not that: its release fence has
// workgroup scope, which on this hardware (waves of a workgroup share their CU's L1) waits for no store at all -- a flag that
// thread 0 raises behind it, however it fences, can overtake the other waves' data (vmcnt is a per-wave counter).
__device__ __forceinline__ void barrier_stores_done() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ bool p2p_wait(const unsigned long long* f, unsigned long long seq, long max_spins, int* fail) {
  long spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
    if (++spins > max_spins) {
      __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  __threadfence_system();  // (acquire: the record behind the flag)
  return true;
}
// dst[i] = src[i], i in [lo, hi), by the calling workgroup: eight loads per thread issued back to back (unconditional, clamped
// index) before the first store -- a load-then-store loop pays one memory round trip per element
__device__ __forceinline__ void p2p_copy(const double* __restrict__ src, double* dst, int64_t lo, int64_t hi) {
  for (int64_t base = lo; base < hi; base += 8 * kBlock) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = base + u * kBlock + threadIdx.x;
      v[u] = src[i < hi ? i : hi - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = base + u * kBlock + threadIdx.x;
      if (i < hi) dst[i] = v[u];
    }
  }
}
// All-gather, ONE WORKGROUP PER PEER (the links of a node are point to point: every pair has its own):
//   workgroup p != rank: rx_p[rank][.] = send[.], release, flag_p[rank] = seq; wait for my flag[p]; local[p][.] = my_rx[p][.]
//   workgroup rank, and the workgroups past P.n (a long record is cut into slices): local[rank][.] = send[.]
// `local` is an ordinary device buffer: what the scalar steps read afterwards -- sixteen leader workgroups at the head of
// every product launch, on its critical path -- never is the fine-grained (uncached) receive area the peers write into.
__global__ __launch_bounds__(kBlock) void k_p2p_gather(const double* __restrict__ send, int64_t count, P2PPeers P, int rank,
                                                       unsigned long long seq, const double* my_rx, double* local, int* fail,
                                                       long max_spins) {
  __shared__ int ok;
  if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
  const int p = blockIdx.x;
  if (p == rank || p >= P.n) {  // my own record: slice 0 (workgroup `rank`), slices 1 .. (the workgroups past P.n)
    const int ns = (int)gridDim.x - P.n + 1, sl = p == rank ? 0 : p - P.n + 1;
    const int64_t per = (count + ns - 1) / ns;
    const int64_t lo = sl * per, hi = lo + per < count ? lo + per : count;
    if (lo < hi) p2p_copy(send, local + (size_t)rank * count, lo, hi);
    return;
  }
  double* loc = local + (size_t)p * count;
  p2p_copy(send, P.rx[p] + (size_t)rank * count, 0, count);
  barrier_stores_done();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(P.flag[p] + rank, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    ok = p2p_wait(P.flag[rank] + p, seq, max_spins, fail) ? 1 : 0;
  }
  __syncthreads();
  if (!ok) return;
  p2p_copy(my_rx + (size_t)p * count, loc, 0, count);
}
// Halo exchange, one workgroup per neighbour (0: left, 1: right): my head region (nl doubles) -> the left neighbour's tail
// slot, my tail region (nr doubles) -> the right neighbour's head slot; their flags; then the wait for that neighbour's record.
struct P2PHalo {
  double *left_dst, *right_dst;                 // where my two regions go (null: no such neighbour)
  unsigned long long *left_flag, *right_flag;   // the neighbours' flag words for records coming from me
  unsigned long long *my_from_left, *my_from_right;
};
__global__ __launch_bounds__(1024) void k_p2p_halo(const double* __restrict__ raw, int64_t nl, int64_t nr, P2PHalo H,
                                                   unsigned long long seq, int* fail, long max_spins) {
  if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
  const bool left = blockIdx.x == 0;
  double* dst = left ? H.left_dst : H.right_dst;
  if (!dst) return;
  const double* src = left ? raw : raw + nl;
  const int64_t cnt = left ? nl : nr;
  // (records are whole [row][lane] pairs of doubles in 16-byte aligned buffers whenever two lanes travel)
  if ((cnt & 1) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
    const double2* s2 = reinterpret_cast<const double2*>(src);
    double2* d2 = reinterpret_cast<double2*>(dst);
    for (int64_t i = threadIdx.x; i < cnt / 2; i += 1024) d2[i] = s2[i];
  } else {
    for (int64_t i = threadIdx.x; i < cnt; i += 1024) dst[i] = src[i];
  }
  barrier_stores_done();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(left ? H.left_flag : H.right_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    p2p_wait(left ? H.my_from_left : H.my_from_right, seq, max_spins, fail);
  }
}

// The exchange AND the finish of the overlap rows in one launch (peer-to-peer route; FPSQ_HALO_FUSE=0: k_p2p_halo, then
// k_halo_finish): workgroups [0, 2 kHaloCopy) push my two regions into the neighbours' slots (kHaloCopy slices per side; the
// last slice to arrive -- a monotone counter per side -- raises the neighbour's flag word) and leave; the others are
// k_halo_finish's workgroups, which first wait (bounded) for the neighbours' records.  A rank's launch e + 1 follows its launch e
// in stream order and launch e ends only when its finish workgroups have consumed slot e: the slot a neighbour overwrites
// with its push e + 2 (after ITS launch e + 1, whose finish workgroups waited for my push e + 1) is free, as with two kernels.
constexpr int kHaloCopy = 4;
struct HaloFinishArgs {
  const double* raw;
  const double* recv;
  int64_t tl, tr, tail0;
  const double* yin;
  double* yout;
  const LaneCtl *ctl0, *ctl1;
  double* partials;
  int32_t pstride;
  // tests only (P2PRoute: FPSQ_DEBUG_P2P_DELAY = r + 1): bit 0 -- this rank's finish workgroups idle ~100 us before they READ the
  // neighbours' records (a slow reader)
  int32_t dbg;
  const LaneCtl *gate0, *gate1;
};
template <int NL>
__global__ __launch_bounds__(kBlock) void k_p2p_halo_finish(P2PHalo H, unsigned long long seq, int* fail, long max_spins,
                                                            unsigned long long* arrive /* [2] */, HaloFinishArgs a) {
  if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
  __shared__ int ok;
  int b = (int)blockIdx.x;
  if (b < 2 * kHaloCopy) {
    const bool left = b < kHaloCopy;
    const int sl = left ? b : b - kHaloCopy;
    double* dst = left ? H.left_dst : H.right_dst;
    if (!dst) return;
    const int64_t nl = a.tl * NL, cnt = left ? nl : a.tr * NL;
    const double* src = left ? a.raw : a.raw + nl;
    const int64_t per = ((cnt + kHaloCopy - 1) / kHaloCopy + 1) & ~(int64_t)1, lo = sl * per, hi = lo + per < cnt ? lo + per : cnt;
    if (lo < hi) p2p_copy(src, dst, lo, hi);
    barrier_stores_done();
    if (threadIdx.x == 0) {
      __threadfence_system();
      const unsigned long long got = __hip_atomic_fetch_add(arrive + (left ? 0 : 1), 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (got % kHaloCopy == 0)
        __hip_atomic_store(left ? H.left_flag : H.right_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  b -= 2 * kHaloCopy;
  const int nb = (int)gridDim.x - 2 * kHaloCopy;
  // The wait comes FIRST and is unconditional -- also when the recurrences have ended or a gate is closed and there is
  // nothing to finish: it is what paces the ranks.  A launch that ended without it would let this rank's NEXT push go out
  // while a slower neighbour is still reading the slot of the same parity (the first build returned early here: a rare wrong
  // last iteration with three ranks sharing one GPU; k_p2p_halo, the two-launch form, has always waited unconditionally).
  const LaneCtl* c[2] = {a.ctl0, a.ctl1};
  bool act[NL];
  double ca[NL], cb[NL];
  bool any = false;
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    act[l] = !(c[l]->done | c[l]->skip);
    ca[l] = c[l]->ca;
    cb[l] = c[l]->cb;
    any |= act[l];
  }
  const bool open = a.gate0 == nullptr || (a.gate0->done && a.gate1->done);
  if (threadIdx.x == 0) {
    bool in = true;
    if (H.my_from_left) in = p2p_wait(H.my_from_left, seq, max_spins, fail);
    if (in && H.my_from_right) in = p2p_wait(H.my_from_right, seq, max_spins, fail);
    ok = in ? 1 : 0;
    if (a.dbg & 1) {  // (tests: a slow reader)
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < 10000ull) __builtin_amdgcn_s_sleep(32);
    }
  }
  __syncthreads();
  if (!ok) return;
  if (!open || !any) return;
  __shared__ double red[4];
  double sq[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = 0.0;
  double sq_tail[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq_tail[l] = 0.0;
  for (int64_t i = (int64_t)b * kBlock + threadIdx.x; i < a.tl + a.tr; i += (int64_t)nb * kBlock) {
    const int64_t row = i < a.tl ? i : a.tail0 + (i - a.tl);
    double acc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] = a.raw[i * NL + l] + a.recv[i * NL + l];
    row_epilogue<NL>((size_t)row, acc, ca, cb, act, a.yin, a.yout, i < a.tl ? sq : sq_tail);  // (see k_halo_finish)
  }
  if (a.partials != nullptr) {
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const double t = block_sum(sq[l], red);
      if (threadIdx.x == 0) a.partials[(size_t)l * a.pstride + b] = t;
    }
  }
}

// out = a * P + b * y (plain vectors, host-given constants), for the p1 = g - A'q1 and J'c products
__global__ __launch_bounds__(kBlock) void k_axpby_plain(const double* __restrict__ P, double a, const double* y,
                                                        double b, double* out, int64_t len) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock)
    out[i] = a * P[i] + (b != 0.0 ? b * y[i] : 0.0);
}

// Local sums of up to four partial arrays into out[0..3] (the payload of the scalar all-reduce). One workgroup.
struct PresumArgs {
  const double* p[4];
  int32_t n[4];
};
__global__ __launch_bounds__(kBlock) void k_presum(PresumArgs a, double* out) {
  __shared__ double red[4];
  for (int k = 0; k < 4; ++k) {
    double s = 0.0;
    if (a.p[k]) s = reduce_partials(a.p[k], a.n[k], red);
    if (threadIdx.x == 0) out[k] = s;
    __syncthreads();
  }
}

// In-process "all-reduce" over up to 8 logical shards that live on the same GPU (validation of the sharded
// code path on a one-GPU box): every buffer receives the sum in rank order.
struct ShardBufs {
  double* b[8];
  int32_t n;
};
__global__ __launch_bounds__(kBlock) void k_local_allreduce(ShardBufs B, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock) {
    double s = 0.0;
    for (int r = 0; r < B.n; ++r) s += B.b[r][i];
    for (int r = 0; r < B.n; ++r) B.b[r][i] = s;
  }
}

// ---- equality-QP user model + penalty epilogues (src/model-Fletcherpenaltynlp.jl:238-248, 385-397, 419-433)

// g = q .* x + d ;  partial of f = sum x (q x / 2 + d) ;  partial of ||x - xk||^2 when xk != null.
// lp != null (fast start of qp_objgrad): also lp[i] = {g_i, x_i} -- the long Golub-Kahan pair with LSQR's u~_1 = g in lane 0
// and, in the still unused CRAIG lane, the point whose constraint values the start-up product will form -- and the
// partials of ||g||^2 in pg.
struct QpGradArgs {
  const double *q, *d, *x, *xk;
  double* g;
  int64_t n;
  double *pf, *pdx, *lp, *pg;
  int64_t n_sum;
  int32_t nblk;  // workgroups of this body (k_startup: the first nblk of the launch; 0 = none)
};
__device__ __forceinline__ void qp_grad_body(const QpGradArgs& a, int blk, double* red) {
  const double* __restrict__ q = a.q;
  const double* __restrict__ d = a.d;
  const double* __restrict__ x = a.x;
  const double* xk = a.xk;
  double* g = a.g;
  double* lp = a.lp;
  double* pf = a.pf;
  double* pdx = a.pdx;
  double* pg = a.pg;
  const int64_t n = a.n, n_sum = a.n_sum;
  double f = 0.0, dx2 = 0.0, gg = 0.0;
  for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < n; i += (int64_t)a.nblk * kBlock) {
    const double xi = x[i], qi = q[i], di = d[i];
    const double gi = qi * xi + di;
    g[i] = gi;
    if (lp) *reinterpret_cast<double2*>(lp + 2 * i) = make_double2(gi, xi);
    if (i < n_sum) {  // (halo mode: sums over the rank's owned prefix of its column window)
      if (lp) gg += gi * gi;
      f += xi * (0.5 * qi * xi + di);
      if (xk) {
        const double t = xi - xk[i];
        dx2 += t * t;
      }
    }
  }
  const double tf = block_sum(f, red);
  if (threadIdx.x == 0) pf[blk] = tf;
  const double td = block_sum(dx2, red);
  if (threadIdx.x == 0) pdx[blk] = td;
  if (lp) {
    const double tg = block_sum(gg, red);
    if (threadIdx.x == 0) pg[blk] = tg;
  }
}
__global__ __launch_bounds__(kBlock) void k_qp_grad(const QpGradArgs a) {
  __shared__ double red[4];
  qp_grad_body(a, blockIdx.x, red);
}

// ys = q1 + sigma q2 ; partials of c'ys and c'c            (m-vectors)
// `flush` (kind UPD_LSQR, two-lane source pair): the LAST x update of the LSQR recurrence whose solution is q1 (= flush.a),
// applied element-wise right here instead of by a launch of its own -- under exactly upd_lsqr's conditions (the recurrence
// ended at iteration flush.it; speculative: the other lane ended too).  w is left alone: nothing reads it any more.
__global__ __launch_bounds__(kBlock) void k_ys(const double* q1, const double* __restrict__ q2,
                                               const double* __restrict__ c, double sigma, double* ys, int64_t m,
                                               double* pcy, double* pcc, double* pack, const LaneCtl* gate0,
                                               const LaneCtl* gate1, const UpdSeg flush) {
  if (gate0 != nullptr && !(gate0->done && gate1->done)) return;
  __shared__ double red[4];
  double cy = 0.0, cc = 0.0;
  bool fl = flush.kind == UPD_LSQR;
  double sg = 0.0;
  if (fl) {
    const LaneCtl* ctl = flush.ctl;
    if (ctl->done && ctl->upd_iter != flush.it) fl = false;
    if (fl && flush.gate != nullptr && !(ctl->done && flush.gate->done)) fl = false;
    sg = ctl->e[0];
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    double q1i = q1[i];
    if (fl) {
      q1i += sg * flush.b[i];
      flush.a[i] = q1i;
    }
    const double y = q1i + sigma * q2[i];
    ys[i] = y;
    if (pack) *reinterpret_cast<double2*>(pack + 2 * i) = make_double2(q1i, c[i]);  // input pair of the A'[q1, c] product
    if (c) {
      const double ci = c[i];
      cy += ci * y;
      cc += ci * ci;
    }
  }
  if (pcy) {
    const double a = block_sum(cy, red);
    if (threadIdx.x == 0) pcy[blockIdx.x] = a;
    const double b = block_sum(cc, red);
    if (threadIdx.x == 0) pcc[blockIdx.x] = b;
  }
}

// gs = p1 + sigma p2                                        (n-vectors; v aliases p2)
__global__ __launch_bounds__(kBlock) void k_gs(const double* __restrict__ p1, const double* __restrict__ p2,
                                               double sigma, double* gs, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    gs[i] = p1[i] + sigma * p2[i];
}

// phi from the partial sums of the evaluation (src/model-Fletcherpenaltynlp.jl:419-433); out = {phi, f, c'c}
struct FxArgs {
  const double *pf, *pdx, *pcy, *pcc;
  int32_t np_n, np_m;
  double rho, eta;
  double* out;  // null: not computed by this launch
  double seq;   // call sequence number stored behind the results (out[3])
  int32_t stride;  // see qp_fx_core
  uint32_t xseq;
  const XchTable* xt;  // non-null: the local sums are added up over the ranks inside this launch (qp_fx_core)
};
constexpr int kFxRed = 16 + 36;  // LDS doubles of qp_fx (the block reduction + xch_sum<4>'s scratch)
__device__ __forceinline__ void qp_fx(const FxArgs& a, double* red /* kFxRed */) {
  qp_fx_core(a.pf, a.pdx, a.pcy, a.pcc, a.np_n, a.np_m, a.rho, a.eta, a.out, a.seq, red, a.stride, a.xt, a.xseq);
}

// One row of grad(phi) (src/model-Fletcherpenaltynlp.jl:403-437 on the eq-QP model), from row i of the raw two-right-hand-side product
// pj = {(A'q1)_i, (A'c)_i}:   gs = (g - A'q1) + sigma v,   gx = gs - q v + sigma v (+ rho A'c) (+ eta (x - xk)).
// ONE spelling of the arithmetic (explicit fused multiply-adds) for the two kernels that compute it -- k_qp_penalty_grad and the row
// epilogue of the raw A' product (k_spmv<.., GRAD>) -- so that the two tails agree bit for bit whatever the compiler would contract.
__device__ __forceinline__ void qp_grad_row(double gi, double lp0, double lp1, double vi, double qi, double sigma, double rho, double eta,
                                            const double* x, const double* xk, int64_t i, double& gs, double& gx) {
  const double p1i = gi - lp0;
  gs = fma(sigma, vi, p1i);
  double gg = fma(sigma, vi, fma(-qi, vi, gs));
  if (rho > 0.0) gg = fma(lp1, rho, gg);
  if (eta > 0.0) gg = fma(eta, x[i] - xk[i], gg);
  gx = gg;
}
// One row of hprod! Val(2)'s result (src/model-Fletcherpenaltynlp.jl:543-562 on the eq-QP model):
//   Hv = p2 - q (v - p1) + 2 sigma (v - p1) (+ rho (J'J v)) (+ eta v).   One spelling, for k_qp_hprod_fin and for the row epilogue of the
// single-lane product A'(A v) (k_spmv<1, .., GRAD>).
__device__ __forceinline__ double qp_hfin_row(double vi, double p1i, double p2i, double qi, double jtjv, double sigma, double rho, double eta) {
  const double pt = vi - p1i;
  double r = fma(2.0 * sigma, pt, fma(-qi, pt, p2i));
  if (rho > 0.0) r = fma(rho, jtjv, r);
  if (eta > 0.0) r = fma(eta, vi, r);
  return r;
}
// what a raw A' product needs to write the END RESULT of its call instead of its own rows (k_spmv<.., GRAD>, one GPU):
//   two lanes  (A'[q1, c], the tail of objgrad!):  gs and gx = grad(phi)                 -- qp_grad_row
//   one lane   (A'(A v),   the tail of hprod!):    hv                                    -- qp_hfin_row (p1, p2 of the two solves)
struct GradEpi {
  const double *g, *v, *q, *x, *xk;
  const double *p1, *p2;
  double sigma, rho, eta;
  double *gs, *gx, *hv;
  FxArgs fx;  // fx.out != null: workgroup 0 of the grid reduces the evaluation's partial sums to phi (k_qp_penalty_grad's last one does)
};

// QP penalty gradient, one pass:  gs = p1 + sigma v;  gx = gs - q.*v + sigma v (+ rho Jc) (+ eta (x - xk)),  v = p2.
// pj != null: p1 and Jc come from ONE two-right-hand-side raw product pj[i] = {(A'q1)_i, (A'c)_i}: p1 = g - pj[.][0].
// The last workgroup of the grid does no streaming: it reduces the evaluation's partial sums to phi (fx.out != null),
// concurrently with the others -- one launch less at the end of every evaluation.
__global__ __launch_bounds__(kBlock) void k_qp_penalty_grad(const double* __restrict__ p1, const double* __restrict__ g,
                                                            const double* __restrict__ pj, const double* __restrict__ v,
                                                            const double* __restrict__ q, const double* jc,
                                                            const double* x, const double* xk, double sigma,
                                                            double rho, double eta, double* gs, double* gx,
                                                            int64_t n, const FxArgs fx, const LaneCtl* gate0,
                                                            const LaneCtl* gate1) {
  if (gate0 != nullptr && !(gate0->done && gate1->done)) return;
  int nb = gridDim.x;
  if (fx.out) {
    --nb;
    if ((int)blockIdx.x == nb) {
      __shared__ double red[kFxRed];
      qp_fx(fx, red);
      return;
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)nb * kBlock) {
    const double vi = v[i];
    double gi, lp0, jci = 0.0;
    if (pj) {
      const double2 t = *reinterpret_cast<const double2*>(pj + 2 * i);
      gi = g[i];
      lp0 = t.x;
      jci = t.y;
    } else {  // (p1 = g - A'q1 was formed by the caller: the same subtraction with a zero)
      gi = p1[i];
      lp0 = 0.0;
      if (rho > 0.0) jci = jc[i];
    }
    double gsi, gg;
    qp_grad_row(gi, lp0, jci, vi, q[i], sigma, rho, eta, x, xk, i, gsi, gg);
    gs[i] = gsi;
    gx[i] = gg;
  }
}

// hprod! Val(2) on the eq-QP model, the two element-wise ends:  Hsv = q .* v   and
//   Hv = p2 - q .* (v - p1) + 2 sigma (v - p1) (+ rho JtJv) (+ eta v)
__global__ __launch_bounds__(kBlock) void k_qp_hsv(const double* __restrict__ q, const double* __restrict__ v, double* hsv,
                                                   int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) hsv[i] = q[i] * v[i];
}
__global__ __launch_bounds__(kBlock) void k_qp_hprod_fin(const double* __restrict__ p1, const double* __restrict__ p2,
                                                         const double* __restrict__ q, const double* __restrict__ v,
                                                         const double* jtjv, double sigma, double rho, double eta,
                                                         double* hv, int64_t n, const LaneCtl* gate0,
                                                         const LaneCtl* gate1) {
  if (gate0 != nullptr && !(gate0->done && gate1->done)) return;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    hv[i] = qp_hfin_row(v[i], p1[i], p2[i], q[i], rho > 0.0 ? jtjv[i] : 0.0, sigma, rho, eta);
  }
}

// ---- MINRES on K = [I A'; A -delta I] itself (fpsq_options.kkt_method = FPSQ_KKT_MINRES_K): both systems of a call are
// lanes of interleaved (n + m)-vectors, stored as a long (n) and a short (m) part.  One launch of a stage covers both
// parts (blocks [0, gl) the long one) and both lanes; partial sums go to part0 / part1 [gl + gs].
struct MkVecs {
  double *Y, *R1, *R2, *W1, *W2, *X;  // [len][2]
};
constexpr int kMkPerBlock = kBlock * 8;

// R1 = R2 = right-hand sides (b0: lane 0, b1: lane 1; null = zero), W1 = W2 = X = 0; partial sums of b^2
__global__ __launch_bounds__(kBlock) void k_mk_init(MkVecs lg, const double* bl0, const double* bl1, int64_t n, MkVecs sh,
                                                    const double* bs0, const double* bs1, int64_t m, int gl,
                                                    double* part0, double* part1) {
  __shared__ double red[2 * 32];
  const int blk = blockIdx.x;
  const bool is_long = blk < gl;
  const MkVecs& v = is_long ? lg : sh;
  const double* b0 = is_long ? bl0 : bs0;
  const double* b1 = is_long ? bl1 : bs1;
  const int64_t len = is_long ? n : m;
  const int64_t beg = (int64_t)(is_long ? blk : blk - gl) * kMkPerBlock;
  double a0 = 0.0, a1 = 0.0;
  for (int64_t i = beg + threadIdx.x; i < beg + kMkPerBlock && i < len; i += kBlock) {
    const double x0 = b0 ? b0[i] : 0.0, x1 = b1 ? b1[i] : 0.0;
    v.R1[2 * i] = x0;
    v.R1[2 * i + 1] = x1;
    v.R2[2 * i] = x0;
    v.R2[2 * i + 1] = x1;
    v.W1[2 * i] = v.W1[2 * i + 1] = 0.0;
    v.W2[2 * i] = v.W2[2 * i + 1] = 0.0;
    v.X[2 * i] = v.X[2 * i + 1] = 0.0;
    a0 += x0 * x0;
    a1 += x1 * x1;
  }
  const double t0 = block_sum(a0, red);
  const double t1 = block_sum(a1, red + 32);
  if (threadIdx.x == 0) {
    part0[blk] = t0;
    part1[blk] = t1;
  }
}

// The three element-wise stages of an iteration (Paige & Saunders; same coefficients e[] as upd_minres):
//   1: y0 = y - e0 r1,                         partial <r2, y0>
//   2: y = y0 - e1 r2; r1 = r2; r2 = y; w1 <- w2, w2 <- e2 r2_old - e3 w2 - e4 w1 (unscaled),   partial <y, y>
//   3: w2 *= e5; x += e6 w2,                    partial <x, x>      (only when stage B of iteration `it` ran)
template <int STAGE>
__global__ __launch_bounds__(kBlock) void k_mk_stage(const LaneCtl* c0, const LaneCtl* c1, int it, MkVecs lg, int64_t n,
                                                     MkVecs sh, int64_t m, int gl, double* part0, double* part1) {
  __shared__ double red[2 * 32];
  const LaneCtl* cs[2] = {c0, c1};
  bool act[2];
  double e[2][7];
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    act[l] = STAGE == 3 ? cs[l]->upd_iter == it : !cs[l]->done;
#pragma unroll
    for (int q = 0; q < 7; ++q) e[l][q] = cs[l]->e[q];
  }
  if (!act[0] && !act[1]) return;
  const int blk = blockIdx.x;
  const bool is_long = blk < gl;
  const MkVecs& v = is_long ? lg : sh;
  const int64_t len = is_long ? n : m;
  const int64_t beg = (int64_t)(is_long ? blk : blk - gl) * kMkPerBlock;
  double acc[2] = {0.0, 0.0};
  for (int64_t i = beg + threadIdx.x; i < beg + kMkPerBlock && i < len; i += kBlock) {
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      if (!act[l]) continue;
      const int64_t k = 2 * i + l;
      if (STAGE == 1) {
        const double y0 = v.Y[k] - (e[l][0] != 0.0 ? e[l][0] * v.R1[k] : 0.0);
        v.Y[k] = y0;
        acc[l] += v.R2[k] * y0;
      } else if (STAGE == 2) {
        const double r = v.R2[k];
        const double y = v.Y[k] - e[l][1] * r;
        v.R1[k] = r;
        v.R2[k] = y;
        const double w2 = v.W2[k];
        v.W2[k] = e[l][2] * r - e[l][3] * w2 - e[l][4] * v.W1[k];
        v.W1[k] = w2;
        acc[l] += y * y;
      } else {
        const double w = v.W2[k] * e[l][5];
        v.W2[k] = w;
        const double xn = v.X[k] + e[l][6] * w;
        v.X[k] = xn;
        acc[l] += xn * xn;
      }
    }
  }
  const double t0 = block_sum(acc[0], red);
  const double t1 = block_sum(acc[1], red + 32);
  if (threadIdx.x == 0) {
    if (act[0]) part0[blk] = t0;
    if (act[1]) part1[blk] = t1;
  }
}

// out0 / out1 = lanes 0 / 1 of an interleaved pair vector
__global__ __launch_bounds__(kBlock) void k_mk_unpack(const double* __restrict__ pair, double* out0, double* out1, int64_t len) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < len; i += (int64_t)gridDim.x * kBlock) {
    out0[i] = pair[2 * i];
    out1[i] = pair[2 * i + 1];
  }
}

}  // namespace fpsq

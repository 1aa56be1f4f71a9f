// fpsq_spmv.hip.h -- the two product kernels of libfpsq (CSR-stream k_spmv for A' and the CSR fallback of A; column-sorted
// row groups k_spmv_rgcs for A; k_spmv_atl, the A' product of a launch with riding leaders): SpMV / SpMM with the fused
// axpby + norm-partial epilogue, the riding vector updates and the riding scalar steps.  Split from fpsq_kernels.hip.h
// because the riding steps need the recurrences of fpsq_krylov.hip.h.
#pragma once
#include "fpsq_krylov.hip.h"

namespace fpsq {

// Halo mode of the row-sharded A' product (k_spmv<.., HALO>): the rows [0, lo) and [hi, nrows) of the product -- the two
// regions of the rank's column window that a neighbour also contributes to -- are NOT finalised: their raw sums go to
// raw[(r < lo ? r : lo + r - hi)][NL] (head region first) and k_halo_finish completes them once the neighbours' sums have
// arrived.  Interior rows get the fused axpby + squared-norm epilogue exactly as on one GPU.
struct HaloRows {
  int64_t lo, hi;
  double* raw;
};

// Column-sorted padded block L (k_spmv<.., CSORT>): thread t owns the entry pairs (2t, 2t + 1) + 512 j, j = 0..3, of the
// block's sorted order.  Values: four 16-byte loads, each wave instruction one contiguous KB.  Index planes: the thread's
// eight 16-bit words (slot | (col & 31) << 11) and eight bytes (col >> 5) are stored contiguously at [8t, 8t + 8) of the
// block -- ONE 16-byte and ONE 8-byte load.  (Twenty-four loads per thread in entry order cost the product more than the
// sorted gathers saved: the texture path is paid per wave instruction.)
struct CsortRaw {
  uint4 w16;
  uint2 w8;
  double2 vv[4];
};
__device__ __forceinline__ void csort_fetch_raw(const CsrView& A, int L, int tid, CsortRaw& r) {
  static_assert(kSpmvNnz == 2048 && kBlock == 256, "column-sorted blocks: 8 entries per thread");
  const size_t b0 = (size_t)L * kSpmvNnz;
  r.w16 = *reinterpret_cast<const uint4*>(A.cs16 + b0 + 8 * tid);
  r.w8 = *reinterpret_cast<const uint2*>(A.cs8 + b0 + 8 * tid);
#pragma unroll
  for (int j = 0; j < 4; ++j) r.vv[j] = *reinterpret_cast<const double2*>(A.vals + b0 + 2 * tid + 512 * j);
}
__device__ __forceinline__ void csort_decode(const CsortRaw& r, int cbase, int (&cidx)[8], int (&slot)[8], double (&v)[8]) {
  const unsigned h16[4] = {r.w16.x, r.w16.y, r.w16.z, r.w16.w};
  const unsigned h8[2] = {r.w8.x, r.w8.y};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int pk = (int)((h16[q >> 1] >> (16 * (q & 1))) & 0xffffu);
    const int hi = (int)((h8[q >> 2] >> (8 * (q & 3))) & 0xffu);
    cidx[q] = cbase + ((hi << 5) | (pk >> 11));
    slot[q] = pk & 2047;
    v[q] = (q & 1) ? r.vv[q >> 1].y : r.vv[q >> 1].x;
  }
}
// SHARED VALUES (CsrView::segdesc != null): thread t owns the entries t + 256 j of the block's stored order (j = 0..7; its eight
// index words are contiguous at [8t, 8t + 8) as above), i.e. lane t % 64 of segment 4 j + t / 64.  A segment's descriptor
// is wave-uniform: four 24-bit bases relative to the block's base (blkdesc.w), three split lanes, the number of valid lanes;
// value address = base of the block + base of my run + lane, or the array's zero entry past the valid lanes.
//
// cshared_head -- everything at the HEAD of a block's life that needs nothing but the block's number, in the order that keeps
// its chain of dependent memory round trips short:
//   1. the index planes (two vector loads per thread) are REQUESTED;
//   2. the block descriptor, its column base and the wave's eight segment descriptors come through the SCALAR cache (one
//      inline-assembly batch of s_load: as vector loads of one address each would cost the texture path a full wave
//      instruction -- eighteen vector loads per thread instead of six made the plain A' product 6 us slower -- and the compiler
//      does not prove the addresses uniform).  They are small, hot in L2, and back long before the index planes;
//   3. the eight VALUE loads leave at once -- they need the descriptors only -- so the values' trip to HBM runs next to the
//      index planes' one instead of behind it;
// the caller then requests row bounds and yin (they need r0 from the scalar descriptor), and cshared_decode waits for the
// index planes and yields the gather addresses: values and gathered x arrive together.  (First build of round 4: descriptor
// and value loads BEHIND the index planes and behind the vector load of blkdesc: two round trips more in the chain.)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// (plain references, no struct: as members of a struct handed through two branches the index words ended up on the stack)
__device__ __forceinline__ void cshared_head(const CsrView& A, int L, int tid, uint4& w16, uint2& w8, int& r0, int& nr, int& s_,
                                             int& cbase, double (&v)[8]) {
  const size_t b0 = (size_t)L * kSpmvNnz;
  w16 = *reinterpret_cast<const uint4*>(A.cs16 + b0 + 8 * tid);
  w8 = *reinterpret_cast<const uint2*>(A.cs8 + b0 + 8 * tid);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int Lu = __builtin_amdgcn_readfirstlane(L);
  typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
  u32x4 bd;
  unsigned int cb;
  u32x16 dA, dB;  // the wave's eight descriptors are contiguous (one 128-byte line): segments w, w + 4, ... = words 4 j .. 4 j + 3
  auto uni = [](const void* q) {
    const unsigned long long pa = reinterpret_cast<unsigned long long>(q);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(pa & 0xffffffffull));
  };
  const unsigned long long pd = uni(A.segdesc + (size_t)Lu * 32 + 8 * w), pb = uni(A.blkdesc + Lu), pc = uni(A.colbase + Lu);
  // (two batches of four descriptors: forty scalar registers live at once made the kernel spill)
  asm volatile(
      "s_load_dwordx4 %1, %4, 0x0\n\t"
      "s_load_dword %2, %5, 0x0\n\t"
      "s_load_dwordx16 %0, %3, 0x0\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(dA), "=&s"(bd), "=&s"(cb)
      : "s"(pd), "s"(pb), "s"(pc)
      : "memory");
  r0 = (int)bd.x;
  nr = (int)bd.y;
  s_ = (int)bd.z;
  cbase = (int)cb;
  const int vbase = (int)bd.w;
  if (vbase <= -128) {  // (block-uniform) a block with values of its own: entry t + 256 j at [2048 own + 256 j + t]
    const double* ov = A.vals_own + (size_t)(-128 - vbase) * kSpmvNnz + tid;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = ov[q * kBlock];
    return;
  }
  auto value_of = [&](unsigned dx, unsigned dy, unsigned dz, unsigned dw) {
    const unsigned s0 = dw & 127u, s1 = (dw >> 7) & 127u, s2 = (dw >> 14) & 127u, nv = (dw >> 21) & 127u;
    const int b0_ = (int)(dx & 0xffffffu), b1_ = (int)((dx >> 24) | ((dy & 0xffffu) << 8));
    const int b2_ = (int)((dy >> 16) | ((dz & 0xffu) << 16)), b3_ = (int)(dz >> 8);
    const int bs = (unsigned)lane < s0 ? b0_ : (unsigned)lane < s1 ? b1_ : (unsigned)lane < s2 ? b2_ : b3_;
    const int pos = (unsigned)lane < nv ? vbase + bs + lane : A.zero_pos;
    return A.vals[pos];
  };
  v[0] = value_of(dA[0], dA[1], dA[2], dA[3]);
  v[1] = value_of(dA[4], dA[5], dA[6], dA[7]);
  v[2] = value_of(dA[8], dA[9], dA[10], dA[11]);
  v[3] = value_of(dA[12], dA[13], dA[14], dA[15]);
  asm volatile(
      "s_load_dwordx16 %0, %1, 0x40\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(dB)
      : "s"(pd)
      : "memory");
  v[4] = value_of(dB[0], dB[1], dB[2], dB[3]);
  v[5] = value_of(dB[4], dB[5], dB[6], dB[7]);
  v[6] = value_of(dB[8], dB[9], dB[10], dB[11]);
  v[7] = value_of(dB[12], dB[13], dB[14], dB[15]);
}
__device__ __forceinline__ void cshared_decode(const uint4& w16, const uint2& w8, int cbase, int (&cidx)[8], int (&slot)[8]) {
  const unsigned h16[4] = {w16.x, w16.y, w16.z, w16.w};
  const unsigned h8[2] = {w8.x, w8.y};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int pk = (int)((h16[q >> 1] >> (16 * (q & 1))) & 0xffffu);
    const int hi = (int)((h8[q >> 2] >> (8 * (q & 3))) & 0xffu);
    cidx[q] = cbase + ((hi << 5) | (pk >> 11));
    slot[q] = pk & 2047;
  }
}
__device__ __forceinline__ void csort_fetch(const CsrView& A, int L, int cbase, int tid, int (&cidx)[8], int (&slot)[8],
                                            double (&v)[8]) {
  CsortRaw r;
  csort_fetch_raw(A, L, tid, r);
  csort_decode(r, cbase, cidx, slot, v);
}

// ------------------------------------------------------------------------------------------------ SpMV / SpMM
//
// out[r][l] = ca_l * sum_k vals[k] * x[colind[k]][l] + cb_l * yin[r][l],  partial[l][blk] = sum_r out[r][l]^2
//
// CSR-stream: a workgroup owns a run of consecutive rows holding <= kSpmvNnz nonzeros.  Phase 1 streams that
// run's (colind, vals) with unit-stride, fully coalesced loads (every lane busy whatever the row lengths), gathers
// x and parks the products in LDS; phase 2 reduces each row's LDS segment with G lanes per row (G = power of two
// chosen from the block's row count: ~100-nnz rows of A get 8 lanes each, ~10-nnz rows of A' one lane each).
// blockIdx is remapped so that each XCD walks a contiguous eighth of the matrix: its private L2 then caches one
// slice of x instead of all of it.  Summation order is a pure function of the sparsity => reproducible.
// PAD: every row block's entries are stored at [L * kSpmvNnz, ...) and zero-padded to kSpmvNnz (CsrView::vals /
// col16 / colind then point to the padded arrays): the matrix stream needs neither the block descriptor (one link
// less in the workgroup's chain of dependent memory round trips) nor bounds checks.  Requires that no block is a
// long row.
// partials[l * pstride + L]: the lane stride of the partial array is the caller's (the workgroup count on one GPU; a
// padded count common to all ranks when the arrays are all-gathered, see run_krylov).
// CSORT (A' of a banded Jacobian: padded layout, every block's columns within 8192 of colbase): the block's entries are
// STORED sorted by column, each carrying its slot in the block's row-major order (slot | (col & 31) << 11 in CsrView::cs16,
// col >> 5 in CsrView::cs8: 11 B per entry; which thread reads what: csort_fetch).  Row-order gathers of ~10-entry rows put ~50 different 128-byte lines into every
// wave instruction and the texture path takes them one after the other: that, not HBM, bounded the A' product (what-if
// builds at the headline size: one address for all lanes 31 -> 20 us, the sorted order's addresses 31 -> 22 us).  In column
// order 64 consecutive entries read ~4 lines; the products are scattered to their row-major LDS slots and phase 2 is
// unchanged -- same values summed in the same order: BITWISE the row-order layout.
// GRAD (the tail of an evaluation on one GPU: NL = 2, the raw product A'[q1, c], round 5): the rows are not written -- each goes
// straight into the gradient row it is needed for (qp_grad_row: gs and gx), which saves writing and re-reading the 16 MB product and
// the launch of k_qp_penalty_grad; the FIRST workgroup of the grid reduces phi (that kernel's last one did).  Bitwise the two-kernel tail
// (FPSQ_FUSE_TAIL=0).
template <int NL, int TAG, bool IDX16 = false, bool PAD = false, bool HALO = false, bool CSORT = false, bool GRAD = false>
__global__ __launch_bounds__(kBlock) void k_spmv(CsrView A, const double* __restrict__ x, const double* yin,
                                                 double* yout, const LaneCtl* ctl0, const LaneCtl* ctl1,
                                                 double* partials, int blk_per_xcd, const UpdSeg u0, const UpdSeg u1,
                                                 const LaneCtl* gate0, const LaneCtl* gate1, int pstride,
                                                 const HaloRows hr, const GradEpi ge) {
  static_assert(!GRAD || (TAG == 1 && !HALO && PAD), "the result epilogues: raw A' products of one GPU, padded blocks");
  // exactly 32 KB of LDS for two right-hand sides (FOUR workgroups per CU -- measured, tools/stream_probe.hip: 30 KB would
  // admit five, 24 KB six; tiles of 1536 entries were slower all the same): the reduction scratch
  // aliases the head of the product buffer
  __shared__ double prod[kSpmvNnz * NL];
  double* red = prod;
  // speculatively enqueued epilogue product: runs only once both recurrences of the call have ended
  if (gate0 != nullptr && !(gate0->done && gate1->done)) return;
  static_assert(!CSORT || (IDX16 && PAD && TAG == 1), "column-sorted blocks: padded A' with block-relative columns only");
  // GRAD: workgroup 0 reduces phi -- everything it sums was complete before this launch, so the scalar is on the host while the
  // product is still streaming (stream-ordered outputs: the caller's next call is enqueued behind it meanwhile) -- and the row
  // blocks follow from workgroup 1 on; no updates ride in this launch (the host checks).
  int shift = 0;
  if constexpr (GRAD) {
    shift = ge.fx.out != nullptr ? 1 : 0;
    if (shift && blockIdx.x == 0) {
      qp_fx(ge.fx, red);
      return;
    }
  } else {
    if (run_fused_updates<NL>(u0, u1, 8 * blk_per_xcd, red)) return;
  }
  // A (TAG 0): XCD-contiguous eighths, so an XCD's L2 holds one slice of the long gathered vector.  A' (TAG 1): the gathered
  // vector is short (L2-resident everywhere) and the identity map keeps all XCDs streaming adjacent addresses: ~1 us faster.
  const int L = TAG == 1 ? (int)blockIdx.x - shift : (int)((blockIdx.x & 7) * blk_per_xcd + (blockIdx.x >> 3));
  if (L >= A.nblk) return;
  // shared values (kernel-uniform): the block's head through the scalar cache, value loads issued first (cshared_head)
  bool shared = false;
  if constexpr (CSORT) shared = A.segdesc != nullptr;
  [[maybe_unused]] uint4 sw16;
  [[maybe_unused]] uint2 sw8;
  [[maybe_unused]] double vsh[kSpmvNnz / kBlock];
  int4 bd;
  int cbase = 0;
  if (CSORT && shared) {
    if constexpr (CSORT) {
      int hr0, hnr, hs;
      cshared_head(A, L, (int)threadIdx.x, sw16, sw8, hr0, hnr, hs, cbase, vsh);
      bd = int4{hr0, hnr, hs, 0};
    }
  } else {
    // issued before the (dependent) done-check below: one memory round trip at the head of the workgroup, not two
    bd = A.blkdesc[L];
    cbase = IDX16 ? A.colbase[L] : 0;
  }
  bool act[NL];
  double ca[NL], cb[NL];
  {
    const LaneCtl* c[2] = {ctl0, ctl1};
    bool any = false;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      act[l] = !(c[l]->done | c[l]->skip);
      ca[l] = c[l]->ca;
      cb[l] = c[l]->cb;
      any |= act[l];
    }
    if (!any) return;
  }
  const int tid = threadIdx.x;
  const int r0 = bd.x, nr = bd.y, s = bd.z, e = bd.w;
  double sq[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = 0.0;

  if (!PAD && e - s > kSpmvNnz) {
    // one long row (nr == 1): every thread strides over it, no LDS staging
    double acc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] = 0.0;
    for (int i = s + tid; i < e; i += kBlock) {
      const int c = IDX16 ? cbase + (int)A.col16[i] : A.colind[i];
      const double v = A.vals[i];
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] += v * x[(size_t)c * NL + l];
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const double t = block_sum(acc[l], red);
      if (tid == 0 && act[l]) {
        if (HALO && (r0 < hr.lo || r0 >= hr.hi)) {
          hr.raw[(size_t)(r0 < hr.lo ? r0 : hr.lo + (r0 - hr.hi)) * NL + l] = t;
        } else {
          const double o = ca[l] * t + (cb[l] != 0.0 ? cb[l] * yin[(size_t)r0 * NL + l] : 0.0);
          yout[(size_t)r0 * NL + l] = o;
          sq[l] = o * o;
        }
      }
    }
  } else {
    // phase 1: coalesced stream of the block's nonzeros -> products in LDS.  Every lane issues all of its loads and
    // all of its gathers unconditionally (out-of-range lanes use column 0 with value 0 and park a zero in an unused
    // slot): a per-element branch would make hipcc wait for each gather before issuing the next one.
    // (the row-segment boundaries of phase 2 are loaded here too, ahead of the barrier that would expose their latency)
    int G = 1;
    while (G < 64 && G * 2 * nr <= kBlock) G <<= 1;
    const int rows_per_pass = kBlock / G;
    const int g = tid / G, gl = tid % G;
    const int rq0 = g < nr ? g : 0;
    const int seg_a0 = A.rowptr[r0 + rq0], seg_b0 = A.rowptr[r0 + rq0 + 1];
    // the yin values of the first row pass, requested now: the epilogue would otherwise expose their latency (a
    // workgroup's life is a chain of ~4 memory round trips; measured +4 % evaluations/s at the headline size)
    double ypre[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) ypre[l] = 0.0;
    if (yin != nullptr) {
      if (NL == 2) {
        const double2 t = *reinterpret_cast<const double2*>(yin + (size_t)(r0 + rq0) * 2);
        ypre[0] = t.x;
        ypre[NL - 1] = t.y;
      } else {
        ypre[0] = yin[r0 + rq0];
      }
    }
    [[maybe_unused]] double gpre[4] = {0.0, 0.0, 0.0, 0.0};  // GRAD: the epilogue's operands of the first row pass, requested here like yin
    if constexpr (GRAD) {
      gpre[0] = NL == 2 ? ge.g[r0 + rq0] : ge.p1[r0 + rq0];
      gpre[1] = ge.v[r0 + rq0];
      gpre[2] = ge.q[r0 + rq0];
      if (NL == 1) gpre[3] = ge.p2[r0 + rq0];
    }
    constexpr int kPer = kSpmvNnz / kBlock;
    int cidx[kPer];
    [[maybe_unused]] int slot[kPer];  // CSORT: where the entry's product goes (its position in the block's row-major order)
    double v[kPer];
    if constexpr (CSORT) {
      if (shared) {
        cshared_decode(sw16, sw8, cbase, cidx, slot);
#pragma unroll
        for (int k = 0; k < kPer; ++k) v[k] = vsh[k];
      } else {
        csort_fetch(A, L, cbase, tid, cidx, slot, v);
      }
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (CSORT) {
      } else if (PAD) {
        const size_t ii = (size_t)L * kSpmvNnz + tid + k * kBlock;
        cidx[k] = IDX16 ? cbase + (int)A.col16[ii] : A.colind[ii];
        v[k] = A.vals[ii];
      } else {
        const int i = s + tid + k * kBlock;
        const bool ok = i < e;
        const int ii = ok ? i : s;
        cidx[k] = IDX16 ? cbase + (int)A.col16[ii] : A.colind[ii];
        v[k] = ok ? A.vals[ii] : 0.0;
      }
    }
    if (NL == 1) {
      double xv[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) xv[k] = x[cidx[k]];
#pragma unroll
      for (int k = 0; k < kPer; ++k) prod[CSORT ? slot[k] : tid + k * kBlock] = v[k] * xv[k];
    } else {
      double2 xv[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        xv[k] = *reinterpret_cast<const double2*>(x + (size_t)cidx[k] * 2);
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k)
        *reinterpret_cast<double2*>(prod + 2 * (CSORT ? slot[k] : tid + k * kBlock)) = make_double2(v[k] * xv[k].x, v[k] * xv[k].y);
    }
    lds_barrier();
    // phase 2: G lanes per row
    for (int base = 0; base < nr; base += rows_per_pass) {
      const int rr = base + g;
      const bool valid = rr < nr;
      double acc[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] = 0.0;
      if (valid) {
        const int a = (base == 0 ? seg_a0 : A.rowptr[r0 + rr]) - s, b = (base == 0 ? seg_b0 : A.rowptr[r0 + rr + 1]) - s;
        row_segment_sum<NL>(prod, a + gl, b, G, acc);
      }
      for (int off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int l = 0; l < NL; ++l) acc[l] += __shfl_down(acc[l], off, 64);
      }
      if (valid && gl == 0) {
        const int64_t row = r0 + rr;
        if (HALO && (row < hr.lo || row >= hr.hi)) {
          double* dst = hr.raw + (size_t)(row < hr.lo ? row : hr.lo + (row - hr.hi)) * NL;
#pragma unroll
          for (int l = 0; l < NL; ++l)
            if (act[l]) dst[l] = acc[l];
        } else if constexpr (GRAD && NL == 2) {
          const double gi = base == 0 ? gpre[0] : ge.g[row], vi = base == 0 ? gpre[1] : ge.v[row], qi = base == 0 ? gpre[2] : ge.q[row];
          double gsi, gg;
          qp_grad_row(gi, acc[0], acc[NL - 1], vi, qi, ge.sigma, ge.rho, ge.eta, ge.x, ge.xk, row, gsi, gg);
          ge.gs[row] = gsi;
          ge.gx[row] = gg;
        } else if constexpr (GRAD) {
          const double p1i = base == 0 ? gpre[0] : ge.p1[row], vi = base == 0 ? gpre[1] : ge.v[row], qi = base == 0 ? gpre[2] : ge.q[row];
          const double p2i = base == 0 ? gpre[3] : ge.p2[row];
          ge.hv[row] = qp_hfin_row(vi, p1i, p2i, qi, acc[0], ge.sigma, ge.rho, ge.eta);
        } else {
          row_epilogue<NL>((size_t)row, acc, ca, cb, act, yin, yout, sq, base == 0 && yin != nullptr ? ypre : nullptr);
        }
      }
    }
  }
  if (partials != nullptr) {
    lds_barrier();  // `red` aliases `prod`: every wave must be past its phase-2 reads
    block_sum_lanes<NL>(sq, red);
    if (tid == 0) {
#pragma unroll
      for (int l = 0; l < NL; ++l) partials[(size_t)l * pstride + L] = sq[l];
    }
  }
}

// ------------------------------------------------------------------------------------------------ steps riding with leaders
//
// (A first form, in which EVERY workgroup of the product recomputed the steps in its prologue, read the partial sums thousands
// of times over on a large grid and was slower than this one on a small grid too: tools/experiments/stepin_all_recompute.patch.)
// The two steps that follow the
// previous product are computed by LEADER workgroups at the head of this launch's grid (see "WHO LEADS"): k_step's body, and
// the step's outcome published as self-validating words (ride_publish) the moment its arithmetic is done.  A product only
// needs the steps' coefficients in its row epilogue -- out = ca (A x) + cb yin is linear in them -- so every other workgroup
// starts streaming at once and picks them up on the way: wave 0 LOOKS for the record (one wave-level device-scope load:
// this XCD's L2 may hold an older image of those lines) behind its matrix stream, again behind its gathers, behind its
// products and behind its row sums -- always behind the loads whose arrival the next step waits for, because device-scope
// loads come back late and the load counter is in-order.  The first resident set of A' workgroups, which starts together
// with the leaders, takes two row blocks and so never gets to an epilogue early (k_spmv_atl); RGCS groups run five tiles.
// Whoever still has nothing when it needs the coefficients polls (ride_settle), a bounded number of times.
// The one-workgroup k_step launch between two products (~5 us in the running pipeline, 32 per evaluation at the headline
// size) disappears; results are bitwise those of the stand-alone steps.
struct RideArgs {
  unsigned long long* rec;  // the leaders' record, one copy per XCC: 16 + 48 self-validating words (four lines), see ride_publish
  unsigned int want;        // this launch's number
  unsigned int pub;         // what the leaders publish: `want` (anything else only in the test of the bounded wait)
  unsigned long long* err;  // host-mapped: set when a bounded wait for the record expired (the call then fails)
  int delay;                // tests only (FPSQ_DEBUG_RIDE_DELAY = c + 1): leader c idles ~100 us before it starts -- what another
                            // kernel holding its XCD would do to it; results must not depend on it
  unsigned int xseq;        // row-sharded handles whose sums over the ranks are formed inside the launch (fpsq_krylov.hip.h xch_sum): the
  const XchTable* xt;       // number of THIS leader set's exchange and the peer table (null: one GPU, a communicator of one, other routes)
  int more;                 // looks granted to the workgroups that wait for the record BEYOND kRidePolls: 0 on one GPU; with leaders that
                            // themselves wait for the peers' sums (xt != null on some leader set of the call), more than those may take
};
struct RideCoef {
  double ca[2], cb[2];
  bool act[2];
};

__device__ __forceinline__ unsigned long long ride_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ride_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (write-through: on its way to memory at once)
}

// The record.  Every 8-byte word carries the launch number in its low half and half a word of payload in its high half;
// a word is read and written in one piece, so whoever sees the number sees the payload that belongs to it -- no flag, no
// ordering between loads, no fence on the leader's critical path, ONE wave instruction per look.
//  * line 0, what a PRODUCT workgroup needs, written by the leader's thread 0 the moment the step is computed:
//      rec[4 l + 0 / 1] = high / low half of ca_l,  rec[4 l + 2 / 3] = of cb_l,  rec[8 + l] = (done | skip) != 0
//  * lines 1-3, the whole control blocks for the riding UPDATE workgroups (dispatched last), written right behind:
//      rec[16 + 24 l + 2 i + 0 / 1] = high / low half of word i (of 12) of lane l's LaneCtl
__device__ __forceinline__ void ride_publish(const LaneCtl* c, int l, unsigned long long* rec, unsigned int want) {
  const unsigned long long a = (unsigned long long)__double_as_longlong(c->ca), b = (unsigned long long)__double_as_longlong(c->cb);
  ride_store(rec + 4 * l + 0, (a & 0xffffffff00000000ull) | want);
  ride_store(rec + 4 * l + 1, (a << 32) | want);
  ride_store(rec + 4 * l + 2, (b & 0xffffffff00000000ull) | want);
  ride_store(rec + 4 * l + 3, (b << 32) | want);
  ride_store(rec + 8 + l, ((unsigned long long)((c->done | c->skip) != 0 ? 1u : 0u) << 32) | want);
}
__device__ __forceinline__ void ride_decode(const unsigned long long* w /* 10 words */, RideCoef& C) {
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    C.ca[l] = __longlong_as_double((long long)((w[4 * l] & 0xffffffff00000000ull) | (w[4 * l + 1] >> 32)));
    C.cb[l] = __longlong_as_double((long long)((w[4 * l + 2] & 0xffffffff00000000ull) | (w[4 * l + 3] >> 32)));
    C.act[l] = (w[8 + l] >> 32) == 0;
  }
}

// WHO LEADS.  The first kRideCand = 16 workgroups of the grid are leaders and nothing else: leader c computes the step of lane
// (c >> 3) & 1 -- eight per lane, redundantly (same inputs, same instructions, same bits) -- and publishes it in the record
// copy OF THE XCC IT RUNS ON (hardware register XCC_ID); a waiting workgroup looks at its own XCC's copy.  One writer per
// copy and lane, no election, no atomics.  The state is committed by leaders 0 and 8 alone (nobody inside the launch waits
// for that: the next launch reads it).  Why not just two leaders: workgroup i is dispatched by XCD i mod 8, in order WITHIN
// that XCD.  When another kernel (another stream, another process sharing the GPU) has an XCD full of workgroups that are
// themselves waiting for their leaders, a leader assigned to that XCD never starts while this kernel's workgroups on the
// other XCDs wait for it: a circular wait across kernels (seen as expired bounded waits with three processes on one GPU).
// With leaders c and c + 8 -- one per lane -- on every XCD, any running workgroup of this launch has both lanes' leaders of
// its own XCD dispatched ahead of it, whatever else fills the device.  (Measured on the way: an election by tickets BEFORE
// the work adds ~1.5 us to every launch -- cfg2 2450 -> 2300 evals/s; tickets in flight with the loads are no better: the
// atomic returns in order AHEAD of them; eight leaders writing ONE record collide on its lines -- headline 920 -> 830.)
constexpr int kRideCand = 16;
__device__ __forceinline__ int ride_xcc() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7; }

// Leader c of lane l.  k_step's body (step_run); line 0 of the record copy is published from the advanced state in LDS BEFORE
// the state is committed: that is all the product workgroups wait for.
struct RideHook {
  const RideArgs* ra;
  const unsigned long long* st80;
  unsigned long long* rec;  // this XCC's copy
  int l;
  int off;                  // StepArgs::prod_ctl_off
  __device__ void advanced() const { ride_publish(reinterpret_cast<const LaneCtl*>(st80 + off), l, rec, ra->pub); }
};
// may_commit = false (the head leaders of a fused launch): nobody here performs the step's side effects -- the mid leaders,
// which redo the step anyway, do, so that the progress words of the launch's two steps reach the host in order whichever
// leader is late.
template <bool XCH = false>
__device__ __forceinline__ void ride_leader(const StepArgs& a, int c, const RideArgs& ra, double* red32, unsigned long long* st80,
                                            bool may_commit = true) {
  const int l = (c >> 3) & 1;
  if (ra.delay != 0 && c == ra.delay - 1) {
    const unsigned long long t0 = wall_clock64();  // (100 MHz)
    while (wall_clock64() - t0 < 10000ull) __builtin_amdgcn_s_sleep(32);
  }
  unsigned long long* rec = ra.rec + 64 * ride_xcc();
  const RideHook hook{&ra, st80, rec, l, a.prod_ctl_off};
  if (a.kind != STEP_NONE) {
    step_run<RideHook, XCH>(a, red32, st80, /*commit=*/may_commit && (c & 7) == 0, &hook, ra.xt, ra.xseq, l);
  } else {
    if (threadIdx.x < 12) st80[threadIdx.x] = reinterpret_cast<const unsigned long long*>(a.state)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) hook.advanced();
  }
  if (threadIdx.x < 12) {  // (step_run's closing barrier: st80 is final for every thread)
    const unsigned long long w = st80[threadIdx.x];
    ride_store(rec + 16 + 24 * l + 2 * threadIdx.x, (w & 0xffffffff00000000ull) | ra.pub);
    ride_store(rec + 16 + 24 * l + 2 * threadIdx.x + 1, (w << 32) | ra.pub);
  }
}

// One look of a PRODUCT workgroup, by the calling wave (one wave per workgroup looks): lanes 0..9 each request one word ...
__device__ __forceinline__ unsigned long long ride_look(const RideArgs& ra) {
  const int lane = threadIdx.x & 63;
  return lane < 10 ? ride_load(ra.rec + 64 * ride_xcc() + lane) : 0ull;
}
// ... and its verdict (wave-uniform); a good look leaves the ten words in `crec` (LDS)
__device__ __forceinline__ bool ride_take(const RideArgs& ra, unsigned long long w, unsigned long long* crec) {
  const int lane = threadIdx.x & 63;
  const bool good = lane >= 10 || (unsigned int)(w & 0xffffffffull) == ra.want;
  if (!__all(good)) return false;
  if (lane < 10) crec[lane] = w;
  return true;
}
// The same for an UPDATE workgroup: lanes 0..47, the two control blocks end up in LDS at img[0..11] and img[80..91].
__device__ __forceinline__ bool ride_take_ctl(const RideArgs& ra, unsigned long long* img) {
  const int lane = threadIdx.x & 63;
  const unsigned long long w = lane < 48 ? ride_load(ra.rec + 64 * ride_xcc() + 16 + lane) : 0ull;
  const bool good = lane >= 48 || (unsigned int)(w & 0xffffffffull) == ra.want;
  const unsigned int nlo = (unsigned int)__shfl_down((unsigned int)(w >> 32), 1, 64);  // the partner's payload (low half of the word)
  if (!__all(good)) return false;
  if (lane < 48 && (lane & 1) == 0) {
    const int k = lane >> 1;  // word k % 12 of lane k / 12
    img[(k / 12) * 80 + k % 12] = (w & 0xffffffff00000000ull) | nlo;
  }
  return true;
}

// The leaders are the first two workgroups of the grid -- dispatched first, running before any workgroup that waits for them
// -- and need a few microseconds.  A wait nevertheless has an end every wave reaches: after kRidePolls looks (~1 us each,
// tens of milliseconds) the workgroup raises the handle's error word and leaves without writing anything (the call returns
// FPSQ_ERR_TIMEOUT).  (Recomputing the steps locally instead -- step_run twice -- was built first: inlined it took the product
// kernels from 81 to 256 registers, as a called function it put their waves on a scratch stack: 64 us per product.)
constexpr int kRidePolls = 1 << 15;

// A workgroup whose looks have all failed by the time it needs the coefficients (every thread calls this; workgroup barriers
// inside): wave 0 keeps looking (CTL: for the whole control blocks, an update workgroup).  Returns false when the bound expired.
template <bool CTL, int SLEEP = 4>
__device__ __forceinline__ bool ride_settle(const RideArgs& ra, unsigned long long* dst, int* got) {
  if (threadIdx.x == 0) *got = 0;
  __syncthreads();
  for (int t = 0; t < kRidePolls + ra.more; ++t) {
    if (threadIdx.x < 64) {
      if (t) __builtin_amdgcn_s_sleep(SLEEP);
      const bool ok = CTL ? ride_take_ctl(ra, dst) : ride_take(ra, ride_look(ra), dst);
      if (ok && threadIdx.x == 0) *got = 1;
    }
    __syncthreads();
    if (*got) return true;
  }
  if (threadIdx.x == 0) __hip_atomic_store(ra.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return false;
}

// The A' product (two lanes, padded blocks with block-relative columns, one GPU) of a launch with riding leaders: k_spmv's
// main path with the coefficients taken as described above.  grid = kRideCand candidates + nwg product workgroups + the riding updates.
// TWO BLOCKS for the first workgroups: the first resident set starts together with the leaders and would reach its row
// epilogue before the record is up (2.6 us of waiting per launch, measured against a build that does not wait).  Product
// workgroup b < n2 therefore takes blocks b and n2 + b: row sums of the first block are HELD in registers, the second block
// is streamed, and both epilogues run behind it -- by then the record has long arrived.  Workgroups b >= n2 (dispatched as
// slots free up, ~10 us into the launch) take the single block n2 + b.  Same blocks, same per-block arithmetic: bitwise.
constexpr int kAtlPass = kMaxRowsPerBlk / kBlock;  // row passes of a block (one lane per row at most kBlock rows per pass)
// HALO: the row-sharded form (see k_spmv<.., HALO>): overlap rows only deposit their raw sums, k_halo_finish completes them.
// Which row blocks product workgroup b of an A' launch with leaders takes (false: none).  The first n2 workgroups take two.
// bpx > 0: XCD-contiguous eighths -- workgroup b runs on XCD b & 7 (whatever precedes the product workgroups in the grid fills a
// multiple of eight slots) and walks blocks [e bpx, (e + 1) bpx) of eighth e = b & 7; its first n2 / 8 workgroups take two
// blocks each.  bpx = 0: grid order.
// rot (tests of the fused launch only, FPSQ_DEBUG_FUSE_ROTATE): the workgroup on XCD b & 7 walks eighth (b + rot) & 7 -- every
// row a row group waits for has then been written on ANOTHER XCD.
__device__ __forceinline__ bool atl_blocks_of(int b, int n2, int bpx, int nblk, int (&Lt)[2], int& nt, int rot = 0) {
  nt = b < n2 ? 2 : 1;
  Lt[0] = b < n2 ? b : n2 + b;
  Lt[1] = n2 + b;
  if (bpx > 0) {
    const int e = (b + rot) & 7, j = b >> 3, n2e = n2 >> 3;
    nt = j < n2e ? 2 : 1;
    Lt[0] = e * bpx + (j < n2e ? j : n2e + j);
    Lt[1] = e * bpx + n2e + j;
    const int end = min((e + 1) * bpx, nblk);
    if (Lt[0] >= end) return false;
    if (Lt[1] >= end) nt = 1;
  }
  return Lt[0] < nblk;
}

// What a product workgroup of a FUSED launch (k_iter_fused: the A' product and the A product of one iteration in one grid) needs
// besides the products' own arguments.  An A' block publishes itself once its rows of the long pair are written through; a row
// group of A starts gathering when the blocks that own the 128-byte lines it reads have done so; the scalar steps behind the
// A' product are computed by a second set of leaders once every block has counted itself in.
struct FuseArgs {
  unsigned int* blkflag;          // [blocks of A']: this launch's number once block L's rows are at their coherence point
  unsigned long long* ptag;       // [blocks][4] self-validating words: the blocks' squared-norm partials (lane 0 high, low; lane 1 high, low)
  const int2* dep;                // per row group of A: first and last A' block owning rows on the lines the group gathers from
  const int2* dep2;               // halo-sharded handles (null otherwise): a second range per group -- the finish workgroups of
                                  // the overlap rows (their flags follow the blocks': index nblk + b), {1, 0} = none
  unsigned int want;              // this launch's number
  unsigned int pub;               // what the blocks publish: `want` (anything else only in the test of the bounded waits)
  unsigned long long* err;        // host-mapped: a bounded wait expired
  unsigned long long* dbg;        // developer probe (null in production): four 100 MHz time stamps per workgroup of the launch, see fuse_stamp
  int more;                       // looks beyond kRidePolls (see RideArgs::more: what a block waits for may be waiting for the peers)
};
// workgroup's stamp k (thread 0; tools/fuse_probe.py reads them): 0 = entry, 1 = dependences met (row group) / record taken (update),
// 2 = tiles done (row group) / partials summed (mid leader), 3 = exit
__device__ __forceinline__ void fuse_stamp(const FuseArgs& fz, int k) {
  if (fz.dbg != nullptr && threadIdx.x == 0) fz.dbg[(size_t)blockIdx.x * 4 + k] = wall_clock64();
}
__device__ __forceinline__ unsigned long long tag_hi(double v, unsigned int want) {
  return ((unsigned long long)__double_as_longlong(v) & 0xffffffff00000000ull) | want;
}
__device__ __forceinline__ unsigned long long tag_lo(double v, unsigned int want) {
  return ((unsigned long long)__double_as_longlong(v) << 32) | want;
}

// ---- SEVERAL joint iterations in one launch (k_iter_multi, fpsq_multi.hip.h; round 5): what a product workgroup of iteration j
// needs beyond the one-launch iteration's FuseArgs.  The A -> A' boundary between two iterations is inside the launch too: an A'
// block of iteration j + 1 starts gathering the short pair when the row groups of iteration j that wrote what it gathers have
// published themselves (rows written through, per-group flags -- the mirror image of blocks -> row groups), and a row group
// leaves its squared-norm partials as tagged words for the head leaders of iteration j + 1.
struct MultiCtx {
  const unsigned int* gflag_prev;  // per row group: launch number of iteration j - 1 once its rows are at their coherence point
                                   // (null: the previous iteration was another launch -- nothing to wait for)
  unsigned int want_prev;
  const int2* bdep;                // per A' block: first and last row group whose rows of the short pair it gathers
  unsigned int* gflag;             // this iteration's row-group flags ...
  unsigned long long* atag;        // ... and tagged partials [group][4] (lane 0 high, low; lane 1 high, low)
  unsigned int want;               // this iteration's number
  const unsigned long long* hdone; // both 32-bit halves non-zero: every recurrence of the call has ended.  A workgroup that finds it
                                   // so AT ITS ENTRY publishes itself (flag / tagged words, payload void) and leaves without streaming --
                                   // nobody who waits for it is left waiting, whatever the order the XCDs dispatch in; past its entry a
                                   // workgroup never looks again (what it waits for was entered, or publishes on leaving, too)
  unsigned long long hd;           // ... the word as this workgroup's FIRST load found it.  A product workgroup examines it only behind
                                   // its first stream loads (requested earlier, it returns first: no round trip of its own at the head --
                                   // measured: a dependent look at the entry of every workgroup cost the launch ~6 % )
};
__device__ __forceinline__ bool multi_over(unsigned long long w) { return (unsigned int)w != 0u && (unsigned int)(w >> 32) != 0u; }
__device__ __forceinline__ bool multi_all_done(const unsigned long long* hdone) {
  const unsigned long long w = __hip_atomic_load(hdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (unsigned int)w != 0u && (unsigned int)(w >> 32) != 0u;
}
// EVERY wave of the workgroup looks for itself (no LDS, no barrier): the row groups [dx, dy] of the previous iteration have published
// themselves.  `first`: what this wave's lanes found with the look requested at the block's head (ahead of the stream loads, so it is
// back before them); only if that was too early does the wave poll (bounded; an expired bound raises the error word and the
// wave goes on -- the call fails anyway, and nobody leaves a barrier behind).
__device__ __forceinline__ unsigned int multi_look_groups(const MultiCtx& mx, int dx, int dy) {
  const int g = dx + (int)(threadIdx.x & 63);
  return g <= dy ? __hip_atomic_load(mx.gflag_prev + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mx.want_prev;
}
__device__ __forceinline__ void multi_wait_groups(const MultiCtx& mx, int dx, int dy, unsigned int first, unsigned long long* err) {
  const int lane = threadIdx.x & 63;
  bool ok = first == mx.want_prev;
  for (int g = dx + 64 + lane; g <= dy; g += 64)  // (a range of more than 64 groups: the rest)
    ok = ok && __hip_atomic_load(mx.gflag_prev + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mx.want_prev;
  if (__all(ok)) return;
  for (int t = 0; t < kRidePolls; ++t) {
    __builtin_amdgcn_s_sleep(4);
    ok = true;
    for (int g = dx + lane; g <= dy; g += 64)
      ok = ok && __hip_atomic_load(mx.gflag_prev + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mx.want_prev;
    if (__all(ok)) return;
  }
  if (lane == 0) __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The A' product of one workgroup: blocks Lt[0 .. nt) of the padded layout, two lanes, coefficients from the leaders' record.
// FUSED (k_iter_fused): rows are written through, and behind each block's epilogue -- every wave has waited for the
// acknowledgement of its stores BEFORE the workgroup barrier of the partial sum -- thread 0 publishes the block: its squared-norm
// partials as self-validating words and its flag.
template <bool CSORT, bool HALO, bool FUSED, bool MULTI = false>
__device__ __forceinline__ void atl_product(const CsrView& A, const double* __restrict__ x, const double* yin, double* yout,
                                            double* partials, int pstride, const int (&Lt)[2], int nt, const RideArgs& ra,
                                            const HaloRows& hr, const FuseArgs& fz, double* prod, unsigned long long* crec,
                                            int* okfp, const MultiCtx* mx = nullptr) {
  static_assert(!MULTI || (FUSED && !HALO), "several iterations per launch: the one-launch iteration's hand-overs, one GPU");
  constexpr int NL = 2;
  double* red = prod;
  const int tid = threadIdx.x;
#define okf (*okfp)
  // Wave 0 looks for the leaders' record at every point where a wait is free anyway: behind the stream (examined behind
  // the gathers), behind the gathers, behind the products, behind the row sums.  Device-scope loads come back later than
  // ordinary ones and the load counter is in-order, so each look is requested BEHIND the loads whose arrival the next
  // step waits for.  peek(): examine the look in flight, request another one if it was too early.
  bool ok = false, flying = false;
  unsigned long long lk = 0;
  auto peek = [&]() {
    if (tid < 64 && !ok) {
      if (flying) ok = ride_take(ra, lk, crec);
      if (!ok) {
        lk = ride_look(ra);
        flying = true;
      }
    }
  };
  // what a block leaves behind for its epilogue
  int hr0[2], hnr[2], hG[2];
  double hacc[2][kAtlPass][NL], hypre[2][NL];
  constexpr int kPer = kSpmvNnz / kBlock;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t >= nt) break;
    const int L = Lt[t];
    int cidx[kPer];
    [[maybe_unused]] int slot[kPer];
    double v[kPer];
    bool shared = false;
    if constexpr (CSORT) shared = A.segdesc != nullptr;  // (kernel-uniform)
    [[maybe_unused]] uint4 sw16;
    [[maybe_unused]] uint2 sw8;
    int r0, nr, s, cbase;
    [[maybe_unused]] unsigned int gl0 = 0;
    [[maybe_unused]] int2 bd2 = make_int2(1, 0);
    if constexpr (MULTI) {  // the first look at the row groups this block waits for: requested ahead of everything else
      if (mx->gflag_prev != nullptr) {
        bd2 = mx->bdep[L];
        gl0 = multi_look_groups(*mx, bd2.x, bd2.y);
      }
    }
    if (CSORT && shared) {  // the block's head through the scalar cache, value loads issued first (cshared_head)
      if constexpr (CSORT) cshared_head(A, L, tid, sw16, sw8, r0, nr, s, cbase, v);
    } else {
      const int4 bd = A.blkdesc[L];
      cbase = A.colbase[L];
      r0 = bd.x, nr = bd.y, s = bd.z;
    }
    int G = 1;
    while (G < 64 && G * 2 * nr <= kBlock) G <<= 1;
    const int rows_per_pass = kBlock / G;
    const int g = tid / G, gl = tid % G;
    const int rq0 = g < nr ? g : 0;
    const int seg_a0 = A.rowptr[r0 + rq0], seg_b0 = A.rowptr[r0 + rq0 + 1];
    hypre[t][0] = hypre[t][1] = 0.0;
    // (HALO: never an overlap row -- this workgroup does not finish those, a finish workgroup on ANOTHER CU writes them later in
    // the same launch, and a line this load left in the CU's L1 would be served, stale, to the long-vector update workgroups that
    // read the finished rows at the end of the launch.  Found by tools/lx_soak_mp.py, round 5: ~1 evaluation in 10^3 with a few
    // 128-byte lines of the solution's overlap rows one iteration behind.)
    if (!MULTI && yin != nullptr && !(HALO && (r0 + rq0 < hr.lo || r0 + rq0 >= hr.hi))) {
      const double2 yy = *reinterpret_cast<const double2*>(yin + (size_t)(r0 + rq0) * 2);
      hypre[t][0] = yy.x;
      hypre[t][1] = yy.y;
    }
    if constexpr (CSORT) {
      if (shared) cshared_decode(sw16, sw8, cbase, cidx, slot);
      else csort_fetch(A, L, cbase, tid, cidx, slot, v);
    } else {
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const size_t ii = (size_t)L * kSpmvNnz + tid + k * kBlock;
        cidx[k] = cbase + (int)A.col16[ii];
        v[k] = A.vals[ii];
      }
    }
    if (t == 0) peek();  // (requests the first look)
    if constexpr (MULTI) {
      // the row groups of the previous iteration that wrote what this block gathers (the stream above does not depend on them).
      // They in turn waited for this block's own previous incarnation: only behind this wait are its `yin` rows -- written one
      // iteration ago into the other long pair -- there to be read (the one-launch iteration prefetches them at the block's head)
      if (t == 0 && multi_over(mx->hd)) {  // (the call has ended: publish the blocks with a void payload, stream nothing more)
        if (tid == 0)
          for (int u = 0; u < nt; ++u) {
            unsigned long long* pt = fz.ptag + (size_t)Lt[u] * 4;
            for (int w = 0; w < 4; ++w) ride_store(pt + w, (w & 1) ? tag_lo(0.0, fz.want) : tag_hi(0.0, fz.want));
            __hip_atomic_store(fz.blkflag + Lt[u], fz.want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        return;
      }
      if (mx->gflag_prev != nullptr) multi_wait_groups(*mx, bd2.x, bd2.y, gl0, fz.err);
      if (yin != nullptr) {
        const double2 yy = ld_pair_ag(yin, r0 + rq0);  // (agent scope: not a line this CU's L1 kept from two iterations ago)
        hypre[t][0] = yy.x;
        hypre[t][1] = yy.y;
      }
    }
    double2 xv[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if constexpr (MULTI) xv[k] = ld_pair_ag(x, cidx[k]);  // (agent scope: rows other XCDs have just written through)
      else xv[k] = *reinterpret_cast<const double2*>(x + (size_t)cidx[k] * 2);
    }
    peek();
#pragma unroll
    for (int k = 0; k < kPer; ++k)
      *reinterpret_cast<double2*>(prod + 2 * (CSORT ? slot[k] : tid + k * kBlock)) = make_double2(v[k] * xv[k].x, v[k] * xv[k].y);
    peek();
    lds_barrier();  // (publishes the products)
    hr0[t] = r0;
    hnr[t] = nr;
    hG[t] = G;
#pragma unroll
    for (int p = 0; p < kAtlPass; ++p) {
      const int rr = p * rows_per_pass + g;
      hacc[t][p][0] = hacc[t][p][1] = 0.0;
      if (p * rows_per_pass < nr) {  // (workgroup-uniform)
        if (rr < nr) {
          const int a = (p == 0 ? seg_a0 : A.rowptr[r0 + rr]) - s, e = (p == 0 ? seg_b0 : A.rowptr[r0 + rr + 1]) - s;
          row_segment_sum<NL>(prod, a + gl, e, G, hacc[t][p]);
        }
        for (int off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
          for (int l = 0; l < NL; ++l) hacc[t][p][l] += __shfl_down(hacc[t][p][l], off, 64);
        }
      }
    }
    if (t + 1 < nt) lds_barrier();  // every wave is past its row sums: the next block's products may overwrite `prod`
  }
  // the coefficients: first needed here (every thread passes this point)
  if (tid < 64) {
    if (!ok && flying) ok = ride_take(ra, lk, crec);
    if (tid == 0) okf = ok ? 1 : 0;
  }
  lds_barrier();  // (publishes the record words and the verdict; `red` aliases `prod`: every wave is past its row sums)
  if (!okf && !ride_settle<false>(ra, crec, &okf)) return;  // (workgroup-uniform; single-block workgroups of a small grid at most)
  RideCoef C;
  ride_decode(crec, C);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t >= nt) break;
    const int G = hG[t], rows_per_pass = kBlock / G, g = tid / G, gl = tid % G;
    double sq[NL] = {0.0, 0.0};
#pragma unroll
    for (int p = 0; p < kAtlPass; ++p) {
      const int rr = p * rows_per_pass + g;
      if (rr < hnr[t] && gl == 0) {
        const int64_t row = hr0[t] + rr;
        if (HALO && (row < hr.lo || row >= hr.hi)) {
          double* dst = hr.raw + (size_t)(row < hr.lo ? row : hr.lo + (row - hr.hi)) * NL;
#pragma unroll
          for (int l = 0; l < NL; ++l)
            if (C.act[l]) {
              // (FUSED: the halo workgroups of the SAME launch read the raw sums -- written through like the rows)
              if constexpr (FUSED) __hip_atomic_store(dst + l, hacc[t][p][l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else dst[l] = hacc[t][p][l];
            }
        } else {
          row_epilogue<NL, FUSED, MULTI>((size_t)row, hacc[t][p], C.ca, C.cb, C.act, yin, yout, sq, p == 0 && yin != nullptr ? hypre[t] : nullptr);
        }
      }
    }
    if constexpr (FUSED) {
      // block_sum_lanes with the store acknowledgements waited for in front of its barrier: behind it every row of the block
      // is at its coherence point, and what thread 0 issues next may tell other XCDs so
      if (t) lds_barrier();
#pragma unroll
      for (int l = 0; l < NL; ++l) sq[l] = wave_sum(sq[l]);
      if ((tid & 63) == 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) red[(tid >> 6) * NL + l] = sq[l];
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (tid == 0) {
        unsigned long long* pt = fz.ptag + (size_t)Lt[t] * 4;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          const double v = (red[l] + red[NL + l]) + (red[2 * NL + l] + red[3 * NL + l]);
          ride_store(pt + 2 * l, tag_hi(v, fz.pub));
          ride_store(pt + 2 * l + 1, tag_lo(v, fz.pub));
        }
        __hip_atomic_store(fz.blkflag + Lt[t], fz.pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (partials != nullptr) {
      if (t) lds_barrier();  // (thread 0 has read `red` for the previous block)
      block_sum_lanes<NL>(sq, red);
      if (tid == 0) {
#pragma unroll
        for (int l = 0; l < NL; ++l) partials[(size_t)l * pstride + Lt[t]] = sq[l];
      }
    }
  }
#undef okf
}

template <bool CSORT, bool HALO = false>
__global__ __launch_bounds__(kBlock) void k_spmv_atl(CsrView A, const double* __restrict__ x, const double* yin, double* yout,
                                                     double* partials, int nwg, int n2, const UpdSeg u0, const UpdSeg u1,
                                                     int pstride, const StepArgs s0, const StepArgs s1, const RideArgs ra,
                                                     const HaloRows hr, int bpx) {
  constexpr int NL = 2;
  __shared__ double prod[kSpmvNnz * NL];
  __shared__ __attribute__((aligned(16))) unsigned long long fst[2 * 80];
  __shared__ double fred[32];
  __shared__ unsigned long long crec[10];
  __shared__ int okf;
  if (blockIdx.x < kRideCand) {
    ride_leader<HALO>(((int)blockIdx.x >> 3) & 1 ? s1 : s0, (int)blockIdx.x, ra, fred, fst);  // (HALO = a sharded handle: may exchange)
    return;
  }
  const int b = (int)blockIdx.x - kRideCand;
  if (b >= nwg) {  // a riding-update workgroup (dispatched last: the record is up long before)
    if (ride_settle<true>(ra, fst, &okf))
      run_fused_updates<NL>(u0, u1, nwg + kRideCand, prod, reinterpret_cast<const LaneCtl*>(fst), reinterpret_cast<const LaneCtl*>(fst + 80));
    return;
  }
  int Lt[2], nt;
  if (!atl_blocks_of(b, n2, bpx, A.nblk, Lt, nt)) return;
  atl_product<CSORT, HALO, false>(A, x, yin, yout, partials, pstride, Lt, nt, ra, hr, FuseArgs{}, prod, crec, &okf);
}

// ------------------------------------------------------------------------------------------------ RGCS product
//
// RGCS = Row Groups, Column-Sorted.  The CSR-stream kernel gathers x in ROW order: for a wide matrix whose rows spread
// over thousands of columns (the constraint Jacobian: 100 nonzeros in an 8192-column window) the 64 gathers of one
// wave instruction land in 64 different 128-byte lines and the L2->L1 line traffic, not HBM, bounds the product
// (rocSPARSE's csrmv hits the same wall: tools/spmv_bench.hip).  Here the entries of a group of consecutive rows
// (<= kRgcsGroupNnz nonzeros, <= kRgcsMaxRows rows) are stored sorted by COLUMN and cut into tiles of kRgcsTile
// entries, so consecutive lanes gather neighbouring columns (~10 lines per wave instruction).  Each entry carries, packed
// with its group-relative column in one 32-bit word (12 B/nnz like CSR), its slot in the tile's ROW-major order:
// products are scattered to LDS by slot and every row's segment is reduced exactly as in the CSR-stream kernel;
// the per-row sums accumulate in registers across the tiles of the group.  Deterministic, no atomics.
#ifndef FPSQ_RGCS_TILE
#define FPSQ_RGCS_TILE 2048
#endif
#ifndef FPSQ_RGCS_GROUP_NNZ
#define FPSQ_RGCS_GROUP_NNZ 12800
#endif
#ifndef FPSQ_RGCS_MAX_ROWS
#define FPSQ_RGCS_MAX_ROWS 128
#endif
constexpr int kRgcsTile = FPSQ_RGCS_TILE;
constexpr int kRgcsColBits = 21;     // group-relative column < 2^21, slot < 2^11
constexpr int kRgcsGroupNnz = FPSQ_RGCS_GROUP_NNZ;
constexpr int kRgcsMaxRows = FPSQ_RGCS_MAX_ROWS;
// row passes of the segment reduction: G lanes per row with G * R <= kBlock, so rows <= kBlock need a single pass
constexpr int kRgcsMaxPass = (kRgcsMaxRows + kBlock - 1) / kBlock;
static_assert(kRgcsTile <= 2048 && kRgcsTile % kBlock == 0, "slot field is 11 bits");

struct RgcsGroup {        // 32 bytes, fetched with two independent 16-byte loads at the head of the workgroup
  int32_t r0, R;          // first row, #rows
  int32_t e0, e1;         // entry range
  int32_t cmin;           // smallest column of the group
  int32_t tp;             // offset into tptr
  int32_t pad[2];
};

struct RgcsView {
  const uint32_t* pidx;   // (slot << kRgcsColBits) | (col - cmin)
  const double* vals;     // same (column-sorted) order
  const RgcsGroup* grp;   // ng group descriptors
  const uint16_t* tptr;   // per tile: R + 1 row-segment boundaries in slot space
  int32_t ng;
  int32_t nrows;
  int32_t stride;         // padded layout: group g's entries start at g * stride (0: compact, start = grp[g].e0)
};

// One row group of the A product (see above).  LEAD: coefficients from the leaders' record `ra`; FUSED (k_iter_fused, implies
// LEAD): the group first waits for the A' blocks of the same launch that own what it gathers, and gathers at agent scope.
template <int NL, bool PAD, bool LEAD, bool FUSED, bool MULTI = false>
__device__ __forceinline__ void rgcs_group(const RgcsView& M, const double* __restrict__ x, const double* yin, double* yout,
                                           const LaneCtl* ctl0, const LaneCtl* ctl1, double* partials, int pstride, int g,
                                           const RideArgs& ra, const FuseArgs& fz, double* prod, unsigned long long* crec,
                                           int* okf, const MultiCtx* mx = nullptr) {
  static_assert(!MULTI || (FUSED && NL == 2), "several iterations per launch: built on the one-launch iteration");
  double* red = prod;  // aliases the product buffer (free again after the tile loop's closing barrier): exactly 32 KB
  if (g >= M.ng) return;
  constexpr int kPer = kRgcsTile / kBlock;
  const int tid = threadIdx.x;
  uint32_t pk[kPer];
  double v[kPer];
  // everything the next tile needs from global memory, issued together and left untouched until it is consumed
  auto fetch_stream = [&](int base, int lo, int hi) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = base + tid + k * kBlock;
      const int ii = PAD ? i : (i < hi ? i : lo);
      pk[k] = M.pidx[ii];
      v[k] = M.vals[ii];
    }
  };
  if (PAD) fetch_stream(g * M.stride, 0, 0);  // ahead of the descriptor
  [[maybe_unused]] unsigned long long look = 0;
  const RgcsGroup gd = M.grp[g];  // before the dependent done-check: one round trip at the head, not two
  double ca[NL], cb[NL];
  bool act[NL];
  [[maybe_unused]] bool ride_ok = false;  // (wave 0's)
  if constexpr (!LEAD) {
    const LaneCtl* c[2] = {ctl0, ctl1};
    bool any = false;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      act[l] = !(c[l]->done | c[l]->skip);
      ca[l] = c[l]->ca;
      cb[l] = c[l]->cb;
      any |= act[l];
    }
    if (!any) return;
  }
  const int r0 = gd.r0, R = gd.R, e0 = PAD ? g * M.stride : gd.e0, e1 = PAD ? g * M.stride + (gd.e1 - gd.e0) : gd.e1;
  const int cmin = gd.cmin;
  const uint16_t* tp = M.tptr + gd.tp;
  int G = 1;
  while (G < 64 && G * 2 * R <= kBlock) G <<= 1;
  const int rpp = kBlock / G, gid = tid / G, gl = tid % G;
  double acc[kRgcsMaxPass][NL];
#pragma unroll
  for (int p = 0; p < kRgcsMaxPass; ++p)
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[p][l] = 0.0;
  uint32_t traw[kRgcsMaxPass];  // this tile's LDS segment [a, b) of each of my rows: two uint16 in one raw dword
  auto fetch_segs = [&](int tile) {
    const uint16_t* tpt = tp + (size_t)tile * (R + 1);
#pragma unroll
    for (int p = 0; p < kRgcsMaxPass; ++p) {
      const int rr = p * rpp + gid;
      const int rq = rr < R ? rr : 0;
      __builtin_memcpy(&traw[p], tpt + rq, 4);
    }
  };
  auto fetch = [&](int base, int tile) {
    fetch_stream(base, e0, e1);
    fetch_segs(tile);
  };
  if (PAD) fetch_segs(0);
  else fetch(e0, 0);
  if constexpr (MULTI) {
    if (multi_over(mx->hd)) {  // (the call has ended: publish the group with a void payload; examined behind the first stream loads)
      if (tid == 0) {
        unsigned long long* pt = mx->atag + (size_t)g * 4;
        for (int w = 0; w < 4; ++w) ride_store(pt + w, (w & 1) ? tag_lo(0.0, mx->want) : tag_hi(0.0, mx->want));
        __hip_atomic_store(mx->gflag + g, mx->want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
  }
  if constexpr (FUSED) {
    // the A' blocks that own the lines this group gathers from: wave 0 looks at their flags (agent-scope loads, requested
    // behind the first tile's stream, which does not depend on them), a bounded number of times
    if (tid < 64) {
      const int2 d = fz.dep[g];
      int d2x = 1, d2y = 0;  // (scalars, not a conditional of structs: that one costs a stack slot)
      if (fz.dep2 != nullptr) {
        const int2 t2 = fz.dep2[g];
        d2x = t2.x;
        d2y = t2.y;
      }
      bool all = false;
      for (int t = 0; t < kRidePolls + fz.more && !all; ++t) {
        if (t) __builtin_amdgcn_s_sleep(8);
        bool ok = true;
        for (int L = d.x + tid; L <= d.y; L += 64)
          ok &= __hip_atomic_load(fz.blkflag + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fz.want;
        for (int L = d2x + tid; L <= d2y; L += 64)
          ok &= __hip_atomic_load(fz.blkflag + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fz.want;
        all = __all(ok);
      }
      if (tid == 0) {
        *okf = all ? 1 : 0;
        if (!all) __hip_atomic_store(fz.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    lds_barrier();
    if (!*okf) return;  // (workgroup-uniform: a block never came -- the call fails with FPSQ_ERR_TIMEOUT)
    fuse_stamp(fz, 1);
  }
  if constexpr (LEAD) {  // first look (wave 0): requested BEHIND the first tile's stream (device-scope loads return late, in order)
    if (tid < 64) look = ride_look(ra);
  }
  int tile = 0;
  for (int base = e0; base < e1; base += kRgcsTile, ++tile) {
    int sa[kRgcsMaxPass], sb[kRgcsMaxPass];
    {
      double2 xv[kPer];
      uint32_t pq[kPer];
      double vq[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        // out-of-range lanes park a zero in an unused slot (PAD: the stored padding entries already say so)
        const bool ok = PAD || base + tid + k * kBlock < e1;
        pq[k] = ok ? pk[k] : ((uint32_t)(tid + k * kBlock) << kRgcsColBits);
        vq[k] = ok ? v[k] : 0.0;
        const int col = cmin + (int)(pq[k] & ((1u << kRgcsColBits) - 1));
        if (NL == 1) xv[k].x = x[col];
        else if (FUSED) xv[k] = ld_pair_ag(x, col);  // (agent scope: rows another XCD has just written through)
        else xv[k] = *reinterpret_cast<const double2*>(x + (size_t)col * 2);
      }
#pragma unroll
      for (int p = 0; p < kRgcsMaxPass; ++p) {
        const bool valid = p * rpp + gid < R;
        sa[p] = (int)(traw[p] & 0xffffu);
        sb[p] = valid ? (int)(traw[p] >> 16) : sa[p];
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const int slot = (int)(pq[k] >> kRgcsColBits);
        if (NL == 1) prod[slot] = vq[k] * xv[k].x;
        else *reinterpret_cast<double2*>(prod + 2 * slot) = make_double2(vq[k] * xv[k].x, vq[k] * xv[k].y);
      }
    }
    if (base + kRgcsTile < e1) fetch(base + kRgcsTile, tile + 1);
    if constexpr (LEAD) {
      // the coefficients are only needed behind the last tile: look at what the previous request brought, ask again
      if (tid < 64 && !ride_ok) {
        ride_ok = ride_take(ra, look, crec);
        if (!ride_ok) look = ride_look(ra);
      }
    }
    lds_barrier();
#pragma unroll
    for (int p = 0; p < kRgcsMaxPass; ++p) row_segment_sum<NL>(prod, sa[p] + gl, sb[p], G, acc[p]);
    lds_barrier();
  }
  if constexpr (FUSED) fuse_stamp(fz, 2);
  if constexpr (LEAD) {
    if (tid < 64) {
      if (!ride_ok) ride_ok = ride_take(ra, look, crec);
      if (tid == 0) *okf = ride_ok ? 1 : 0;
    }
    lds_barrier();
    if (!*okf && !ride_settle<false>(ra, crec, okf)) return;  // (workgroup-uniform, rare)
    RideCoef RC;
    ride_decode(crec, RC);
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      ca[l] = RC.ca[l];
      cb[l] = RC.cb[l];
      act[l] = RC.act[l];
    }
  }
  double sq[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = 0.0;
#pragma unroll
  for (int p = 0; p < kRgcsMaxPass; ++p) {
    for (int off = G >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[p][l] += __shfl_down(acc[p][l], off, 64);
    }
    const int rr = p * rpp + gid;
    if (rr < R && gl == 0) row_epilogue<NL, MULTI, MULTI>((size_t)(r0 + rr), acc[p], ca, cb, act, yin, yout, sq);
  }
  if constexpr (MULTI) {
    // the A' blocks of the NEXT iteration of this launch gather these rows: written through (row_epilogue<.., WT>), their
    // acknowledgements waited for in front of the barrier of the partial sum (block_sum_lanes' arithmetic), then thread 0
    // publishes the group: partials (plain, for a consumer in a later launch, and tagged, for the next head leaders) and its flag
#pragma unroll
    for (int l = 0; l < NL; ++l) sq[l] = wave_sum(sq[l]);
    if ((tid & 63) == 0) {
#pragma unroll
      for (int l = 0; l < NL; ++l) red[(tid >> 6) * NL + l] = sq[l];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid == 0) {
      unsigned long long* pt = mx->atag + (size_t)g * 4;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const double v = (red[l] + red[NL + l]) + (red[2 * NL + l] + red[3 * NL + l]);
        if (partials != nullptr) partials[(size_t)l * pstride + g] = v;
        ride_store(pt + 2 * l, tag_hi(v, mx->want));
        ride_store(pt + 2 * l + 1, tag_lo(v, mx->want));
      }
      __hip_atomic_store(mx->gflag + g, mx->want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (partials != nullptr) {
    block_sum_lanes<NL>(sq, red);
    if (tid == 0) {
#pragma unroll
      for (int l = 0; l < NL; ++l) partials[(size_t)l * pstride + g] = sq[l];
    }
  }
}

// PAD: the groups' entries are stored at a fixed stride and zero-padded to whole tiles, so the first tile's stream
// does not wait for the group descriptor and no load needs a bounds check.
// LEAD: the launch's first kRideCand workgroups are the candidates for leading the riding steps (see "steps riding with leaders" above); the
// groups and the riding updates follow, the coefficients are picked up between the tiles.
// XCH (with LEAD): the leaders may form their sums over the ranks in the launch (a sharded handle; see step_run)
template <int NL, bool PAD = false, bool LEAD = false, bool XCH = false>
__global__ __launch_bounds__(kBlock) void k_spmv_rgcs(RgcsView M, const double* __restrict__ x, const double* yin,
                                                      double* yout, const LaneCtl* ctl0, const LaneCtl* ctl1,
                                                      double* partials, int grp_per_xcd, const UpdSeg u0,
                                                      const UpdSeg u1, const LaneCtl* gate0, const LaneCtl* gate1,
                                                      int pstride, const StepArgs s0, const StepArgs s1, const RideArgs ra) {
  __shared__ double prod[kRgcsTile * NL];
  static_assert(!LEAD || NL == 2, "riding leaders: two lanes");
  [[maybe_unused]] unsigned long long* fst = nullptr;
  [[maybe_unused]] double* fred = nullptr;
  [[maybe_unused]] unsigned long long* crec = nullptr;
  [[maybe_unused]] int* okf = nullptr;
  if (gate0 != nullptr && !(gate0->done && gate1->done)) return;
  if constexpr (LEAD) {
    __shared__ __attribute__((aligned(16))) unsigned long long fst_[2 * 80];
    __shared__ double fred_[32];
    __shared__ unsigned long long crec_[10];
    __shared__ int okf_;
    fst = fst_;
    fred = fred_;
    crec = crec_;
    okf = &okf_;
    if (blockIdx.x < kRideCand) {
      ride_leader<XCH>(((int)blockIdx.x >> 3) & 1 ? s1 : s0, (int)blockIdx.x, ra, fred, fst);
      return;
    }
  }
  const int bid = LEAD ? (int)blockIdx.x - kRideCand : (int)blockIdx.x;
  // (XCD-contiguous eighths: essential here -- with the identity map the product takes 43 us instead of 29 us)
  const int g = (bid & 7) * grp_per_xcd + (bid >> 3);
  if constexpr (LEAD) {
    if (bid >= 8 * grp_per_xcd) {  // a riding-update workgroup
      if (ride_settle<true>(ra, fst, okf))
        run_fused_updates<NL>(u0, u1, 8 * grp_per_xcd + kRideCand, prod, reinterpret_cast<const LaneCtl*>(fst),
                              reinterpret_cast<const LaneCtl*>(fst + 80));
      return;
    }
  } else {
    if (run_fused_updates<NL>(u0, u1, 8 * grp_per_xcd, prod)) return;
  }
  rgcs_group<NL, PAD, LEAD, false>(M, x, yin, yout, ctl0, ctl1, partials, pstride, g, ra, FuseArgs{}, prod, crec, okf);
}

// ------------------------------------------------------------------------------------------------ one iteration, one launch
//
// k_iter_fused: the A' product and the A product of a joint iteration in ONE grid.  Two launches per iteration end twice with a
// resident set of workgroups draining while nothing new may start, and begin twice with ~3 us until the first streams have
// landed (profiles/r03_at_phase_probe.txt).  The dependence across the first of the two boundaries is local: a row group of A
// gathers the long pair on its column window only -- the output of ~40 row blocks of A' -- and needs the scalars behind the
// A' product (the global norm) only in its row epilogue.  Here the row groups follow the A' blocks in the same grid and start as
// the blocks drain:
//
//   [ 16 head leaders | A' product workgroups | 16 mid leaders | row groups of A | updates riding with A' | updates riding with A ]
//
//   * head leaders: the steps behind the PREVIOUS A product, as in k_spmv_atl (record `ra`), but without side effects;
//   * A' workgroups: atl_product<.., FUSED>: rows written through, then flag + tagged partials per block;
//   * mid leaders (leader c: lane (c >> 3) & 1, one per lane on every XCD like the head leaders): the head step AGAIN (same
//     inputs, same bits -- so the state it leaves never travels between workgroups inside the launch), then the step behind the
//     A' product from that state in LDS and the blocks' tagged partials (bounded looks until every word is this launch's), published in the second record `rb`; leaders 0 and 8 commit it to a THIRD copy of the state (a head
//     leader held up by another kernel may still be reading the first);
//   * row groups: rgcs_group<.., FUSED>: wait for the flags of the blocks owning their lines, gather at agent scope, coefficients
//     from `rb`; their updates likewise.
// No fence anywhere (an agent-scope release or acquire costs an L2 write-back / invalidate per workgroup: 5-7 x slower,
// profiles/r04_coherence_whatif.txt).  What it rests on instead: (1) a store with the agent-scope bit is written through, and
// vmcnt counts it only when it is at its coherence point; (2) nobody reads a line of the long pair before the block that owns it
// -- blocks start on 128-byte boundaries (make_rowblocks, align = 8 rows) and a group waits for every block owning a line it
// touches -- is complete, so no XCD's L2 can hold an earlier version; (3) self-validating words wherever a value travels
// without an ordering guarantee (records, partials).  Every wait is bounded and ends in the handle's error word.
// Dependences point to workgroups earlier in the grid.  Results: BITWISE the two-launch iteration (same per-block and per-group
// arithmetic, the partials summed in the same order by the same step code).

// squared-norm partials of lane l of the A' blocks [0, n) from their tagged words, summed like reduce_two sums a plain array
// (same association per thread, same tree); false: a word did not carry this launch's number after kRidePolls looks
__device__ __forceinline__ bool reduce_tagged(const unsigned long long* ptag, int n, int l, unsigned int want, double* red,
                                              int* flag, double& s0, int wpb = 4 /* words per block */, int more = 0) {
  const int t = threadIdx.x;
  auto part = [&](int i, bool& good) {
    const unsigned long long hi = ride_load(ptag + (size_t)i * wpb + 2 * l), lo = ride_load(ptag + (size_t)i * wpb + 2 * l + 1);
    good = good && (unsigned int)(hi & 0xffffffffull) == want && (unsigned int)(lo & 0xffffffffull) == want;
    return __longlong_as_double((long long)((hi & 0xffffffff00000000ull) | (lo >> 32)));
  };
  double a = 0.0;
  for (int look = 0; look < kRidePolls + more; ++look) {
    bool good = true;
    a = 0.0;
    if (n <= kStepThreads * 24) {  // reduce_two's single batch: a thread adds its entries t, t + 256, ... in that order
      for (int base = 0; base < n; base += kStepThreads * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * kStepThreads + t;
          const double w = part(i < n ? i : n - 1, good);
          v[u] = i < n ? w : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
      }
    } else {  // ... its batches of eight, each summed from zero
      for (int base = 0; base < n; base += kStepThreads * 8) {
        double v[8], b = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + u * kStepThreads + t;
          const double w = part(i < n ? i : n - 1, good);
          v[u] = i < n ? w : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) b += v[u];
        a += b;
      }
    }
    if (t == 0) *flag = 0;
    __syncthreads();
    if (!good) *flag = 1;
    __syncthreads();
    const bool again = *flag != 0;
    __syncthreads();
    if (!again) {
      double dummy = 0.0, s1 = 0.0;
      block_reduce_two(a, dummy, red, s0, s1);
      return true;
    }
    __builtin_amdgcn_s_sleep(16);
  }
  return false;
}

// mid leader c (see above)
template <bool XCH = false>
__device__ __forceinline__ void fuse_mid_leader(const StepArgs& sh, const StepArgs& sm, int c, const RideArgs& ra, const RideArgs& rb,
                                                const FuseArgs& fz, double* red32, unsigned long long* st80, int* flag) {
  const int l = (c >> 3) & 1;
  const bool commit = (c & 7) == 0;
  if (rb.delay != 0 && c == rb.delay - 1) {  // (tests: this mid leader starts ~100 us late, as behind another kernel's workgroups)
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 10000ull) __builtin_amdgcn_s_sleep(32);
  }
  // 1. the state the head step leaves, recomputed; leaders 0 and 8 perform its side effects (progress word, final statistics,
  // the second copy of the state) -- here, ahead of the next step's, and not in the head leaders
  if (sh.kind != STEP_NONE) {
    step_run<NoStepHook, XCH>(sh, red32, st80, commit, nullptr, ra.xt, ra.xseq, l);  // (the head step's exchange again: its rows are still in the ring)
  } else {
    const int nq = state_bytes(sm.kind) / 8;
    if ((int)threadIdx.x < nq) st80[threadIdx.x] = reinterpret_cast<const unsigned long long*>(sh.state)[threadIdx.x];
    __syncthreads();
  }
  // 2. + 3. the step behind the A' product, from the state in LDS and the blocks' tagged partials: looking at them until every
  // word carries this launch's number IS the wait for the blocks (a shared counter of completed blocks was built first: five
  // thousand agent-scope atomics on one address took the launch from ~60 to 106 us)
  unsigned long long* rec = rb.rec + 64 * ride_xcc();
  const bool skip = reinterpret_cast<const LaneCtl*>(st80)->done != 0;  // (uniform: LDS)
  double s0 = 0.0;
  bool fine = true;
  if (!skip) fine = reduce_tagged(fz.ptag, sm.n0, l, fz.want, red32, flag, s0, 4, XCH ? fz.more : 0);
  fuse_stamp(fz, 2);
  if (!fine) {
    if (threadIdx.x == 0) __hip_atomic_store(fz.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;  // (nothing published: whoever waits for this record runs into its own bound)
  }
  if constexpr (XCH) {
    if (!skip && rb.xt != nullptr) {  // (row-sharded: the sum over the ranks, formed here -- fpsq_krylov.hip.h xch_sum; uniform)
      double v[2] = {s0, 0.0};
      xch_sum<2>(rb.xt, rb.xseq, l, v, red32);
      s0 = v[0];
    }
  }
  if (threadIdx.x == 0) {
    if (!skip) step_advance(sm, st80, s0, 0.0, commit ? sm.prog : nullptr);
    ride_publish(reinterpret_cast<const LaneCtl*>(st80 + sm.prod_ctl_off), l, rec, rb.pub);
    if (commit && !skip) step_final_stats(sm, st80);
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const unsigned long long w = st80[threadIdx.x];
    ride_store(rec + 16 + 24 * l + 2 * threadIdx.x, (w & 0xffffffff00000000ull) | rb.pub);
    ride_store(rec + 16 + 24 * l + 2 * threadIdx.x + 1, (w << 32) | rb.pub);
  }
  if (commit) {
    const int nq = state_bytes(sm.kind) / 8;
    unsigned long long* gdst = reinterpret_cast<unsigned long long*>(sm.state_out);
    if ((int)threadIdx.x < nq) gdst[threadIdx.x] = st80[threadIdx.x];
  }
}

// ---- the halo of a row-sharded handle INSIDE the one-launch iteration (round 5).  Behind the A' workgroups the grid carries
//   2 kHaloCopy push workgroups: wait (bounded) for the flags of the A' blocks that deposit the raw sums of my head / tail region,
//     then write their slice into the neighbour's slot; the last slice to arrive raises the neighbour's flag word (as
//     k_p2p_halo_finish does between two product launches);
//   gf finish workgroups: wait for those blocks AND (bounded, unconditionally: it paces the ranks) for the neighbours' records,
//     complete the overlap rows -- yout = ca (own + neighbour's) + cb yin, the coefficients from the head leaders' record,
//     written through -- and publish themselves like A' blocks: flag + tagged squared-norm partials at index nblk + b.
// Row groups whose lines reach into an overlap region wait for the finish workgroups too (FuseArgs::dep2); the mid leaders sum
// nblk + gf tagged partials -- the order of the two-launch form's array.  The regions start and end on 128-byte lines of the
// long pair (8 rows: distributed.halo_plan rounds the windows), so a line has one owner here as well.
struct FuseHalo {
  P2PHalo H;
  unsigned long long seq;
  int* fail;
  long max_spins;
  unsigned long long* arrive;  // [2]: arrival counters of the push slices, per side
  const double* raw;           // [(tl + tr)][2]: my raw sums on the two regions (head first)
  const double* recv;          // the neighbours' (this exchange's half of the slots)
  int64_t tl, tr, tail0;
  int32_t gf, nwg;             // finish workgroups; halo workgroups in the grid (2 kHaloCopy + gf, padded to a multiple of 8)
  int2 depL, depR;             // the A' blocks holding rows of the head / tail region
};
__device__ __forceinline__ double ld_ag(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
// wave 0 of the workgroup: every A' block of [d.x, d.y] has published this launch's number (bounded); the verdict in *okf
__device__ __forceinline__ bool fuse_wait_blocks(const FuseArgs& fz, int dx, int dy, int ex, int ey, int* okf) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    bool all = false;
    for (int t = 0; t < kRidePolls + fz.more && !all; ++t) {
      if (t) __builtin_amdgcn_s_sleep(8);
      bool ok = true;
      for (int L = dx + tid; L <= dy; L += 64)
        ok &= __hip_atomic_load(fz.blkflag + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fz.want;
      for (int L = ex + tid; L <= ey; L += 64)
        ok &= __hip_atomic_load(fz.blkflag + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fz.want;
      all = __all(ok);
    }
    if (tid == 0) {
      *okf = all ? 1 : 0;
      if (!all) __hip_atomic_store(fz.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  return *okf != 0;
}
__device__ __forceinline__ void fuse_halo_wg(int b, const FuseHalo& fh, const FuseArgs& fz, const RideArgs& ra, int nblk, double* lp,
                                             double* red, unsigned long long* crec, int* okf) {
  constexpr int NL = 2;
  const int tid = threadIdx.x;
  if (__hip_atomic_load(fh.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;  // (an earlier exchange of the call gave up)
  if (b < 2 * kHaloCopy) {  // ---- push
    const bool left = b < kHaloCopy;
    const int sl = left ? b : b - kHaloCopy;
    double* dst = left ? fh.H.left_dst : fh.H.right_dst;
    if (!dst) return;
    if (!fuse_wait_blocks(fz, left ? fh.depL.x : fh.depR.x, left ? fh.depL.y : fh.depR.y, 1, 0, okf)) return;
    const int64_t nl = fh.tl * NL, cnt = left ? nl : fh.tr * NL;
    const double* src = left ? fh.raw : fh.raw + nl;
    const int64_t per = ((cnt + kHaloCopy - 1) / kHaloCopy + 1) & ~(int64_t)1, lo = sl * per, hi = lo + per < cnt ? lo + per : cnt;
    for (int64_t base = lo; base < hi; base += 8 * kBlock) {  // (p2p_copy with agent-scope loads: the sums were written through by other XCDs)
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = base + u * kBlock + tid;
        v[u] = ld_ag(src + (i < hi ? i : hi - 1));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = base + u * kBlock + tid;
        if (i < hi) dst[i] = v[u];
      }
    }
    barrier_stores_done();
    if (tid == 0) {
      __threadfence_system();
      const unsigned long long got = __hip_atomic_fetch_add(fh.arrive + (left ? 0 : 1), 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
      if (got % kHaloCopy == 0)
        __hip_atomic_store(left ? fh.H.left_flag : fh.H.right_flag, fh.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  b -= 2 * kHaloCopy;
  if (b >= fh.gf) return;  // (padding)
  // ---- finish.  Both waits come first and are unconditional (see k_p2p_halo_finish: the wait for the neighbours paces the ranks)
  bool fine = fuse_wait_blocks(fz, fh.tl > 0 ? fh.depL.x : 1, fh.tl > 0 ? fh.depL.y : 0, fh.tr > 0 ? fh.depR.x : 1, fh.tr > 0 ? fh.depR.y : 0, okf);
  if (fine) {
    if (tid == 0) {
      bool in = true;
      if (fh.H.my_from_left) in = p2p_wait(fh.H.my_from_left, fh.seq, fh.max_spins, fh.fail);
      if (in && fh.H.my_from_right) in = p2p_wait(fh.H.my_from_right, fh.seq, fh.max_spins, fh.fail);
      *okf = in ? 1 : 0;
    }
    __syncthreads();
    fine = *okf != 0;
  }
  if (!fine) return;  // (nothing published: whoever waits for this workgroup runs into its own bound; the call fails)
  __syncthreads();
  if (!ride_settle<false>(ra, crec, okf)) return;
  RideCoef C;
  ride_decode(crec, C);
  double sq[NL] = {0.0, 0.0}, sq_tail[NL] = {0.0, 0.0};
  for (int64_t i = (int64_t)b * kBlock + tid; i < fh.tl + fh.tr; i += (int64_t)fh.gf * kBlock) {
    const int64_t row = i < fh.tl ? i : fh.tail0 + (i - fh.tl);
    double acc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l)
      acc[l] = ld_ag(fh.raw + i * NL + l) +
               __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(fh.recv + i * NL + l),
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    row_epilogue<NL, true>((size_t)row, acc, C.ca, C.cb, C.act, lp, lp, i < fh.tl ? sq : sq_tail);
  }
#pragma unroll
  for (int l = 0; l < NL; ++l) sq[l] = wave_sum(sq[l]);
  __syncthreads();  // (`red` may alias what ride_settle used)
  if ((tid & 63) == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) red[(tid >> 6) * NL + l] = sq[l];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (every row is at its coherence point)
  if (tid == 0) {
    unsigned long long* pt = fz.ptag + (size_t)(nblk + b) * 4;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const double v = ((red[l] + red[NL + l]) + red[2 * NL + l]) + red[3 * NL + l];  // (block_sum's order: k_p2p_halo_finish's bits)
      ride_store(pt + 2 * l, tag_hi(v, fz.pub));
      ride_store(pt + 2 * l + 1, tag_lo(v, fz.pub));
    }
    __hip_atomic_store(fz.blkflag + nblk + b, fz.pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

struct FuseGrid {
  int32_t nwg_t, n2, bpx;  // A' product workgroups (a multiple of 8), of which the first n2 take two blocks; blocks per XCD
  int32_t nupd_t;          // update workgroups riding with A' (padded to a multiple of 8)
  int32_t gpx;             // row groups per XCD
  int32_t rot;             // atl_blocks_of's rotation (0; tests: 1..7)
};

template <bool CSORT, bool HALO = false>
__global__ __launch_bounds__(kBlock) void k_iter_fused(CsrView AT, RgcsView RA, const double* sp_in, double* lp, double* sp_out,
                                                       double* part_a, int pstride_a, const FuseGrid fg, const UpdSeg ut0,
                                                       const UpdSeg ut1, const UpdSeg ua0, const UpdSeg ua1, const StepArgs sh0,
                                                       const StepArgs sh1, const StepArgs sm0, const StepArgs sm1,
                                                       const RideArgs ra, const RideArgs rb, const FuseArgs fz, const HaloRows hr,
                                                       const FuseHalo fh) {
  constexpr int NL = 2;
  __shared__ double prod[kSpmvNnz * NL];
  static_assert(kSpmvNnz == kRgcsTile, "one product buffer serves both products");
  __shared__ __attribute__((aligned(16))) unsigned long long fst[2 * 80];
  __shared__ double fred[32];
  __shared__ unsigned long long crec[10];
  __shared__ int okf;
  int b = (int)blockIdx.x;
  fuse_stamp(fz, 0);
  if (b < kRideCand) {
    ride_leader<HALO>((b >> 3) & 1 ? sh1 : sh0, b, ra, fred, fst, /*may_commit=*/false);  // (HALO = a sharded handle: may exchange)
    fuse_stamp(fz, 3);
    return;
  }
  b -= kRideCand;
  if (b < fg.nwg_t) {
    int Lt[2], nt;
    if (!atl_blocks_of(b, fg.n2, fg.bpx, AT.nblk, Lt, nt, fg.rot)) return;
    atl_product<CSORT, HALO, true>(AT, sp_in, lp, lp, nullptr, 0, Lt, nt, ra, hr, fz, prod, crec, &okf);
    fuse_stamp(fz, 3);
    return;
  }
  b -= fg.nwg_t;
  if constexpr (HALO) {  // (a multiple of 8 workgroups: the row groups behind keep their XCDs)
    if (b < fh.nwg) {
      fuse_halo_wg(b, fh, fz, ra, AT.nblk, lp, prod, crec, &okf);
      fuse_stamp(fz, 3);
      return;
    }
    b -= fh.nwg;
  }
  if (b < kRideCand) {
    fuse_mid_leader<HALO>((b >> 3) & 1 ? sh1 : sh0, (b >> 3) & 1 ? sm1 : sm0, b, ra, rb, fz, fred, fst, &okf);
    fuse_stamp(fz, 3);
    return;
  }
  b -= kRideCand;
  if (b < 8 * fg.gpx) {
    const int g = (b & 7) * fg.gpx + (b >> 3);
    rgcs_group<NL, true, true, true>(RA, lp, sp_in, sp_out, nullptr, nullptr, part_a, pstride_a, g, rb, fz, prod, crec, &okf);
    fuse_stamp(fz, 3);
    return;
  }
  b -= 8 * fg.gpx;
  if (b < fg.nupd_t) {  // (behind the row groups: nothing in this launch waits for them, and the row groups should enter as the A' blocks drain)
    if (b >= ut0.nblk + ut1.nblk) return;  // (padding)
    if (ride_settle<true>(ra, fst, &okf)) {
      fuse_stamp(fz, 1);
      const UpdSeg& u = b < ut0.nblk ? ut0 : ut1;
      upd_run<NL>(u, b < ut0.nblk ? b : b - ut0.nblk, prod, reinterpret_cast<const LaneCtl*>(u.lane == 0 ? fst : fst + 80));
    }
    fuse_stamp(fz, 3);
    return;
  }
  b -= fg.nupd_t;
  if (b >= ua0.nblk + ua1.nblk) return;
  if (ride_settle<true, 48>(rb, fst, &okf)) {  // (released late in the launch: look rarely, the record's lines are busy)
    fuse_stamp(fz, 1);
    const UpdSeg& u = b < ua0.nblk ? ua0 : ua1;
    upd_run<NL>(u, b < ua0.nblk ? b : b - ua0.nblk, prod, reinterpret_cast<const LaneCtl*>(u.lane == 0 ? fst : fst + 80));
  }
  fuse_stamp(fz, 3);
}

// ------------------------------------------------------------------------------------------------ MINRES: E1 -> step A -> E2, one launch
//
// A MINRES lane (solve_two_extras: hprod! Val(1)) needs two global sums per iteration beyond the products': alpha = <r2, y0> / beta
// between its element-wise stages E1 and E2, beta_new = ||y|| behind E2.  E1, the scalar step A and E2 were three launches
// (~5 + 6 + 5 us on m-vectors); here they are one: workgroup 0 is the leader, workgroups 1 .. nblk do stage E1 on their elements,
// publish their partial of <r2, y0> as self-validating words, wait for the leader's record (it sums the partials in
// reduce_two's order, runs minres_a_step, publishes the coefficients of E2) and go on with stage E2 on the SAME elements --
// what a thread wrote in E1 it reads itself in E2.  Every workgroup of the grid must be resident at once (the waiting ones
// hold their slots): <= 1025 workgroups of 256 threads without LDS to speak of -- the host checks.  Bounded waits as everywhere.
// Bitwise the three launches.
__global__ __launch_bounds__(kBlock) void k_minres_mid(const UpdSeg e1, const UpdSeg e2, const StepArgs sa, unsigned long long* ptag,
                                                       unsigned long long* rec /* 10 words */, unsigned int want,
                                                       unsigned long long* err) {
  constexpr int NL = 2;
  __shared__ __attribute__((aligned(16))) unsigned long long st[80];
  __shared__ double red[32];
  __shared__ unsigned long long crec[10];
  __shared__ int flag;
  const LaneCtl* gctl = reinterpret_cast<const LaneCtl*>(sa.state);
  const bool done = gctl->done != 0;  // (uniform; the state was committed before this launch)
  if (blockIdx.x == 0) {  // the leader
    const int nq = state_bytes(sa.kind) / 8;
    if ((int)threadIdx.x < nq) st[threadIdx.x] = reinterpret_cast<const unsigned long long*>(sa.state)[threadIdx.x];
    __syncthreads();
    double s0 = 0.0;
    bool fine = true;
    if (!done) fine = reduce_tagged(ptag, sa.n0, 0, want, red, &flag, s0, 2);
    if (!fine) {
      if (threadIdx.x == 0) __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    if (threadIdx.x == 0) {
      if (!done) step_advance(sa, st, s0, 0.0, sa.prog);
      const LaneCtl* c = reinterpret_cast<const LaneCtl*>(st);
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // e[1..4]: the coefficients of E2
        ride_store(rec + 2 * k, tag_hi(c->e[1 + k], want));
        ride_store(rec + 2 * k + 1, tag_lo(c->e[1 + k], want));
      }
      ride_store(rec + 8, ((unsigned long long)(c->done != 0 ? 1u : 0u) << 32) | want);
    }
    __syncthreads();
    if (!done) {
      unsigned long long* gdst = reinterpret_cast<unsigned long long*>(sa.state);
      if ((int)threadIdx.x < nq) gdst[threadIdx.x] = st[threadIdx.x];
    }
    return;
  }
  if (done) return;  // (stages E1 and E2 of a finished lane do nothing)
  const int blk = (int)blockIdx.x - 1;
  double* sp = const_cast<double*>(e1.src);
  {  // stage E1 (upd_minres<1>): y0 = q - e0 r1, partial <r2, y0>
    const double e0 = gctl->e[0];
    double acc = 0.0;
    for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < e1.len; i += (int64_t)e1.nblk * kBlock) {
      const double y0 = sp[i * NL + e1.lane] - (e0 != 0.0 ? e0 * e1.a[i] : 0.0);
      sp[i * NL + e1.lane] = y0;
      acc += e1.b[i] * y0;
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) {
      ride_store(ptag + 2 * (size_t)blk, tag_hi(t, want));
      ride_store(ptag + 2 * (size_t)blk + 1, tag_lo(t, want));
      flag = 0;
    }
  }
  __syncthreads();
  // the leader's record
  for (int t = 0; t < kRidePolls; ++t) {
    if (threadIdx.x < 64) {
      if (t) __builtin_amdgcn_s_sleep(4);
      const int lane = threadIdx.x & 63;
      const unsigned long long w = lane < 9 ? ride_load(rec + lane) : 0ull;
      const bool good = lane >= 9 || (unsigned int)(w & 0xffffffffull) == want;
      if (__all(good)) {
        if (lane < 9) crec[lane] = w;
        if (lane == 0) flag = 1;
      }
    }
    __syncthreads();
    if (flag) break;
  }
  if (!flag) {
    if (threadIdx.x == 0) __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  if ((crec[8] >> 32) != 0) return;  // (the step ended the recurrence: no E2)
  double e[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = __longlong_as_double((long long)((crec[2 * k] & 0xffffffff00000000ull) | (crec[2 * k + 1] >> 32)));
  {  // stage E2 (upd_minres<2>): y = y0 - e1 r2; r_new = y; w~ = e2 r2 - e3 w2 - e4 w1; partial ||y||^2
    double acc = 0.0;
    for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < e2.len; i += (int64_t)e2.nblk * kBlock) {
      const double r = e2.a[i];
      const double y = sp[i * NL + e2.lane] - e[0] * r;
      sp[i * NL + e2.lane] = y;
      e2.b[i] = y;
      e2.d[i] = e[1] * r - e[2] * e2.c[i] - e[3] * e2.d[i];
      acc += y * y;
    }
    __syncthreads();  // (`red` of stage E1)
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) e2.partials[blk] = t;
  }
}

}  // namespace fpsq

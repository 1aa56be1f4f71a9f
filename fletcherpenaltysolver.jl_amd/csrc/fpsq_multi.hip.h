// fpsq_multi.hip.h -- SEVERAL joint Krylov iterations in ONE launch (k_iter_multi; round 5).
//
// k_iter_fused put the A' -> A boundary of an iteration inside a launch; the A -> A' boundary BETWEEN two iterations stayed a kernel
// boundary: the last row groups drain at half occupancy while nothing new may start, and the next launch's first streams need
// ~3 us to land -- 5-6 us of ~60 per iteration at the headline size (measured with tools/fuse_probe.py).  Here the
// grid is K iteration grids one behind the other,
//
//   [ head leaders | A' blocks | mid leaders | row groups | LSQR updates | CRAIG updates ]  x K
//
// and the A' blocks of iteration j + 1 start as the row groups of iteration j drain.  What that takes beyond the one-launch iteration
// (every dependence still points to a workgroup EARLIER in the grid; every wait is bounded and ends in the handle's error word):
//  * the REVERSE dependence -- an A' block of iteration j + 1 overwrites rows of the long pair that row groups and CRAIG's long
//    update of iteration j still read -- is removed by a second long pair: iteration j reads its `yin` rows from one buffer and
//    writes the other (the short pair has alternated since round 3).  A buffer comes round again two iterations later, behind
//    leader sets that have seen every reader of it finish (below);
//  * the row groups write their rows through and publish a flag and tagged partials per group (rgcs_group<.., MULTI>); an A' block
//    waits for the groups whose rows it gathers (MultiCtx::bdep) and gathers at agent scope (atl_product<.., MULTI>);
//  * the recurrence state travels from leader set to leader set as self-validating words (state records: one copy per XCC and lane,
//    written by that XCC's leader of the lane, read by the same XCC's leader of the next set), the coefficient records live in a
//    ring of four -- a late update workgroup of iteration j may still be looking at its record while iteration j + 1's is written;
//  * the riding updates publish their squared-norm partial (or a void word) as tagged words when they are done: the next head
//    leaders sum the partials from them, and the next mid leaders have seen EVERY update workgroup of the previous iteration finish
//    before they release the workgroups that rewrite what those read or write (the old short pair, x / w in place);
//  * block flags, tagged partials and update tags alternate between two copies by the parity of the iteration number: a late mid
//    leader of iteration j is still reading iteration j's words while iteration j + 1's blocks publish theirs (iteration j + 2 cannot
//    get that far: its blocks wait for row groups that wait for the mid leaders of j + 1, which wait for every block of j + 1, ...);
//  * operands that cross an iteration inside the launch are read at agent scope wherever the writer may sit on another XCD (the
//    gathers, the updates' `src`); what a workgroup re-reads from the workgroup of the SAME index of the previous iteration (its own
//    `yin` rows, x / w of an update) stays on one XCD -- every role group starts on a multiple of 8 workgroups;
//  * once every recurrence has ended, a workgroup that finds that out AT ITS ENTRY publishes itself with a void payload and leaves
//    (MultiCtx::hdone): iterations enqueued past convergence cost their dispatch, not their streams.
// Arithmetic: per block, per group, per update workgroup and per step exactly the one-launch iteration's, the partials summed in the
// same order -- results BITWISE those of K one-launch iterations (tests/test_gpu_parity.py).
#pragma once
#include "fpsq_spmv.hip.h"

namespace fpsq {

constexpr int kMultiMax = 8;      // iterations per launch at most
constexpr int kRecRing = 4;       // coefficient / state record slots: iteration number & 3
constexpr int kStateWords = 80;   // a recurrence state is at most 640 bytes (static_asserts in fpsq_krylov.hip.h)
constexpr int kSrecLane = 2 * kStateWords;   // tagged words of one lane's state
constexpr int kSrecXcc = 2 * kSrecLane;      // ... of an XCC's copy (two lanes)
constexpr int kSrecSlot = 8 * kSrecXcc;      // ... of a slot (eight XCCs); slot = (seq & 3) * 2 + (mid ? 1 : 0)

struct MultiArgs {
  int32_t K, per_iter;         // iterations in this launch; workgroups per iteration (a multiple of 8)
  int32_t it0;                 // iteration number of j = 0
  uint32_t seq0;               // launch number of iteration 0 (iteration j: seq0 + j)
  FuseGrid fg;                 // one iteration's layout: A' workgroups, LSQR-update workgroups (padded), row groups per XCD
  int32_t nupd_a;              // workgroups of CRAIG's SHORT update (whose partials the next head leaders wait for), padded to a multiple of 8
  int32_t nlong;               // ... of its LONG update, padded: they run ONE ITERATION LATER, behind the next iteration's A' blocks (and behind the
                               // last iteration's grid), so that only the small m-vector updates stand between an iteration's row groups and
                               // the next A' blocks in the dispatch order; 0: CRAIG's long update rides with the short one
  int32_t sp0, lp0;            // which of the two buffers iteration 0 READS (short pair: gathers, yin of the row groups; long pair: yin of the blocks)
  int32_t pstride_a;
  double* sp[2];
  double* lp[2];
  double* part_last;           // plain A partials of the LAST iteration (what the next launch's leaders, or a stand-alone step, read)
  StepArgs sh[2], sm[2];       // the head and mid steps of iteration 0 per lane (kind, it, prog, host_stats; sh: state, p0 / p1 plain arrays)
  void* commit[2][2];          // [lane][j & 1]: where the committing mid leader of iteration j leaves the state
  int32_t p1seg[2];            // per lane: the update segment whose partials the head step reads as its second array (0, 1: ut; 3: ua[1])
  int32_t n1;                  // ... and their count
  UpdSeg ut[2], ua[2];         // iteration 0's update segments (it, src, partials are re-derived per iteration)
  double* pw[2][2];            // [lane][parity]: the plain update-partial arrays (KrylovRun::upd_part)
  unsigned long long *rec_h, *rec_m;  // coefficient records: kRecRing x 8 XCC x 64 words each
  unsigned long long* srec;    // state records: kRecRing x 2 x kSrecSlot words
  unsigned int* flag[2];       // A' blocks: [parity][block]
  unsigned long long* ptag[2];
  unsigned int* gflag[2];      // row groups: [parity][group]
  unsigned long long* atag[2];
  unsigned long long* utag[2]; // update workgroups: [parity][4 segments][kEwBlocksMax][2 words]
  const int2* dep;             // per row group: A' blocks it waits for
  const int2* bdep;            // per A' block: row groups it waits for
  unsigned long long* hdone;   // two 32-bit halves, one per lane: != 0 once that recurrence has ended
  unsigned long long* err;
  int32_t delay_h, delay_m;    // tests: leader c + 1 of every head / mid set starts ~100 us late
  uint32_t break_pub;          // tests: XOR-ed into what the A' blocks publish (0 in production)
};

// ---- state records
// every thread of the workgroup calls; st <- the nq state words of `rec` once each carries `want` (bounded looks)
__device__ __forceinline__ bool srec_take(const unsigned long long* rec, int nq, unsigned int want, unsigned long long* st, int* flag) {
  const int t = threadIdx.x;
  for (int look = 0; look < kRidePolls; ++look) {
    bool good = true;
    unsigned long long w = 0;
    if (t < nq) {
      const unsigned long long hi = ride_load(rec + 2 * t), lo = ride_load(rec + 2 * t + 1);
      good = (unsigned int)(hi & 0xffffffffull) == want && (unsigned int)(lo & 0xffffffffull) == want;
      w = (hi & 0xffffffff00000000ull) | (lo >> 32);
    }
    if (t == 0) *flag = 0;
    __syncthreads();
    if (!good) *flag = 1;
    __syncthreads();
    const bool again = *flag != 0;
    __syncthreads();
    if (!again) {
      if (t < nq) st[t] = w;
      __syncthreads();
      return true;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  return false;
}
__device__ __forceinline__ void srec_put(unsigned long long* rec, const unsigned long long* st, int nq, unsigned int want) {
  const int t = threadIdx.x;
  if (t < nq) {
    const unsigned long long w = st[t];
    ride_store(rec + 2 * t, (w & 0xffffffff00000000ull) | want);
    ride_store(rec + 2 * t + 1, (w << 32) | want);
  }
}
__device__ __forceinline__ unsigned long long* srec_at(const MultiArgs& M, unsigned int seq, int mid, int xcc, int l) {
  return M.srec + (size_t)((seq & (kRecRing - 1)) * 2 + mid) * kSrecSlot + (size_t)xcc * kSrecXcc + (size_t)l * kSrecLane;
}
__device__ __forceinline__ unsigned long long* utag_at(const MultiArgs& M, int parity, int seg, int blk) {
  return M.utag[parity] + ((size_t)seg * kEwBlocksMax + blk) * 2;
}

// a leader's coefficient record from the state in LDS: line 0 (what a product workgroup needs) and the whole control block
// (update workgroups) -- ride_leader's two publications
__device__ __forceinline__ void multi_publish_coef(const unsigned long long* st, int l, unsigned long long* rec, unsigned int want) {
  if (threadIdx.x == 0) ride_publish(reinterpret_cast<const LaneCtl*>(st), l, rec, want);
  if (threadIdx.x < 12) {
    const unsigned long long w = st[threadIdx.x];
    ride_store(rec + 16 + 24 * l + 2 * threadIdx.x, (w & 0xffffffff00000000ull) | want);
    ride_store(rec + 16 + 24 * l + 2 * threadIdx.x + 1, (w << 32) | want);
  }
}
// ... and of a leader that found the call ended at its entry: a finished lane (done = 1, nothing left to apply)
__device__ __forceinline__ void multi_publish_void(int l, unsigned long long* rec, unsigned int want) {
  if (threadIdx.x < 12) {
    LaneCtl c{};
    c.done = 1;
    c.upd_iter = -1;
    const unsigned long long w = reinterpret_cast<const unsigned long long*>(&c)[threadIdx.x];
    ride_store(rec + 16 + 24 * l + 2 * threadIdx.x, (w & 0xffffffff00000000ull) | want);
    ride_store(rec + 16 + 24 * l + 2 * threadIdx.x + 1, (w << 32) | want);
    if (threadIdx.x == 0) ride_publish(&c, l, rec, want);
  }
}

// sums of a tagged array like reduce_two sums a plain one (see reduce_tagged), `single` = reduce_two's one-batch branch
__device__ __forceinline__ bool sum_tagged(const unsigned long long* tag, int n, int wpb, int off, unsigned int want, bool single, double& a) {
  const int t = threadIdx.x;
  auto part = [&](int i, bool& good) {
    const unsigned long long hi = ride_load(tag + (size_t)i * wpb + off), lo = ride_load(tag + (size_t)i * wpb + off + 1);
    good = good && (unsigned int)(hi & 0xffffffffull) == want && (unsigned int)(lo & 0xffffffffull) == want;
    return __longlong_as_double((long long)((hi & 0xffffffff00000000ull) | (lo >> 32)));
  };
  bool good = true;
  a = 0.0;
  for (int base = 0; base < n; base += kStepThreads * 8) {
    double v[8], b = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * kStepThreads + t;
      const double w = part(i < n ? i : n - 1, good);
      v[u] = i < n ? w : 0.0;
    }
    if (single) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) b += v[u];
      a += b;
    }
  }
  return good;
}
// both sums of a head step from tagged words (A partials: four words per group, this lane's at 2 l; update partials: two words per
// workgroup); results in thread 0; bounded
__device__ __forceinline__ bool reduce_two_tagged(const unsigned long long* t0, int n0, int l, const unsigned long long* t1, int n1,
                                                  unsigned int want, double* red, int* flag, double& s0, double& s1) {
  const bool single = n0 > 0 && n0 <= kStepThreads * 24 && n1 <= kStepThreads * 4;
  for (int look = 0; look < kRidePolls; ++look) {
    double a = 0.0, b = 0.0;
    bool good = sum_tagged(t0, n0, 4, 2 * l, want, single, a);
    if (n1 > 0) good = sum_tagged(t1, n1, 2, 0, want, single, b) && good;
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();
    if (!good) *flag = 1;
    __syncthreads();
    const bool again = *flag != 0;
    __syncthreads();
    if (!again) {
      block_reduce_two(a, b, red, s0, s1);
      return true;
    }
    __builtin_amdgcn_s_sleep(16);
  }
  return false;
}

// ---- the two leader sets of iteration j.  Leader c: lane (c >> 3) & 1, one per lane on every XCD (see "who leads", fpsq_spmv.hip.h)
__device__ __forceinline__ void multi_leader_delay(int delay, int c) {
  if (delay != 0 && c == delay - 1) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 10000ull) __builtin_amdgcn_s_sleep(32);
  }
}
__device__ __forceinline__ void multi_fail(const MultiArgs& M) {
  if (threadIdx.x == 0) __hip_atomic_store(M.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void multi_head_leader(const MultiArgs& M, int j, int c, double* red32, unsigned long long* st80, int* flag) {
  const int l = (c >> 3) & 1;
  const bool commit = (c & 7) == 0;
  const unsigned int seq = M.seq0 + (unsigned int)j;
  const int xcc = ride_xcc();
  unsigned long long* rec = M.rec_h + (size_t)(seq & (kRecRing - 1)) * 512 + 64 * xcc;
  multi_leader_delay(M.delay_h, c);
  StepArgs a = M.sh[l];
  a.it += j;
  const int nq = state_bytes(a.kind) / 8;
  double s0 = 0.0, s1 = 0.0;
  bool skip;
  if (j == 0) {  // the state and the partials of the previous LAUNCH: plain memory (step_run's first half)
    const unsigned long long* gsrc = reinterpret_cast<const unsigned long long*>(a.state);
    if ((int)threadIdx.x < nq) st80[threadIdx.x] = gsrc[threadIdx.x];
    skip = lane_done(a);
    if (!skip) reduce_two(a.p0, a.n0, a.p1, a.p1 ? a.n1 : 0, red32, s0, s1);  // (contains the barrier that publishes st80)
    else __syncthreads();
  } else {  // what the mid leader of this XCC left behind iteration j - 1, and that iteration's tagged partials
    if (!srec_take(srec_at(M, seq - 1, 1, xcc, l), nq, seq - 1, st80, flag)) return multi_fail(M);
    skip = reinterpret_cast<const LaneCtl*>(st80)->done != 0;
    if (!skip) {
      const int pq = (M.it0 + j - 1) & 1;  // the previous iteration's parity
      const unsigned long long* t1 = a.p1 ? utag_at(M, pq, M.p1seg[l], 0) : nullptr;
      if (!reduce_two_tagged(M.atag[pq], a.n0, l, t1, a.p1 ? M.n1 : 0, seq - 1, red32, flag, s0, s1)) return multi_fail(M);
    }
  }
  if (threadIdx.x == 0 && !skip) {
    step_advance(a, st80, s0, s1, commit ? a.prog : nullptr);
    if (commit) step_final_stats(a, st80);
  }
  __syncthreads();
  multi_publish_coef(st80, l, rec, seq);
  srec_put(srec_at(M, seq, 0, xcc, l), st80, nq, seq);
}

__device__ __forceinline__ void multi_mid_leader(const MultiArgs& M, int j, int c, int nblk, double* red32, unsigned long long* st80,
                                                 int* flag) {
  const int l = (c >> 3) & 1;
  const bool commit = (c & 7) == 0;
  const unsigned int seq = M.seq0 + (unsigned int)j;
  const int xcc = ride_xcc();
  const int q = (M.it0 + j) & 1;
  unsigned long long* rec = M.rec_m + (size_t)(seq & (kRecRing - 1)) * 512 + 64 * xcc;
  multi_leader_delay(M.delay_m, c);
  StepArgs a = M.sm[l];
  a.it += j;
  const int nq = state_bytes(a.kind) / 8;
  if (!srec_take(srec_at(M, seq, 0, xcc, l), nq, seq, st80, flag)) return multi_fail(M);
  if (j > 0) {
    // every update workgroup of the previous iteration has finished: what this record releases -- the row groups' epilogues (they
    // rewrite the short pair those updates read) and CRAIG's updates (x / w in place) -- comes after them
    const int pq = (M.it0 + j - 1) & 1;
    for (int look = 0;; ++look) {
      bool good = true;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int cnt = s == 0 ? M.ut[0].nblk : s == 1 ? M.ut[1].nblk : s == 2 ? M.ua[0].nblk : M.ua[1].nblk;
        for (int b = threadIdx.x; b < cnt; b += kStepThreads)
          good = good && (unsigned int)(ride_load(utag_at(M, pq, s, b)) & 0xffffffffull) == seq - 1;
      }
      if (threadIdx.x == 0) *flag = 0;
      __syncthreads();
      if (!good) *flag = 1;
      __syncthreads();
      const bool again = *flag != 0;
      __syncthreads();
      if (!again) break;
      if (look >= kRidePolls) return multi_fail(M);
      __builtin_amdgcn_s_sleep(8);
    }
  }
  const bool skip = reinterpret_cast<const LaneCtl*>(st80)->done != 0;
  double s0 = 0.0;
  if (!skip && !reduce_tagged(M.ptag[q], a.n0, l, seq, red32, flag, s0)) return multi_fail(M);
  if (threadIdx.x == 0 && !skip) {
    step_advance(a, st80, s0, 0.0, commit ? a.prog : nullptr);
    if (commit) step_final_stats(a, st80);
  }
  __syncthreads();
  multi_publish_coef(st80, l, rec, seq);
  srec_put(srec_at(M, seq, 1, xcc, l), st80, nq, seq);
  if (commit) {
    unsigned long long* gdst = reinterpret_cast<unsigned long long*>(M.commit[l][j & 1]);
    if ((int)threadIdx.x < nq) gdst[threadIdx.x] = st80[threadIdx.x];
    if (threadIdx.x == 0 && reinterpret_cast<const LaneCtl*>(st80)->done != 0)
      __hip_atomic_store(reinterpret_cast<unsigned int*>(M.hdone) + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---- a riding update workgroup of iteration j: segment s (0, 1: LSQR updates, released by the head record; 2, 3: CRAIG long /
// short, released by the mid record), workgroup blk of it
__device__ __forceinline__ void multi_update_wg(const MultiArgs& M, int j, int s, int blk, bool over, double* prod, unsigned long long* fst,
                                                int* okf) {
  const unsigned int seq = M.seq0 + (unsigned int)j;
  const int it = M.it0 + j, q = it & 1;
  unsigned long long* tg = utag_at(M, q, s, blk);
  if (over) {  // (the call has ended: published, nothing applied)
    if (threadIdx.x == 0) {
      ride_store(tg, tag_hi(0.0, seq));
      ride_store(tg + 1, tag_lo(0.0, seq));
    }
    return;
  }
  UpdSeg u;  // (static selects: a dynamic index into the kernel arguments would put the segment on the stack)
  if (s == 0) u = M.ut[0];
  else if (s == 1) u = M.ut[1];
  else if (s == 2) u = M.ua[0];
  else u = M.ua[1];
  RideArgs ra{};
  ra.rec = (s < 2 ? M.rec_h : M.rec_m) + (size_t)(seq & (kRecRing - 1)) * 512;
  ra.want = seq;
  ra.err = M.err;
  if (!(s < 2 ? ride_settle<true>(ra, fst, okf) : ride_settle<true, 48>(ra, fst, okf))) return;
  if (s < 2) {  // LSQR's update of iteration it - 1 (KrylovRun::lsqr_upd_seg)
    u.it = it - 1;
    u.src = M.sp[(M.sp0 + j) & 1];
    u.partials = M.pw[u.lane][(it - 1) & 1];
  } else if (s == 2) {  // CRAIG long: reads the long pair this iteration's A' product has written
    u.it = it;
    u.src = M.lp[((M.lp0 + j) & 1) ^ 1];
  } else {  // CRAIG short: reads the OLD short pair
    u.it = it;
    u.src = M.sp[(M.sp0 + j) & 1];
    u.partials = M.pw[u.lane][(it - 1) & 1];
  }
  if (threadIdx.x == 0 && u.partials != nullptr) u.partials[blk] = 0.0;  // (a finished lane's update returns without writing it)
  upd_run_ag<2>(u, blk, prod, reinterpret_cast<const LaneCtl*>(u.lane == 0 ? fst : fst + 80));
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (every store of the workgroup has reached the L2 of its XCD)
  if (threadIdx.x == 0) {
    const double v = u.partials != nullptr ? u.partials[blk] : 0.0;
    ride_store(tg, tag_hi(v, seq));
    ride_store(tg + 1, tag_lo(v, seq));
  }
}

template <bool CSORT>
__global__ __launch_bounds__(kBlock) void k_iter_multi(CsrView AT, RgcsView RA, const MultiArgs M) {
  constexpr int NL = 2;
  __shared__ double prod[kSpmvNnz * NL];
  __shared__ __attribute__((aligned(16))) unsigned long long fst[2 * 80];
  __shared__ double fred[32];
  __shared__ unsigned long long crec[10];
  __shared__ int okf;
  const int j = (int)blockIdx.x / M.per_iter;
  int b = (int)blockIdx.x - j * M.per_iter;
  if (j >= M.K) {  // (behind the last iteration: its long update)
    const unsigned long long hd_ = __hip_atomic_load(M.hdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int vb = b; vb < M.ua[0].nblk; vb += M.nlong) {
      multi_update_wg(M, M.K - 1, 2, vb, multi_over(hd_), prod, fst, &okf);
      __syncthreads();
    }
    return;
  }
  const unsigned int seq = M.seq0 + (unsigned int)j;
  const int it = M.it0 + j, q = it & 1;
  // (the first load of every workgroup; leaders and update workgroups -- which wait for records anyway -- act on it at once, product
  // workgroups behind their first stream loads: MultiCtx::hd)
  const unsigned long long hd = __hip_atomic_load(M.hdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const FuseGrid& fg = M.fg;
  if (b < kRideCand) {
    if (multi_over(hd)) multi_publish_void((b >> 3) & 1, M.rec_h + (size_t)(seq & (kRecRing - 1)) * 512 + 64 * ride_xcc(), seq);
    else multi_head_leader(M, j, b, fred, fst, &okf);
    return;
  }
  b -= kRideCand;
  RideArgs ra{};
  ra.rec = M.rec_h + (size_t)(seq & (kRecRing - 1)) * 512;
  ra.want = seq;
  ra.pub = seq;
  ra.err = M.err;
  RideArgs rb = ra;
  rb.rec = M.rec_m + (size_t)(seq & (kRecRing - 1)) * 512;
  FuseArgs fz{};
  fz.blkflag = M.flag[q];
  fz.ptag = M.ptag[q];
  fz.dep = M.dep;
  fz.want = seq;
  fz.pub = seq ^ M.break_pub;
  fz.err = M.err;
  MultiCtx mx{};
  mx.gflag_prev = j > 0 ? M.gflag[q ^ 1] : nullptr;
  mx.want_prev = seq - 1;
  mx.bdep = M.bdep;
  mx.gflag = M.gflag[q];
  mx.atag = M.atag[q];
  mx.want = seq;
  mx.hdone = M.hdone;
  mx.hd = hd;
  const int rd = (M.sp0 + j) & 1;          // the short pair this iteration gathers (and its updates read)
  const int ly = (M.lp0 + j) & 1;          // the long pair its A' blocks read their yin rows from; they write the other
  if (b < fg.nwg_t) {
    int Lt[2], nt;
    if (!atl_blocks_of(b, fg.n2, fg.bpx, AT.nblk, Lt, nt, fg.rot)) return;
    atl_product<CSORT, false, true, true>(AT, M.sp[rd], M.lp[ly], M.lp[ly ^ 1], nullptr, 0, Lt, nt, ra, HaloRows{}, fz, prod, crec, &okf, &mx);
    return;
  }
  b -= fg.nwg_t;
  if (b < M.nlong) {  // CRAIG's long update of the PREVIOUS iteration (iteration 0: of the previous launch -- done there)
    if (j > 0)
      for (int vb = b; vb < M.ua[0].nblk; vb += M.nlong) {
        multi_update_wg(M, j - 1, 2, vb, multi_over(hd), prod, fst, &okf);
        __syncthreads();
      }
    return;
  }
  b -= M.nlong;
  if (b < kRideCand) {
    if (multi_over(hd)) multi_publish_void((b >> 3) & 1, rb.rec + 64 * ride_xcc(), seq);
    else multi_mid_leader(M, j, b, AT.nblk, fred, fst, &okf);
    return;
  }
  b -= kRideCand;
  if (b < 8 * fg.gpx) {
    const int g = (b & 7) * fg.gpx + (b >> 3);
    if (g >= RA.ng) return;
    rgcs_group<NL, true, true, true, true>(RA, M.lp[ly ^ 1], M.sp[rd], M.sp[rd ^ 1], nullptr, nullptr, j == M.K - 1 ? M.part_last : nullptr,
                                           M.pstride_a, g, rb, fz, prod, crec, &okf, &mx);
    return;
  }
  b -= 8 * fg.gpx;
  // The riding updates: FEW real workgroups, each walking several of the segments' (virtual) workgroups -- same arithmetic, same
  // partials per virtual workgroup.  They sit between this iteration's row groups and the next iteration's leaders and A' blocks in
  // the grid: a thousand of them would take every slot the draining row groups free for ~10 us, and the next A' blocks -- whose early
  // start is the point of this kernel -- would be dispatched only behind them.
  if (b < fg.nupd_t) {
    const int tot = M.ut[0].nblk + M.ut[1].nblk;
    for (int vb = b; vb < tot; vb += fg.nupd_t) {
      const int s = vb < M.ut[0].nblk ? 0 : 1;
      multi_update_wg(M, j, s, s == 0 ? vb : vb - M.ut[0].nblk, multi_over(hd), prod, fst, &okf);
      __syncthreads();
    }
    return;
  }
  b -= fg.nupd_t;
  // CRAIG: the short update (and the long one too when it is not deferred: M.nlong == 0), the short one first
  const int nshort = M.ua[1].nblk, tot = nshort + (M.nlong == 0 ? M.ua[0].nblk : 0);
  for (int vb = b; vb < tot; vb += M.nupd_a) {
    const int s = vb < nshort ? 3 : 2;
    multi_update_wg(M, j, s, s == 3 ? vb : vb - nshort, multi_over(hd), prod, fst, &okf);
    __syncthreads();
  }
}

}  // namespace fpsq

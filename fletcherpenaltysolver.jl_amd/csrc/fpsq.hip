// fpsq.hip -- host side of libfpsq.so: the C ABI of include/fpsq.h, Jacobian storage (CSR of A and of A'),
// and the stream orchestration of the device-resident Krylov recurrences.
//
// Reference path replaced: src/solve_linear_system.jl:45-140 + src/solve_two_systems_struct.jl:167-244
// (FletcherPenaltySolver.jl v0.3.0), whose arithmetic runs in Krylov.jl on the CPU.
#include "../../include/fpsq.h"
#include "fpsq_spmv.hip.h"
#include "fpsq_multi.hip.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>  // types only; the library is dlopen'ed on first use

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

using namespace fpsq;

namespace {

thread_local std::string g_create_error;

struct DevCsr {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  int32_t* rowptr = nullptr;
  int32_t* colind = nullptr;
  double* vals = nullptr;
  int32_t* rowblk = nullptr;
  int32_t nblk = 0;
  int32_t row_align = 1;       // make_rowblocks' alignment of the block boundaries (8 for the A' of a fused-iteration handle)
  uint16_t* col16 = nullptr;   // compressed columns (see CsrView), null when not representable
  int32_t* colbase = nullptr;
  int4* blkdesc = nullptr;
  bool padded = false;         // vals / col16 / colind hold nblk blocks of kSpmvNnz slots (see k_spmv<.., PAD>)
  int64_t nstore = 0;          // stored value slots: nnz, or nblk * kSpmvNnz when padded
  int32_t win = 0;             // with col16: widest column span of a row block
  uint16_t* cs16 = nullptr;    // column-sorted padded blocks (k_spmv<.., CSORT>): slot | (col & 31) << 11 ...
  uint8_t* cs8 = nullptr;      // ... and col >> 5 of every stored entry; col16 is then not kept
  bool sorted = false;
  // SHARED VALUES (A' only; see pad_blocks): the blocks hold no values of their own -- every entry is read from the row-group
  // copy of A (`vals_ext` = DevRgcs::vals), located through one 16-byte descriptor per 64 consecutive entries
  uint4* segdesc = nullptr;
  const double* vals_ext = nullptr;
  int64_t zero_pos = 0;
  bool shared = false;
  CsrView view() const {
    return CsrView{rowptr, colind, shared ? vals_ext : vals, rowblk, nblk, (int32_t)nrows, col16, colbase, blkdesc, cs16, cs8, segdesc,
                   (int32_t)zero_pos, vals};
  }
};

struct EventPair {
  hipEvent_t a, b;
};

struct DevRgcs {
  bool ok = false;
  RgcsView view{};
  double* vals = nullptr;
  int32_t* vperm = nullptr;  // vals[t] = A.vals[vperm[t]]
  int64_t nnz = 0;
  int64_t nstore = 0;        // stored value slots (> nnz in the padded layout)
};

// ------------------------------------------------------------------ communicators (row-sharded A)
// Collectives are enqueued on the solver's stream; every rank issues the same sequence (the Krylov loop takes
// its exit decision from replicated, bitwise-identical device state at fixed iteration boundaries).
struct Comm {
  int nranks = 1, rank = 0;
  std::string err;
  virtual int allreduce_sum(double* buf, size_t count, hipStream_t s) = 0;
  // Halo mode: vec is this rank's [n_loc][NL] window of raw partial products.  Its first tl rows are the same global
  // columns as the last tl rows of rank - 1's window, its last tr rows the first tr rows of rank + 1's.  On return
  // (stream order) recvL / recvR hold the neighbours' partials on those regions; vec itself is untouched.
  virtual int halo_exchange(const double* vec, int64_t n_loc, int NL, int64_t tl, int64_t tr, double* recvL,
                            double* recvR, hipStream_t s) = 0;
  // recv[r * count + i] = rank r's send[i].  Data movement only: the sums are formed by the step kernel in a fixed
  // rank-major order, so replicated scalars are bitwise identical on every rank by construction.
  virtual int allgather(const double* send, double* recv, size_t count, hipStream_t s) = 0;
  // halo mode, once the handle's exchange buffers exist (collective): a peer-to-peer communicator learns its peers' here
  struct Buffers {
    double* gath[2];      // the two (parity) receive buffers of the all-gathers, [nranks][seg_len] each
    double* halo_recv;    // [(ovl + ovr)][2]
    int64_t ovl, ovr;
  };
  // peer-to-peer routes: the exchange and the finish of the overlap rows as ONE launch (k_p2p_halo_finish); false: not here
  virtual bool halo_exchange_finish(int NL, const HaloFinishArgs& fa, int finish_wgs, hipStream_t s) { return false; }
  // ... and both INSIDE the one-launch iteration (k_iter_fused<.., HALO>): fills the peers' part of the launch's FuseHalo (slots,
  // flag words, the exchange's sequence number); false: this communicator cannot (RCCL: the exchange is a library call)
  virtual bool halo_fused_args(const double* recv, int64_t tl, int64_t tr, FuseHalo& fh) { return false; }
  virtual int arm(const Buffers&, hipStream_t) { return 0; }
  virtual bool failed() { return false; }  // a bounded wait of the peer-to-peer route expired
  // Memory a peer may write into (the gather buffers, the halo slots): a communicator that exports it to other processes
  // or devices decides how it is allocated (fine-grained: visible to a polling kernel across devices).  Freed with hipFree.
  virtual hipError_t alloc_exchange(void** p, size_t bytes) { return hipMalloc(p, bytes); }
  // how the exchanges of the Krylov loop travel (fpsq_info.comm_route)
  virtual int route() const { return FPSQ_ROUTE_RCCL; }
  // Sums over the ranks formed INSIDE the launches that need them (fpsq_krylov.hip.h xch_sum): the device-resident peer table,
  // null when this communicator does not do that (RCCL route; ranks sharing a device).  Known after arm().
  virtual const XchTable* xch_table() const { return nullptr; }
  // ... and the looks the OTHER workgroups of such a launch get beyond kRidePolls (RideArgs::more / FuseArgs::more): they wait for
  // leaders that may be waiting for a late peer, so their bound has to outlast the leaders' (4 x: a follower's look is shorter)
  virtual int wait_more() const { return 0; }
  virtual ~Comm() {}
};

// ---- the peer-to-peer exchange route (halo-sharded loop): NO collective call inside the Krylov loop.  A rank WRITES its
// record straight into its peers' buffers, then its sequence number into their flag words, and waits -- in the same
// one-workgroup kernel, a bounded number of polls -- until its own flag words carry that number (k_p2p_gather, k_p2p_halo).
// Who the peers are is the communicator's business: the other shards of one process (P2PLocalComm: pointers on the same
// device) or the other ranks of a node (IpcComm: their buffers mapped with hipIpcOpenMemHandle; the stores then travel over
// xGMI).  Ordering: gathers alternate between two buffers -- a peer can be at most one reduction ahead, and what it then
// overwrites was consumed before this rank's previous push (which the peer's current one waited for); halo slots alternate
// the same way.
// Every peer table of the peer-to-peer route (here, IpcComm::opened, LocalGroup, P2PPeers in the kernel arguments) has this many
// entries: the GPUs of one node.  More ranks (two nodes, 16 logical ranks) stay on RCCL -- decided in arm(), unanimously.
constexpr int kMaxP2PRanks = 8;
static_assert(sizeof(P2PPeers::rx) / sizeof(double*) == kMaxP2PRanks && sizeof(P2PPeers::flag) / sizeof(unsigned long long*) == kMaxP2PRanks,
              "k_p2p_gather's peer table");
struct P2PRoute {
  int nranks = 1, rank = 0;
  bool armed = false;
  Comm::Buffers mine{};
  // receive area of the all-gathers: [2 parities][rx_half doubles], rx_half >= nranks x the longest record.  The peers write
  // into it; the gather kernel copies what arrived into the handle's ordinary buffer (Buffers::gath), which is what the
  // scalar steps read.  Allocated by the communicator at arm() (exported / fine-grained when the peers are other processes).
  double* rx = nullptr;
  int64_t rx_half = 0;
  double* peer_rx[2][kMaxP2PRanks] = {};
  unsigned long long* peer_flags[kMaxP2PRanks] = {};  // 8 gather words (one per sender), then "from left", "from right"
  double* peer_halo[kMaxP2PRanks] = {};
  int64_t peer_ovl[kMaxP2PRanks] = {}, peer_ovr[kMaxP2PRanks] = {};
  unsigned long long* flags = nullptr;  // mine (device; sequence numbers, monotone); behind the 16 flag words: the receive
                                        // area of the in-launch sums (xch_sum), so that ONE mapped allocation serves both
  static constexpr size_t kFlagWords = 16 + (size_t)kXchRing * kXchRanks * kXchWords;
  XchTable* xt_dev = nullptr;           // non-null: the sums over the ranks are formed inside the launches (lx)
  int lx_want = 1;                      // FPSQ_LX: 0 never, 1 (default) when every rank has a device of its own, 2 always (tests with small grids)
  int xch_delay_rank = 0;               // FPSQ_DEBUG_XCH_DELAY (tests)
  int halo_dbg = 0;                     // HaloFinishArgs::dbg (tests: FPSQ_DEBUG_P2P_DELAY = r + 1)
  int halo_delay_rank = 0;
  int* fail_host = nullptr;             // host-mapped: a bounded wait expired
  int* fail_dev = nullptr;
  unsigned long long gather_seq = 0, halo_seq = 0;
  long max_spins = 50000000L;           // bound of every in-kernel wait (FPSQ_P2P_POLLS; ~1-2 us per poll)
  bool failed() const { return fail_host && *fail_host != 0; }
  int wait_more_dbg = -1;               // FPSQ_DEBUG_WAIT_MORE (tests: 0 = the bound of one GPU)
  int wait_more() const {
    if (wait_more_dbg >= 0) return wait_more_dbg;
    return (int)std::min<long>(4 * std::min<long>(max_spins, (long)INT32_MAX / 8), (long)INT32_MAX / 2);
  }
  int xch_long_delay_ms = 0;            // FPSQ_DEBUG_XCH_LONG_DELAY_MS (tests; with FPSQ_DEBUG_XCH_DELAY naming the rank)
  int alloc_fail_word(std::string& err) {
    if (const char* ev = std::getenv("FPSQ_P2P_POLLS")) max_spins = std::max(1L, std::atol(ev));
    if (const char* ev = std::getenv("FPSQ_HALO_FUSE")) fuse_halo = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("FPSQ_LX")) lx_want = std::atoi(ev);
    if (const char* ev = std::getenv("FPSQ_DEBUG_XCH_DELAY")) xch_delay_rank = std::atoi(ev);
    if (const char* ev = std::getenv("FPSQ_DEBUG_XCH_LONG_DELAY_MS")) xch_long_delay_ms = std::max(0, std::min(2000, std::atoi(ev)));
    if (const char* ev = std::getenv("FPSQ_DEBUG_WAIT_MORE")) wait_more_dbg = std::max(0, std::atoi(ev));
    if (const char* ev = std::getenv("FPSQ_DEBUG_P2P_DELAY")) halo_delay_rank = std::atoi(ev);
    halo_dbg = halo_delay_rank == rank + 1 ? 1 : 0;
    if (hipHostMalloc((void**)&fail_host, 4, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&fail_dev, fail_host, 0) != hipSuccess) {
      err = "p2p arm: allocation failed";
      return FPSQ_ERR_HIP;
    }
    *fail_host = 0;
    return 0;
  }
  // the peer table of the in-launch sums, once peer_flags[] is known
  int make_xch_table(std::string& err) {
    XchTable T{};
    for (int r = 0; r < nranks; ++r) T.peer[r] = peer_flags[r] + 16;
    T.nranks = nranks;
    T.rank = rank;
    T.max_polls = (int32_t)std::min<long>(max_spins, (long)INT32_MAX);
    T.delay_rank = xch_delay_rank;
    T.long_delay_ticks = (unsigned int)xch_long_delay_ms * 100000u;  // (100 MHz)
    T.fail = fail_dev;
    if (hipMalloc((void**)&xt_dev, sizeof(XchTable)) != hipSuccess ||
        hipMemcpy(xt_dev, &T, sizeof T, hipMemcpyHostToDevice) != hipSuccess) {
      err = "p2p arm: allocation failed";
      return FPSQ_ERR_HIP;
    }
    return 0;
  }
  bool is_gather_buffer(const double* recv) const { return armed && (recv == mine.gath[0] || recv == mine.gath[1]); }
  void allgather(const double* send, double* recv, size_t count, hipStream_t s) {
    const int par = recv == mine.gath[1];
    P2PPeers P{};
    P.n = nranks;
    for (int r = 0; r < nranks; ++r) {
      P.rx[r] = peer_rx[par][r];
      P.flag[r] = peer_flags[r];
    }
    // (a long record -- few ranks, many row blocks each -- gets extra workgroups for the copy of the rank's own part)
    const int extra = (int)std::min<size_t>(7, count / (8 * kBlock));
    hipLaunchKernelGGL(k_p2p_gather, dim3(nranks + extra), dim3(kBlock), 0, s, send, (int64_t)count, P, rank, ++gather_seq,
                       rx + (size_t)par * rx_half, recv, fail_dev, max_spins);
  }
  void halo_exchange(const double* vec, int NL, int64_t tl, int64_t tr, const double* recvL, hipStream_t s) {
    const P2PHalo H = halo_peers(NL, tl, tr, recvL);
    hipLaunchKernelGGL(k_p2p_halo, dim3(2), dim3(1024), 0, s, vec, tl * NL, tr * NL, H, ++halo_seq, fail_dev, max_spins);
  }
  P2PHalo halo_peers(int NL, int64_t tl, int64_t tr, const double* recvL) const {
    P2PHalo H{};
    const int par = recvL != mine.halo_recv;  // which half of the (double-buffered) slots this exchange uses: the same on
                                              // every rank (all ranks make the same sequence of exchanges)
    if (rank > 0 && tl > 0) {  // my head region = the left neighbour's tail slot (behind its own head slot)
      const int L = rank - 1;
      H.left_dst = peer_halo[L] + (size_t)par * (size_t)(peer_ovl[L] + peer_ovr[L]) * 2 + (size_t)peer_ovl[L] * NL;
      H.left_flag = peer_flags[L] + 9;  // its "from right" word
      H.my_from_left = flags + 8;
    }
    if (rank < nranks - 1 && tr > 0) {
      const int R = rank + 1;
      H.right_dst = peer_halo[R] + (size_t)par * (size_t)(peer_ovl[R] + peer_ovr[R]) * 2;
      H.right_flag = peer_flags[R] + 8;  // its "from left" word
      H.my_from_right = flags + 9;
    }
    return H;
  }
  bool fuse_halo = true;  // FPSQ_HALO_FUSE=0: exchange and finish as two launches
  bool halo_fused_args(const double* recv, int64_t tl, int64_t tr, FuseHalo& fh) {
    if (!armed) return false;
    fh.H = halo_peers(2, tl, tr, recv);
    fh.seq = ++halo_seq;
    fh.fail = fail_dev;
    fh.max_spins = max_spins;
    fh.arrive = flags + 12;
    return true;
  }
  bool halo_exchange_finish(int NL, const HaloFinishArgs& fa, int finish_wgs, hipStream_t s) {
    if (!armed || !fuse_halo) return false;
    const P2PHalo H = halo_peers(NL, fa.tl, fa.tr, fa.recv);
    const dim3 grid(2 * kHaloCopy + finish_wgs);
    unsigned long long* arrive = flags + 12;  // (words 12, 13 of my flag block: arrival counters of the copy slices, per side)
    HaloFinishArgs fb = fa;
    fb.dbg = halo_dbg;
    if (NL == 2)
      hipLaunchKernelGGL(k_p2p_halo_finish<2>, grid, dim3(kBlock), 0, s, H, ++halo_seq, fail_dev, max_spins, arrive, fb);
    else
      hipLaunchKernelGGL(k_p2p_halo_finish<1>, grid, dim3(kBlock), 0, s, H, ++halo_seq, fail_dev, max_spins, arrive, fb);
    return true;
  }
  void release() {
    if (xt_dev) hipFree(xt_dev);
    xt_dev = nullptr;
    if (rx) hipFree(rx);
    rx = nullptr;
    if (flags) hipFree(flags);
    if (fail_host) hipHostFree(fail_host);
    flags = nullptr;
    fail_host = nullptr;
  }
};

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string& err) {
    if (lib) return true;
    // by SONAME first: a process that imported torch already holds librccl.so.1 and must keep using that copy
    // FPSQ_RCCL_LIB: another build of the collectives library (a site build; the multi-process loopback stand-in of
    // tests/shim, which lets the multi-rank path run on a one-GPU box) -- then that one or nothing
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (const char* ov = std::getenv("FPSQ_RCCL_LIB")) {
      lib = dlopen(ov, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char* nm : names)
        if ((lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    }
    if (!lib) {
      err = std::string("cannot dlopen librccl: ") + dlerror();
      return false;
    }
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    Send = (decltype(Send))dlsym(lib, "ncclSend");
    Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !AllGather || !CommDestroy || !GetErrorString || !Send || !Recv || !GroupStart ||
        !GroupEnd) {
      err = "librccl is missing a required symbol";
      return false;
    }
    return true;
  }
};
RcclApi g_rccl;

struct RcclComm : Comm {
  ncclComm_t c = nullptr;
  int allreduce_sum(double* buf, size_t count, hipStream_t s) override {
    ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, c, s);
    if (r != ncclSuccess) {
      err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r);
      return FPSQ_ERR_COMM;
    }
    return 0;
  }
  int allgather(const double* send, double* recv, size_t count, hipStream_t s) override {
    ncclResult_t r = g_rccl.AllGather(send, recv, count, ncclDouble, c, s);
    if (r != ncclSuccess) {
      err = std::string("ncclAllGather: ") + g_rccl.GetErrorString(r);
      return FPSQ_ERR_COMM;
    }
    return 0;
  }
  // neighbour-to-neighbour exchange over xGMI: one grouped send/recv pair per neighbour (<= 2 x window x NL doubles)
  int halo_exchange(const double* vec, int64_t n_loc, int NL, int64_t tl, int64_t tr, double* recvL, double* recvR,
                    hipStream_t s) override {
    ncclResult_t r = g_rccl.GroupStart();
    if (r == ncclSuccess && rank > 0 && tl > 0) {
      r = g_rccl.Send(vec, (size_t)tl * NL, ncclDouble, rank - 1, c, s);
      if (r == ncclSuccess) r = g_rccl.Recv(recvL, (size_t)tl * NL, ncclDouble, rank - 1, c, s);
    }
    if (r == ncclSuccess && rank < nranks - 1 && tr > 0) {
      r = g_rccl.Send(vec + (size_t)(n_loc - tr) * NL, (size_t)tr * NL, ncclDouble, rank + 1, c, s);
      if (r == ncclSuccess) r = g_rccl.Recv(recvR, (size_t)tr * NL, ncclDouble, rank + 1, c, s);
    }
    const ncclResult_t e = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) {
      err = std::string("halo exchange (ncclSend/ncclRecv): ") + g_rccl.GetErrorString(r);
      return FPSQ_ERR_COMM;
    }
    return 0;
  }
  ~RcclComm() override {
    if (c) g_rccl.CommDestroy(c);
  }
};

// The ranks of ONE NODE, one process per GPU: RCCL for the set-up collectives and as the fallback, the peer-to-peer route
// (P2PRoute) for the exchanges of the halo-sharded Krylov loop.  At arm() every rank exports its two gather buffers, its
// halo slots and its flag words with hipIpcGetMemHandle, the handles travel through one RCCL all-gather, every rank maps
// its peers' with hipIpcOpenMemHandle (peer access enabled lazily: the stores of k_p2p_gather / k_p2p_halo then go over
// xGMI), and a second all-gather makes the decision unanimous: if ANY rank could not export or open, all stay on RCCL.
// At the headline size the RCCL route pays three collective calls (15-30 us each) per joint iteration against ~8 us of
// products on 8 GPUs; this one pays three one-workgroup kernels.  IPC handles open between processes sharing ONE device
// too, which is how the route is tested here (tests/test_gpu_p2p_ipc.py: 2 and 3 processes on one GPU).
struct IpcComm : RcclComm {
  int want = FPSQ_ROUTE_AUTO;   // fpsq_comm_set_route / FPSQ_COMM_ROUTE
  P2PRoute rt;
  std::string note;             // why the route fell back to RCCL (fpsq_last_error after a FPSQ_ROUTE_P2P request)
  void* opened[kMaxP2PRanks][3] = {};
  struct Blob {                 // what a rank tells its peers (padded to whole doubles)
    hipIpcMemHandle_t h[3];     // receive area of the gathers (one allocation, both parities), halo slots, flag words
    int64_t ovl, ovr, rx_half;  // rx_half: doubles between the two parities of the receive area
    int32_t ok, pid;
    int32_t lx_want, pad;       // FPSQ_LX of that rank (the in-launch sums are switched on unanimously)
    char dev[48];               // PCI bus id of its device: two ranks on ONE device keep the exchange kernels (see xch_sum)
  };
  static constexpr size_t kBlobDoubles = (sizeof(Blob) + 7) / 8;
  hipError_t alloc_exchange(void** p, size_t bytes) override {
    if (want == FPSQ_ROUTE_RCCL) return hipMalloc(p, bytes);
    // fine-grained: a peer's stores must become visible to a kernel of this device that is polling / about to read
    hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      e = hipMalloc(p, bytes);
    }
    return e;
  }
  int route() const override { return rt.armed ? FPSQ_ROUTE_P2P : FPSQ_ROUTE_RCCL; }
  const XchTable* xch_table() const override { return rt.armed ? rt.xt_dev : nullptr; }
  int wait_more() const override { return rt.armed && rt.xt_dev ? rt.wait_more() : 0; }
  bool failed() override { return rt.failed(); }
  int arm(const Buffers& b, hipStream_t s) override {
    if (want == FPSQ_ROUTE_RCCL) return 0;
    if (nranks > kMaxP2PRanks) {  // (every rank sees the same nranks: the same decision everywhere, no exchange needed)
      note = "more than " + std::to_string(kMaxP2PRanks) + " ranks: the peer tables of the peer-to-peer route hold one node's GPUs";
      if (want == FPSQ_ROUTE_P2P) {
        err = "peer-to-peer route requested but not available: " + note;
        return FPSQ_ERR_COMM;
      }
      return 0;
    }
    rt.nranks = nranks;
    rt.rank = rank;
    rt.mine = b;
    Blob me{};
    me.ok = 1;
    me.pid = (int32_t)getpid();
    me.ovl = b.ovl;
    me.ovr = b.ovr;
    rt.rx_half = b.gath[1] - b.gath[0];
    me.rx_half = rt.rx_half;
    if (hipExtMallocWithFlags((void**)&rt.flags, P2PRoute::kFlagWords * 8, hipDeviceMallocFinegrained) != hipSuccess ||
        hipExtMallocWithFlags((void**)&rt.rx, (size_t)rt.rx_half * 2 * 8, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      if (rt.flags) hipFree(rt.flags);
      rt.flags = nullptr;
      rt.rx = nullptr;
      me.ok = 0;
      note = "fine-grained allocation of the flag words / receive area failed";
      // (the kernels are never launched without them: the route stays unarmed)
    } else {
      hipMemset(rt.flags, 0, P2PRoute::kFlagWords * 8);
    }
    if (int rc = rt.alloc_fail_word(err)) return rc;
    me.lx_want = rt.lx_want;
    {
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(me.dev, (int)sizeof me.dev, dev) != hipSuccess) {
        (void)hipGetLastError();
        std::snprintf(me.dev, sizeof me.dev, "?");  // (unknown: counts as shared)
      }
    }
    if (me.ok && nranks > 1) {
      void* base[3] = {rt.rx, b.halo_recv, rt.flags};
      for (int k = 0; k < 3 && me.ok; ++k)
        if (hipIpcGetMemHandle(&me.h[k], base[k]) != hipSuccess) {
          (void)hipGetLastError();
          me.ok = 0;
          note = "hipIpcGetMemHandle failed";
        }
    }
    hipDeviceSynchronize();
    // round 1: everybody's blob
    std::vector<double> all(kBlobDoubles * nranks), mine_d(kBlobDoubles, 0.0);
    std::memcpy(mine_d.data(), &me, sizeof me);
    double *dsend = nullptr, *drecv = nullptr;
    if (hipMalloc((void**)&dsend, kBlobDoubles * 8) != hipSuccess || hipMalloc((void**)&drecv, all.size() * 8) != hipSuccess) {
      err = "p2p arm: allocation failed";
      return FPSQ_ERR_HIP;
    }
    auto gather_round = [&](const std::vector<double>& snd, size_t cnt) -> int {
      if (hipMemcpyAsync(dsend, snd.data(), cnt * 8, hipMemcpyHostToDevice, s) != hipSuccess) return FPSQ_ERR_HIP;
      if (int rc = RcclComm::allgather(dsend, drecv, cnt, s)) return rc;
      if (hipMemcpyAsync(all.data(), drecv, cnt * nranks * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess)
        return FPSQ_ERR_HIP;
      return 0;
    };
    int rc = gather_round(mine_d, kBlobDoubles);
    std::vector<Blob> blobs(nranks);
    bool ok = rc == 0;
    if (rc == 0) {
      for (int r = 0; r < nranks; ++r) {
        std::memcpy(&blobs[r], all.data() + kBlobDoubles * r, sizeof(Blob));
        if (!blobs[r].ok) {
          ok = false;
          if (note.empty()) note = "rank " + std::to_string(r) + " could not export its buffers";
        }
      }
    }
    // map the peers' buffers
    if (ok) {
      for (int r = 0; r < nranks && ok; ++r) {
        if (r == rank) continue;
        for (int k = 0; k < 3 && ok; ++k)
          if (hipIpcOpenMemHandle(&opened[r][k], blobs[r].h[k], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            opened[r][k] = nullptr;
            ok = false;
            note = "hipIpcOpenMemHandle failed for rank " + std::to_string(r) +
                   (blobs[r].pid == me.pid ? " (same process: use the in-process group instead)" : "");
          }
      }
    }
    // round 2: unanimous or not at all
    if (rc == 0) {
      std::vector<double> v(1, ok ? 1.0 : 0.0);
      rc = gather_round(v, 1);
      if (rc == 0)
        for (int r = 0; r < nranks; ++r)
          if (all[r] == 0.0) {
            if (ok && note.empty()) note = "rank " + std::to_string(r) + " could not map its peers' buffers";
            ok = false;
          }
    }
    hipFree(dsend);
    hipFree(drecv);
    if (rc) return rc;
    if (!ok) {
      close_peers();
      if (want == FPSQ_ROUTE_P2P) {
        err = "peer-to-peer route requested but not available: " + note;
        return FPSQ_ERR_COMM;
      }
      return 0;  // (every rank took the same decision: the RCCL route)
    }
    for (int r = 0; r < nranks; ++r) {
      const bool self = r == rank;
      double* g0 = self ? rt.rx : (double*)opened[r][0];
      rt.peer_rx[0][r] = g0;
      rt.peer_rx[1][r] = g0 + blobs[r].rx_half;
      rt.peer_halo[r] = self ? b.halo_recv : (double*)opened[r][1];
      rt.peer_flags[r] = self ? rt.flags : (unsigned long long*)opened[r][2];
      rt.peer_ovl[r] = blobs[r].ovl;
      rt.peer_ovr[r] = blobs[r].ovr;
    }
    // In-launch sums over the ranks (xch_sum): every rank must want them, and either every rank has a device of its own or every
    // rank forces them (FPSQ_LX=2: tests whose grids are resident all at once).  Every rank sees the same blobs: same decision.
    {
      bool all_on = true, all_force = true, distinct = true;
      for (int r = 0; r < nranks; ++r) {
        all_on = all_on && blobs[r].lx_want >= 1;
        all_force = all_force && blobs[r].lx_want >= 2;
        for (int q = 0; q < r; ++q)
          if (std::strncmp(blobs[r].dev, blobs[q].dev, sizeof blobs[r].dev) == 0 || blobs[r].dev[0] == '?') distinct = false;
      }
      if (nranks > 1 && all_on && (distinct || all_force))
        if (int rc2 = rt.make_xch_table(err)) return rc2;
    }
    rt.armed = true;
    return 0;
  }
  int allgather(const double* send, double* recv, size_t count, hipStream_t s) override {
    if (!rt.is_gather_buffer(recv)) return RcclComm::allgather(send, recv, count, s);
    rt.allgather(send, recv, count, s);
    return 0;
  }
  int halo_exchange(const double* vec, int64_t n_loc, int NL, int64_t tl, int64_t tr, double* recvL, double* recvR,
                    hipStream_t s) override {
    if (!rt.armed) return RcclComm::halo_exchange(vec, n_loc, NL, tl, tr, recvL, recvR, s);
    rt.halo_exchange(vec, NL, tl, tr, recvL, s);
    return 0;
  }
  bool halo_exchange_finish(int NL, const HaloFinishArgs& fa, int finish_wgs, hipStream_t s) override {
    return rt.halo_exchange_finish(NL, fa, finish_wgs, s);
  }
  bool halo_fused_args(const double* recv, int64_t tl, int64_t tr, FuseHalo& fh) override { return rt.halo_fused_args(recv, tl, tr, fh); }
  void close_peers() {
    for (int r = 0; r < kMaxP2PRanks; ++r)
      for (int k = 0; k < 3; ++k)
        if (opened[r][k]) {
          hipIpcCloseMemHandle(opened[r][k]);
          opened[r][k] = nullptr;
        }
  }
  ~IpcComm() override {
    close_peers();
    rt.release();
  }
};

// P logical shards in ONE process on ONE device (each handle driven by its own host thread): the sum is a kernel.
struct LocalGroup {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  double* bufs[8] = {};
  const double* vecs[8] = {};  // halo exchange: every shard's window of partial products and its length
  int64_t nloc[8] = {};
  hipEvent_t ready[8] = {};
  hipEvent_t copied[8] = {};
  hipEvent_t done = nullptr;
  // peer-to-peer route (fpsq_local_group_set_p2p): what every shard published at arm()
  bool p2p = false;
  struct Pub {
    double* rx[2];
    unsigned long long* flags;  // 8 gather flag words (one per sender), then "from left", "from right"
    double* halo_recv;
    int64_t ovl, ovr;
  } pub[8] = {};
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long gen = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

struct LocalComm : Comm {
  LocalGroup* g = nullptr;
  int route() const override { return FPSQ_ROUTE_LOCAL; }
  int allreduce_sum(double* buf, size_t count, hipStream_t s) override {
    g->bufs[rank] = buf;
    hipEventRecord(g->ready[rank], s);
    g->barrier();
    if (rank == 0) {
      ShardBufs B;
      B.n = g->n;
      for (int r = 0; r < g->n; ++r) {
        hipStreamWaitEvent(s, g->ready[r], 0);
        B.b[r] = g->bufs[r];
      }
      const int grid = (int)std::max<size_t>(1, std::min<size_t>((count + kBlock - 1) / kBlock, 2048));
      hipLaunchKernelGGL(k_local_allreduce, dim3(grid), dim3(kBlock), 0, s, B, (int64_t)count);
      hipEventRecord(g->done, s);
    }
    g->barrier();
    hipStreamWaitEvent(s, g->done, 0);
    g->barrier();  // nobody may start the next collective (and overwrite bufs[] / re-record events) before all queued the wait
    return 0;
  }
  int allgather(const double* send, double* recv, size_t count, hipStream_t s) override {
    g->vecs[rank] = send;
    hipEventRecord(g->ready[rank], s);
    g->barrier();  // every shard's source pointer and `ready` event are published
    GatherSrc S;
    S.n = g->n;
    for (int r = 0; r < g->n; ++r) {
      if (r != rank) hipStreamWaitEvent(s, g->ready[r], 0);
      S.s[r] = g->vecs[r];
    }
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((count * g->n + kBlock - 1) / kBlock, 256));
    hipLaunchKernelGGL(k_local_allgather, dim3(grid), dim3(kBlock), 0, s, S, recv, (int64_t)count);
    hipEventRecord(g->copied[rank], s);
    g->barrier();  // every `copied` event is recorded
    // a shard's next kernels rewrite its source array: every other shard must have taken its copy first
    for (int r = 0; r < g->n; ++r)
      if (r != rank) hipStreamWaitEvent(s, g->copied[r], 0);
    g->barrier();  // the events may be re-recorded by the next collective only after everyone queued its waits
    return 0;
  }
  int halo_exchange(const double* vec, int64_t n_loc, int NL, int64_t tl, int64_t tr, double* recvL, double* recvR,
                    hipStream_t s) override {
    g->vecs[rank] = vec;
    g->nloc[rank] = n_loc;
    hipEventRecord(g->ready[rank], s);
    g->barrier();  // every shard's pointer and `ready` event are published
    if (rank > 0 && tl > 0) {
      hipStreamWaitEvent(s, g->ready[rank - 1], 0);
      hipMemcpyAsync(recvL, g->vecs[rank - 1] + (size_t)(g->nloc[rank - 1] - tl) * NL, (size_t)tl * NL * 8,
                     hipMemcpyDeviceToDevice, s);
    }
    if (rank < nranks - 1 && tr > 0) {
      hipStreamWaitEvent(s, g->ready[rank + 1], 0);
      hipMemcpyAsync(recvR, g->vecs[rank + 1], (size_t)tr * NL * 8, hipMemcpyDeviceToDevice, s);
    }
    hipEventRecord(g->copied[rank], s);
    g->barrier();  // every `copied` event is recorded
    // the caller's next kernel modifies vec: both neighbours must have taken their copies of it first
    if (rank > 0) hipStreamWaitEvent(s, g->copied[rank - 1], 0);
    if (rank < nranks - 1) hipStreamWaitEvent(s, g->copied[rank + 1], 0);
    g->barrier();  // the events may be re-recorded by the next collective only after everyone queued its waits
    return 0;
  }
};

// The same logical shards on the peer-to-peer route (P2PRoute): the peers are the other shards' buffers on the same
// device, which exercises the protocol (ordering, double buffering, bounded waits), not a link.  Set-up collectives
// (before arm()) use LocalComm's.
struct P2PLocalComm : LocalComm {
  P2PRoute rt;
  int route() const override { return FPSQ_ROUTE_LOCAL_P2P; }
  const XchTable* xch_table() const override { return rt.armed ? rt.xt_dev : nullptr; }
  int wait_more() const override { return rt.armed && rt.xt_dev ? rt.wait_more() : 0; }
  int arm(const Buffers& b, hipStream_t) override {
    rt.nranks = nranks;
    rt.rank = rank;
    rt.mine = b;
    rt.rx_half = b.gath[1] - b.gath[0];
    if (hipMalloc((void**)&rt.flags, P2PRoute::kFlagWords * 8) != hipSuccess ||
        hipMemset(rt.flags, 0, P2PRoute::kFlagWords * 8) != hipSuccess ||
        hipMalloc((void**)&rt.rx, (size_t)rt.rx_half * 2 * 8) != hipSuccess) {
      err = "p2p arm: allocation failed";
      return FPSQ_ERR_HIP;
    }
    if (int rc = rt.alloc_fail_word(err)) return rc;
    hipDeviceSynchronize();
    LocalGroup::Pub& me = g->pub[rank];
    me.rx[0] = rt.rx;
    me.rx[1] = rt.rx + rt.rx_half;
    me.flags = rt.flags;
    me.halo_recv = b.halo_recv;
    me.ovl = b.ovl;
    me.ovr = b.ovr;
    g->barrier();  // every shard has published
    for (int r = 0; r < nranks; ++r) {
      const LocalGroup::Pub& q = g->pub[r];
      rt.peer_rx[0][r] = q.rx[0];
      rt.peer_rx[1][r] = q.rx[1];
      rt.peer_flags[r] = q.flags;
      rt.peer_halo[r] = q.halo_recv;
      rt.peer_ovl[r] = q.ovl;
      rt.peer_ovr[r] = q.ovr;
    }
    // (the shards share ONE device: the in-launch sums only when a test with small grids forces them -- one environment, one decision)
    if (nranks > 1 && rt.lx_want >= 2)
      if (int rc = rt.make_xch_table(err)) return rc;
    rt.armed = true;
    g->barrier();
    return 0;
  }
  bool failed() override { return rt.failed(); }
  int allgather(const double* send, double* recv, size_t count, hipStream_t s) override {
    if (!rt.is_gather_buffer(recv)) return LocalComm::allgather(send, recv, count, s);
    rt.allgather(send, recv, count, s);
    return 0;
  }
  int halo_exchange(const double* vec, int64_t n_loc, int NL, int64_t tl, int64_t tr, double* recvL, double* recvR,
                    hipStream_t s) override {
    if (!rt.armed) return LocalComm::halo_exchange(vec, n_loc, NL, tl, tr, recvL, recvR, s);
    rt.halo_exchange(vec, NL, tl, tr, recvL, s);
    return 0;
  }
  bool halo_exchange_finish(int NL, const HaloFinishArgs& fa, int finish_wgs, hipStream_t s) override {
    return rt.halo_exchange_finish(NL, fa, finish_wgs, s);
  }
  bool halo_fused_args(const double* recv, int64_t tl, int64_t tr, FuseHalo& fh) override { return rt.halo_fused_args(recv, tl, tr, fh); }
  ~P2PLocalComm() override { rt.release(); }
};

}  // namespace

struct fpsq_solver_s {
  int64_t n = 0, m = 0, nnz = 0;
  fpsq_options opt{};
  double delta = 0.0;
  hipStream_t stream = nullptr;
  bool in_stream_on = false;     // fpsq_set_input_stream: producer stream of device-resident arguments
  hipStream_t in_stream = nullptr;
  // One GPU, a registered producer stream (FPSQ_ADOPT_STREAM=0 switches it off): the library enqueues ON that stream instead of on one of its own -- inputs and outputs
  // are then ordered by the stream itself: no event record / wait pair at either end of a call, and no hops between two queues from
  // the last kernel of an evaluation to the first of the next (the caller's stream waits for the tail, the library's for the caller's)
  bool adopt_streams = true, adopted = false;
  hipStream_t own_stream = nullptr;
  hipEvent_t ev_in = nullptr;
  bool have_structure = false, have_values = false;
  std::string err;

  DevCsr A, AT;
  DevRgcs RA;                   // column-sorted row-group copy of A used by the A product when eligible
  int32_t* permT = nullptr;     // AT.vals[t] = A.vals[permT[t]]
  int64_t nnz_in = 0;           // length of the caller's value array (COO entries or CSR nnz)
  bool perms_to_input = false;  // COO structure without duplicates: permT / RA.vperm are composed down to the caller's array
  bool refresh_3pass = false;   // FPSQ_JAC_REFRESH=3: the three grid-stride gathers of rounds 1-3 (A/B, test)
  int32_t* in_perm = nullptr;   // COO path: sorted position -> caller index
  int32_t* in_slotptr = nullptr;// COO path with duplicates: CSR slot -> range of sorted positions
  double* in_vals = nullptr;    // staging of the caller's values (COO path)

  std::vector<void*> allocs;
  // Golub-Kahan vectors, [len][2] interleaved: LP = "long" (n) pair, SP = "short" (m) pair
  double *LP, *SP;
  double* SP2;                  // alternate short pair: the A product ping-pongs SP so fused updates may read the old one
  // n-vectors
  double *Cx, *Cw2, *in_n1, *in_n2, *p1, *p2b, *gs, *gx, *jc, *g, *xin, *xk;
  // m-vectors
  double *Lw[2], *Lx[2], *Cw, *Cy, *in_m, *ys, *c, *Mr[2], *Mw[2], *Mx;
  // partial-sum buffers
  double *pS, *pS2, *pW[2], *pE, *pE2, *pE3, *pQ[2], *pC[2];
  double* pS2b = nullptr;  // second array for the A product's partials: fused launches alternate (KrylovRun::pa_last)
  // second halves of the update partials.  A riding step and a riding update of ONE launch must never share an array:
  // the leaders of the step (sixteen workgroups, any of which another kernel may hold up) read, the update workgroups --
  // released by the record of their own XCC's leader -- write.  LSQR's / CRAIG's update partials therefore alternate
  // between pW[l] and pWalt[l] by iteration (run_krylov: upd_part), MINRES' stage E3 writes pWalt where E2 writes pW.
  double* pWalt[2];
  double* pEm[2];               // squared-norm partials of the m-vector right-hand sides (pE / pE2: of the n-vector ones)
  int npS = 0;
  int strT = 0, strA = 0;       // lane strides of pS (A' product partials) and pS2 (A product partials)
  LsqrState* lsqr[2];
  CraigState* craig;
  LsqrState* lsqr_alt[2];       // second copies: the target of a step that rides in a product launch (see run_krylov)
  CraigState* craig_alt;
  bool at_sorted = true;        // A' blocks stored column-sorted where representable (FPSQ_AT_SORTED=0: row order)
  bool at_shared = true;        // ... and without values of their own where the row groups of A can serve them (FPSQ_AT_SHARED=0)
  // steps riding with LEADERS (large grids, see fpsq_spmv.hip.h): the leaders' record (one line of device memory), a launch counter
  unsigned long long* ride_rec = nullptr;
  unsigned long long ride_seq = 0;
  bool ride_lead = true;        // FPSQ_RIDE_LEAD=0: large grids keep the stand-alone k_step
  bool ride_break = false;      // FPSQ_DEBUG_RIDE_BREAK=1 (tests): the leaders publish a wrong launch number, every wait expires
  int ride_delay = 0;           // FPSQ_DEBUG_RIDE_DELAY=c+1 (tests): leader c of every launch starts ~100 us late
  int resident_wgs = 1024;      // product workgroups (32 KB of LDS) the device holds at once: 4 per CU, measured
  bool atl_two = true;          // k_spmv_atl: two row blocks for the first resident set (FPSQ_ATL_TWO=0: one each)
  // one launch per joint iteration (k_iter_fused; FPSQ_FUSE_ITER=0: two launches)
  int ride_delay_mid = 0;       // FPSQ_DEBUG_RIDE_DELAY_MID=c+1 (tests): mid leader c of every fused launch starts ~100 us late
  int fuse_rotate = 0;          // FPSQ_DEBUG_FUSE_ROTATE=r (tests): the A' blocks of eighth e are written on XCD (e - r) & 7, gathered on XCD e
  bool minres_merge = true;     // MINRES lane: stage E1, step A and stage E2 as one launch (k_minres_mid; FPSQ_MINRES_MERGE=0: three)
  unsigned long long* mm_ptag = nullptr;  // its tagged partials (two words per element-wise workgroup)
  bool fuse_fell_back = false;  // an expired wait of a fused launch has just switched the handle to two launches per iteration
  bool fuse_break = false;      // FPSQ_DEBUG_FUSE_BREAK=1 (tests): the A' blocks of a fused launch publish a wrong number, every wait for them expires
  int fuse_iter = 1;            // 0: never; 1: where it pays (setup_fused_iteration); 2: wherever it is possible (tests)
  bool fuse_ok = false;
  // The in-launch hand-overs (riding leaders' records per XCC, written-through rows, blocks dealt to XCDs by blockIdx & 7) were
  // validated on gfx942 / gfx950 in SPX mode with 8 XCCs (tools/coherence_probe.hip): anything else keeps two launches per
  // iteration from the start instead of finding out through expired waits (advisor, round 4)
  bool fuse_hw_ok = false;
  bool verbose = false;         // FPSQ_VERBOSE=1: one line on stderr when a call is repeated on two launches per iteration
  int64_t mmid_launches = 0;    // k_minres_mid launches of the current call
  int mmid_cap = 0;             // workgroups of k_minres_mid the device holds at once (occupancy x CUs): its grid must fit with a margin
  int64_t loop_launches = 0, loop_iters = 0;  // the Krylov loop(s) of the current call (fpsq_info.last_loop_*)
  int2* fz_dep = nullptr;                 // per row group: the A' blocks it waits for
  // halo-sharded handles (setup_fused_halo, at fpsq_comm_set_halo): the finish workgroups a row group waits for, the A' blocks
  // that deposit the raw sums of the two overlap regions; what the set-up needs again then (block boundaries, column ranges)
  int2* fz_dep2 = nullptr;
  int2 fz_depL{1, 0}, fz_depR{1, 0};
  // several iterations per launch (k_iter_multi, fpsq_multi.hip.h; FPSQ_MULTI_ITER=k: at most k per launch, 1: off)
  // Default 1 = off.  Measured at the headline size (profiles/r05_multi_iter.txt): bitwise the one-launch iterations, 9 launches per
  // evaluation instead of 21, and NO gain -- 0.99-1.00 x: the kernel boundary it removes (2.6 us per iteration) is paid back inside the
  // launch (agent-scope gathers of the short pair, the second long pair, tagged publications: +2.8 us per iteration)
  int multi_max = 1;
  // real workgroups of the LSQR / CRAIG updates in a multi launch (FPSQ_MULTI_UPD=t,a; multiples of 8).  Default: one per
  // segment workgroup.  Fewer, each walking several -- so that the next iteration's A' blocks are dispatched sooner -- was measured
  // SLOWER at the headline size (64 / 192: 854 evals/s, 128 / 384: 926, all: 940 against 952 with one iteration per launch: the long
  // update then cannot keep up and the next mid leaders wait for it)
  int multi_upd_t = 1 << 20, multi_upd_a = 1 << 20;
  // FPSQ_MULTI_DEFER_LONG=1: CRAIG's long update one iteration later, behind the NEXT iteration's A' blocks, so that only the small
  // m-vector updates stand between an iteration's row groups and the next A' blocks in the dispatch order.  Measured SLOWER (909
  // against 949 evals/s at 8 iterations per launch: the long update then competes with the A' phase and holds the mid leaders up)
  bool multi_defer_long = false;
  bool multi_ok = false;
  int2* mz_bdep = nullptr;
  unsigned int* mz_flag2 = nullptr;          // second parity of fz_flag / fz_ptag
  unsigned long long* mz_ptag2 = nullptr;
  unsigned int* mz_gflag[2] = {nullptr, nullptr};
  unsigned long long* mz_atag[2] = {nullptr, nullptr};
  unsigned long long* mz_utag[2] = {nullptr, nullptr};
  unsigned long long *mz_rec_h = nullptr, *mz_rec_m = nullptr, *mz_srec = nullptr, *mz_hdone = nullptr;
  double* LP2 = nullptr;                      // the second long pair
  int64_t multi_launches = 0, multi_iters = 0;
  bool fuse_halo_ok = false;
  bool fuse_halo_on = true;               // FPSQ_FUSE_HALO=0: a handle with shared rows keeps the halo launch between two product launches
  std::vector<int32_t> fz_rb;
  std::vector<int2> fz_colrange;
  unsigned int* fz_flag = nullptr;        // per A' block: launch number of its last completion
  unsigned long long* fz_ptag = nullptr;  // per A' block: four tagged words (its squared-norm partials)
  unsigned long long* ride_rec2 = nullptr;  // the mid leaders' record
  void* state3[3] = {nullptr, nullptr, nullptr};  // third copies of the LSQR (x 2) / CRAIG / LNLQ states: lsqr, craig, lnlq
  int64_t fused_launches = 0, fused_total = 0;
  // developer probe (FPSQ_FUSE_PROBE=<file>, FPSQ_FUSE_PROBE_AT=<n-th fused launch of the handle>): per-workgroup time stamps of one launch
  int64_t fuse_probe_at = 0;
  bool fuse_tail = true;               // FPSQ_FUSE_TAIL=0: the raw A'[q1, c] product and k_qp_penalty_grad as two launches (one GPU; bitwise the same)
  const GradEpi* tail_grad = nullptr;  // set around the tail's product launch: launch_spmv then picks k_spmv<.., GRAD>
  bool tail_grad_used = false;         // ... and says so (a layout it has no GRAD variant for: the caller launches k_qp_penalty_grad)
  unsigned long long* fuse_probe_buf = nullptr;
  int fuse_probe_grid = 0;
  std::vector<int> fuse_probe_layout;
  std::string fuse_probe_path;
  bool at_xcd = true;           // k_spmv_atl: every XCD walks a contiguous eighth of the row blocks (FPSQ_AT_XCD=0: grid order)
  MinresState* minres;
  LnlqState* lnlq;
  LnlqState* lnlq_alt;          // (second copy, see lsqr_alt)
  MinresState* minres_alt;
  LaneCtl* ctl_tmp;
  LaneCtl* ctl_raw;             // constant {ca = 1, cb = 0, done = 0}: raw partial products before an all-reduce
  LaneCtl* ctl_pm;              // constant {1, -1}
  LaneCtl* ctl_mp;              // constant {-1, 1}
  Comm* comm = nullptr;         // null: single GPU
  // Halo mode of the sharded handle (fpsq_comm_set_halo): n is the length of this rank's COLUMN WINDOW; its first
  // `ovl` entries are shared with rank - 1, its last `ovr` with rank + 1; sums over n-vectors run over the owned prefix
  // [0, n - ovr) and are all-reduced like the sums over the (row-sharded) m-vectors.
  bool halo = false;
  int64_t ovl = 0, ovr = 0;
  double* halo_recv = nullptr;  // 2 x [(ovl + ovr)][2]: the neighbours' raw sums on the two overlap regions; consecutive
                                // exchanges alternate between the two halves (a neighbour that is one exchange ahead --
                                // the epilogue runs several without a reduction in between -- never overwrites a record
                                // this rank has not consumed yet)
  uint64_t halo_calls = 0;
  double* halo_raw = nullptr;   // [(ovl + ovr)][2]: this rank's raw sums there (k_spmv<.., HALO>), head region first
  int halo_gf = 0;              // workgroups of k_halo_finish (0: no overlap at all)
  // Halo mode keeps every partial-sum array of the Krylov loop in ONE per-rank segment `seg`, laid out
  //   [E0 | E1 | M0 | M1 | T0 | T1 | V0 | V1 | A0 | A1 | W0 | W1 | E3]
  //   (pE, pE2, pEm[0..1], pS lanes, pWalt[0..1], pS2 lanes, pW[0..1], pE3: the steps behind an A product read the A partials
  //   and ONE half of the update partials -- with a half on either side of A both ranges are contiguous)
  // with counts cE / cW / cT / cA padded to the maxima over the ranks (zeros beyond a rank's own count -- every array is
  // always written with the same local count, so the padding stays zero): the arrays a
  // scalar step reads are then one contiguous range, which is all-gathered into `gath` ([nranks][range]) right before
  // the step; the step kernel sums the ranks' copies itself (StepArgs::nseg).
  double* seg = nullptr;
  double* gath = nullptr;
  int64_t seg_len = 0;
  int cE = 0, cT = 0, cA = 0, cW = 0;
  bool gather_ready = false;
  uint32_t xch_seq = 0;         // sequence number of the last in-launch sum over the ranks (xch_sum; the same on every rank)
  uint32_t last_xseq = 0;       // what prepare_step gave the pair it has just prepared (0: no exchange): travels NEXT to the steps --
  uint32_t ride_xseq = 0;       // k_step's arguments, the RideArgs of the launch whose leaders compute them (pre_args sets ride_xseq)
  uint64_t gather_calls = 0;    // the all-gathers alternate between the two halves of `gath`: a peer that is one reduction
                                // ahead never overwrites a record its neighbour has not read yet
  double* comm_vec = nullptr;   // [n][2] all-reduce payload (partial A' products)
  double* comm_scal = nullptr;  // 8 doubles: scalar all-reduce payload
  double* dscal;               // small device scalar scratch
  Progress* prog_host = nullptr;  // host-mapped
  Progress* prog_dev = nullptr;
  fpsq_stats* hstats = nullptr;   // host-mapped: written by the step kernel that ends a recurrence
  fpsq_stats* hstats_dev = nullptr;
  double* hscal = nullptr;        // host-mapped: scalar results (phi, f, c'c) written by the kernel that computes them
  double* hscal_dev = nullptr;
  // MINRES on K itself (kkt_method = FPSQ_KKT_MINRES_K): allocated at the first call
  bool mk_ready = false;
  MkVecs mk_long{}, mk_short{};
  MinresState* mk_state = nullptr;  // [2]
  double* mk_part[2] = {nullptr, nullptr};
  int mk_gl = 0, mk_gs = 0;
  // Speculative epilogue (run_krylov): kernels launched while gate0 is set only act once BOTH lane controls say `done`.
  const LaneCtl* gate0 = nullptr;
  const LaneCtl* gate1 = nullptr;
  bool tail_was_run = false;   // the caller's epilogue was enqueued (gated) inside run_krylov and the gates were open
  int64_t expect_iters[5][5][2] = {};  // [kind of lane 0][kind of lane NL-1]: iterations the last two such runs needed
  bool adaptive_runahead = true;    // FPSQ_ADAPTIVE_RUNAHEAD=0 disables (A/B)
  // FPSQ_HOST_TRACE=1: host timestamps at fixed points of fpsq_qp_objgrad, averaged and printed at destroy (developer aid)
  bool host_trace = false;
  double ht_sum[12] = {};
  int64_t ht_calls = 0;
  std::chrono::steady_clock::time_point ht_last, ht_exit;
  bool ht_have_exit = false;
  // start-up launch of the next run also evaluates the eq-QP gradient (qp_objgrad's fast start): nblk > 0
  QpGradArgs startup_qg{};
  // The final LSQR x update may be left to the caller's epilogue when its FIRST kernel is k_ys (absorb_flush, set by
  // qp_objgrad): run_krylov then parks the segment here instead of launching it.
  bool absorb_flush = false;
  UpdSeg pending_flush{};
  // stream-ordered outputs (fpsq_set_output_ordering)
  bool out_ordered = false;
  hipEvent_t ev_out = nullptr;
  double call_seq = 0.0;            // sequence number the phi reduction stores behind its results (hscal[3])
  // FPSQ_AB_MASK (developer A/B, tools/ab_modes.py): 1 = gradient kernel not merged into the start-up launch, 2 = final
  // LSQR update not absorbed by k_ys, 4 = phi reduced by a launch of its own right behind k_ys (default: the gradient
  // kernel's extra workgroup), 8 = no stream-ordered return
  int ab_mask = 0;
  bool ab_dynamic = false;          // FPSQ_AB_DYNAMIC=1: the mask is re-read from the environment at every qp_objgrad call
  int64_t force_expect = -1;        // fpsq_debug_expect_iterations: overrides the expected count of the next run (test hook)

  // instrumentation
  bool profile = false;
  std::vector<EventPair> ev_pool;
  size_t ev_used = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::chrono::steady_clock::time_point t_call;
  fpsq_info info{};
  int64_t launches = 0, spmv_launches = 0;
  int64_t prod_a[2] = {0, 0}, prod_at[2] = {0, 0};
};

struct fpsq_qp_s {
  fpsq_handle h;
  double *q, *d, *b;
};

namespace {

#define HIPCHK(h, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return FPSQ_ERR_HIP;                                                                       \
    }                                                                                            \
  } while (0)

template <class T>
int dalloc(fpsq_handle h, T** p, size_t count) {
  void* q = nullptr;
  HIPCHK(h, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// a buffer the peers of a sharded handle may write into: allocated the way the communicator needs it (Comm::alloc_exchange)
template <class T>
int xalloc(fpsq_handle h, T** p, size_t count) {
  void* q = nullptr;
  HIPCHK(h, h->comm->alloc_exchange(&q, std::max<size_t>(count, 1) * sizeof(T)));
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// release one dalloc'ed buffer before the handle dies
template <class T>
void dfree(fpsq_handle h, T** p) {
  auto it = std::find(h->allocs.begin(), h->allocs.end(), (void*)*p);
  if (it != h->allocs.end()) h->allocs.erase(it);
  hipFree(*p);
  *p = nullptr;
}

inline int ew_grid(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, kEwBlocksMax));
}

// ------------------------------------------------------------------ host-side sparse set-up

struct HostCsr {
  int64_t nrows, ncols;
  std::vector<int32_t> rowptr, colind;
};

// align > 1 (the A' blocks of a handle whose iterations run as one launch, k_iter_fused): a block that holds at least `align` rows
// ends on a multiple of `align` rows -- with 16-byte rows of the long pair and align = 8 every 128-byte line of the product's
// output then belongs to ONE block.  (Blocks of fewer rows -- very long rows -- stay as they are: rowblocks_aligned() says so.)
std::vector<int32_t> make_rowblocks(const std::vector<int32_t>& rowptr, int64_t nrows, int align = 1) {
  std::vector<int32_t> rb;
  rb.push_back(0);
  int64_t r = 0;
  while (r < nrows) {
    int64_t r1 = r;
    int64_t nz = 0;
    while (r1 < nrows && (r1 - r) < kMaxRowsPerBlk) {
      const int64_t len = rowptr[r1 + 1] - rowptr[r1];
      if (nz + len > kSpmvNnz) break;
      nz += len;
      ++r1;
    }
    if (r1 == r) r1 = r + 1;  // a single row longer than kSpmvNnz gets a block of its own
    else if (align > 1 && r1 < nrows && r1 - r >= align) r1 = r + (r1 - r) / align * align;
    rb.push_back((int32_t)r1);
    r = r1;
  }
  return rb;
}

bool rowblocks_aligned(const std::vector<int32_t>& rb, int align) {
  for (size_t i = 0; i + 1 < rb.size(); ++i)
    if (rb[i] % align) return false;
  return true;
}

// transpose structure: returns CSR of A' and perm with AT slot t <- A slot perm[t]
void transpose_structure(const HostCsr& A, HostCsr& T, std::vector<int32_t>& perm) {
  const int64_t nnz = (int64_t)A.colind.size();
  T.nrows = A.ncols;
  T.ncols = A.nrows;
  T.rowptr.assign(T.nrows + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) T.rowptr[A.colind[k] + 1]++;
  for (int64_t j = 0; j < T.nrows; ++j) T.rowptr[j + 1] += T.rowptr[j];
  T.colind.resize(nnz);
  perm.resize(nnz);
  std::vector<int32_t> next(T.rowptr.begin(), T.rowptr.end() - 1);
  for (int64_t i = 0; i < A.nrows; ++i)
    for (int32_t k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k) {
      const int32_t t = next[A.colind[k]]++;
      T.colind[t] = (int32_t)i;
      perm[t] = k;
    }
}

int upload_csr(fpsq_handle h, const HostCsr& H, DevCsr& D) {
  D.nrows = H.nrows;
  D.ncols = H.ncols;
  D.nnz = (int64_t)H.colind.size();
  D.nstore = D.nnz;
  std::vector<int32_t> rb = make_rowblocks(H.rowptr, H.nrows, D.row_align);
  D.nblk = (int32_t)rb.size() - 1;
  if (int rc = dalloc(h, &D.rowptr, H.rowptr.size())) return rc;
  // one padding entry (column 0, value 0): the product kernels read index `s` of an empty row block unconditionally
  if (int rc = dalloc(h, &D.colind, H.colind.size() + 1)) return rc;
  if (int rc = dalloc(h, &D.vals, H.colind.size() + 1)) return rc;
  HIPCHK(h, hipMemset(D.colind + H.colind.size(), 0, 4));
  HIPCHK(h, hipMemset(D.vals + H.colind.size(), 0, 8));
  if (int rc = dalloc(h, &D.rowblk, rb.size())) return rc;
  HIPCHK(h, hipMemcpy(D.rowptr, H.rowptr.data(), H.rowptr.size() * 4, hipMemcpyHostToDevice));
  if (!H.colind.empty())
    HIPCHK(h, hipMemcpy(D.colind, H.colind.data(), H.colind.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(D.rowblk, rb.data(), rb.size() * 4, hipMemcpyHostToDevice));
  {
    std::vector<int4> bd(std::max(D.nblk, 1));
    for (int b = 0; b < D.nblk; ++b) bd[b] = int4{rb[b], rb[b + 1] - rb[b], H.rowptr[rb[b]], H.rowptr[rb[b + 1]]};
    if (int rc = dalloc(h, &D.blkdesc, bd.size())) return rc;
    HIPCHK(h, hipMemcpy(D.blkdesc, bd.data(), bd.size() * sizeof(int4), hipMemcpyHostToDevice));
  }
  // 16-bit block-relative columns when every row block spans < 65536 columns
  if (h->opt.jac_format != 1 && D.nnz > 0) {
    std::vector<int32_t> base(D.nblk, 0);
    std::vector<uint16_t> c16(D.nnz + 1, 0);
    bool ok = true;
    int span = 0;
    for (int b = 0; b < D.nblk && ok; ++b) {
      const int s = H.rowptr[rb[b]], e = H.rowptr[rb[b + 1]];
      int lo = INT32_MAX, hi = -1;
      for (int k = s; k < e; ++k) {
        lo = std::min(lo, H.colind[k]);
        hi = std::max(hi, H.colind[k]);
      }
      if (e == s) lo = hi = 0;
      if (hi - lo > 65535) ok = false;
      span = std::max(span, hi - lo + 1);
      base[b] = lo;
      for (int k = s; k < e && ok; ++k) c16[k] = (uint16_t)(H.colind[k] - lo);
    }
    if (ok) {
      if (int rc = dalloc(h, &D.col16, c16.size())) return rc;
      if (int rc = dalloc(h, &D.colbase, base.size())) return rc;
      HIPCHK(h, hipMemcpy(D.col16, c16.data(), c16.size() * 2, hipMemcpyHostToDevice));
      HIPCHK(h, hipMemcpy(D.colbase, base.data(), base.size() * 4, hipMemcpyHostToDevice));
      D.win = span;
    }
  }
  return 0;
}

// Re-store an uploaded CSR in the padded block layout of k_spmv<.., PAD>.  `perm` (value source of every compact entry)
// is rewritten to the padded numbering with -1 in the padding slots.  No-op when some block is one long row.
// csr_pos (optional, A' only): for every CSR slot of A its position in the padded row-group copy of A (build_rgcs), and
// zero_pos, a position of that array that always holds 0.0.  When given -- and the blocks qualify for the column-sorted
// layout -- the blocks are stored WITHOUT VALUES: an A' block's entries come from the ~10 row groups whose column windows
// reach its columns, and inside a group (column-sorted) they are a contiguous run; the block's entries are therefore
// stored in the order of their POSITIONS in the row-group array, 64 consecutive entries (one wave instruction) read at most
// four runs, and one 16-byte descriptor per such segment says where: four 24-bit bases relative to the block's base
// (blkdesc.w), three split lanes, the number of valid lanes.  The index planes keep the column-sorted format (slot in the
// block's row-major order | column relative to colbase), so the products land in the same LDS slots and are summed in the
// same order: BITWISE the other layouts.  What it buys: the Krylov loop streams ONE copy of the values (80 MB less
// working set next to a 256 MB Infinity Cache: measured as a what-if in round 3, +3.9 % evaluations/s at the headline
// size), a Jacobian refresh writes one array instead of two, 82 MB less memory.  The value addresses need the descriptors
// first -- but so do the gathers of x need the index words, and values and gathers then travel in the same round trip: the
// workgroup's chain of dependent memory round trips is no longer.
int pad_blocks(fpsq_handle h, const HostCsr& H, std::vector<int32_t>& perm, DevCsr& D,
               const std::vector<int32_t>* csr_pos = nullptr, int64_t zero_pos = 0, const double* ext_vals = nullptr) {
  if (D.nnz == 0 || h->opt.jac_format == 1) return 0;
  std::vector<int32_t> rb = make_rowblocks(H.rowptr, H.nrows, D.row_align);
  const int nblk = (int)rb.size() - 1;
  for (int b = 0; b < nblk; ++b)
    if (H.rowptr[rb[b + 1]] - H.rowptr[rb[b]] > kSpmvNnz) return 0;
  const size_t slots = (size_t)nblk * kSpmvNnz;
  if (slots >= (size_t)INT32_MAX) return 0;
  std::vector<int32_t> pperm(slots, -1), pcol;
  std::vector<uint16_t> pc16;
  std::vector<uint8_t> pc8;
  std::vector<int32_t> base, ord;
  const bool idx16 = D.col16 != nullptr;
  // column-sorted blocks (k_spmv<.., CSORT>): 13 bits of block-relative column next to the 11-bit slot
  const bool sorted = idx16 && D.win <= 8192 && h->at_sorted;
  static_assert(kSpmvNnz <= 2048, "slot field of the column-sorted layout is 11 bits");
  if (sorted) {
    pc16.resize(slots);
    for (size_t q = 0; q < slots; ++q) {  // padding: an unused slot (its own sorted position), column 0, value 0
      const size_t t = q % kSpmvNnz;
      pc16[q - t + 8 * ((t % 512) / 2) + 2 * (t / 512) + (t & 1)] = (uint16_t)t;
    }
    pc8.assign(slots, 0);
    base.resize(nblk);
    HIPCHK(h, hipMemcpy(base.data(), D.colbase, (size_t)nblk * 4, hipMemcpyDeviceToHost));
  } else if (idx16) {
    pc16.assign(slots, 0);
    base.resize(nblk);
    HIPCHK(h, hipMemcpy(base.data(), D.colbase, (size_t)nblk * 4, hipMemcpyDeviceToHost));
  } else {
    pcol.assign(slots, 0);
  }
  // ---- shared values.  A block whose entries cannot be described that way (a 64-entry segment touching more than four
  // runs: the first blocks of the headline generators, where the clamped windows of the top rows pile several groups' last
  // few columns into one block) keeps 2048 values of its OWN in a small side array (D.vals, refreshed like before); its
  // blkdesc.w = -128 - (its index there) tells the kernel.  More than a quarter of the blocks like that: not worth it.
  bool shared = sorted && csr_pos != nullptr && h->at_shared && zero_pos < (int64_t)INT32_MAX;
  std::vector<uint4> segd;
  std::vector<int32_t> vbase, own_perm;
  std::vector<uint16_t> sc16;
  std::vector<uint8_t> sc8;
  int nown = 0;
  if (shared) {
    segd.assign((size_t)nblk * 32, uint4{0u, 0u, 0u, 0u});
    vbase.assign(nblk, 0);
    sc16.resize(slots);
    sc8.assign(slots, 0);
    std::vector<int64_t> pos;
    std::vector<uint4> sd(32);
    for (int b = 0; b < nblk; ++b) {
      const int s = H.rowptr[rb[b]], e = H.rowptr[rb[b + 1]], cnt = e - s;
      ord.resize(cnt);
      pos.resize(cnt);
      for (int k = 0; k < cnt; ++k) {
        ord[k] = k;
        pos[k] = (*csr_pos)[perm[s + k]];
      }
      std::sort(ord.begin(), ord.end(), [&](int a, int c) { return pos[a] < pos[c]; });
      const int64_t base64 = (cnt ? pos[ord[0]] : 0) - 64;
      bool ok = true;
      for (int sg = 0; sg < 32 && ok; ++sg) {
        const int lo = sg * 64, nvalid = std::max(0, std::min(64, cnt - lo));
        int64_t vb[4] = {0, 0, 0, 0};
        int split[3] = {64, 64, 64};
        int np = 0;
        for (int l = 0; l < nvalid && ok; ++l) {
          const int64_t p = pos[ord[lo + l]];
          if (l == 0 || p != pos[ord[lo + l - 1]] + 1) {  // a new run starts at lane l
            if (np == 4) {
              ok = false;
              break;
            }
            if (np > 0) split[np - 1] = l;
            vb[np++] = p - l - base64;
          }
        }
        for (int i = 0; i < 4; ++i)
          if (vb[i] < 0 || vb[i] >= (1ll << 24)) ok = false;
        const uint64_t a0 = (uint64_t)vb[0] | ((uint64_t)vb[1] << 24) | ((uint64_t)vb[2] << 48);
        uint4 d;
        d.x = (uint32_t)a0;
        d.y = (uint32_t)(a0 >> 32);
        d.z = (uint32_t)(((uint64_t)vb[2] >> 16) | ((uint64_t)vb[3] << 8));
        d.w = (uint32_t)split[0] | ((uint32_t)split[1] << 7) | ((uint32_t)split[2] << 14) | ((uint32_t)nvalid << 21);
        sd[(sg % 4) * 8 + sg / 4] = d;  // (stored per wave: segment 4 j + w at [8 w + j], a wave's eight in one 128-byte line)
      }
      if (ok) {
        vbase[b] = (int32_t)base64;
        for (int sg = 0; sg < 32; ++sg) segd[(size_t)b * 32 + sg] = sd[sg];
      } else {  // its own values, in column-sorted order
        vbase[b] = -128 - nown;
        std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return H.colind[s + a] < H.colind[s + c]; });
        own_perm.resize((size_t)(nown + 1) * kSpmvNnz, -1);
        for (int t = 0; t < cnt; ++t) own_perm[(size_t)nown * kSpmvNnz + t] = perm[s + ord[t]];
        ++nown;
      }
      for (int t = 0; t < kSpmvNnz; ++t) {  // entry t of the stored order belongs to thread t % 256, its j-th word (j = t / 256)
        const size_t qi = (size_t)b * kSpmvNnz + 8 * (t % kBlock) + t / kBlock;
        if (t < cnt) {
          const int k = ord[t], col = H.colind[s + k] - base[b];
          sc16[qi] = (uint16_t)(k | ((col & 31) << 11));
          sc8[qi] = (uint8_t)(col >> 5);
        } else {
          sc16[qi] = (uint16_t)t;  // an unused slot of the product buffer; its value is 0 (zero_pos / the side array's padding)
        }
      }
    }
    if (std::getenv("FPSQ_VERBOSE")) std::fprintf(stderr, "fpsq: shared A' values: %d of %d blocks keep their own\n", nown, nblk);
    if (nown > nblk / 4) shared = false;
  }
  if (shared) {
    dfree(h, &D.vals);
    dfree(h, &D.col16);
    dfree(h, &D.colind);
    const size_t nown_slots = (size_t)nown * kSpmvNnz;
    if (int rc = dalloc(h, &D.vals, nown_slots + 1)) return rc;
    HIPCHK(h, hipMemset(D.vals, 0, (nown_slots + 1) * 8));
    if (int rc = dalloc(h, &D.cs16, slots)) return rc;
    if (int rc = dalloc(h, &D.cs8, slots)) return rc;
    if (int rc = dalloc(h, &D.segdesc, segd.size())) return rc;
    HIPCHK(h, hipMemcpy(D.cs16, sc16.data(), slots * 2, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(D.cs8, sc8.data(), slots, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(D.segdesc, segd.data(), segd.size() * sizeof(uint4), hipMemcpyHostToDevice));
    std::vector<int4> bd(nblk);
    for (int b = 0; b < nblk; ++b) bd[b] = int4{rb[b], rb[b + 1] - rb[b], H.rowptr[rb[b]], vbase[b]};
    HIPCHK(h, hipMemcpy(D.blkdesc, bd.data(), bd.size() * sizeof(int4), hipMemcpyHostToDevice));
    D.sorted = true;
    D.shared = true;
    D.vals_ext = ext_vals;
    D.zero_pos = zero_pos;
    D.padded = true;
    D.nstore = (int64_t)nown_slots;  // (what a refresh still has to fill: the side array)
    perm.swap(own_perm);
    return 0;
  }
  for (int b = 0; b < nblk; ++b) {
    const int s = H.rowptr[rb[b]], e = H.rowptr[rb[b + 1]];
    if (sorted) {
      ord.resize(e - s);
      for (int k = s; k < e; ++k) ord[k - s] = k;
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return H.colind[a] < H.colind[c]; });
      for (int t = 0; t < e - s; ++t) {
        const int k = ord[t], col = H.colind[k] - base[b];
        const size_t q = (size_t)b * kSpmvNnz + t;
        pperm[q] = perm[k];
        // (index planes: the eight entries of a thread contiguously, see csort_fetch)
        const size_t qi = (size_t)b * kSpmvNnz + 8 * ((t % 512) / 2) + 2 * (t / 512) + (t & 1);
        pc16[qi] = (uint16_t)((k - s) | ((col & 31) << 11));
        pc8[qi] = (uint8_t)(col >> 5);
      }
      continue;
    }
    for (int k = s; k < e; ++k) {
      const size_t q = (size_t)b * kSpmvNnz + (k - s);
      pperm[q] = perm[k];
      if (idx16) pc16[q] = (uint16_t)(H.colind[k] - base[b]);
      else pcol[q] = H.colind[k];
    }
  }
  dfree(h, &D.vals);
  if (int rc = dalloc(h, &D.vals, slots)) return rc;
  HIPCHK(h, hipMemset(D.vals, 0, slots * 8));
  if (sorted) {
    dfree(h, &D.col16);
    dfree(h, &D.colind);
    if (int rc = dalloc(h, &D.cs16, slots)) return rc;
    if (int rc = dalloc(h, &D.cs8, slots)) return rc;
    HIPCHK(h, hipMemcpy(D.cs16, pc16.data(), slots * 2, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(D.cs8, pc8.data(), slots, hipMemcpyHostToDevice));
    D.sorted = true;
  } else if (idx16) {
    dfree(h, &D.col16);
    dfree(h, &D.colind);  // the 16-bit form is the only one the padded kernel reads
    if (int rc = dalloc(h, &D.col16, slots)) return rc;
    HIPCHK(h, hipMemcpy(D.col16, pc16.data(), slots * 2, hipMemcpyHostToDevice));
  } else {
    dfree(h, &D.colind);
    if (int rc = dalloc(h, &D.colind, slots)) return rc;
    HIPCHK(h, hipMemcpy(D.colind, pcol.data(), slots * 4, hipMemcpyHostToDevice));
  }
  D.padded = true;
  D.nstore = (int64_t)slots;
  perm.swap(pperm);
  return 0;
}

// Row-group column-sorted copy of A (see k_spmv_rgcs).  Not built (ok = false) when a group spans >= 2^21 columns.
int build_rgcs(fpsq_handle h, const HostCsr& H, DevRgcs& D, std::vector<int32_t>* csr_pos = nullptr,
               std::vector<int2>* col_range = nullptr) {
  const int64_t nnz = (int64_t)H.colind.size();
  D.ok = false;
  if (csr_pos) csr_pos->clear();
  if (nnz == 0 || h->opt.jac_format == 1) return 0;
  std::vector<RgcsGroup> groups;
  std::vector<uint32_t> pidx(nnz);
  std::vector<int32_t> vperm(nnz);
  std::vector<uint16_t> tptr;
  std::vector<int32_t> ord, lrow, cntr, nxt;
  // Nonzero budget of a group = a whole number of tiles (a workgroup pays the same latency for a partly filled
  // tile), chosen so that the groups fill the GPU's resident-workgroup slots (4 per CU at 32 KB of LDS) about once:
  // measured at the headline size, 1000 groups of 5 tiles run the product in ~30 us, 782 groups of 6.2 tiles in 37 us.
  int budget = kRgcsGroupNnz;
  {
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, h->opt.device) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    const int64_t slots = (int64_t)cus * 4;
    const double avg = (double)nnz / (double)std::max<int64_t>(H.nrows, 1);
    const int64_t kmax = std::max<int64_t>(1, (int64_t)(std::min<double>(kRgcsMaxRows * avg, kRgcsGroupNnz) / kRgcsTile));
    const int64_t k = std::min(kmax, std::max<int64_t>(1, (nnz + slots * kRgcsTile - 1) / (slots * kRgcsTile)));
    budget = (int)(k * kRgcsTile);
    if (const char* ev = std::getenv("FPSQ_RGCS_TILES"))  // tuning override: tiles per group
      budget = (int)(std::max<int64_t>(1, std::min<int64_t>(kmax, std::atoi(ev))) * kRgcsTile);
  }
  auto group_end = [&](int64_t r) {
    int64_t r1 = r, nz = 0;
    while (r1 < H.nrows && r1 - r < kRgcsMaxRows) {
      const int64_t len = H.rowptr[r1 + 1] - H.rowptr[r1];
      if (nz + len > budget && r1 > r) break;
      nz += len;
      ++r1;
    }
    return r1;
  };
  // ORDER OF THE ENTRIES INSIDE A GROUP: by column PHASE, (col mod P), P = the typical width of a group's column window.
  // A workgroup sweeps its window tile by tile while all the groups of an XCD are resident together.  Sorted by column proper,
  // group g reads column c when its sweep gets there -- (c - cmin_g) / width of the way through the launch -- and the ~9
  // neighbouring groups whose windows overlap in c (PDE-like rows: the window moves by a fraction of its width from group to
  // group) read it at nine different times, spread over the whole launch, while the matrix streams through the same L2:
  // the x window was fetched 2.6 times (profiles/r03_pmc_traffic.json: 1.15 x the product's algorithmic bytes).  Sorted by
  // phase every group is at the same ABSOLUTE columns at the same time -- a rotation of its column order, any order is valid
  // -- and the overlap is served by the L2.  Windows as wide as the matrix (random patterns): P covers it, plain column order.
  int64_t P = INT64_MAX;
  {
    std::vector<int64_t> widths;
    for (int64_t r = 0; r < H.nrows;) {
      const int64_t r1 = group_end(r);
      int64_t cmin = INT64_MAX, cmax = -1;
      for (int64_t k = H.rowptr[r]; k < H.rowptr[r1]; ++k) {
        cmin = std::min<int64_t>(cmin, H.colind[k]);
        cmax = std::max<int64_t>(cmax, H.colind[k]);
      }
      if (cmax >= cmin) widths.push_back(cmax - cmin + 1);
      r = r1;
    }
    if (!widths.empty()) {
      std::nth_element(widths.begin(), widths.begin() + widths.size() / 2, widths.end());
      const int64_t med = std::max<int64_t>(1, widths[widths.size() / 2]);
      // the period: the WIDEST of the typical windows (those within 1.5 x the median), so that (col mod P) is one-to-one on
      // every typical group's window -- a pure rotation of its column order.  (The median itself -- rounds 3 -- left half of the
      // groups a little wider than the period: the first and last few columns of such a window share phases and their
      // entries INTERLEAVE in the sorted order, which cuts the contiguous per-group runs the shared-value layout of A'
      // builds on into slivers.)
      P = med;
      for (const int64_t w : widths)
        if (w <= med + med / 2) P = std::max(P, w);
    }
    if (const char* ev = std::getenv("FPSQ_RGCS_PHASE"))  // 0: plain column order (A/B)
      if (std::atoi(ev) == 0) P = INT64_MAX;
  }
  int64_t r = 0;
  while (r < H.nrows) {
    const int64_t r1 = group_end(r);
    const int R = (int)(r1 - r);
    const int e0 = H.rowptr[r], e1 = H.rowptr[r1], cnt = e1 - e0;
    int cmin = INT32_MAX, cmax = -1;
    for (int k = e0; k < e1; ++k) {
      cmin = std::min(cmin, H.colind[k]);
      cmax = std::max(cmax, H.colind[k]);
    }
    if (cnt == 0) cmin = cmax = 0;
    if ((int64_t)cmax - cmin >= (1ll << kRgcsColBits)) return 0;  // not representable: keep CSR-stream
    lrow.resize(cnt);
    for (int rr = 0; rr < R; ++rr)
      for (int k = H.rowptr[r + rr]; k < H.rowptr[r + rr + 1]; ++k) lrow[k - e0] = rr;
    ord.resize(cnt);
    for (int k = 0; k < cnt; ++k) ord[k] = k;
    // (a group much wider than the typical window would interleave several column ranges in one tile: plain order for it)
    const int64_t Pg = (int64_t)cmax - cmin + 1 > P + P / 2 ? INT64_MAX : P;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
      const int64_t ca = H.colind[e0 + a], cb = H.colind[e0 + b];
      const int64_t pa = ca % Pg, pb = cb % Pg;
      return pa != pb ? pa < pb : ca < cb;
    });
    const int ntile = (cnt + kRgcsTile - 1) / kRgcsTile;
    const int32_t tp_start = (int32_t)tptr.size();
    for (int t = 0; t < ntile; ++t) {
      const int a = t * kRgcsTile, b = std::min(cnt, a + kRgcsTile);
      cntr.assign(R + 1, 0);
      for (int k = a; k < b; ++k) cntr[lrow[ord[k]] + 1]++;
      for (int rr = 0; rr < R; ++rr) cntr[rr + 1] += cntr[rr];
      for (int rr = 0; rr <= R; ++rr) tptr.push_back((uint16_t)cntr[rr]);
      nxt.assign(cntr.begin(), cntr.end() - 1);
      for (int k = a; k < b; ++k) {
        const int src = ord[k];
        const int slot = nxt[lrow[src]]++;
        pidx[e0 + k] = ((uint32_t)slot << kRgcsColBits) | (uint32_t)(H.colind[e0 + src] - cmin);
        vperm[e0 + k] = e0 + src;
      }
    }
    if (col_range) col_range->push_back(make_int2(cmin, cmax));
    RgcsGroup gd{};
    gd.r0 = (int32_t)r;
    gd.R = R;
    gd.e0 = e0;
    gd.e1 = e1;
    gd.cmin = cmin;
    gd.tp = tp_start;
    groups.push_back(gd);
    if (ntile == 0)
      for (int rr = 0; rr <= R; ++rr) tptr.push_back(0);
    r = r1;
  }
  tptr.push_back(0);
  tptr.push_back(0);  // the kernel reads two uint16 at once
  // Padded layout (k_spmv_rgcs<.., PAD>): group g at [g * budget, ...), zero entries up to the end of its last tile.
  bool padded = (int64_t)groups.size() * budget < (int64_t)INT32_MAX;
  for (const RgcsGroup& gd : groups) padded = padded && gd.e1 - gd.e0 <= budget;
  int64_t nstore = nnz;
  if (padded) {
    nstore = (int64_t)groups.size() * budget;
    std::vector<uint32_t> pp((size_t)nstore, 0u);
    std::vector<int32_t> vp((size_t)nstore, -1);
    for (size_t gi = 0; gi < groups.size(); ++gi) {
      RgcsGroup& gd = groups[gi];
      const int cnt = gd.e1 - gd.e0;
      const size_t dst = gi * (size_t)budget;
      for (int k = 0; k < cnt; ++k) {
        pp[dst + k] = pidx[gd.e0 + k];
        vp[dst + k] = vperm[gd.e0 + k];
      }
      const int full = (cnt + kRgcsTile - 1) / kRgcsTile * kRgcsTile;
      for (int k = cnt; k < full; ++k) pp[dst + k] = (uint32_t)(k % kRgcsTile) << kRgcsColBits;  // unused slot, value 0
    }
    pidx.swap(pp);
    vperm.swap(vp);
  }
  uint32_t* dp;
  RgcsGroup* dg;
  uint16_t* d5;
  if (int rc = dalloc(h, &dp, (size_t)nstore + 1)) return rc;
  if (int rc = dalloc(h, &D.vals, (size_t)nstore + 1)) return rc;
  if (int rc = dalloc(h, &D.vperm, (size_t)nstore)) return rc;
  if (int rc = dalloc(h, &dg, groups.size())) return rc;
  if (int rc = dalloc(h, &d5, tptr.size() + 2)) return rc;
  HIPCHK(h, hipMemcpy(dp, pidx.data(), (size_t)nstore * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemset(dp + nstore, 0, 4));
  HIPCHK(h, hipMemset(D.vals, 0, ((size_t)nstore + 1) * 8));
  HIPCHK(h, hipMemcpy(D.vperm, vperm.data(), (size_t)nstore * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(dg, groups.data(), groups.size() * sizeof(RgcsGroup), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(d5, tptr.data(), tptr.size() * 2, hipMemcpyHostToDevice));
  if (csr_pos && padded) {  // where every CSR slot of A sits in the (padded) row-group array: pad_blocks builds A' on it
    csr_pos->assign((size_t)nnz, -1);
    for (int64_t p = 0; p < nstore; ++p)
      if (vperm[p] >= 0) (*csr_pos)[vperm[p]] = (int32_t)p;
  }
  D.view = RgcsView{dp, D.vals, dg, d5, (int32_t)groups.size(), (int32_t)H.nrows, padded ? budget : 0};
  D.nstore = nstore;
  D.nnz = nnz;
  D.ok = true;
  return 0;
}

inline int npart_A(fpsq_handle h) { return h->RA.ok ? h->RA.view.ng : h->A.nblk; }
// Sums over the ranks need no launch of their own: one GPU; a communicator of ONE rank (nobody to add to); or the halo-sharded
// layout on a peer-to-peer route whose ranks form them inside the launches that need them (Comm::xch_table, known after arm()).
inline bool insum(fpsq_handle h) {
  return !h->comm || (h->halo && (h->comm->nranks == 1 || h->comm->xch_table() != nullptr));
}
// ... and when other ranks exist: the table the kernels are given (null: one GPU, or a communicator of one)
inline const XchTable* insum_table(fpsq_handle h) { return h->comm && h->comm->nranks > 1 ? h->comm->xch_table() : nullptr; }
inline int64_t n_owned(fpsq_handle h) { return h->halo ? h->n - h->ovr : h->n; }

int alloc_workspaces(fpsq_handle h) {
  const size_t n = (size_t)h->n, m = (size_t)h->m;
  if (int rc = dalloc(h, &h->LP, 2 * n)) return rc;
  if (int rc = dalloc(h, &h->SP, 2 * m)) return rc;
  if (int rc = dalloc(h, &h->SP2, 2 * m)) return rc;
  if (int rc = dalloc(h, &h->comm_vec, 2 * n)) return rc;
  double** nv[] = {&h->Cx, &h->Cw2, &h->in_n1, &h->in_n2, &h->p1, &h->p2b,
                   &h->gs, &h->gx, &h->jc, &h->g, &h->xin, &h->xk};
  for (auto p : nv)
    if (int rc = dalloc(h, p, n)) return rc;
  double** mv[] = {&h->Lw[0], &h->Lw[1], &h->Lx[0], &h->Lx[1], &h->Cw, &h->Cy, &h->in_m, &h->ys, &h->c,
                   &h->Mr[0], &h->Mr[1], &h->Mw[0], &h->Mw[1], &h->Mx};
  for (auto p : mv)
    if (int rc = dalloc(h, p, m)) return rc;
  h->npS = std::max(std::max(std::max(h->A.nblk, h->AT.nblk), kEwBlocksMax), npart_A(h));
  if (int rc = dalloc(h, &h->pS, (size_t)h->npS * 2)) return rc;
  if (int rc = dalloc(h, &h->pS2, (size_t)h->npS * 2)) return rc;
  if (int rc = dalloc(h, &h->pS2b, (size_t)h->npS * 2)) return rc;
  h->strT = h->AT.nblk;
  h->strA = npart_A(h);
  double** ev[] = {&h->pW[0], &h->pW[1], &h->pWalt[0], &h->pWalt[1], &h->pE, &h->pE2, &h->pE3, &h->pQ[0], &h->pQ[1], &h->pC[0], &h->pC[1],
                   &h->pEm[0], &h->pEm[1]};
  for (auto p : ev)
    if (int rc = dalloc(h, p, (size_t)kEwBlocksMax * 2)) return rc;
  if (!h->mm_ptag) {
    if (int rc = dalloc(h, &h->mm_ptag, (size_t)kEwBlocksMax * 2)) return rc;
    HIPCHK(h, hipMemset(h->mm_ptag, 0, (size_t)kEwBlocksMax * 16));
  }
  return 0;
}

// One launch per joint iteration (k_iter_fused) -- what it needs beyond the two products' layouts: every A' block boundary on a
// 128-byte line of the long pair, the main layouts of both products (column-sorted padded blocks, padded row groups), 32-bit
// byte offsets into the long pair, and per row group the range of A' blocks that own the lines it gathers from.
int setup_fused_iteration(fpsq_handle h, const HostCsr& HT, const std::vector<int2>& col_range) {
  h->fuse_ok = false;
  if (!h->fuse_iter || !h->fuse_hw_ok || !h->RA.ok || h->RA.view.stride == 0 || !h->AT.padded || !(h->AT.sorted || h->AT.col16) || h->AT.nblk < 1)
    return 0;
  if ((int64_t)h->n * 16 >= (int64_t)INT32_MAX || (int64_t)h->m * 16 >= (int64_t)INT32_MAX) return 0;
  if ((int)col_range.size() != h->RA.view.ng) return 0;
  const std::vector<int32_t> rb = make_rowblocks(HT.rowptr, HT.nrows, h->AT.row_align);
  if ((int)rb.size() - 1 != h->AT.nblk || !rowblocks_aligned(rb, 8)) return 0;
  std::vector<int2> dep(col_range.size());
  for (size_t g = 0; g < col_range.size(); ++g) {
    const int64_t lo = col_range[g].x & ~7, hi = std::min<int64_t>((int64_t)col_range[g].y | 7, h->n - 1);
    const int b0 = (int)(std::upper_bound(rb.begin(), rb.end(), (int32_t)lo) - rb.begin()) - 1;
    const int b1 = (int)(std::upper_bound(rb.begin(), rb.end(), (int32_t)hi) - rb.begin()) - 1;
    dep[g] = make_int2(std::max(b0, 0), std::min(std::max(b1, 0), h->AT.nblk - 1));
  }
  if (h->fuse_iter == 1) {
    // Where it pays (measured): a grid of several resident sets -- the row groups then enter as the last A'
    // blocks drain and find most of what they wait for done -- whose row groups depend on a small part of the A' blocks.
    // A grid that is resident at once gains nothing from sharing a launch and pays for the flags (cfg2, random columns: every
    // group waits for every block; 2500 -> 2230 evals/s).
    double width = 0.0;
    for (const int2& d : dep) width += d.y - d.x + 1;
    width /= (double)std::max<size_t>(dep.size(), 1);
    // (the size threshold: profiles/r04_fused_sizes.txt -- headline generator, one launch against two: -3.2 % at 1225 blocks,
    // -2.4 % at 1617, +2.3 % at 1764, +7.7 % at 1862, +8.4 % at 1960, +7.3 % at 2450, +2.5 to +4 % at 4900, -0.5 % at 9800)
    if (h->AT.nblk < 17 * h->resident_wgs / 10 || width > h->AT.nblk / 8.0) return 0;
  }
  dfree(h, &h->fz_dep);
  dfree(h, &h->fz_flag);
  dfree(h, &h->fz_ptag);
  if (int rc = dalloc(h, &h->fz_dep, dep.size())) return rc;
  // (+ kEwBlocksMax entries: the finish workgroups of a halo-sharded handle publish themselves behind the blocks)
  if (int rc = dalloc(h, &h->fz_flag, (size_t)h->AT.nblk + kEwBlocksMax)) return rc;
  if (int rc = dalloc(h, &h->fz_ptag, ((size_t)h->AT.nblk + kEwBlocksMax) * 4)) return rc;
  HIPCHK(h, hipMemcpy(h->fz_dep, dep.data(), dep.size() * sizeof(int2), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemset(h->fz_flag, 0, ((size_t)h->AT.nblk + kEwBlocksMax) * 4));
  HIPCHK(h, hipMemset(h->fz_ptag, 0, ((size_t)h->AT.nblk + kEwBlocksMax) * 32));
  h->fz_rb = rb;
  h->fz_colrange = col_range;
  h->fuse_ok = true;
  // ---- several iterations per launch (fpsq_multi.hip.h): per A' block the row groups whose rows of the short pair it gathers
  // (the mirror image of `dep`; a cover by ONE range -- waiting for more is safe; a block nobody gathers from -- empty columns --
  // waits for every group: its own previous incarnation is then complete too), second copies of the flags and tagged words,
  // records, the second long pair
  h->multi_ok = false;
  if (h->multi_max > 1) {
    const int ng = h->RA.view.ng, nb = h->AT.nblk;
    std::vector<int2> bdep((size_t)nb, make_int2(INT32_MAX, -1));
    for (int g = 0; g < ng; ++g)
      for (int L = dep[g].x; L <= dep[g].y; ++L) {
        bdep[L].x = std::min(bdep[L].x, g);
        bdep[L].y = std::max(bdep[L].y, g);
      }
    for (int L = 0; L < nb; ++L)
      if (bdep[L].y < bdep[L].x) bdep[L] = make_int2(0, ng - 1);
    dfree(h, &h->mz_bdep);
    dfree(h, &h->mz_flag2);
    dfree(h, &h->mz_ptag2);
    if (int rc = dalloc(h, &h->mz_bdep, bdep.size())) return rc;
    HIPCHK(h, hipMemcpy(h->mz_bdep, bdep.data(), bdep.size() * sizeof(int2), hipMemcpyHostToDevice));
    if (int rc = dalloc(h, &h->mz_flag2, (size_t)nb + kEwBlocksMax)) return rc;
    if (int rc = dalloc(h, &h->mz_ptag2, ((size_t)nb + kEwBlocksMax) * 4)) return rc;
    HIPCHK(h, hipMemset(h->mz_flag2, 0, ((size_t)nb + kEwBlocksMax) * 4));
    HIPCHK(h, hipMemset(h->mz_ptag2, 0, ((size_t)nb + kEwBlocksMax) * 32));
    for (int q = 0; q < 2; ++q) {
      dfree(h, &h->mz_gflag[q]);
      dfree(h, &h->mz_atag[q]);
      if (int rc = dalloc(h, &h->mz_gflag[q], (size_t)ng)) return rc;
      if (int rc = dalloc(h, &h->mz_atag[q], (size_t)ng * 4)) return rc;
      HIPCHK(h, hipMemset(h->mz_gflag[q], 0, (size_t)ng * 4));
      HIPCHK(h, hipMemset(h->mz_atag[q], 0, (size_t)ng * 32));
      if (!h->mz_utag[q]) {
        if (int rc = dalloc(h, &h->mz_utag[q], (size_t)4 * kEwBlocksMax * 2)) return rc;
        HIPCHK(h, hipMemset(h->mz_utag[q], 0, (size_t)4 * kEwBlocksMax * 16));
      }
    }
    if (!h->mz_rec_h) {
      const size_t words = (size_t)2 * kRecRing * 512 + (size_t)kRecRing * 2 * kSrecSlot + 8;
      if (int rc = dalloc(h, &h->mz_rec_h, words)) return rc;
      HIPCHK(h, hipMemset(h->mz_rec_h, 0, words * 8));
      h->mz_rec_m = h->mz_rec_h + (size_t)kRecRing * 512;
      h->mz_srec = h->mz_rec_m + (size_t)kRecRing * 512;
      h->mz_hdone = h->mz_srec + (size_t)kRecRing * 2 * kSrecSlot;
    }
    if (!h->LP2)
      if (int rc = dalloc(h, &h->LP2, 2 * (size_t)h->n)) return rc;
    h->multi_ok = true;
  }
  return 0;
}

// The one-launch iteration of a halo-sharded handle (fpsq_comm_set_halo, or a new structure on such a handle): which A' blocks
// deposit the raw sums of the two overlap regions, and which row groups gather from a region (they wait for the finish
// workgroups).  Needs the regions on 128-byte lines of the long pair (8 rows): distributed.halo_plan rounds its windows so.
int setup_fused_halo(fpsq_handle h) {
  h->fuse_halo_ok = false;
  if (!h->fuse_ok || !h->halo || h->ovl + h->ovr == 0) return 0;
  if (h->ovl % 8 != 0 || (h->n - h->ovr) % 8 != 0 || h->halo_gf > kEwBlocksMax) return 0;
  const std::vector<int32_t>& rb = h->fz_rb;
  auto block_of = [&](int64_t row) { return (int)(std::upper_bound(rb.begin(), rb.end(), (int32_t)row) - rb.begin()) - 1; };
  h->fz_depL = h->ovl > 0 ? make_int2(0, std::max(block_of(h->ovl - 1), 0)) : make_int2(1, 0);
  h->fz_depR = h->ovr > 0 ? make_int2(std::max(block_of(h->n - h->ovr), 0), h->AT.nblk - 1) : make_int2(1, 0);
  std::vector<int2> dep2(h->fz_colrange.size());
  for (size_t g = 0; g < dep2.size(); ++g) {
    const int64_t lo = h->fz_colrange[g].x & ~7, hi = std::min<int64_t>((int64_t)h->fz_colrange[g].y | 7, h->n - 1);
    const bool touches = lo < h->ovl || hi >= h->n - h->ovr;
    dep2[g] = touches ? make_int2(h->AT.nblk, h->AT.nblk + h->halo_gf - 1) : make_int2(1, 0);
  }
  dfree(h, &h->fz_dep2);
  if (int rc = dalloc(h, &h->fz_dep2, dep2.size())) return rc;
  HIPCHK(h, hipMemcpy(h->fz_dep2, dep2.data(), dep2.size() * sizeof(int2), hipMemcpyHostToDevice));
  h->fuse_halo_ok = true;
  return 0;
}

// after the structure (host CSR of A) is known: transposed copy, uploads, workspaces
int finish_structure(fpsq_handle h, const HostCsr& HA) {
  HostCsr HT;
  std::vector<int32_t> perm;
  transpose_structure(HA, HT, perm);
  if (int rc = upload_csr(h, HA, h->A)) return rc;
  h->AT.row_align = h->fuse_iter ? 8 : 1;  // (whether a sharded handle may use the launch is decided per run: KrylovRun::setup)
  if (const char* ev = std::getenv("FPSQ_AT_ROW_ALIGN")) h->AT.row_align = std::max(1, std::atoi(ev));  // (tests: the fused layout without the fused launch)
  if (int rc = upload_csr(h, HT, h->AT)) return rc;
  std::vector<int32_t> csr_pos;
  std::vector<int2> col_range;
  if (int rc = build_rgcs(h, HA, h->RA, &csr_pos, &col_range)) return rc;
  const bool can_share = h->RA.ok && !csr_pos.empty() && !h->refresh_3pass;
  if (int rc = pad_blocks(h, HT, perm, h->AT, can_share ? &csr_pos : nullptr, h->RA.nstore, h->RA.vals)) return rc;
  if (int rc = dalloc(h, &h->permT, perm.size())) return rc;
  if (!perm.empty()) HIPCHK(h, hipMemcpy(h->permT, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
  h->nnz = h->A.nnz;
  // COO input without duplicates: the value permutations of A' and of the row groups are composed with the COO -> CSR order
  // once, here, so that a refresh gathers straight from the caller's jac_coord! output (k_refresh) -- no CSR staging pass.
  // (With duplicates the slots are summed into the CSR array first and the permutations keep pointing there.)
  h->perms_to_input = false;
  if (h->in_perm && !h->in_slotptr && !h->refresh_3pass && h->nnz > 0) {
    if (h->AT.nstore > 0)
      hipLaunchKernelGGL(k_compose_perm, dim3(ew_grid(h->AT.nstore)), dim3(kBlock), 0, nullptr, h->permT, h->in_perm, h->AT.nstore);
    if (h->RA.ok)
      hipLaunchKernelGGL(k_compose_perm, dim3(ew_grid(h->RA.nstore)), dim3(kBlock), 0, nullptr, h->RA.vperm, h->in_perm, h->RA.nstore);
    h->perms_to_input = true;
  }
  if (int rc = alloc_workspaces(h)) return rc;
  if (int rc = setup_fused_iteration(h, HT, col_range)) return rc;
  if (int rc = setup_fused_halo(h)) return rc;  // (a halo-sharded handle given a new structure)
  HIPCHK(h, hipDeviceSynchronize());  // the set-up used null-stream copies/memsets; the solver stream is non-blocking
  h->have_structure = true;
  h->have_values = false;
  h->info.n = h->n;
  h->info.m = h->m;
  h->info.nnz = h->nnz;
  h->info.spmv_a_blocks = npart_A(h);
  h->info.spmv_at_blocks = h->AT.nblk;
  h->info.at_sorted = h->AT.shared ? 2 : h->AT.sorted ? 1 : 0;
  return 0;
}

// ------------------------------------------------------------------ launch helpers

inline void ht_mark(fpsq_handle h, int k) {
  if (!h->host_trace) return;
  const auto now = std::chrono::steady_clock::now();
  h->ht_sum[k] += std::chrono::duration<double>(now - h->ht_last).count();
  h->ht_last = now;
}


enum { TAG_A = 0, TAG_AT = 1 };

// Profiled product launches attach the event pair to the dispatch itself (hipExtLaunchKernelGGL): the elapsed time
// is the kernel's own start-to-end time, as rocprofv3 reports it.  Two hipEventRecord markers around the launch add
// ~5 us of marker processing to every sample.
template <typename K, typename... Args>
void launch_product(fpsq_handle h, K kernel, dim3 grid, Args... args) {
  if (!h->profile) {
    hipLaunchKernelGGL(kernel, grid, dim3(kBlock), 0, h->stream, args...);
    return;
  }
  if (h->ev_used == h->ev_pool.size()) {
    EventPair p;
    hipEventCreate(&p.a);
    hipEventCreate(&p.b);
    h->ev_pool.push_back(p);
  }
  EventPair& e = h->ev_pool[h->ev_used++];
  hipExtLaunchKernelGGL(kernel, grid, dim3(kBlock), 0, h->stream, e.a, e.b, 0, args...);
}

UpdSeg seg_none() {
  UpdSeg s{};
  s.kind = UPD_NONE;
  s.nblk = 0;
  return s;
}

// u0/u1: vector-update segments that ride in the product launch (run_fused_updates); they may only read what the
// product reads.
// halo_rows (A' products of a halo-mode handle): the overlap rows of the rank's column window only get their raw sums,
// see HaloRows / halo_finish.
// pre (two entries): the scalar steps of the two lanes that follow the previous product ride in this launch, with leader
// workgroups (k_spmv_atl / k_spmv_rgcs<.., LEAD>; run_krylov only hands steps over where both products have those variants)
template <int NL>
void launch_spmv(fpsq_handle h, int tag, const double* x, const double* yin, double* yout, const LaneCtl* c0,
                 const LaneCtl* c1, double* partials, const UpdSeg& u0 = seg_none(), const UpdSeg& u1 = seg_none(),
                 bool halo_rows = false, const StepArgs* pre = nullptr) {
  const int nupd = u0.nblk + u1.nblk;
  const HaloRows hr{h->ovl, h->n - h->ovr, h->halo_raw};
  StepArgs z0{}, z1{};
  if (pre) {
    z0 = pre[0];
    z1 = pre[1];
  }
  RideArgs ra{};
  const bool lead = pre != nullptr;
  if (lead) {
    ra.rec = h->ride_rec;
    ra.want = (unsigned int)++h->ride_seq;
    ra.pub = h->ride_break ? ~ra.want : ra.want;
    ra.err = reinterpret_cast<unsigned long long*>(h->hscal_dev + 15);
    ra.delay = h->ride_delay;
    ra.xseq = h->ride_xseq;
    ra.xt = h->ride_xseq ? insum_table(h) : nullptr;
    ra.more = ra.xt ? h->comm->wait_more() : 0;  // (the leaders may be waiting for a late peer: whoever waits for them outlasts that)
  }
  if (tag == TAG_A && h->RA.ok) {
    const int per_xcd = (h->RA.view.ng + 7) / 8;
#define FPSQ_LAUNCH_RGCS(...) \
    launch_product(h, k_spmv_rgcs<__VA_ARGS__>, dim3(per_xcd * 8 + nupd + (lead ? kRideCand : 0)), h->RA.view, x, yin, yout, c0, c1, partials, \
                   per_xcd, u0, u1, h->gate0, h->gate1, h->strA, z0, z1, ra)
    if constexpr (NL == 2) {
      // (a sharded handle whose leaders form their sums over the ranks in the launch: the variants with the exchange compiled in)
      if (lead && ra.xt != nullptr && h->RA.view.stride) FPSQ_LAUNCH_RGCS(2, true, true, true);
      else if (lead && ra.xt != nullptr) FPSQ_LAUNCH_RGCS(2, false, true, true);
      else if (lead && h->RA.view.stride) FPSQ_LAUNCH_RGCS(2, true, true);
      else if (lead) FPSQ_LAUNCH_RGCS(2, false, true);
    }
    if (!pre) {
      if (h->RA.view.stride) FPSQ_LAUNCH_RGCS(NL, true);
      else FPSQ_LAUNCH_RGCS(NL, false);
    }
#undef FPSQ_LAUNCH_RGCS
  } else {
    const DevCsr& M = tag == TAG_A ? h->A : h->AT;
    const int per_xcd = (M.nblk + 7) / 8;
    // the tail of a call on one GPU: the raw product's rows go straight into the call's result (grad(phi): two lanes; Hv: one)
    const bool grad = h->tail_grad != nullptr && tag == TAG_AT && !lead && !halo_rows && M.sorted && nupd == 0;
    const GradEpi ge = grad ? *h->tail_grad : GradEpi{};
    const dim3 grid(per_xcd * 8 + nupd + (grad && ge.fx.out != nullptr ? 1 : 0));
    const int ps = tag == TAG_A ? h->strA : h->strT;
#define FPSQ_LAUNCH_SPMV(...) \
    launch_product(h, k_spmv<__VA_ARGS__>, grid, M.view(), x, yin, yout, c0, c1, partials, per_xcd, u0, u1, h->gate0, h->gate1, ps, hr, ge)
    bool done_pre = false;
    if constexpr (NL == 2) {
      if (lead) {  // (tag == TAG_AT: padded blocks with block-relative columns)
        done_pre = true;
        // the first resident set of workgroups takes two row blocks each (see k_spmv_atl)
        const int R = h->resident_wgs - kRideCand;
        int n2 = !h->atl_two || M.nblk <= R ? 0 : std::min(R, M.nblk - R);
        int nwg = M.nblk - n2;
        int bpx = 0;
        if (h->at_xcd) {  // XCD-contiguous eighths of the row blocks (FPSQ_AT_XCD=0: grid order)
          bpx = (M.nblk + 7) / 8;
          const int n2e = std::min(n2 / 8, bpx / 2);
          n2 = 8 * n2e;
          nwg = 8 * (bpx - n2e);
        }
        const dim3 lgrid(kRideCand + nwg + nupd);
        if (M.sorted && halo_rows)
          launch_product(h, k_spmv_atl<true, true>, lgrid, M.view(), x, yin, yout, partials, nwg, n2, u0, u1, ps, z0, z1, ra, hr, bpx);
        else if (M.sorted)
          launch_product(h, k_spmv_atl<true, false>, lgrid, M.view(), x, yin, yout, partials, nwg, n2, u0, u1, ps, z0, z1, ra, hr, bpx);
        else if (halo_rows)
          launch_product(h, k_spmv_atl<false, true>, lgrid, M.view(), x, yin, yout, partials, nwg, n2, u0, u1, ps, z0, z1, ra, hr, bpx);
        else
          launch_product(h, k_spmv_atl<false, false>, lgrid, M.view(), x, yin, yout, partials, nwg, n2, u0, u1, ps, z0, z1, ra, hr, bpx);
      }
    }
    if (done_pre) {
    } else if (tag == TAG_A && M.col16) FPSQ_LAUNCH_SPMV(NL, TAG_A, true);
    else if (tag == TAG_A) FPSQ_LAUNCH_SPMV(NL, TAG_A, false);
    else if (halo_rows) {
      if (M.sorted) FPSQ_LAUNCH_SPMV(NL, TAG_AT, true, true, true, true);
      else if (M.col16 && M.padded) FPSQ_LAUNCH_SPMV(NL, TAG_AT, true, true, true);
      else if (M.col16) FPSQ_LAUNCH_SPMV(NL, TAG_AT, true, false, true);
      else if (M.padded) FPSQ_LAUNCH_SPMV(NL, TAG_AT, false, true, true);
      else FPSQ_LAUNCH_SPMV(NL, TAG_AT, false, false, true);
    } else if (grad) {
      FPSQ_LAUNCH_SPMV(NL, TAG_AT, true, true, false, true, true);
      h->tail_grad_used = true;
    } else if (M.sorted) FPSQ_LAUNCH_SPMV(NL, TAG_AT, true, true, false, true);
    else if (M.col16 && M.padded) FPSQ_LAUNCH_SPMV(NL, TAG_AT, true, true);
    else if (M.col16) FPSQ_LAUNCH_SPMV(NL, TAG_AT, true);
    else if (M.padded) FPSQ_LAUNCH_SPMV(NL, TAG_AT, false, true);
    else FPSQ_LAUNCH_SPMV(NL, TAG_AT, false);
#undef FPSQ_LAUNCH_SPMV
  }
  h->launches++;
  h->spmv_launches++;
  (tag == TAG_A ? h->prod_a : h->prod_at)[NL - 1]++;
}

__global__ void k_set_ctl(LaneCtl* c, double ca, double cb) {
  c->ca = ca;
  c->cb = cb;
  c->done = 0;
  c->skip = 0;
  c->upd_iter = -1;
}

// control block holding the host-given coefficient pair (ca, cb)
const LaneCtl* const_ctl(fpsq_handle h, double ca, double cb) {
  // the coefficient pairs of the hot path are resident constants: no set-up launch
  if (ca == 1.0 && cb == 0.0) return h->ctl_raw;
  if (ca == 1.0 && cb == -1.0) return h->ctl_pm;
  if (ca == -1.0 && cb == 1.0) return h->ctl_mp;
  hipLaunchKernelGGL(k_set_ctl, dim3(1), dim3(1), 0, h->stream, h->ctl_tmp, ca, cb);
  h->launches++;
  return h->ctl_tmp;
}

// out = ca * op(A) x + cb * yin with host-given constants
void spmv_const(fpsq_handle h, int tag, double ca, const double* x, double cb, const double* yin, double* yout) {
  const LaneCtl* c = const_ctl(h, ca, cb);
  launch_spmv<1>(h, tag, x, yin, yout, c, c, nullptr);
}

int comm_allreduce(fpsq_handle h, double* buf, size_t count) {
  if (int rc = h->comm->allreduce_sum(buf, count, h->stream)) {
    h->err = h->comm->err;
    return rc;
  }
  return 0;
}

// Sum over the ranks of the raw partial A' products in buf ([n][NL]): all-reduce of the replicated n-vector (the
// replicated layout; halo mode never comes here: see halo_finish)
int comm_reduce_long(fpsq_handle h, double* buf, int NL) { return comm_allreduce(h, buf, (size_t)h->n * NL); }

// LP <- ca A' SP + cb LP with norm partials (count returned in *np).  Sharded: every rank holds a row block A_r, so
// A'x = sum_r A_r' x_r: raw partial product -> all-reduce -> fused axpby + norm on the replicated result.
// Halo mode, after k_spmv<.., HALO>: exchange the raw sums of the two overlap regions with the neighbours, then finish
// those rows (yout = ca (own + neighbour's) + cb yin, squared-norm partials of the owned head region behind the product's).
template <int NL>
int halo_finish(fpsq_handle h, const double* yin, double* yout, const LaneCtl* c0, const LaneCtl* c1, double* partials) {
  const int64_t t = h->ovl + h->ovr;
  if (t == 0) return 0;
  double* rl = h->halo_recv + (size_t)(h->halo_calls++ & 1) * (size_t)t * 2;
  {  // peer-to-peer routes: exchange + finish in one launch
    const HaloFinishArgs fa{h->halo_raw, rl, h->ovl, h->ovr, h->n - h->ovr, yin, yout, c0, c1,
                            partials ? partials + h->AT.nblk : nullptr, h->strT, 0 /* dbg: the route's */, h->gate0, h->gate1};
    if (h->comm->halo_exchange_finish(NL, fa, h->halo_gf, h->stream)) {
      h->launches++;
      return 0;
    }
  }
  if (int rc = h->comm->halo_exchange(h->halo_raw, t, NL, h->ovl, h->ovr, rl, rl + (size_t)h->ovl * NL, h->stream)) {
    h->err = h->comm->err;
    return rc;
  }
  hipLaunchKernelGGL(k_halo_finish<NL>, dim3(h->halo_gf), dim3(kBlock), 0, h->stream, h->halo_raw, rl, h->ovl,
                     h->ovr, h->n - h->ovr, yin, yout, c0, c1, partials ? partials + h->AT.nblk : nullptr, h->strT, h->gate0,
                     h->gate1);
  h->launches++;
  return 0;
}

template <int NL>
int at_product(fpsq_handle h, const double* x, double* y, const LaneCtl* c0, const LaneCtl* c1, double* partials,
               int* np, const UpdSeg& u0 = seg_none(), const UpdSeg& u1 = seg_none(), const StepArgs* pre = nullptr) {
  if (!h->comm) {
    launch_spmv<NL>(h, TAG_AT, x, y, y, c0, c1, partials, u0, u1, false, pre);
    *np = h->AT.nblk;
    return 0;
  }
  if (h->halo) {
    // every row the rank alone contributes to is finished by the product kernel exactly as on one GPU (so the vector
    // updates may ride in the launch); only the overlap rows wait for the neighbours
    launch_spmv<NL>(h, TAG_AT, x, y, y, c0, c1, partials, u0, u1, /*halo_rows=*/true, pre);
    // (steps riding in that launch: the control blocks k_halo_finish must read are the ones the leaders have just written)
    const LaneCtl* f0 = pre ? reinterpret_cast<const LaneCtl*>(pre[0].state_out) : c0;
    const LaneCtl* f1 = pre ? reinterpret_cast<const LaneCtl*>(pre[NL - 1].state_out) : c1;
    if (int rc = halo_finish<NL>(h, y, y, f0, f1, partials)) return rc;
    *np = h->AT.nblk + h->halo_gf;
    return 0;
  }
  launch_spmv<NL>(h, TAG_AT, x, nullptr, h->comm_vec, h->ctl_raw, h->ctl_raw, nullptr);
  if (int rc = comm_reduce_long(h, h->comm_vec, NL)) return rc;
  const int g = ew_grid(h->n);
  h->strT = g;  // replicated layout: the norm partials of the A' product come from this kernel, g per lane
  hipLaunchKernelGGL(k_axpby_norm<NL>, dim3(g), dim3(kBlock), 0, h->stream, h->comm_vec, y, c0, c1, h->n, n_owned(h),
                     partials);
  h->launches++;
  *np = g;
  return 0;
}

// out = ca A' x + cb yin (plain vectors, host constants), all-reduced when sharded
int at_product_const(fpsq_handle h, double ca, const double* x, double cb, const double* yin, double* yout) {
  if (!h->comm) {
    spmv_const(h, TAG_AT, ca, x, cb, yin, yout);
    return 0;
  }
  if (h->halo) {
    const LaneCtl* c = const_ctl(h, ca, cb);
    launch_spmv<1>(h, TAG_AT, x, yin, yout, c, c, nullptr, seg_none(), seg_none(), /*halo_rows=*/true);
    return halo_finish<1>(h, yin, yout, c, c, nullptr);
  }
  launch_spmv<1>(h, TAG_AT, x, nullptr, h->comm_vec, h->ctl_raw, h->ctl_raw, nullptr);
  if (int rc = comm_reduce_long(h, h->comm_vec, 1)) return rc;
  hipLaunchKernelGGL(k_axpby_plain, dim3(ew_grid(h->n)), dim3(kBlock), 0, h->stream, h->comm_vec, ca, yin, cb, yout, h->n);
  h->launches++;
  return 0;
}

// Bounded wait until the device has reached `target` iterations (or finished).  The progress word lives in
// host-mapped memory and is stored by the scalar kernels; if the stream drains without the word moving (which
// would mean the mapped store is not visible) we fall back to reading the device state explicitly.
// one consistent snapshot {iter, done} of a lane's progress word (a single 8-byte load: see publish())
inline Progress load_progress(const Progress* p) {
  const uint64_t v = *reinterpret_cast<const volatile uint64_t*>(p);
  Progress r;
  r.iter = (int32_t)(uint32_t)(v & 0xffffffffu);
  r.done = (int32_t)(uint32_t)(v >> 32);
  return r;
}

int wait_progress(fpsq_handle h, int lane, int target, const int32_t* dev_done, const int32_t* dev_iter) {
  Progress* p = &h->prog_host[lane];
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  auto reached = [&]() {
    const Progress s = load_progress(p);
    return s.done || s.iter >= target;
  };
  while (!reached()) {
    if ((++spins & 63) == 0) {
      // a bounded wait inside a launch has expired (the handle's error word): nothing later in this call can be right, and the
      // recurrences' progress words will not move any more -- leave the loop now, not at itmax (advisor, round 4)
      // (stop WAITING, not the call: the end of the call reads the word, switches the handle to two launches per iteration and has
      // the entry point repeat the call -- ride_failed / with_fuse_fallback; pace_single ends the loop on the same word)
      if (*reinterpret_cast<volatile uint64_t*>(h->hscal + 15) != 0) return 0;
      hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) {
        if (reached()) break;
        int32_t d = 0, it = 0;
        HIPCHK(h, hipMemcpy(&d, dev_done, 4, hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(&it, dev_iter, 4, hipMemcpyDeviceToHost));
        p->done = d;
        p->iter = it;
        break;
      } else if (q != hipErrorNotReady) {
        h->err = std::string("stream failed while iterating: ") + hipGetErrorString(q);
        return FPSQ_ERR_HIP;
      }
      const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (el > 120.0) {
        h->err = "timeout waiting for device progress";
        return FPSQ_ERR_TIMEOUT;
      }
      if (el > 0.002) std::this_thread::yield();
    }
  }
  return 0;
}

// ------------------------------------------------------------------ Krylov drivers

struct LsqrParams {
  double lambda, atol, rtol, axtol, btol, etol, conlim;
  int64_t itmax;
  int32_t pub_from;
};

__device__ __forceinline__ void lsqr_set_params(LsqrState* S, const LsqrParams& P) {
  S->lambda = P.lambda;
  S->atol = P.atol;
  S->rtol = P.rtol;
  S->axtol = P.axtol;
  S->btol = P.btol;
  S->etol = P.etol;
  S->ctol = P.conlim > 0.0 ? 1.0 / P.conlim : 0.0;
  S->itmax = P.itmax;
  S->pub_from = P.pub_from;
  S->ctl.done = 0;
  S->ctl.skip = 0;
  S->ctl.upd_iter = -1;
}

struct CraigParams {
  double mu, lambda, atol, rtol, btol, conlim, xsign;
  int64_t itmax;
  int32_t start_skipped;
  int32_t pub_from;
};

__device__ __forceinline__ void craig_set_params(CraigState* S, const CraigParams& P) {
  S->mu = P.mu;
  S->lambda = P.lambda;
  S->atol = P.atol;
  S->rtol = P.rtol;
  S->btol = P.btol;
  S->ctol = P.conlim > 0.0 ? 1.0 / P.conlim : 0.0;
  S->xsign = P.xsign;
  S->itmax = P.itmax;
  S->pub_from = P.pub_from;
  S->ctl.done = 0;
  S->ctl.skip = P.start_skipped;  // stays out of the LSQR lane's start-up product; craig_begin clears it
  S->ctl.upd_iter = -1;
}

struct LnlqParams {
  double mu, atol, rtol, xsign;
  int64_t itmax;
  int32_t start_skipped, pub_from;
};

__device__ __forceinline__ void lnlq_set_params(LnlqState* S, const LnlqParams& P) {
  S->mu = P.mu;
  S->atol = P.atol;
  S->rtol = P.rtol;
  S->xsign = P.xsign;
  S->itmax = P.itmax;
  S->pub_from = P.pub_from;
  S->ctl.done = 0;
  S->ctl.skip = P.start_skipped;  // stays out of the LSQR lane's start-up product; lnlq_begin_step clears it
  S->ctl.upd_iter = -1;
}

struct MinresParams {
  double lambda, atol, rtol, etol, conlim;
  int64_t itmax;
  int32_t pub_from;
};

__device__ __forceinline__ void minres_set_params(MinresState* S, const MinresParams& P) {
  S->lambda = P.lambda;
  S->atol = P.atol;
  S->rtol = P.rtol;
  S->etol = P.etol;
  S->ctol = P.conlim > 0.0 ? 1.0 / P.conlim : 0.0;
  S->itmax = P.itmax;
  S->pub_from = P.pub_from;
  S->ctl.done = 0;
  S->ctl.skip = 1;  // stays out of the LSQR lane's start-up product; minres_begin_step clears it
  S->ctl.upd_iter = -1;
  S->ctlT.done = 0;
  S->ctlT.skip = 0;
  S->ctlT.upd_iter = -1;
  S->ctlT.ca = 1.0;  // tmp = A' r2, raw
  S->ctlT.cb = 0.0;
  S->kmode = 0;
  S->kdelta = 0.0;
}

// Start-up of a run in ONE launch: lane parameters (workgroup 0), the right-hand sides loaded into their interleaved
// lanes with the squared-norm partials, and the vectors that start at zero.
struct LoadSeg {
  const double* src;
  double scale;
  double* dst;
  double* dst2;  // optional plain copy of the scaled vector (MINRES keeps r2 = b next to the pair's lane)
  int32_t lane, nblk;
  int64_t len;
  int64_t sum_len;  // the squared-norm partials run over [0, sum_len) (halo mode: the owned prefix of an n-vector)
  double* partials;
};
template <int NL>
__global__ __launch_bounds__(kBlock) void k_startup(LsqrState* S0, LsqrParams P0, LsqrState* S1, LsqrParams P1, CraigState* C,
                                                    CraigParams PC, MinresState* M, MinresParams PM, LnlqState* Q,
                                                    LnlqParams PQ, LoadSeg l0, LoadSeg l1, ZeroArgs z, int nzblk,
                                                    const QpGradArgs qg) {
  __shared__ double red[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (S0) lsqr_set_params(S0, P0);
    if (S1) lsqr_set_params(S1, P1);
    if (C) craig_set_params(C, PC);
    if (M) minres_set_params(M, PM);
    if (Q) lnlq_set_params(Q, PQ);
  }
  // qp_objgrad's fast start: the first qg.nblk workgroups evaluate g = q .* x + d, write the long pair {g, x} and the
  // partial sums of f and ||g||^2 (the user-model evaluation of _compute_ys_gs!, model:238-240) -- one launch, no
  // kernel boundary between the model evaluation and the start-up of the recurrences
  if ((int)blockIdx.x < qg.nblk) {
    qp_grad_body(qg, blockIdx.x, red);
    return;
  }
  int blk = blockIdx.x - qg.nblk;
  if (blk < l0.nblk + l1.nblk) {
    const bool first = blk < l0.nblk;
    if (!first) blk -= l0.nblk;
    const double* src = first ? l0.src : l1.src;
    double* dst = first ? l0.dst : l1.dst;
    double* dst2 = first ? l0.dst2 : l1.dst2;
    const double scale = first ? l0.scale : l1.scale;
    const int lane = first ? l0.lane : l1.lane;
    const int nb = first ? l0.nblk : l1.nblk;
    const int64_t len = first ? l0.len : l1.len;
    const int64_t sum_len = first ? l0.sum_len : l1.sum_len;
    double* partials = first ? l0.partials : l1.partials;
    double sq = 0.0;
    for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < len; i += (int64_t)nb * kBlock) {
      const double v = scale * src[i];
      dst[i * NL + lane] = v;
      if (dst2) dst2[i] = v;
      if (i < sum_len) sq += v * v;
    }
    const double t = block_sum(sq, red);
    if (threadIdx.x == 0) partials[blk] = t;
    return;
  }
  blk -= l0.nblk + l1.nblk;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!z.p[k]) continue;
    for (int64_t i = (int64_t)blk * kBlock + threadIdx.x; i < z.n[k]; i += (int64_t)nzblk * kBlock) z.p[k][i] = 0.0;
  }
}

enum { LANE_LSQR = 1, LANE_CRAIG = 2, LANE_MINRES = 3, LANE_LNLQ = 4 };
// the two least-norm recurrences share their vector plumbing (short Mu~, w, y; long v~, x)
inline bool is_ln(int kind) { return kind == LANE_CRAIG || kind == LANE_LNLQ; }

// One Krylov recurrence of a (possibly fused) run.
struct Lane {
  int kind = 0;
  const double* rhs = nullptr;  // LSQR: n-vector b;  CRAIG, MINRES: m-vector b
  double rhs_scale = 1.0;
  double lambda = 0.0;          // LSQR regularisation; MINRES: shift of A A' + lambda I
  double delta = 0.0;           // CRAIG: M = (1/delta) I, sqd when != 0
  double xsign = 1.0;           // CRAIG: xs accumulates xsign * x
  double* x = nullptr;          // LSQR, MINRES: solution (m).  CRAIG: xs (n)
  double* y = nullptr;          // CRAIG: y (m)
  fpsq_stats* st = nullptr;     // destination of the final stats: an element of the host-mapped h->hstats
  fpsq_stats* st_dev = nullptr; // its device alias (filled by run_krylov / run_minres)
  // fast start (qp_objgrad, fused single-GPU runs):
  bool preloaded = false;               // LSQR: the caller already wrote rhs into the long pair's lane and ||rhs||^2 partials to pE
  const double* affine_shift = nullptr; // CRAIG: rhs = -(A z - shift) with z already in the long pair's lane: formed by the
  double* affine_out = nullptr;         //        LSQR start-up product (the lane is otherwise parked there); A z - shift -> affine_out
  // filled by run_krylov
  void* state = nullptr;
  void* state_alt = nullptr;    // the other copy of the state (riding steps alternate between the two)
  void* state_alt2 = nullptr;   // a third one (fused iterations: the step behind the A' product lands there, see k_iter_fused)
  LaneCtl* ctl = nullptr;       // coefficients of the A product (and of the A' product for LSQR / CRAIG)
  LaneCtl* ctlT = nullptr;      // coefficients of the A' product (MINRES: the raw tmp = A' r2)
  int64_t itmax = 0;
};

StepArgs step_args(int kind, const Lane& L, int it, const double* p0, int n0, const double* p1, int n1, Progress* prog) {
  StepArgs a{};
  a.kind = kind;
  a.it = it;
  a.state = L.state;
  a.p0 = p0;
  a.p1 = p1;
  a.n0 = n0;
  a.n1 = n1;
  a.prog = prog;
  a.host_stats = L.st_dev;
  return a;
}

void launch_step_raw(fpsq_handle h, const StepArgs& a0, const StepArgs& a1, uint32_t xseq = 0) {
  const int nb = a1.kind != STEP_NONE ? 2 : 1;
  const XchTable* xt = xseq ? insum_table(h) : nullptr;
  if (xt) hipLaunchKernelGGL(k_step<true>, dim3(nb), dim3(kStepThreads), 0, h->stream, a0, a1, xt, (unsigned int)xseq);
  else hipLaunchKernelGGL(k_step<false>, dim3(nb), dim3(kStepThreads), 0, h->stream, a0, a1, xt, 0u);
  h->launches++;
}

// `sharded`: the partial arrays of these steps are sums over m-vectors, of which a rank only holds its rows:
// local sums -> one scalar all-reduce (4 doubles) -> the step kernel reads the global sums.
// padded (common to all ranks) count of the segment array that starts at p; 0: not an array of the segment
int seg_count(fpsq_handle h, const double* p) {
  if (p == h->pE || p == h->pE2) return h->cE;
  if (p == h->pEm[0] || p == h->pEm[1]) return h->cW;
  if (p == h->pS || p == h->pS + h->strT) return h->cT;
  if (p == h->pS2 || p == h->pS2 + h->strA) return h->cA;
  if (p == h->pW[0] || p == h->pW[1] || p == h->pWalt[0] || p == h->pWalt[1] || p == h->pE3) return h->cW;
  return 0;
}

// Everything a step needs BEFORE its kernel: row-sharded runs gather (halo mode) or pre-sum + all-reduce the partial sums
// its arguments point to, and the arguments are redirected to the gathered / reduced numbers.  A step that rides in the next
// product launch is prepared when it is handed over (the collective must precede that launch in the stream).
int prepare_step(fpsq_handle h, StepArgs& a0, StepArgs& a1, bool sharded = false, int sharded1 = -1) {
  const bool sh[2] = {sharded, sharded1 < 0 ? sharded : sharded1 != 0};  // per step: its partials are per-rank sums
  h->last_xseq = 0;
  if (h->comm && h->halo && insum(h)) {
    // the step's workgroup forms the sum over the ranks itself (xch_sum): the arguments stay the rank's local arrays, and the pair
    // gets an exchange number -- the same sequence on every rank -- which travels next to the steps (h->last_xseq: the caller
    // hands it to k_step or to the launch whose leaders compute the pair).  A communicator of one: nothing at all.
    if (insum_table(h) != nullptr && ((sh[0] && a0.kind != STEP_NONE) || (sh[1] && a1.kind != STEP_NONE))) h->last_xseq = ++h->xch_seq;
    return 0;
  }
  if (h->comm && h->halo && (sh[0] || sh[1])) {
    // Halo mode: ONE all-gather of the contiguous segment range holding the arrays these steps read; the step kernel then
    // sums the nranks copies of every array in rank-major order (no local pre-sum launch, no reduction by the library).
    StepArgs* w[2] = {&a0, &a1};
    const double *lo = nullptr, *hi = nullptr;
    for (int k = 0; k < 2; ++k) {
      if (w[k]->kind == STEP_NONE || !sh[k]) continue;
      const double* ps[2] = {w[k]->p0, w[k]->p1};
      for (const double* q : ps) {
        if (!q) continue;
        const int c = seg_count(h, q);
        if (c == 0) {
          h->err = "internal: a sharded step reads a partial array outside the gather segment";
          return FPSQ_ERR_STATE;
        }
        if (!lo || q < lo) lo = q;
        if (!hi || q + c > hi) hi = q + c;
      }
    }
    const int64_t len = hi - lo;
    double* gbuf = h->gath + (size_t)(h->gather_calls++ & 1) * (size_t)h->seg_len * h->comm->nranks;
    if (int rc = h->comm->allgather(lo, gbuf, (size_t)len, h->stream)) {
      h->err = h->comm->err;
      return rc;
    }
    for (int k = 0; k < 2; ++k) {
      if (w[k]->kind == STEP_NONE || !sh[k]) continue;
      w[k]->n0 = seg_count(h, w[k]->p0);
      w[k]->p0 = gbuf + (w[k]->p0 - lo);
      if (w[k]->p1) {
        w[k]->n1 = seg_count(h, w[k]->p1);
        w[k]->p1 = gbuf + (w[k]->p1 - lo);
      }
      w[k]->nseg = h->comm->nranks;
      w[k]->seg_stride = (int32_t)len;
    }
    return 0;
  }
  if (h->comm && (sh[0] || sh[1])) {
    PresumArgs P{};
    const StepArgs* a[2] = {&a0, &a1};
    for (int k = 0; k < 2; ++k) {
      if (a[k]->kind == STEP_NONE || !sh[k]) continue;
      P.p[2 * k] = a[k]->p0;
      P.n[2 * k] = a[k]->n0;
      P.p[2 * k + 1] = a[k]->p1;
      P.n[2 * k + 1] = a[k]->n1;
    }
    hipLaunchKernelGGL(k_presum, dim3(1), dim3(kBlock), 0, h->stream, P, h->comm_scal);
    h->launches++;
    if (int rc = comm_allreduce(h, h->comm_scal, 4)) return rc;
    StepArgs* w[2] = {&a0, &a1};
    for (int k = 0; k < 2; ++k) {
      if (w[k]->kind == STEP_NONE || !sh[k]) continue;
      w[k]->p0 = h->comm_scal + 2 * k;
      w[k]->n0 = 1;
      if (w[k]->p1) {
        w[k]->p1 = h->comm_scal + 2 * k + 1;
        w[k]->n1 = 1;
      }
    }
  }
  return 0;
}

int launch_step(fpsq_handle h, StepArgs a0, StepArgs a1, bool sharded = false, int sharded1 = -1) {
  if (int rc = prepare_step(h, a0, a1, sharded, sharded1)) return rc;
  launch_step_raw(h, a0, a1, h->last_xseq);
  return 0;
}

template <int NL>
void launch_updates(fpsq_handle h, const UpdSeg& s0, const UpdSeg& s1, const UpdSeg& s2) {
  const int nb = s0.nblk + s1.nblk + s2.nblk;
  if (nb == 0) return;
  hipLaunchKernelGGL(k_updates<NL>, dim3(nb), dim3(kBlock), 0, h->stream, s0, s1, s2);
  h->launches++;
}


// Runs 1 or 2 recurrences in lock-step on the interleaved Golub-Kahan pairs LP (n) / SP (m):
//   A' product: LP <- ca A' SP + cb LP      (LSQR: u~ <- B v - alpha u;     CRAIG: v~ <- B'u - beta v)
//   A  product: SP <- ca A  LP + cb SP      (LSQR: v~ <- B'u - beta v;      CRAIG: Mu~ <- B v - alpha Mu)
// Each lane is exactly Krylov.jl's lsqr! / craig! on its own right-hand side (its results do not depend on the
// other lane); running them side by side turns two SpMVs into one SpMM with k = 2.
// `tail` (optional, single GPU): enqueues the caller's epilogue kernels.  When the iteration count of the previous call
// of the same kind is known, the final LSQR flush and the tail are enqueued SPECULATIVELY right behind iteration
// `expect`, gated on the lanes' `done` flags (h->gate0/1): if the recurrences do end there -- consecutive evaluations of
// a line search mostly repeat their counts -- the epilogue runs without the host first having to see `done` and only
// then launching it (a ~30 us bubble per evaluation); if not, the gated kernels exit at once and the loop goes on.
// h->tail_was_run tells the caller whether its epilogue has been taken care of.
//
// Structure (round 4; one 640-line function before): KrylovRun::run() is the loop and knows three things -- a PRODUCT is
// launched (with whatever rides in it), the STEPS behind it are posted (PendingSteps: they ride in the next product launch
// or get a launch of their own), the host PACES itself (exchange boundaries of a sharded run, run-ahead, speculation).
// What a recurrence of a given kind contributes at each of those points -- which step kinds, which update segments,
// which partial arrays -- is in the builders (lane_*, *_seg, steps_after_*); nothing outside them switches on a lane's kind.
using TailFn = std::function<int()>;

// ---- what depends on the KIND of a recurrence
inline int lane_begin_kind(const Lane& L) {
  return L.kind == LANE_LSQR ? STEP_LSQR_BEGIN : L.kind == LANE_CRAIG ? STEP_CRAIG_BEGIN : L.kind == LANE_LNLQ ? STEP_LNLQ_BEGIN : STEP_MINRES_BEGIN;
}
// the step behind the A' product of an iteration (a MINRES lane runs the stopping tests of the previous iteration there)
inline int lane_kind_after_at(const Lane& L) {
  return L.kind == LANE_LSQR ? STEP_LSQR_SA : L.kind == LANE_CRAIG ? STEP_CRAIG_SA : L.kind == LANE_LNLQ ? STEP_LNLQ_SA : STEP_MINRES_C;
}
// ... and behind the A product (MINRES: step A, between its stages E1 and E2)
inline int lane_kind_after_a(const Lane& L) {
  return L.kind == LANE_LSQR ? STEP_LSQR_SB : L.kind == LANE_CRAIG ? STEP_CRAIG_SB : L.kind == LANE_LNLQ ? STEP_LNLQ_SB : STEP_MINRES_A;
}
inline const int32_t* lane_iter_ptr(const Lane& L) {
  return L.kind == LANE_LSQR ? &((LsqrState*)L.state)->iter
         : L.kind == LANE_CRAIG ? &((CraigState*)L.state)->iter
         : L.kind == LANE_LNLQ  ? &((LnlqState*)L.state)->iter
                                : &((MinresState*)L.state)->iter;
}
// a MINRES / LNLQ lane reports iteration k (step C; pass k) while the host is enqueueing iteration k + 1
inline int lane_lag(const Lane& L) { return L.kind == LANE_MINRES || L.kind == LANE_LNLQ ? 1 : 0; }
// after a product launch that carried the lane's step: the lane lives in its other state copy now
inline void lane_swap_state(Lane& L) {
  std::swap(L.state, L.state_alt);
  L.ctl = reinterpret_cast<LaneCtl*>(L.state);  // LaneCtl is the first member of every state
  L.ctlT = L.kind == LANE_MINRES ? &reinterpret_cast<MinresState*>(L.state)->ctlT : L.ctl;
}

template <int NL>
struct KrylovRun {
  fpsq_handle h;
  Lane* lanes;
  const TailFn* tail;
  const int64_t n, m;
  const fpsq_options& o;
  hipStream_t s;
  const int gn, gm, nbA;
  double *LP, *SP;
  double* LPalt = nullptr;  // the second long pair (several iterations per launch alternate; nullptr: not available)
  bool can_multi = false;   // ... whenever the previous product's steps are pending and the expected count leaves room for >= 2
  // the run-ahead's expectation (see run())
  int64_t* expect_slot;
  const bool local_vec;    // vector updates touch rank-local data only (one GPU, or the halo-sharded layout)
  int64_t expect = 0;
  int32_t pub_from = 0;
  // the lanes
  bool any_lsqr = false;
  int64_t itmax_all = 0;
  Progress* prog[2] = {nullptr, nullptr};
  int minres_lane = -1, affine_lane = -1;
  LsqrState* lsS[2] = {nullptr, nullptr};
  LsqrParams lsP[2] = {};
  CraigState* crS = nullptr;
  CraigParams crP{};
  MinresState* mrS = nullptr;
  MinresParams mrP{};
  LnlqState* lqS = nullptr;
  LnlqParams lqP{};
  bool lead = false;        // the steps ride in the next product launch (leader workgroups)
  bool fuse_upd = false;    // the vector updates ride in the product launches
  bool split_steps = false; // replicated n-sums and per-rank m-sums cannot share a presum launch
  bool can_fuse = false;    // a joint iteration is ONE launch (k_iter_fused) whenever the previous product's steps are pending
  // Where the last A product left its squared-norm partials.  A fused launch READS them (head leaders; mid leaders redoing the
  // head step, any of which another kernel may hold up) while its own row groups -- released per XCC -- WRITE theirs: the
  // launch writes the other array (found by test_one_launch_iterations_with_a_late_mid_leader, which fails with one array)
  double* pa_last = nullptr;
  StepArgs none{};
  // the steps behind the last product, not launched yet
  StepArgs pend[2];
  bool have_pend = false;
  uint32_t pend_xseq = 0;  // ... and the number of their exchange (sharded, in-launch sums; 0: none)
  // the loop
  double *SPcur, *SPalt;
  int look = 1;
  int64_t it = 0;
  int64_t spec_it = -1;  // iteration behind which the gated flush + tail were enqueued
  int64_t tail_launches = 0;
  UpdSeg winit[2] = {seg_none(), seg_none()};
  UpdSeg lu[2] = {seg_none(), seg_none()};  // what rides in (or precedes) this iteration's products: LSQR's update of the previous one
  int nlu = 0;

  KrylovRun(fpsq_handle h_, Lane* lanes_, const TailFn* tail_)
      : h(h_), lanes(lanes_), tail(tail_), n(h_->n), m(h_->m), o(h_->opt), s(h_->stream), gn(ew_grid(h_->n)), gm(ew_grid(h_->m)),
        nbA(npart_A(h_)), LP(h_->LP), SP(h_->SP), expect_slot(h_->expect_iters[lanes_[0].kind][lanes_[NL - 1].kind]),
        local_vec(!h_->comm || h_->halo), SPcur(h_->SP), SPalt(h_->SP2) {
    none.kind = STEP_NONE;
  }

  // coefficients of the A product / the A' product
  LaneCtl* c0() const { return lanes[0].ctl; }
  LaneCtl* c1() const { return lanes[NL - 1].ctl; }
  LaneCtl* t0() const { return lanes[0].ctlT; }
  LaneCtl* t1() const { return lanes[NL - 1].ctlT; }

  // ------------------------------------------------------------------ set-up of the lanes
  void setup() {
    h->tail_was_run = false;
    // iteration count of the previous runs with the same pair of recurrences (0: unknown).  The scalar steps publish their
    // progress to the host only from that iteration on (and when a recurrence ends): see publish().
    // (sharded: only in halo mode, where every rank derives the same count from the replicated recurrence state)
    // The LARGER of the last two counts.  The two ways of being wrong cost very differently: one iteration too many is two
    // launches that exit at their first instruction (~7 us); one too few is a speculative epilogue enqueued for nothing, a host
    // round trip before the loop goes on and another before the epilogue is enqueued again (measured with evaluations
    // alternating between a 14- and a 15-iteration regime, bench.py --alternate-delta: +11 % per evaluation when the last
    // count alone is the expectation, profiles/r04_alternate_delta.txt).
    int64_t expect_v = (h->adaptive_runahead && local_vec) ? std::max(expect_slot[0], expect_slot[1]) : 0;
    if (h->force_expect >= 0 && local_vec) expect_v = h->force_expect;
    h->force_expect = -1;
    expect = expect_v;
    pub_from = (int32_t)std::min<int64_t>(expect, INT32_MAX);
    int nlsqr = 0;
    for (int l = 0; l < NL; ++l) {
      Lane& L = lanes[l];
      prog[l] = &h->prog_dev[l];
      h->prog_host[l].iter = 0;
      h->prog_host[l].done = 0;
      L.st_dev = h->hstats_dev + (L.st - h->hstats);
      *L.st = fpsq_stats{};
      if (L.kind == LANE_LSQR) {
        any_lsqr = true;
        LsqrState* S = h->lsqr[nlsqr];
        L.state_alt2 = reinterpret_cast<LsqrState*>(h->state3[0]) + nlsqr;
        L.state_alt = h->lsqr_alt[nlsqr++];
        L.state = S;
        L.ctl = &S->ctl;
        L.itmax = o.ls_itmax == 0 ? n + m : o.ls_itmax;
        lsP[nlsqr - 1] = LsqrParams{L.lambda, o.ls_atol, o.ls_rtol, o.ls_axtol, o.ls_btol, o.ls_etol, o.ls_conlim, L.itmax,
                                    pub_from};
        lsS[nlsqr - 1] = S;
      } else if (L.kind == LANE_MINRES) {
        MinresState* S = h->minres;
        L.state_alt = h->minres_alt;
        L.state = S;
        L.ctl = &S->ctl;
        L.ctlT = &S->ctlT;
        L.itmax = o.ne_itmax == 0 ? 2 * m : o.ne_itmax;
        // (its stopping tests of iteration k run one product later than the other recurrences': see the main loop)
        mrP = MinresParams{L.lambda, o.ne_atol, o.ne_rtol, o.ne_etol, o.ne_conlim, L.itmax, std::max(pub_from - 1, 0)};
        mrS = S;
        minres_lane = l;
      } else if (L.kind == LANE_LNLQ) {
        LnlqState* S = h->lnlq;
        L.state_alt = h->lnlq_alt;
        L.state_alt2 = h->state3[2];
        L.state = S;
        L.ctl = &S->ctl;
        // pass k of lnlq!'s loop is completed (and tested) by the step after the A' product of iteration k + 1
        L.itmax = (o.ln_itmax == 0 ? n + m : o.ln_itmax) + 1;
        lqP = LnlqParams{L.delta != 0.0 ? 1.0 / L.delta : 1.0, o.ln_atol, o.ln_rtol, L.xsign, L.itmax - 1, NL == 2 ? 1 : 0,
                         std::max(pub_from - 1, 0)};
        lqS = S;
      } else {
        CraigState* S = h->craig;
        L.state_alt = h->craig_alt;
        L.state_alt2 = h->state3[1];
        L.state = S;
        L.ctl = &S->ctl;
        L.itmax = o.ln_itmax == 0 ? n + m : o.ln_itmax;
        const bool reg = L.delta != 0.0;
        crP = CraigParams{reg ? 1.0 / L.delta : 1.0, reg ? 1.0 : 0.0, o.ln_atol, o.ln_rtol, o.ln_btol, o.ln_conlim,
                          L.xsign, L.itmax, NL == 2 ? 1 : 0, pub_from};
        crS = S;
      }
      if (!L.ctlT) L.ctlT = L.ctl;
      itmax_all = std::max(itmax_all, L.itmax);
    }
    // Riding steps (two LSQR / CRAIG lanes; one GPU or the halo-sharded layout): instead of a one-workgroup k_step launch
    // behind every product, the two steps are handed to the NEXT product launch, where leader workgroups compute them and the
    // others pick the coefficients up on their way to the row epilogue (k_spmv_atl, k_spmv_rgcs<.., LEAD>).  Such a step reads
    // the lane's current state copy and writes the other one; the lane's pointers (state, ctl) switch to it once the launch
    // is enqueued.
    lead = NL == 2 && (!h->comm || h->halo) && h->ride_lead && h->AT.padded && (h->AT.sorted || h->AT.col16) && h->RA.ok;
    // (a MINRES lane -- solve_two_extras -- on one GPU only: its sums run over row-sharded m-vectors)
    for (int l = 0; l < NL; ++l)
      lead = lead && (lanes[l].kind == LANE_LSQR || is_ln(lanes[l].kind) || (lanes[l].kind == LANE_MINRES && !h->comm));
    // fast start: the CRAIG lane whose right-hand side the LSQR start-up product forms
    for (int l = 0; l < NL; ++l)
      if (is_ln(lanes[l].kind) && lanes[l].affine_shift && any_lsqr && NL == 2 && local_vec) affine_lane = l;
    // Single GPU: the vector updates ride in the product launches (run_fused_updates).  An update may only read what
    // its host product reads: the LSQR x/w update of iteration it-1 (reads the short pair) goes with the A' product of
    // iteration it; CRAIG's updates of iteration it (read the long pair and the OLD short pair) go with the A product,
    // which therefore writes the alternate short pair (ping-pong).  The same holds for a row-sharded handle in halo mode
    // (every vector a rank updates is its own).  Sharded with replicated n-vectors: separate update launch, in place.
    fuse_upd = local_vec;
    split_steps = h->comm && !h->halo;
    pa_last = h->pS2;
    // (a halo-sharded handle: when its sums over the ranks need no launch of their own and -- for now -- no row of its window is
    // shared with a neighbour: a communicator of one, a block-diagonal Jacobian)
    can_fuse = NL == 2 && h->fuse_ok && h->at_xcd && lead && fuse_upd && minres_lane < 0 && !h->ride_break &&
               (!h->comm || (h->halo && insum(h) && (h->ovl + h->ovr == 0 || (h->fuse_halo_ok && h->fuse_halo_on))));
    look = std::max(1, o.lookahead);
    // several iterations per launch: LSQR / CRAIG lanes of a single-GPU handle whose iterations may share a launch at all
    can_multi = can_fuse && !h->comm && h->multi_ok && h->multi_max > 1 && h->fuse_probe_at == 0;
    for (int l = 0; l < NL; ++l) can_multi = can_multi && (lanes[l].kind == LANE_LSQR || lanes[l].kind == LANE_CRAIG);
    LPalt = can_multi ? h->LP2 : nullptr;
    if (can_multi) hipMemsetAsync(h->mz_hdone, 0, 8, s);  // (nobody has ended yet)
  }

  // ------------------------------------------------------------------ the steps behind a product
  // hands the pending steps to a stand-alone launch (needed whenever the host or a gated kernel must see their effect now)
  int flush_pend(bool sharded) {
    if (!have_pend) return 0;
    have_pend = false;
    if (h->comm) {  // (prepared -- gathered, or numbered -- when they were handed over)
      launch_step_raw(h, pend[0], pend[1], pend_xseq);
      return 0;
    }
    return launch_step(h, pend[0], pend[1], sharded);
  }
  // after a product launch that carried the pending steps: the lanes live in their other state copies now
  void adopt_pend() {
    for (int l = 0; l < NL; ++l)
      if (pend[l].kind != STEP_NONE) lane_swap_state(lanes[l]);
    have_pend = false;
  }
  // the steps behind a product: riding in the next product launch when both lanes have one, else their own launch now
  int post_step(const StepArgs& a0, const StepArgs& a1, bool sharded) {
    if (lead && a0.kind != STEP_NONE && a1.kind != STEP_NONE) {
      pend[0] = a0;
      pend[1] = a1;
      pend_xseq = 0;
      if (h->comm) {
        if (int rc = prepare_step(h, pend[0], pend[1], sharded)) return rc;
        pend_xseq = h->last_xseq;
      }
      have_pend = true;
      return 0;
    }
    return launch_step(h, a0.kind ? a0 : a1, a0.kind ? a1 : none, sharded);
  }
  // the pending steps as the next product launch takes them (null: nothing pending)
  const StepArgs* pre_args(bool for_at) {
    h->ride_xseq = have_pend ? pend_xseq : 0;  // (the launch that takes the steps also takes their exchange's number: launch_spmv)
    if (!have_pend) return nullptr;
    for (int l = 0; l < NL; ++l) {
      pend[l].state = lanes[l].state;
      pend[l].state_out = lanes[l].state_alt;
      pend[l].prod_ctl_off = for_at && lanes[l].kind == LANE_MINRES ? (int32_t)(offsetof(MinresState, ctlT) / 8) : 0;
    }
    return pend;
  }

  // ------------------------------------------------------------------ builders: update segments and step arguments
  // Where the vector update of iteration k leaves its squared-norm partials (read by the step behind the NEXT A product):
  // halves alternate, because the A' launch of iteration k + 1 carries both that step -- riding, computed by sixteen
  // leaders of which any may be late -- and the update of iteration k + 1, whose workgroups only wait for the record of
  // their own XCC's leader before they write.  (CRAIG's update rides one launch later than the step that reads its
  // partials and would be safe in one array; it follows the same parity so that a sharded step gathers one range.)
  double* upd_part(int l, int64_t k) const { return (k & 1) ? h->pWalt[l] : h->pW[l]; }
  UpdSeg lsqr_upd_seg(int l, int64_t it_of_update) const {
    UpdSeg u{};
    u.kind = UPD_LSQR;
    u.it = (int)it_of_update;
    u.ctl = lanes[l].ctl;
    u.src = SPcur;
    u.lane = l;
    u.nblk = gm;
    u.a = lanes[l].x;
    u.b = h->Lw[l];
    u.len = m;
    u.partials = upd_part(l, it_of_update);
    return u;
  }
  UpdSeg lsqr_winit_seg(int l) const {  // w_1 = v_1, x_0 = 0
    UpdSeg u{};
    u.kind = UPD_LSQR_WINIT;
    u.it = 0;
    u.ctl = lanes[l].ctl;
    u.src = SP;
    u.lane = l;
    u.nblk = gm;
    u.a = lanes[l].x;
    u.b = h->Lw[l];
    u.len = m;
    u.partials = upd_part(l, 0);
    return u;
  }
  // the least-norm lane's updates of iteration `it`: long (x, w2) and short (w, y)
  void ln_upd_segs(int l, UpdSeg& lng, UpdSeg& sht) const {
    const Lane& L = lanes[l];
    UpdSeg u{};
    u.kind = L.kind == LANE_LNLQ ? UPD_LNLQ_LONG : L.delta != 0.0 ? UPD_CRAIG_LONG_REG : UPD_CRAIG_LONG;
    u.it = (int)it;
    u.ctl = L.ctl;
    u.src = LP;
    u.lane = l;
    u.nblk = gn;
    u.a = L.x;
    u.b = h->Cw2;
    u.len = n;
    lng = u;
    UpdSeg v{};
    v.kind = L.kind == LANE_LNLQ ? UPD_LNLQ_SHORT : UPD_CRAIG_SHORT;
    v.it = (int)it;
    v.ctl = L.ctl;
    v.src = SPcur;
    v.lane = l;
    v.nblk = gm;
    v.a = h->Cw;
    v.b = L.y;
    v.len = m;
    v.partials = upd_part(l, it - 1);
    sht = v;
  }
  // MINRES stage segments of iteration `k` (the Lanczos vector under construction sits in lane l of `pair`)
  UpdSeg minres_seg(int stage, int64_t k, double* pair) const {
    const int l = minres_lane;
    UpdSeg u{};
    u.kind = stage == 1 ? UPD_MINRES_E1 : stage == 2 ? UPD_MINRES_E2 : UPD_MINRES_E3;
    u.it = (int)k;
    u.ctl = lanes[l].ctl;
    u.src = pair;
    u.lane = l;
    u.nblk = gm;
    u.len = m;
    double* r2 = h->Mr[k % 2];
    double* r1 = h->Mr[(k + 1) % 2];  // also receives the new r2
    double* w1 = h->Mw[k % 2];        // w_{k-2}, overwritten by w_k
    double* w2 = h->Mw[(k + 1) % 2];
    if (stage == 1) {
      u.a = r1;
      u.b = r2;
      u.partials = h->pE3;
    } else if (stage == 2) {
      u.a = r2;
      u.b = r1;
      u.c = w2;
      u.d = w1;
      u.partials = h->pW[l];
    } else {
      u.a = w1;
      u.b = lanes[l].x;
      u.partials = h->pWalt[l];  // (rides in the launch whose leaders compute step B from E2's partials in pW[l])
    }
    return u;
  }
  StepArgs minres_step(int kind, int64_t k) const {  // B: after E2 (partials in pW); C: after E3 (partials in pWalt)
    const int l = minres_lane;
    return step_args(kind, lanes[l], (int)k, kind == STEP_MINRES_C ? h->pWalt[l] : kind == STEP_MINRES_A ? h->pE3 : h->pW[l], gm,
                     nullptr, 0, prog[l]);
  }
  // lane l's step behind the A' product of iteration `it` (npT partials per lane)
  StepArgs step_after_at(int l, int npT) const {
    const Lane& L = lanes[l];
    if (L.kind == LANE_MINRES) return it > 1 ? minres_step(STEP_MINRES_C, it - 1) : none;  // the stopping tests of iteration it - 1
    return step_args(lane_kind_after_at(L), L, (int)it, h->pS + (size_t)l * h->strT, npT, nullptr, 0, prog[l]);
  }
  // ... and behind the A product
  StepArgs step_after_a(int l) const {
    const Lane& L = lanes[l];
    if (L.kind == LANE_MINRES) return minres_step(STEP_MINRES_A, it);
    return step_args(lane_kind_after_a(L), L, (int)it, pa_last + (size_t)l * h->strA, nbA,
                     L.kind == LANE_LNLQ ? nullptr : upd_part(l, it - 1), gm, prog[l]);
  }
  bool all_done() const {
    for (int l = 0; l < NL; ++l)
      if (!load_progress(&h->prog_host[l]).done) return false;
    return true;
  }

  // ------------------------------------------------------------------ start-up
  // parameters, right-hand sides, beta_1 (one launch), then (LSQR) alpha_1 and w_1
  int startup() {
    StepArgs b0 = none, b1 = none;
    LoadSeg ld[2] = {};
    ZeroArgs z{};
    int nzblk = 0;
    for (int l = 0; l < NL; ++l) {
      Lane& L = lanes[l];
      double* pe = L.kind == LANE_LSQR ? (l == 0 ? h->pE : h->pE2) : h->pEm[l];
      LoadSeg& g = ld[l];
      g.src = L.rhs;
      g.scale = L.rhs_scale;
      g.lane = l;
      g.partials = pe;
      if (L.kind == LANE_LSQR) {
        // x = 0 is written by the w_1 start-up update (also when the recurrence ends at start-up)
        g.dst = LP;
        g.len = n;
        g.sum_len = n_owned(h);
        g.nblk = L.preloaded ? 0 : gn;  // fast start: the caller wrote the lane and the ||rhs||^2 partials already
        (l == 0 ? b0 : b1) = step_args(STEP_LSQR_BEGIN, L, 0, pe, gn, nullptr, 0, prog[l]);
      } else if (L.kind == LANE_MINRES) {
        // r1 = r2 = b: r2 sits in Mr[1] (iteration 1 reads r2 from Mr[it % 2]) and in the short pair's lane
        g.dst = SP;
        g.dst2 = h->Mr[1];
        g.len = m;
        g.sum_len = m;
        g.nblk = gm;
        z.p[0] = L.x;
        z.p[1] = h->Mw[0];
        z.p[2] = h->Mw[1];
        z.p[3] = h->Mr[0];
        z.n[0] = z.n[1] = z.n[2] = z.n[3] = m;
        nzblk = gm;
      } else {
        if (L.affine_shift) {  // fast start: the lane receives `shift`; the start-up product turns it into -(A z - shift)
          g.src = L.affine_shift;
          g.scale = 1.0;
        }
        g.dst = SP;
        g.len = m;
        g.sum_len = m;
        g.nblk = gm;
        z.p[0] = L.x;
        z.n[0] = n;
        z.p[1] = L.y;
        z.n[1] = m;
        z.p[2] = h->Cw;
        z.n[2] = m;
        if (L.delta != 0.0) {
          z.p[3] = h->Cw2;
          z.n[3] = n;
        }
        nzblk = gn;
      }
    }
    ht_mark(h, 3);
    hipLaunchKernelGGL(k_startup<NL>, dim3(h->startup_qg.nblk + ld[0].nblk + ld[1].nblk + nzblk), dim3(kBlock), 0, s, lsS[0],
                       lsP[0], lsS[1], lsP[1], crS, crP, mrS, mrP, lqS, lqP, ld[0], ld[1], z, nzblk, h->startup_qg);
    h->startup_qg.nblk = 0;
    h->launches++;
    bool ln_begun = false, minres_begun = false;
    if (any_lsqr) {
      // v~_1 = B'u_1 = A u~_1 / beta_1 for the LSQR lanes.  The CRAIG lane is parked by ctl.skip -- unless its
      // right-hand side is still to be formed (fast start): then it rides along with the constant pair (-1, +1):
      // SP[.][l] <- -A z + shift, and the norm partials of the launch are those of its right-hand side.
      const LaneCtl* s0c = c0();
      const LaneCtl* s1c = c1();
      if (affine_lane == 0) s0c = h->ctl_mp;
      if (affine_lane == NL - 1 && affine_lane >= 0) s1c = h->ctl_mp;
      if (lead && insum(h) && fuse_upd) {
        // riding steps: beta_1 of the LSQR lanes goes with THIS product's leaders too; a lane without a step of its own has the
        // control block it brings to this product published as it is (ride_leader, kind NONE)
        pend[0] = b0;
        pend[1] = b1;
        pend_xseq = 0;
        if (h->comm) {  // (halo mode: ||rhs||^2 runs over the ranks' owned parts)
          if (int rc = prepare_step(h, pend[0], pend[1], /*sharded=*/true)) return rc;
          pend_xseq = h->last_xseq;
        }
        have_pend = true;
        const StepArgs* pre = pre_args(false);
        if (pend[0].kind == STEP_NONE) pend[0].state = const_cast<LaneCtl*>(s0c);
        if (pend[1].kind == STEP_NONE) pend[1].state = const_cast<LaneCtl*>(s1c);
        launch_spmv<NL>(h, TAG_A, LP, SP, SP, s0c, s1c, h->pS2, seg_none(), seg_none(), false, pre);
        adopt_pend();
      } else {
        if (int rc = launch_step(h, b0.kind ? b0 : b1, b0.kind ? b1 : none, /*sharded=*/h->halo)) return rc;
        launch_spmv<NL>(h, TAG_A, LP, SP, SP, s0c, s1c, h->pS2);
      }
      StepArgs s0 = none, s1 = none;
      UpdSeg w0 = seg_none(), w1 = seg_none();
      for (int l = 0; l < NL; ++l) {
        Lane& L = lanes[l];
        if (L.kind != LANE_LSQR) continue;
        (s0.kind ? s1 : s0) = step_args(STEP_LSQR_BEGIN2, L, 0, h->pS2 + (size_t)l * h->strA, nbA, nullptr, 0, prog[l]);
        (w0.nblk ? w1 : w0) = lsqr_winit_seg(l);
      }
      if (fuse_upd && !s1.kind) {
        // the least-norm lane's beta_1 step shares the launch (it un-parks the lane: must follow the start-up product)
        for (int l = 0; l < NL; ++l)
          if (is_ln(lanes[l].kind)) {
            if (l == affine_lane)  // ||rhs||^2 came out of the start-up product
              s1 = step_args(lane_begin_kind(lanes[l]), lanes[l], 0, h->pS2 + (size_t)l * h->strA, nbA, nullptr, 0, prog[l]);
            else
              s1 = step_args(lane_begin_kind(lanes[l]), lanes[l], 0, h->pEm[l], gm, nullptr, 0, prog[l]);
            ln_begun = true;
          }
        // (riding steps: a MINRES lane's beta_1 step -- it un-parks the lane: must follow the start-up product -- pairs up too)
        if (!s1.kind && lead && !h->comm && minres_lane == 1) {
          s1 = step_args(STEP_MINRES_BEGIN, lanes[1], 0, h->pEm[1], gm, nullptr, 0, prog[1]);
          minres_begun = true;
        }
      }
      if (affine_lane >= 0) {  // keep A z - shift = -rhs before the first A product overwrites the lane
        UpdSeg u = seg_none();
        u.kind = UPD_NEG_COPY;
        u.src = SP;
        u.lane = affine_lane;
        u.nblk = gm;
        u.a = lanes[affine_lane].affine_out;
        u.len = m;
        (w0.nblk ? w1 : w0) = u;
      }
      // (riding steps: alpha_1 / the least-norm lane's beta_1 go with the first A' product of the loop)
      if (lead) {
        if (int rc = post_step(s0, s1, /*sharded=*/true)) return rc;
      } else {
        if (int rc = launch_step(h, s0, s1, /*sharded=*/true)) return rc;
      }
      if (fuse_upd) {  // w_1 rides in the first A' product
        winit[0] = w0;
        winit[1] = w1;
      } else {
        launch_updates<NL>(h, w0, w1, seg_none());
      }
    }
    for (int l = 0; l < NL; ++l)
      if (is_ln(lanes[l].kind) && !ln_begun)
        if (int rc = launch_step(h, step_args(lane_begin_kind(lanes[l]), lanes[l], 0, h->pEm[l], gm, nullptr, 0, prog[l]), none,
                                 /*sharded=*/true))
          return rc;
    if (minres_lane >= 0 && !minres_begun)  // (un-parks the lane: must follow the LSQR lane's start-up product)
      if (int rc = launch_step(h, step_args(STEP_MINRES_BEGIN, lanes[minres_lane], 0, h->pEm[minres_lane], gm, nullptr, 0,
                                            prog[minres_lane]),
                               none, /*sharded=*/true))
        return rc;
    return 0;
  }

  // ------------------------------------------------------------------ one joint iteration
  // A MINRES lane (solve_two_extras) shares the two products of an iteration with the other recurrence: tmp = A' r2
  // rides in the A' product, q = (A tmp + lambda r2) / beta in the A product; then its element-wise stages E1 -> scalar
  // step A -> E2 -> step B.  Stage E3 (w, x) only needs the scalars of step B: it rides in the A' product of the NEXT
  // iteration and its stopping tests (step C) share the step launch that follows that product -- one short
  // element-wise launch and one scalar launch more per iteration than the other recurrence alone.
  //
  // first half-step of every lane: the A' product (LSQR's update of the previous iteration and MINRES' stage E3 riding), its steps
  int half_step_at() {
    lu[0] = lu[1] = seg_none();
    nlu = 0;
    if (it > 1) {
      for (int l = 0; l < NL; ++l)
        if (lanes[l].kind == LANE_LSQR) lu[nlu++] = lsqr_upd_seg(l, it - 1);
    } else {
      lu[0] = winit[0];  // fused runs: w_1 = v_1 (empty segments otherwise)
      lu[1] = winit[1];
    }
    // MINRES: stage E3 of the PREVIOUS iteration (w, x and ||x||^2 for its stopping tests)
    if (minres_lane >= 0 && it > 1) {
      const UpdSeg e3 = minres_seg(3, it - 1, SPcur);
      if (fuse_upd) lu[nlu < 2 ? nlu : 1] = e3;
      else launch_updates<NL>(h, e3, seg_none(), seg_none());
    }
    int npT = 0;
    if (fuse_upd) {
      const StepArgs* pre = pre_args(true);
      if (int rc = at_product<NL>(h, SPcur, LP, t0(), t1(), h->pS, &npT, lu[0], lu[1], pre)) return rc;
      if (pre) adopt_pend();
    } else {
      if (int rc = at_product<NL>(h, SPcur, LP, t0(), t1(), h->pS, &npT)) return rc;
    }
    StepArgs sa[2] = {none, none};
    for (int l = 0; l < NL; ++l) sa[l] = step_after_at(l, npT);
    // sums over n-vectors: replicated (no all-reduce) unless the n-vectors are column windows (halo mode); MINRES' sums
    // run over (row-sharded) m-vectors
    const bool sh0 = lanes[0].kind == LANE_MINRES ? true : h->halo;
    const bool sh1 = lanes[NL - 1].kind == LANE_MINRES ? true : h->halo;
    if (!h->comm) return post_step(sa[0], NL == 2 ? sa[1] : none, false);
    if (lead && sh0 && sh1) return post_step(sa[0], sa[1], true);  // (halo mode, LSQR / CRAIG lanes: both steps sum gathered n-sums)
    if (NL == 2 && split_steps && sh0 != sh1) {
      if (int rc = launch_step(h, sa[0], none, sh0)) return rc;
      return launch_step(h, sa[1], none, sh1);
    }
    if (NL == 2) return launch_step(h, sa[0].kind ? sa[0] : sa[1], sa[0].kind ? sa[1] : none, sa[0].kind ? sh0 : sh1, sa[0].kind ? sh1 : 0);
    if (sa[0].kind) return launch_step(h, sa[0], none, sh0);
    return 0;
  }
  // second half-step: the A product (the least-norm lane's updates of this iteration riding), its steps, MINRES' stages
  int half_step_a() {
    UpdSeg cu[2] = {seg_none(), seg_none()};
    for (int l = 0; l < NL; ++l)
      if (is_ln(lanes[l].kind)) ln_upd_segs(l, cu[0], cu[1]);
    if (fuse_upd) {
      const StepArgs* pre = pre_args(false);
      launch_spmv<NL>(h, TAG_A, LP, SPcur, SPalt, c0(), c1(), h->pS2, cu[0], cu[1], false, pre);
      pa_last = h->pS2;
      if (pre) adopt_pend();
      std::swap(SPcur, SPalt);
    } else {
      // (at most three segments: lanes <= 2 and only one of them can be CRAIG)
      if (nlu == 2) launch_updates<NL>(h, lu[0], lu[1], seg_none());
      else launch_updates<NL>(h, lu[0], cu[0], cu[1]);
      launch_spmv<NL>(h, TAG_A, LP, SPcur, SPcur, c0(), c1(), h->pS2);
      pa_last = h->pS2;
    }
    // A MINRES lane the host has SEEN finished (a zero right-hand side -- hprod! Val(1) on a model without curvature in the
    // constraints --, or an early convergence): its stand-alone launches would exit at once, ~3.5 us each; skipped.  One GPU
    // only: sharded, every rank would have to see it at the same iteration.  (Its riding / shared steps stay: they cost nothing.)
    const bool mdead = minres_lane >= 0 && !h->comm && load_progress(&h->prog_host[minres_lane]).done;
    // MINRES: E1 on q (now in the current pair's lane) before its scalar step A -- with riding steps on one GPU, E1, the step
    // and E2 are ONE launch (k_minres_mid: every workgroup does E1, waits for the leader's record, does E2 on the same elements)
    // (every workgroup of that launch must be resident at once -- the waiting ones hold their slots: the grid has to fit the
    // device with a margin for whatever else runs; should another kernel take the slots all the same, the bounded waits end the
    // call, ride_failed() switches the merge off and the call is repeated on three launches)
    const bool mmid = minres_lane >= 0 && !mdead && lead && NL == 2 && !h->comm && h->minres_merge && h->mm_ptag != nullptr &&
                      4 * (1 + gm) <= 3 * h->mmid_cap;
    if (minres_lane >= 0 && !mdead && !mmid) launch_updates<NL>(h, minres_seg(1, it, SPcur), seg_none(), seg_none());
    StepArgs sb[2] = {none, none};
    for (int l = 0; l < NL; ++l) sb[l] = step_after_a(l);
    if (minres_lane >= 0 && lead && NL == 2) {
      // MINRES' step A must run before E2; the other lane's step is only needed by the NEXT A' launch (its epilogue and its
      // riding update) and waits for MINRES' step B to ride there with it
      if (mmid) {
        const UpdSeg e1 = minres_seg(1, it, SPcur), e2 = minres_seg(2, it, SPcur);
        hipLaunchKernelGGL(k_minres_mid, dim3(1 + e1.nblk), dim3(kBlock), 0, s, e1, e2, sb[minres_lane], h->mm_ptag, h->ride_rec2,
                           (unsigned int)++h->ride_seq, reinterpret_cast<unsigned long long*>(h->hscal_dev + 15));
        h->launches++;
        h->mmid_launches++;
      } else if (!mdead) {
        if (int rc = launch_step(h, sb[minres_lane], none, /*sharded=*/true)) return rc;
        launch_updates<NL>(h, minres_seg(2, it, SPcur), seg_none(), seg_none());
      }
      StepArgs pair[2];
      pair[minres_lane] = minres_step(STEP_MINRES_B, it);
      pair[1 - minres_lane] = sb[1 - minres_lane];
      return post_step(pair[0], pair[1], true);
    }
    if (!h->comm || lead) {
      if (int rc = post_step(sb[0], NL == 2 ? sb[1] : none, true)) return rc;
    } else {
      if (int rc = launch_step(h, sb[0], sb[1], /*sharded=*/true)) return rc;
    }
    if (minres_lane >= 0 && !mdead) {  // E2 -> scalar step B (beta, the rotation, the coefficients of E3 and of the next products)
      launch_updates<NL>(h, minres_seg(2, it, SPcur), seg_none(), seg_none());
      if (int rc = launch_step(h, minres_step(STEP_MINRES_B, it), none, /*sharded=*/true)) return rc;
    }
    return 0;
  }

  // One launch for both half-steps (k_iter_fused): the A' product with what rides in it, the steps behind it (mid leaders), the A
  // product with what rides in it.  Needs the previous product's steps pending (they are the head leaders' work).
  int iteration_fused() {
    lu[0] = lu[1] = seg_none();
    nlu = 0;
    if (it > 1) {
      for (int l = 0; l < NL; ++l)
        if (lanes[l].kind == LANE_LSQR) lu[nlu++] = lsqr_upd_seg(l, it - 1);
    } else {
      lu[0] = winit[0];
      lu[1] = winit[1];
    }
    const StepArgs* pre = pre_args(true);
    StepArgs sh[2] = {pre[0], pre[NL - 1]}, sm[2];
    // halo-sharded with rows shared with the neighbours: the exchange and the finish of the overlap rows ride in this launch
    // (fuse_halo_wg); the finish workgroups' partials follow the blocks'
    // (the HALO kernel also whenever the leaders exchange -- its leaders have the exchange compiled in; no shared rows: no halo workgroups)
    const bool shared_rows = h->comm && h->halo && h->ovl + h->ovr > 0;
    const bool with_halo = shared_rows || (h->comm && h->halo && insum_table(h) != nullptr);
    FuseHalo fh{};
    HaloRows hr{};
    if (with_halo && !shared_rows) hr = HaloRows{0, h->n, h->halo_raw};
    if (shared_rows) {
      const int64_t t = h->ovl + h->ovr;
      double* rl = h->halo_recv + (size_t)(h->halo_calls++ & 1) * (size_t)t * 2;
      if (!h->comm->halo_fused_args(rl, h->ovl, h->ovr, fh)) {
        h->err = "internal: one-launch iteration on a communicator without in-launch halo exchange";
        return FPSQ_ERR_STATE;
      }
      fh.raw = h->halo_raw;
      fh.recv = rl;
      fh.tl = h->ovl;
      fh.tr = h->ovr;
      fh.tail0 = h->n - h->ovr;
      fh.gf = h->halo_gf;
      fh.nwg = (2 * kHaloCopy + h->halo_gf + 7) / 8 * 8;
      fh.depL = h->fz_depL;
      fh.depR = h->fz_depR;
      hr = HaloRows{h->ovl, h->n - h->ovr, h->halo_raw};
    }
    for (int l = 0; l < NL; ++l) {
      sm[l] = step_after_at(l, h->AT.nblk + (shared_rows ? h->halo_gf : 0));
      sm[l].state = sh[l].state_out;  // (what the head step leaves: the mid leaders recompute it, nobody reads this pointer)
      sm[l].state_out = lanes[l].state_alt2;
      sm[l].prod_ctl_off = 0;
    }
    uint32_t mid_xseq = 0;
    if (h->comm) {  // (halo mode: the mid leaders' sums run over the ranks -- the exchange behind the head steps')
      if (int rc = prepare_step(h, sm[0], sm[1], true)) return rc;
      mid_xseq = h->last_xseq;
    }
    UpdSeg cu[2] = {seg_none(), seg_none()};
    for (int l = 0; l < NL; ++l)
      if (is_ln(lanes[l].kind)) ln_upd_segs(l, cu[0], cu[1]);
    FuseGrid fg{};
    fg.bpx = (h->AT.nblk + 7) / 8;
    {
      const int R = h->resident_wgs - kRideCand;
      const int n2 = !h->atl_two || h->AT.nblk <= R ? 0 : std::min(R, h->AT.nblk - R);
      const int n2e = std::min(n2 / 8, fg.bpx / 2);
      fg.n2 = 8 * n2e;
      fg.nwg_t = 8 * (fg.bpx - n2e);
    }
    fg.nupd_t = (lu[0].nblk + lu[1].nblk + 7) / 8 * 8;
    fg.gpx = (h->RA.view.ng + 7) / 8;
    fg.rot = h->fuse_rotate;
    RideArgs ra{}, rb{};
    ra.rec = h->ride_rec;
    ra.want = (unsigned int)++h->ride_seq;
    ra.pub = ra.want;
    ra.err = reinterpret_cast<unsigned long long*>(h->hscal_dev + 15);
    ra.delay = h->ride_delay;
    rb = ra;
    rb.rec = h->ride_rec2;
    rb.delay = h->ride_delay_mid;
    ra.xseq = h->ride_xseq;  // (pre_args: the pending head steps' exchange)
    ra.xt = ra.xseq ? insum_table(h) : nullptr;
    rb.xseq = mid_xseq;
    rb.xt = mid_xseq ? insum_table(h) : nullptr;
    // (leaders that wait for a late peer keep everything behind them waiting: blocks, row groups, the other leader set, updates)
    const int more = ra.xt || rb.xt ? h->comm->wait_more() : 0;
    ra.more = rb.more = more;
    FuseArgs fz{};
    fz.more = more;
    fz.blkflag = h->fz_flag;
    fz.ptag = h->fz_ptag;
    fz.dep = h->fz_dep;
    fz.dep2 = shared_rows ? h->fz_dep2 : nullptr;
    fz.want = ra.want;
    fz.pub = h->fuse_break ? ~ra.want : ra.want;
    fz.err = ra.err;
    double* part_a = pa_last == h->pS2 ? h->pS2b : h->pS2;  // (not the array this launch's leaders read)
    const dim3 grid(kRideCand + fg.nwg_t + fh.nwg + kRideCand + fg.nupd_t + 8 * fg.gpx + cu[0].nblk + cu[1].nblk);
    if (h->fuse_probe_at > 0 && h->fused_total + 1 == h->fuse_probe_at) {  // developer probe: this launch leaves time stamps
      h->fuse_probe_grid = (int)grid.x;
      h->fuse_probe_layout = {kRideCand, fg.nwg_t + fh.nwg, kRideCand, 8 * fg.gpx, fg.nupd_t, cu[0].nblk + cu[1].nblk};
      if (dalloc(h, &h->fuse_probe_buf, (size_t)grid.x * 4) == 0) {
        hipMemsetAsync(h->fuse_probe_buf, 0, (size_t)grid.x * 32, h->stream);
        fz.dbg = h->fuse_probe_buf;
      }
    }
    h->fused_total++;
#define FPSQ_LAUNCH_FUSED(...)                                                                                                        \
    launch_product(h, k_iter_fused<__VA_ARGS__>, grid, h->AT.view(), h->RA.view, (const double*)SPcur, LP, SPalt, part_a, h->strA, fg, \
                   lu[0], lu[1], cu[0], cu[1], sh[0], sh[1], sm[0], sm[1], ra, rb, fz, hr, fh)
    if (h->AT.sorted && with_halo) FPSQ_LAUNCH_FUSED(true, true);
    else if (h->AT.sorted) FPSQ_LAUNCH_FUSED(true, false);
    else if (with_halo) FPSQ_LAUNCH_FUSED(false, true);
    else FPSQ_LAUNCH_FUSED(false, false);
#undef FPSQ_LAUNCH_FUSED
    h->launches++;
    h->spmv_launches++;
    h->prod_a[1]++;
    h->prod_at[1]++;
    h->fused_launches++;
    pa_last = part_a;
    // the lanes live in their third copies now; the other two are free for the next launch's two steps
    for (int l = 0; l < NL; ++l) {
      Lane& L = lanes[l];
      void* s0 = L.state;
      L.state = L.state_alt2;
      L.state_alt2 = L.state_alt;
      L.state_alt = s0;
      L.ctl = reinterpret_cast<LaneCtl*>(L.state);
      L.ctlT = L.ctl;
    }
    have_pend = false;
    std::swap(SPcur, SPalt);
    StepArgs sb[2] = {none, none};
    for (int l = 0; l < NL; ++l) sb[l] = step_after_a(l);
    return post_step(sb[0], sb[1], true);
  }

  // K joint iterations in ONE launch (k_iter_multi, fpsq_multi.hip.h): iterations it .. it + K - 1.  Needs what iteration_fused
  // needs (the previous product's steps pending) and it >= 2 (iteration 1 carries the start-up's w_1 segments).
  int iteration_multi(int K) {
    MultiArgs M{};
    M.K = K;
    M.it0 = (int32_t)it;
    const StepArgs* pre = pre_args(true);
    int nut = 0;
    for (int l = 0; l < NL; ++l) {
      M.sh[l] = pre[l];
      M.sm[l] = step_after_at(l, h->AT.nblk);
      M.sm[l].state = nullptr;
      M.sm[l].state_out = nullptr;
      M.sm[l].prod_ctl_off = 0;
      M.commit[l][0] = lanes[l].state_alt;
      M.commit[l][1] = lanes[l].state_alt2;
      M.pw[l][0] = h->pW[l];
      M.pw[l][1] = h->pWalt[l];
      M.p1seg[l] = 3;
      if (lanes[l].kind == LANE_LSQR) {
        M.p1seg[l] = nut;
        M.ut[nut++] = lsqr_upd_seg(l, it - 1);
      }
    }
    for (int k = nut; k < 2; ++k) M.ut[k] = seg_none();
    M.ua[0] = M.ua[1] = seg_none();
    for (int l = 0; l < NL; ++l)
      if (is_ln(lanes[l].kind)) ln_upd_segs(l, M.ua[0], M.ua[1]);
    M.n1 = gm;
    FuseGrid& fg = M.fg;
    fg.bpx = (h->AT.nblk + 7) / 8;
    {
      const int R = h->resident_wgs - kRideCand;
      const int n2 = !h->atl_two || h->AT.nblk <= R ? 0 : std::min(R, h->AT.nblk - R);
      const int n2e = std::min(n2 / 8, fg.bpx / 2);
      fg.n2 = 8 * n2e;
      fg.nwg_t = 8 * (fg.bpx - n2e);
    }
    // (few real update workgroups, each walking several virtual ones: see k_iter_multi)
    fg.nupd_t = std::min((M.ut[0].nblk + M.ut[1].nblk + 7) / 8 * 8, h->multi_upd_t);
    fg.gpx = (h->RA.view.ng + 7) / 8;
    fg.rot = h->fuse_rotate;
    // CRAIG's long update one iteration later, behind the next A' blocks (FPSQ_MULTI_DEFER_LONG=0: with the short one)
    M.nlong = h->multi_defer_long ? (M.ua[0].nblk + 7) / 8 * 8 : 0;
    M.nupd_a = std::min(((M.nlong ? 0 : M.ua[0].nblk) + M.ua[1].nblk + 7) / 8 * 8, h->multi_upd_a);
    M.per_iter = kRideCand + fg.nwg_t + M.nlong + kRideCand + 8 * fg.gpx + fg.nupd_t + M.nupd_a;
    M.seq0 = (uint32_t)(h->ride_seq + 1);
    h->ride_seq += (unsigned long long)K;
    M.sp[0] = SPcur;
    M.sp[1] = SPalt;
    M.sp0 = 0;
    M.lp[0] = LP;
    M.lp[1] = LPalt;
    M.lp0 = 0;
    M.part_last = pa_last == h->pS2 ? h->pS2b : h->pS2;  // (not the array this launch's first leaders read)
    M.pstride_a = h->strA;
    M.rec_h = h->mz_rec_h;
    M.rec_m = h->mz_rec_m;
    M.srec = h->mz_srec;
    M.flag[0] = h->fz_flag;
    M.flag[1] = h->mz_flag2;
    M.ptag[0] = h->fz_ptag;
    M.ptag[1] = h->mz_ptag2;
    for (int q = 0; q < 2; ++q) {
      M.gflag[q] = h->mz_gflag[q];
      M.atag[q] = h->mz_atag[q];
      M.utag[q] = h->mz_utag[q];
    }
    M.dep = h->fz_dep;
    M.bdep = h->mz_bdep;
    M.hdone = h->mz_hdone;
    M.err = reinterpret_cast<unsigned long long*>(h->hscal_dev + 15);
    M.delay_h = h->ride_delay;
    M.delay_m = h->ride_delay_mid;
    M.break_pub = h->fuse_break ? ~0u : 0u;
    const dim3 grid((unsigned)M.per_iter * (unsigned)K + (unsigned)M.nlong);
    if (h->AT.sorted) launch_product(h, k_iter_multi<true>, grid, h->AT.view(), h->RA.view, M);
    else launch_product(h, k_iter_multi<false>, grid, h->AT.view(), h->RA.view, M);
    h->launches++;
    h->spmv_launches++;
    h->prod_a[1] += K;
    h->prod_at[1] += K;
    h->fused_launches += K;
    h->fused_total += K;
    h->multi_launches++;
    h->multi_iters += K;
    pa_last = M.part_last;
    // the lanes live where the last iteration's mid leaders committed; the other two copies are free for the next launch
    for (int l = 0; l < NL; ++l) {
      Lane& L = lanes[l];
      void* cur = L.state;
      void* fin = M.commit[l][(K - 1) & 1];
      void* oth = M.commit[l][K & 1];
      L.state = fin;
      L.state_alt = cur;
      L.state_alt2 = oth;
      L.ctl = reinterpret_cast<LaneCtl*>(L.state);
      L.ctlT = L.ctl;
    }
    have_pend = false;
    if (K & 1) {
      std::swap(SPcur, SPalt);
      std::swap(LP, LPalt);
    }
    it += K - 1;  // (run() counted the first one)
    StepArgs sb[2] = {none, none};
    for (int l = 0; l < NL; ++l) sb[l] = step_after_a(l);
    return post_step(sb[0], sb[1], true);
  }

  // ------------------------------------------------------------------ the host's pacing
  // the gated final LSQR flush + the caller's epilogue behind iteration `it` (see the comment above)
  int enqueue_speculative() {
    if (tail == nullptr || !fuse_upd) return 0;
    UpdSeg seg[2] = {seg_none(), seg_none()};
    int ns = 0;
    for (int l = 0; l < NL; ++l)
      if (lanes[l].kind == LANE_LSQR) {
        seg[ns] = lsqr_upd_seg(l, it);
        seg[ns++].gate = lanes[NL - 1 - l].ctl;  // the other lane of the call (NL = 1: itself)
      }
    if (h->absorb_flush && ns == 1) h->pending_flush = seg[0];  // applied by the tail's first kernel (k_ys)
    else launch_updates<NL>(h, seg[0], seg[1], seg_none());
    h->gate0 = lanes[0].ctl;
    h->gate1 = lanes[NL - 1].ctl;
    const int64_t l0 = h->launches;
    const int rc = (*tail)();
    tail_launches += h->launches - l0;  // (the caller's epilogue, not the loop: fpsq_info.last_loop_launches)
    h->gate0 = h->gate1 = nullptr;
    h->pending_flush.kind = UPD_NONE;
    if (rc) return rc;
    spec_it = it;
    return 0;
  }
  // the host waits until every unfinished lane has reported iteration `target` (minus its lag) or has ended
  int wait_lanes(int64_t target) {
    for (int l = 0; l < NL; ++l) {
      if (load_progress(&h->prog_host[l]).done) continue;
      const int32_t* ddone = &lanes[l].ctl->done;
      if (int rc = wait_progress(h, l, (int)target - lane_lag(lanes[l]), ddone, lane_iter_ptr(lanes[l]))) return rc;
    }
    return 0;
  }
  // Sharded: every rank must enqueue the same collectives: decide at fixed iteration boundaries from the (replicated,
  // bitwise identical) device state, never from the timing of the progress word.  With the iteration count of the
  // previous call known (halo mode; the same on every rank) the first look is AT that count, with the gated flush
  // and epilogue already enqueued behind it: a repeating count costs no stream synchronisation inside the loop.
  int pace_sharded(bool& stop) {
    bool boundary = it == itmax_all;
    if (expect > 0) {
      if (it == expect) {
        if (int rc = flush_pend(true)) return rc;  // (the gated kernels must see this iteration's verdict)
        if (int rc = enqueue_speculative()) return rc;
        boundary = true;
      } else if (it > expect && (it - expect) % look == 0) {
        boundary = true;
      }
    } else if (it % look == 0) {
      boundary = true;
    }
    if (boundary) {
      if (int rc = flush_pend(true)) return rc;  // (so must the host; the same launches on every rank)
      HIPCHK(h, hipStreamSynchronize(s));
      if (h->comm->failed()) {  // (peer-to-peer route: a peer's record never came; nothing later in this call can be right)
        h->info.p2p_timeouts++;
        h->err = "peer-to-peer exchange: a peer's record did not arrive (bounded wait expired)";
        return FPSQ_ERR_TIMEOUT;
      }
      if (all_done()) stop = true;
    }
    return 0;
  }
  int pace_single(bool& stop) {
    if (all_done()) {
      stop = true;
      return 0;
    }
    if (*reinterpret_cast<volatile uint64_t*>(h->hscal + 15) != 0) {  // (an expired wait inside a launch: see wait_progress)
      stop = true;
      return 0;  // (call_end reports it -- and switches the handle to two launches per iteration: ride_failed)
    }
    // before the expected count the steps publish nothing (but the end of a recurrence): enqueue on
    if (it < expect) return 0;
    // bound the run-ahead of the host on the slowest unfinished lane
    int slow = INT32_MAX;
    for (int l = 0; l < NL; ++l) {
      const Progress ps = load_progress(&h->prog_host[l]);
      if (!ps.done) slow = std::min(slow, (int)ps.iter + lane_lag(lanes[l]));
    }
    if (it > expect && it - slow >= look) {
      if (int rc = flush_pend(true)) return rc;  // (the host is about to wait for the pending steps' progress)
      if (int rc = wait_lanes(it - look + 1)) return rc;
      if (all_done()) {
        stop = true;
        return 0;
      }
    }
    // Consecutive calls of one kind (the evaluations of a line search, the CG steps of a Newton iteration) mostly take
    // the same number of iterations: do not enqueue iteration expect + 1 before the device has finished iteration
    // `expect`.  When the count repeats, no launch is enqueued past convergence (each costs ~3.5 us of GPU time even
    // though it exits at once: ~50 us per evaluation at lookahead 4); when it does not, this is one short bubble.
    if (expect > 0 && it == expect) {
      if (int rc = flush_pend(true)) return rc;  // (the gated kernels and the host must see this iteration's verdict)
      if (int rc = enqueue_speculative()) return rc;
      if (int rc = wait_lanes(it)) return rc;
      if (all_done()) stop = true;
    }
    return 0;
  }

  // ------------------------------------------------------------------ behind the loop
  int finish() {
    if (!h->comm && !all_done()) {
      // The loop ran out of iterations (itmax) before the host saw every lane end.  The steps still in the stream will publish
      // those ends into the progress words -- which the NEXT run of this call (the second lane of an unfused call, the extras
      // lanes of hprod! Val(1)) resets on the host and then polls: a late "done" of THIS run would make it stop enqueueing at
      // once and leave its recurrence unfinished (found by the fixed-iteration tests: statistics of the second lane all zero).
      // Drain the stream, so that every word says what this run ended with.  (Only the itmax exit comes here: the other exits
      // of the loop have seen `done`; a sharded run has synchronised at this boundary already.)
      if (int rc = flush_pend(true)) return rc;
      HIPCHK(h, hipStreamSynchronize(s));
    }
    if (all_done()) {  // the iteration at which the last recurrence finished (its progress word says so)
      int64_t e = 0;
      for (int l = 0; l < NL; ++l) e = std::max<int64_t>(e, h->prog_host[l].iter + lane_lag(lanes[l]));
      expect_slot[1] = expect_slot[0];
      expect_slot[0] = e;
    }
    if (int rc = flush_pend(true)) return rc;
    ht_mark(h, 4);
    if (spec_it >= 0 && spec_it == it && all_done()) {
      // every recurrence ended at or before the iteration the speculative flush + tail were enqueued behind: their gates
      // were open, the call's epilogue is already in the stream
      h->tail_was_run = true;
      return 0;
    }
    // the last LSQR update (iteration `it`) has not been enqueued yet
    UpdSeg seg[2] = {seg_none(), seg_none()};
    int ns = 0;
    for (int l = 0; l < NL; ++l)
      if (lanes[l].kind == LANE_LSQR && it >= 1) seg[ns++] = lsqr_upd_seg(l, it);
    if (it == 0) {  // no iteration ran (itmax = 0): the pending w_1 / x = 0 start-up still has to happen
      seg[0] = winit[0];
      seg[1] = winit[1];
    }
    // MINRES: stage E3 and the stopping tests of the last enqueued iteration (no-ops when it ended earlier)
    if (h->absorb_flush && tail != nullptr && ns == 1 && it >= 1 && minres_lane < 0)
      h->pending_flush = seg[0];  // the caller's epilogue starts with k_ys, which applies it
    else
      launch_updates<NL>(h, seg[0], seg[1], minres_lane >= 0 && it >= 1 ? minres_seg(3, it, SPcur) : seg_none());
    if (minres_lane >= 0 && it >= 1)
      if (int rc = launch_step(h, minres_step(STEP_MINRES_C, it), none, /*sharded=*/true)) return rc;
    return 0;  // the final stats were left in lanes[l].st by the step that ended each recurrence
  }

  int run() {
    setup();
    if (int rc = startup()) return rc;
    const int64_t launches0 = h->launches;
    while (it < itmax_all) {
      ++it;
      // several iterations per launch while the expected count (or itmax) leaves room for at least two; never across the count:
      // the gated flush and the epilogue go right behind it
      int K = 1;
      if (can_multi && have_pend && it >= 2 && expect > 0 && it <= expect)
        K = (int)std::min<int64_t>(std::min<int64_t>(h->multi_max, expect - it + 1), itmax_all - it + 1);
      if (K >= 2) {
        if (int rc = iteration_multi(K)) return rc;
      } else if (can_fuse && have_pend) {
        if (int rc = iteration_fused()) return rc;
      } else {
        if (int rc = half_step_at()) return rc;
        if (int rc = half_step_a()) return rc;
      }
      bool stop = false;
      if (int rc = h->comm ? pace_sharded(stop) : pace_single(stop)) return rc;
      if (stop) break;
    }
    h->loop_iters += it;
    h->loop_launches += h->launches - launches0 - tail_launches;
    return finish();
  }
};

template <int NL>
int run_krylov(fpsq_handle h, Lane* lanes, const TailFn* tail = nullptr) {
  return KrylovRun<NL>(h, lanes, tail).run();
}

int run_lanes(fpsq_handle h, Lane* lanes, int nlanes, const TailFn* tail = nullptr) {
  h->tail_was_run = false;
  if (nlanes == 2 && h->opt.fuse_two_rhs) return run_krylov<2>(h, lanes, (!h->comm || h->halo) ? tail : nullptr);
  for (int l = 0; l < nlanes; ++l)
    if (int rc = run_krylov<1>(h, lanes + l)) return rc;
  return 0;
}

// Halo mode, once, at the first (collective) solve call: the padded counts of the gather segment = the maxima over
// the ranks of the local partial counts, then the segment itself (see fpsq_solver_s::seg).
int ensure_gather_layout(fpsq_handle h) {
  if (!h->comm || !h->halo || h->gather_ready) return 0;
  const int P = h->comm->nranks;
  const int gn = ew_grid(h->n), gm = ew_grid(h->m);
  const double mine[4] = {(double)gn, (double)(h->AT.nblk + h->halo_gf), (double)npart_A(h), (double)gm};
  double* dsend = h->dscal;      // 4 doubles
  double* drecv = nullptr;       // 4 P doubles
  if (int rc = dalloc(h, &drecv, (size_t)4 * P)) return rc;
  HIPCHK(h, hipMemcpyAsync(dsend, mine, sizeof mine, hipMemcpyHostToDevice, h->stream));
  if (int rc = h->comm->allgather(dsend, drecv, 4, h->stream)) {
    h->err = h->comm->err;
    return rc;
  }
  std::vector<double> all((size_t)4 * P);
  HIPCHK(h, hipMemcpyAsync(all.data(), drecv, all.size() * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  dfree(h, &drecv);
  int c[4] = {0, 0, 0, 0};
  for (int r = 0; r < P; ++r)
    for (int k = 0; k < 4; ++k) c[k] = std::max(c[k], (int)all[(size_t)4 * r + k]);
  h->cE = c[0];
  h->cT = c[1];
  h->cA = c[2];
  h->cW = c[3];
  h->seg_len = 2 * (int64_t)h->cE + 2 * (int64_t)h->cT + 2 * (int64_t)h->cA + 7 * (int64_t)h->cW;
  if (int rc = dalloc(h, &h->seg, (size_t)h->seg_len)) return rc;
  if (int rc = dalloc(h, &h->gath, (size_t)h->seg_len * P * 2)) return rc;
  HIPCHK(h, hipMemsetAsync(h->seg, 0, (size_t)h->seg_len * 8, h->stream));  // the padding entries stay zero for good
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double* q = h->seg;
  h->pE = q;
  q += h->cE;
  h->pE2 = q;
  q += h->cE;
  h->pEm[0] = q;
  q += h->cW;
  h->pEm[1] = q;
  q += h->cW;
  h->pS = q;
  h->strT = h->cT;
  q += 2 * h->cT;
  h->pWalt[0] = q;
  q += h->cW;
  h->pWalt[1] = q;
  q += h->cW;
  h->pS2 = q;
  h->strA = h->cA;
  q += 2 * h->cA;
  h->pW[0] = q;
  q += h->cW;
  h->pW[1] = q;
  q += h->cW;
  h->pE3 = q;
  {
    Comm::Buffers B{};
    B.gath[0] = h->gath;
    B.gath[1] = h->gath + (size_t)h->seg_len * P;
    B.halo_recv = h->halo_recv;
    B.ovl = h->ovl;
    B.ovr = h->ovr;
    if (int rc = h->comm->arm(B, h->stream)) {
      h->err = h->comm->err;
      return rc;
    }
  }
  h->gather_ready = true;
  h->info.comm_route = h->comm->route();
  h->info.comm_in_launch_sums = insum(h) ? 1 : 0;
  return 0;
}

int check_ready(fpsq_handle h) {
  if (!h) return FPSQ_ERR_ARG;
  if (!h->have_structure || !h->have_values) {
    h->err = "Jacobian structure/values not set";
    return FPSQ_ERR_STATE;
  }
  hipSetDevice(h->opt.device);
  return ensure_gather_layout(h);
}

// Device-resident arguments are produced on the caller's stream: everything queued there so far must be complete
// before the first kernel / copy of this call touches them (include/fpsq.h, "INPUT READINESS").
// back to the handle's own stream (see fpsq_set_input_stream)
void unadopt_stream(fpsq_handle h) {
  if (!h->adopted) return;
  hipStreamSynchronize(h->stream);
  h->stream = h->own_stream;
  h->adopted = false;
}
void order_inputs(fpsq_handle h) {
  if (!h->in_stream_on || h->adopted) return;
  hipEventRecord(h->ev_in, h->in_stream);
  hipStreamWaitEvent(h->stream, h->ev_in, 0);
}

void call_begin(fpsq_handle h) {
  h->absorb_flush = false;
  h->pending_flush.kind = UPD_NONE;
  h->startup_qg.nblk = 0;
  h->launches = 0;
  h->spmv_launches = 0;
  h->prod_a[0] = h->prod_a[1] = h->prod_at[0] = h->prod_at[1] = 0;
  h->fused_launches = 0;
  h->mmid_launches = 0;
  h->multi_launches = h->multi_iters = 0;
  h->loop_launches = h->loop_iters = 0;
  h->ev_used = 0;
  // device-side timing of the whole call only when profiling is on: an event record is a marker packet the GPU has to
  // process (a few us each); otherwise last_solve_ms is the host's wall time of the call
  if (h->profile) hipEventRecord(h->ev0, h->stream);
  h->t_call = std::chrono::steady_clock::now();
}

// steps riding with leaders: a workgroup's bounded wait for the leaders' record expired (never observed; see kRidePolls)
bool ride_failed(fpsq_handle h) {
  // (the device stores the INTEGER 1 there -- as a double a denormal, which a host running with flush-to-zero would not see)
  volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(h->hscal + 15);
  if (*w == 0) return false;
  *w = 0;
  h->info.wait_timeouts++;
  h->err = "a bounded wait inside a product launch expired (the leaders' record did not arrive, or -- one-launch iterations -- a block's "
           "flag / partials did not): FPSQ_FUSE_ITER=0 keeps two launches per iteration, FPSQ_RIDE_LEAD=0 the stand-alone steps";
  // (something else held the device for longer than the bound: this handle goes on with two launches per iteration, whose
  // waits involve the leaders only)
  if (h->fused_launches > 0 && h->fuse_ok) {
    h->fuse_ok = false;
    h->fuse_fell_back = true;  // (the entry point repeats the call once: with_fuse_fallback)
  }
  // the same for a MINRES lane's merged launch (k_minres_mid: every workgroup of its grid must be resident at once): back to
  // three launches, whose workgroups wait for nobody, and the call is repeated
  if (h->mmid_launches > 0 && h->minres_merge) {
    h->minres_merge = false;
    h->fuse_fell_back = true;
  }
  return true;
}

int call_end(fpsq_handle h) {
  if (h->profile) hipEventRecord(h->ev1, h->stream);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->comm && h->comm->failed()) {
    h->info.p2p_timeouts++;
    h->err = "peer-to-peer exchange: a peer's record did not arrive (bounded wait expired)";
    return FPSQ_ERR_TIMEOUT;
  }
  if (ride_failed(h)) return FPSQ_ERR_TIMEOUT;
  float ms = 0.f;
  if (h->profile) hipEventElapsedTime(&ms, h->ev0, h->ev1);
  else ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - h->t_call).count();
  h->info.last_solve_ms = ms;
  h->info.last_kernel_launches = h->launches;
  h->info.last_spmv_launches = h->spmv_launches;
  for (int i = 0; i < 2; ++i) {
    h->info.last_prod_a[i] = h->prod_a[i];
    h->info.last_prod_at[i] = h->prod_at[i];
  }
  h->info.last_fused_launches = h->fused_launches;
  h->info.last_multi_launches = h->multi_launches;
  h->info.last_multi_iterations = h->multi_iters;
  h->info.last_loop_iterations = h->loop_iters;
  h->info.last_loop_launches = h->loop_launches;
  double sp = 0.0;
  for (size_t i = 0; i < h->ev_used; ++i) {
    float t = 0.f;
    hipEventElapsedTime(&t, h->ev_pool[i].a, h->ev_pool[i].b);
    sp += t;
  }
  h->info.last_spmv_ms = sp;
  return 0;
}

// End of a call with stream-ordered outputs: the caller's stream waits (event, no host block) for everything enqueued so
// far; the host only waits until the phi reduction -- which rides in the first kernel of the epilogue's tail -- has stored
// the call's sequence number behind its results (host-mapped memory, release store: the values and, from the earlier
// step kernels, the final statistics are there when the number is).
int call_end_ordered(fpsq_handle h, double seq) {
  if (!h->adopted) {
    HIPCHK(h, hipEventRecord(h->ev_out, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->in_stream, h->ev_out, 0));
  }
  volatile double* flag = h->hscal + 3;
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (*flag != seq) {
    if ((++spins & 255) == 0) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) break;  // everything ran (the mapped store is then visible too; if not, the values below are read after a full drain anyway)
      if (q != hipErrorNotReady) {
        h->err = std::string("stream failed in the epilogue: ") + hipGetErrorString(q);
        return FPSQ_ERR_HIP;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
        h->err = "timeout waiting for the evaluation's scalar results";
        return FPSQ_ERR_TIMEOUT;
      }
    }
  }
  if (h->comm && h->comm->failed()) {
    h->info.p2p_timeouts++;
    h->err = "peer-to-peer exchange: a peer's record did not arrive (bounded wait expired)";
    return FPSQ_ERR_TIMEOUT;
  }
  if (ride_failed(h)) return FPSQ_ERR_TIMEOUT;
  h->info.last_solve_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - h->t_call).count();
  h->info.last_kernel_launches = h->launches;
  h->info.last_spmv_launches = h->spmv_launches;
  for (int i = 0; i < 2; ++i) {
    h->info.last_prod_a[i] = h->prod_a[i];
    h->info.last_prod_at[i] = h->prod_at[i];
  }
  h->info.last_fused_launches = h->fused_launches;
  h->info.last_multi_launches = h->multi_launches;
  h->info.last_multi_iterations = h->multi_iters;
  h->info.last_loop_iterations = h->loop_iters;
  h->info.last_loop_launches = h->loop_launches;
  h->info.last_spmv_ms = 0.0;
  return 0;
}

// true when p is device memory of the handle's GPU (kernels may then use it in place)
bool on_this_device(fpsq_handle h, const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // plain host memory: not an error
    return false;
  }
  return a.type == hipMemoryTypeDevice && a.device == h->opt.device;
}

int soft_rc(const fpsq_stats st[2]) { return (st[0].solved ? 0 : 1) | (st[1].solved ? 0 : 2); }

__global__ void k_mk_params(MinresState* S, MinresParams P, double kdelta) {
  MinresState* s = S + blockIdx.x;
  minres_set_params(s, P);
  s->kmode = 1;
  s->kdelta = kdelta;
  s->lambda = 0.0;
  s->ctl.skip = 0;
}

// Both systems K [p; q] = [bp_l; bq_l], l = 0, 1, by MINRES on K = [I A'; A -delta I] (order n + m), the two recurrences
// in lock-step on interleaved vectors: per iteration one two-RHS A' product, one two-RHS A product, three element-wise
// stages over n + m and three scalar steps.  Not a path of the reference (its `solve_two_mixed` is LSQR + CRAIG, the
// default here): BASELINE.json's north_star / configs[1] name it ("MINRES matrix-free") and SURVEY 8(b) lists it in the
// method enum.  Tolerances: the reference's MINRES set (ne_atol, ne_rtol, ne_etol, ne_conlim; ne_itmax = 0 -> 2 (n + m)).
// Null right-hand sides are zero.  Solutions to (p0, q0, p1, q1); stats to h->hstats[0 / 1].
int minres_k_device(fpsq_handle h, const double* bp0, const double* bq0, const double* bp1, const double* bq1, double* p0,
                    double* q0, double* p1, double* q1) {
  if (h->comm) {
    h->err = "kkt_method = MINRES_K is single-GPU (use the default LSQR + CRAIG method on a sharded handle)";
    return FPSQ_ERR_STATE;
  }
  hipStream_t s = h->stream;
  const int64_t n = h->n, m = h->m;
  if (!h->mk_ready) {
    double** lv[] = {&h->mk_long.Y, &h->mk_long.R1, &h->mk_long.R2, &h->mk_long.W1, &h->mk_long.W2, &h->mk_long.X};
    for (auto p : lv)
      if (int rc = dalloc(h, p, 2 * (size_t)n)) return rc;
    double** sv[] = {&h->mk_short.Y, &h->mk_short.R1, &h->mk_short.R2, &h->mk_short.W1, &h->mk_short.W2, &h->mk_short.X};
    for (auto p : sv)
      if (int rc = dalloc(h, p, 2 * (size_t)m)) return rc;
    h->mk_gl = (int)((n + kMkPerBlock - 1) / kMkPerBlock);
    h->mk_gs = (int)((m + kMkPerBlock - 1) / kMkPerBlock);
    for (int l = 0; l < 2; ++l)
      if (int rc = dalloc(h, &h->mk_part[l], (size_t)(h->mk_gl + h->mk_gs))) return rc;
    if (int rc = dalloc(h, &h->mk_state, 2)) return rc;
    h->mk_ready = true;
  }
  const int gl = h->mk_gl, gt = h->mk_gl + h->mk_gs;
  const fpsq_options& o = h->opt;
  const int64_t itmax = o.ne_itmax > 0 ? o.ne_itmax : 2 * (n + m);
  MinresParams P{0.0, o.ne_atol, o.ne_rtol, o.ne_etol, o.ne_conlim, itmax, INT32_MAX};
  hipLaunchKernelGGL(k_mk_params, dim3(2), dim3(1), 0, s, h->mk_state, P, h->delta);
  hipLaunchKernelGGL(k_mk_init, dim3(gt), dim3(kBlock), 0, s, h->mk_long, bp0, bp1, n, h->mk_short, bq0, bq1, m, gl,
                     h->mk_part[0], h->mk_part[1]);
  h->launches += 2;
  MinresState* S0 = h->mk_state;
  MinresState* S1 = h->mk_state + 1;
  for (int l = 0; l < 2; ++l) {
    h->prog_host[l].iter = 0;
    h->prog_host[l].done = 0;
    h->hstats[l] = fpsq_stats{};
  }
  auto sargs = [&](int kind, int l, int it) {
    StepArgs a{};
    a.kind = kind;
    a.it = it;
    a.state = h->mk_state + l;
    a.p0 = h->mk_part[l];
    a.n0 = gt;
    a.p1 = nullptr;
    a.n1 = 0;
    a.prog = h->prog_dev + l;
    a.host_stats = h->hstats_dev + l;
    return a;
  };
  launch_step_raw(h, sargs(STEP_MINRES_BEGIN, 0, 0), sargs(STEP_MINRES_BEGIN, 1, 0));
  const LaneCtl* gsave0 = h->gate0;
  const LaneCtl* gsave1 = h->gate1;
  h->gate0 = h->gate1 = nullptr;
  int64_t it = 0, chunk = 8;
  int rc = 0;
  while (true) {
    for (int64_t k = 0; k < chunk && it < itmax; ++k) {
      ++it;
      // y = K r2 / beta: long part through A', short part through A (both read the pair r2, write the pair y)
      launch_spmv<2>(h, TAG_AT, h->mk_short.R2, h->mk_long.R2, h->mk_long.Y, &S0->ctlT, &S1->ctlT, nullptr);
      launch_spmv<2>(h, TAG_A, h->mk_long.R2, h->mk_short.R2, h->mk_short.Y, &S0->ctl, &S1->ctl, nullptr);
      hipLaunchKernelGGL(k_mk_stage<1>, dim3(gt), dim3(kBlock), 0, s, &S0->ctl, &S1->ctl, (int)it, h->mk_long, n, h->mk_short,
                         m, gl, h->mk_part[0], h->mk_part[1]);
      launch_step_raw(h, sargs(STEP_MINRES_A, 0, (int)it), sargs(STEP_MINRES_A, 1, (int)it));
      hipLaunchKernelGGL(k_mk_stage<2>, dim3(gt), dim3(kBlock), 0, s, &S0->ctl, &S1->ctl, (int)it, h->mk_long, n, h->mk_short,
                         m, gl, h->mk_part[0], h->mk_part[1]);
      launch_step_raw(h, sargs(STEP_MINRES_B, 0, (int)it), sargs(STEP_MINRES_B, 1, (int)it));
      hipLaunchKernelGGL(k_mk_stage<3>, dim3(gt), dim3(kBlock), 0, s, &S0->ctl, &S1->ctl, (int)it, h->mk_long, n, h->mk_short,
                         m, gl, h->mk_part[0], h->mk_part[1]);
      launch_step_raw(h, sargs(STEP_MINRES_C, 0, (int)it), sargs(STEP_MINRES_C, 1, (int)it));
      h->launches += 3;
    }
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      h->err = std::string("minres_k: ") + hipGetErrorString(e);
      rc = FPSQ_ERR_HIP;
      break;
    }
    if ((h->prog_host[0].done && h->prog_host[1].done) || it >= itmax) break;
    chunk = std::min<int64_t>(chunk * 2, 64);
  }
  h->gate0 = gsave0;
  h->gate1 = gsave1;
  if (rc) return rc;
  if (!(h->prog_host[0].done && h->prog_host[1].done)) {  // (the mapped words lag: read the states)
    MinresState hs[2];
    HIPCHK(h, hipMemcpy(hs, h->mk_state, sizeof hs, hipMemcpyDeviceToHost));
    for (int l = 0; l < 2; ++l) h->hstats[l] = hs[l].stats;
  }
  hipLaunchKernelGGL(k_mk_unpack, dim3(ew_grid(n)), dim3(kBlock), 0, s, h->mk_long.X, p0, p1, n);
  hipLaunchKernelGGL(k_mk_unpack, dim3(ew_grid(m)), dim3(kBlock), 0, s, h->mk_short.X, q0, q1, m);
  h->launches += 2;
  return 0;
}

// device-side solve_two_mixed: g (n), c (m) device pointers; results left in h->p1, h->Lx[0] (q1), h->Cx (p2), h->Cy (q2)
// defer_p1: the caller forms p1 = g - A'q1 itself (qp_objgrad pairs that product with A'c in one two-RHS launch)
// affine_shift != null (fast start): c is NOT formed yet; CRAIG's right-hand side -(A z - shift), z in the long pair's
// CRAIG lane, comes out of the LSQR start-up product and A z - shift is left in `c` (see run_krylov)
int two_mixed_device(fpsq_handle h, const double* g, double* c, bool defer_p1 = false,
                     const double* affine_shift = nullptr, const TailFn* tail = nullptr) {
  if (h->opt.kkt_method == FPSQ_KKT_MINRES_K) {
    if (defer_p1 || affine_shift || tail) {
      h->err = "kkt_method = MINRES_K serves fpsq_solve_two_mixed / fpsq_solve_two_least_squares / fpsq_ys_gs only";
      return FPSQ_ERR_STATE;
    }
    // K [p1; q1] = [g; 0], K [p2; q2] = [0; c]
    return minres_k_device(h, g, nullptr, nullptr, c, h->p1, h->Lx[0], h->Cx, h->Cy);
  }
  Lane lanes[2];
  // (q1, stats1) = solve_least_square(qds, Aop', rhs1, sqrt(delta))      src/solve_linear_system.jl:123
  lanes[0].kind = LANE_LSQR;
  lanes[0].rhs = g;
  lanes[0].lambda = std::sqrt(h->delta);
  lanes[0].x = h->Lx[0];
  lanes[0].st = &h->hstats[0];
  // (p2, q2, stats2) = solve_least_norm(qds, Aop, -rhs2, delta); p2 = -p2 :132-133
  lanes[1].kind = h->opt.ln_method == FPSQ_LN_LNLQ ? LANE_LNLQ : LANE_CRAIG;
  lanes[1].rhs = c;
  lanes[1].rhs_scale = -1.0;
  if (affine_shift) {
    lanes[0].preloaded = true;
    lanes[1].affine_shift = affine_shift;
    lanes[1].affine_out = c;
  }
  lanes[1].delta = h->delta;
  lanes[1].xsign = -1.0;
  lanes[1].x = h->Cx;
  lanes[1].y = h->Cy;
  lanes[1].st = &h->hstats[1];
  // p1 = rhs1 - Aop' q1                                                   :126-127
  TailFn full = [&]() -> int {
    if (!defer_p1)
      if (int rc = at_product_const(h, -1.0, h->Lx[0], 1.0, g, h->p1)) return rc;
    return tail ? (*tail)() : 0;
  };
  if (int rc = run_lanes(h, lanes, 2, tail ? &full : nullptr)) return rc;
  if (!h->tail_was_run)
    if (int rc = full()) return rc;
  return 0;
}

// device-side solve_two_least_squares: results in h->p1, h->Lx[0], h->p2b, h->Lx[1]
int two_least_squares_device(fpsq_handle h, const double* r1, const double* r2, const TailFn* tail = nullptr) {
  if (h->opt.kkt_method == FPSQ_KKT_MINRES_K) {
    if (tail) {
      h->err = "kkt_method = MINRES_K serves fpsq_solve_two_mixed / fpsq_solve_two_least_squares / fpsq_ys_gs only";
      return FPSQ_ERR_STATE;
    }
    return minres_k_device(h, r1, nullptr, r2, nullptr, h->p1, h->Lx[0], h->p2b, h->Lx[1]);
  }
  Lane lanes[2];
  const double* rhs[2] = {r1, r2};
  for (int l = 0; l < 2; ++l) {
    lanes[l].kind = LANE_LSQR;
    lanes[l].rhs = rhs[l];
    lanes[l].lambda = std::sqrt(h->delta);
    lanes[l].x = h->Lx[l];
    lanes[l].st = &h->hstats[l];
  }
  // src/solve_linear_system.jl:90-91 and :99-100
  TailFn full = [&]() -> int {
    if (int rc = at_product_const(h, -1.0, h->Lx[0], 1.0, r1, h->p1)) return rc;
    if (int rc = at_product_const(h, -1.0, h->Lx[1], 1.0, r2, h->p2b)) return rc;
    return tail ? (*tail)() : 0;
  };
  if (int rc = run_lanes(h, lanes, 2, tail ? &full : nullptr)) return rc;
  if (!h->tail_was_run)
    if (int rc = full()) return rc;
  return 0;
}

}  // namespace

// ===================================================================================== C ABI

// One-launch iterations and workgroup slots.  The waiting workgroups of a fused launch hold slots that the workgroups they wait for
// may still need.  WITHIN ONE LAUNCH that cannot deadlock: every dependence points to a workgroup earlier in the grid, a queue
// dispatches its grid in order (workgroup i through XCD i mod 8, in order within the XCD), and the leaders are one per lane on
// EVERY XCD -- so whatever a running workgroup waits for has been dispatched ahead of it on the XCD that dispatches it, and
// the earliest unfinished workgroup of the grid waits for nothing.  ACROSS LAUNCHES that argument does not hold -- two handles
// of this process iterating from two host threads are two queues, exactly like two processes: each queue's waiting workgroups
// can fill slots the OTHER queue's not-yet-dispatched workgroups need.  No circular wait was ever observed inside one process
// (tools/fuse_soak_two.py: 2.4 M fused launches of two handles sharing the device), three ranks rehearsed on ONE GPU did run into
// it; the answer is the same for both: every wait is bounded, the call ends in FPSQ_ERR_TIMEOUT with every kernel gone, the
// handle keeps two launches per iteration from then on (ride_failed -- their waits involve the leaders only) and the call is
// REPEATED once -- its inputs are untouched -- so the caller sees a delay and fpsq_info.fuse_fallbacks, not an error.
template <class F>
int with_fuse_fallback(fpsq_handle h, F&& call) {
  int rc = call();
  // (a rank of several cannot repeat a call on its own -- its peers are not repeating theirs: there the expired wait is the call's
  // result, FPSQ_ERR_TIMEOUT on this rank and, through the peers' own bounded waits, on the others; the job decides what next --
  // bench.py starts over on the collectives.  Their waits are long for that reason: RideArgs::more.)
  if (rc == FPSQ_ERR_TIMEOUT && h && h->fuse_fell_back && !(h->comm && h->comm->nranks > 1)) {
    h->fuse_fell_back = false;
    h->info.fuse_fallbacks++;  // (fpsq_info: a benchmark or a test sees that it happened)
    if (h->verbose)
      std::fprintf(stderr, "fpsq: a bounded wait of a one-launch iteration expired (is the GPU shared with other processes?); this handle "
                           "continues with two launches per iteration, the call is repeated\n");
    rc = call();
  }
  if (h) h->fuse_fell_back = false;
  return rc;
}

extern "C" {

const char* fpsq_version(void) { return "fpsq 0.1.0 (gfx950)"; }

void fpsq_default_options(int64_t n, int64_t m, fpsq_options* o) {
  const double se = std::sqrt(2.220446049250313e-16);
  std::memset(o, 0, sizeof *o);
  o->ls_atol = se;
  o->ls_rtol = se;
  o->ls_itmax = 5 * (m + n);
  o->ln_atol = se;
  o->ln_rtol = se;
  o->ln_btol = se;
  o->ln_conlim = 1.0 / se;
  o->ln_itmax = 5 * (m + n);
  o->ne_atol = se;
  o->ne_rtol = se;
  o->ne_etol = se;
  o->ne_itmax = 0;
  o->ne_conlim = 1.0 / se;
  o->ls_axtol = se;
  o->ls_btol = se;
  o->ls_etol = se;
  o->ls_conlim = 1.0 / se;
  o->fuse_two_rhs = 1;
  o->lookahead = 4;
  o->device = 0;
}

const char* fpsq_last_error(fpsq_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int fpsq_create(fpsq_handle* out, int64_t n, int64_t m, const fpsq_options* opts) {
  if (!out || n <= 0 || m <= 0 || n >= INT32_MAX || m >= INT32_MAX) {
    g_create_error = "fpsq_create: bad arguments";
    return FPSQ_ERR_ARG;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    g_create_error = std::string("fpsq_create: no HIP device (") + hipGetErrorString(e) +
                     "); libfpsq has no CPU fallback";
    return FPSQ_ERR_HIP;
  }
  fpsq_handle h = new fpsq_solver_s();
  h->n = n;
  h->m = m;
  if (opts) h->opt = *opts; else fpsq_default_options(n, m, &h->opt);
  auto fail = [&](const char* what, hipError_t err) {
    g_create_error = std::string("fpsq_create: ") + what + ": " + hipGetErrorString(err);
    delete h;
    return FPSQ_ERR_HIP;
  };
  if ((e = hipSetDevice(h->opt.device)) != hipSuccess) return fail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
  if ((e = hipHostMalloc((void**)&h->prog_host, 4 * sizeof(Progress), hipHostMallocMapped | hipHostMallocCoherent)) !=
      hipSuccess)
    return fail("hipHostMalloc", e);
  std::memset(h->prog_host, 0, 4 * sizeof(Progress));
  if ((e = hipHostGetDevicePointer((void**)&h->prog_dev, h->prog_host, 0)) != hipSuccess)
    return fail("hipHostGetDevicePointer", e);
  if ((e = hipHostMalloc((void**)&h->hstats, 4 * sizeof(fpsq_stats), hipHostMallocMapped | hipHostMallocCoherent)) !=
      hipSuccess)
    return fail("hipHostMalloc", e);
  if ((e = hipHostMalloc((void**)&h->hscal, 16 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent)) !=
      hipSuccess)
    return fail("hipHostMalloc", e);
  if (const char* ev = std::getenv("FPSQ_ADAPTIVE_RUNAHEAD")) h->adaptive_runahead = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_HOST_TRACE")) h->host_trace = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_AT_SORTED")) h->at_sorted = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_AT_SHARED")) h->at_shared = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_RIDE_LEAD")) h->ride_lead = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_ATL_TWO")) h->atl_two = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_AT_XCD")) h->at_xcd = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_FUSE_ITER")) h->fuse_iter = std::atoi(ev);
  if (const char* ev = std::getenv("FPSQ_DEBUG_FUSE_BREAK")) h->fuse_break = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_FUSE_HALO")) h->fuse_halo_on = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_MULTI_ITER")) h->multi_max = std::min(std::max(std::atoi(ev), 1), kMultiMax);
  if (const char* ev = std::getenv("FPSQ_MULTI_DEFER_LONG")) h->multi_defer_long = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_MULTI_UPD")) {
    int a = 0, b = 0;
    if (std::sscanf(ev, "%d,%d", &a, &b) == 2 && a >= 8 && b >= 8) {
      h->multi_upd_t = a / 8 * 8;
      h->multi_upd_a = b / 8 * 8;
    }
  }
  if (const char* ev = std::getenv("FPSQ_MINRES_MERGE")) h->minres_merge = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_DEBUG_FUSE_ROTATE")) h->fuse_rotate = std::atoi(ev) & 7;
  if (const char* ev = std::getenv("FPSQ_DEBUG_RIDE_DELAY_MID")) h->ride_delay_mid = std::atoi(ev);
  if (const char* ev = std::getenv("FPSQ_FUSE_PROBE")) {
    h->fuse_probe_path = ev;
    h->fuse_probe_at = 100;
    if (const char* at = std::getenv("FPSQ_FUSE_PROBE_AT")) h->fuse_probe_at = std::atoll(at);
  }
  if (const char* ev = std::getenv("FPSQ_FUSE_TAIL")) h->fuse_tail = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_ADOPT_STREAM")) h->adopt_streams = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_DEBUG_RIDE_BREAK")) h->ride_break = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_DEBUG_RIDE_DELAY")) h->ride_delay = std::atoi(ev);
  if (const char* ev = std::getenv("FPSQ_JAC_REFRESH")) h->refresh_3pass = std::atoi(ev) == 3;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->opt.device) == hipSuccess && prop.multiProcessorCount > 0) {
      h->resident_wgs = 4 * prop.multiProcessorCount;
      int xccs = 0;
      if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, h->opt.device) != hipSuccess) {
        (void)hipGetLastError();
        xccs = 0;
      }
      int per_cu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_minres_mid, kBlock, 0) == hipSuccess && per_cu > 0)
        h->mmid_cap = per_cu * prop.multiProcessorCount;
      else
        (void)hipGetLastError();
      const std::string arch = prop.gcnArchName;
      h->fuse_hw_ok = xccs == 8 && (arch.rfind("gfx950", 0) == 0 || arch.rfind("gfx942", 0) == 0);
    }
  }
  if (const char* ev = std::getenv("FPSQ_FUSE_ANY_DEVICE")) h->fuse_hw_ok = h->fuse_hw_ok || std::atoi(ev) != 0;  // (bring-up on other parts)
  if (const char* ev = std::getenv("FPSQ_VERBOSE")) h->verbose = std::atoi(ev) != 0;
  if (const char* ev = std::getenv("FPSQ_AB_MASK")) h->ab_mask = std::atoi(ev);
  if (const char* ev = std::getenv("FPSQ_AB_DYNAMIC")) h->ab_dynamic = std::atoi(ev) != 0;
  std::memset(h->hstats, 0, 4 * sizeof(fpsq_stats));
  std::memset(h->hscal, 0, 16 * sizeof(double));
  if ((e = hipHostGetDevicePointer((void**)&h->hstats_dev, h->hstats, 0)) != hipSuccess)
    return fail("hipHostGetDevicePointer", e);
  if ((e = hipHostGetDevicePointer((void**)&h->hscal_dev, h->hscal, 0)) != hipSuccess)
    return fail("hipHostGetDevicePointer", e);
  hipEventCreate(&h->ev0);
  hipEventCreate(&h->ev1);
  void* p = nullptr;
  const size_t state_bytes = sizeof(LsqrState) * 6 + sizeof(CraigState) * 3 + sizeof(MinresState) * 2 + sizeof(LnlqState) * 3 +
                             4 * sizeof(LaneCtl) + 64 * sizeof(double);
  if ((e = hipMalloc(&p, state_bytes)) != hipSuccess) return fail("hipMalloc", e);
  h->allocs.push_back(p);
  hipMemset(p, 0, state_bytes);
  {
    void* q = nullptr;
    // (a record copy of 64 words per XCC; a second record for the mid leaders of fused iterations)
    if ((e = hipMalloc(&q, 2 * 8 * 512 + 64)) != hipSuccess) return fail("hipMalloc", e);
    h->allocs.push_back(q);
    hipMemset(q, 0, 2 * 8 * 512 + 64);
    h->ride_rec = (unsigned long long*)q;
    h->ride_rec2 = h->ride_rec + 512;
  }
  char* cp = (char*)p;
  h->lsqr[0] = (LsqrState*)cp;
  h->lsqr[1] = h->lsqr[0] + 1;
  cp += sizeof(LsqrState) * 2;
  h->craig = (CraigState*)cp;
  cp += sizeof(CraigState);
  h->lsqr_alt[0] = (LsqrState*)cp;
  h->lsqr_alt[1] = h->lsqr_alt[0] + 1;
  cp += sizeof(LsqrState) * 2;
  h->craig_alt = (CraigState*)cp;
  cp += sizeof(CraigState);
  h->minres = (MinresState*)cp;
  cp += sizeof(MinresState);
  h->lnlq = (LnlqState*)cp;
  cp += sizeof(LnlqState);
  h->lnlq_alt = (LnlqState*)cp;
  cp += sizeof(LnlqState);
  h->minres_alt = (MinresState*)cp;
  cp += sizeof(MinresState);
  h->state3[0] = cp;  // (two LSQR states)
  cp += sizeof(LsqrState) * 2;
  h->state3[1] = cp;
  cp += sizeof(CraigState);
  h->state3[2] = cp;
  cp += sizeof(LnlqState);
  h->ctl_tmp = (LaneCtl*)cp;
  cp += sizeof(LaneCtl);
  h->ctl_raw = (LaneCtl*)cp;
  cp += sizeof(LaneCtl);
  h->ctl_pm = (LaneCtl*)cp;
  cp += sizeof(LaneCtl);
  h->ctl_mp = (LaneCtl*)cp;
  cp += sizeof(LaneCtl);
  h->dscal = (double*)cp;
  h->comm_scal = h->dscal + 32;
  {
    LaneCtl raw{};
    raw.ca = 1.0;
    raw.cb = 0.0;
    raw.upd_iter = -1;
    hipMemcpy(h->ctl_raw, &raw, sizeof raw, hipMemcpyHostToDevice);
    raw.ca = 1.0;
    raw.cb = -1.0;
    hipMemcpy(h->ctl_pm, &raw, sizeof raw, hipMemcpyHostToDevice);
    raw.ca = -1.0;
    raw.cb = 1.0;
    hipMemcpy(h->ctl_mp, &raw, sizeof raw, hipMemcpyHostToDevice);
  }
  hipDeviceSynchronize();
  *out = h;
  return FPSQ_OK;
}

int fpsq_destroy(fpsq_handle h) {
  if (!h) return FPSQ_ERR_ARG;
  if (h->host_trace && h->ht_calls > 0) {
    static const char* nm[12] = {"between calls", "entry->inputs ordered", "->first launch", "->krylov start", "->krylov end",
                                 "->epilogue enqueued", "->synchronised", "->exit", "", "", "", ""};
    std::fprintf(stderr, "fpsq host trace (%lld qp_objgrad calls), us per call:", (long long)h->ht_calls);
    for (int k = 0; k < 8; ++k) std::fprintf(stderr, "  %s %.1f", nm[k], 1e6 * h->ht_sum[k] / (double)h->ht_calls);
    std::fprintf(stderr, "\n");
  }
  hipSetDevice(h->opt.device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->fuse_probe_buf && !h->fuse_probe_path.empty()) {  // developer probe: "grid n0 n1 .. n5" then one line of four stamps per workgroup
    std::vector<unsigned long long> st((size_t)h->fuse_probe_grid * 4);
    hipMemcpy(st.data(), h->fuse_probe_buf, st.size() * 8, hipMemcpyDeviceToHost);
    if (FILE* f = std::fopen(h->fuse_probe_path.c_str(), "w")) {
      std::fprintf(f, "%d", h->fuse_probe_grid);
      for (int v : h->fuse_probe_layout) std::fprintf(f, " %d", v);
      std::fprintf(f, "\n");
      for (int i = 0; i < h->fuse_probe_grid; ++i)
        std::fprintf(f, "%llu %llu %llu %llu\n", st[4 * (size_t)i], st[4 * (size_t)i + 1], st[4 * (size_t)i + 2], st[4 * (size_t)i + 3]);
      std::fclose(f);
    }
  }
  delete h->comm;
  for (void* p : h->allocs) hipFree(p);
  for (auto& e : h->ev_pool) {
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->ev_in) hipEventDestroy(h->ev_in);
  if (h->ev_out) hipEventDestroy(h->ev_out);
  if (h->prog_host) hipHostFree(h->prog_host);
  if (h->hstats) hipHostFree(h->hstats);
  if (h->hscal) hipHostFree(h->hscal);
  if (h->own_stream && h->own_stream != h->stream) {  // (an adopted stream is the caller's)
    hipStreamDestroy(h->own_stream);
  } else if (h->stream && !h->adopted) {
    hipStreamDestroy(h->stream);
  }
  delete h;
  return FPSQ_OK;
}

int fpsq_set_jacobian_structure_csr(fpsq_handle h, const int32_t* rowptr, const int32_t* colind) {
  if (!h || !rowptr || h->have_structure) {
    if (h) h->err = "set_jacobian_structure: bad arguments or structure already set";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  HostCsr HA;
  HA.nrows = h->m;
  HA.ncols = h->n;
  HA.rowptr.resize(h->m + 1);
  HIPCHK(h, hipMemcpy(HA.rowptr.data(), rowptr, (size_t)(h->m + 1) * 4, hipMemcpyDefault));
  const int64_t nnz = HA.rowptr[h->m];
  if (HA.rowptr[0] != 0 || nnz < 0) {
    h->err = "set_jacobian_structure_csr: rowptr must be 0-based";
    return FPSQ_ERR_ARG;
  }
  HA.colind.resize(nnz);
  if (nnz) HIPCHK(h, hipMemcpy(HA.colind.data(), colind, (size_t)nnz * 4, hipMemcpyDefault));
  for (int64_t i = 0; i < h->m; ++i)
    if (HA.rowptr[i + 1] < HA.rowptr[i]) {
      h->err = "set_jacobian_structure_csr: rowptr not monotone";
      return FPSQ_ERR_ARG;
    }
  for (int64_t k = 0; k < nnz; ++k)
    if (HA.colind[k] < 0 || HA.colind[k] >= h->n) {
      h->err = "set_jacobian_structure_csr: column index out of range";
      return FPSQ_ERR_ARG;
    }
  h->nnz_in = nnz;
  return finish_structure(h, HA);
}

int fpsq_set_jacobian_structure_coo(fpsq_handle h, int64_t nnz, const int64_t* rows, const int64_t* cols,
                                    int32_t index_base) {
  if (!h || nnz < 0 || nnz >= INT32_MAX || (nnz > 0 && (!rows || !cols)) || h->have_structure) {
    if (h) h->err = "set_jacobian_structure_coo: bad arguments or structure already set";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  std::vector<int64_t> r(nnz), c(nnz);
  if (nnz) {
    HIPCHK(h, hipMemcpy(r.data(), rows, (size_t)nnz * 8, hipMemcpyDefault));
    HIPCHK(h, hipMemcpy(c.data(), cols, (size_t)nnz * 8, hipMemcpyDefault));
  }
  const int64_t m = h->m, n = h->n;
  std::vector<int32_t> cnt(m + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) {
    r[k] -= index_base;
    c[k] -= index_base;
    if (r[k] < 0 || r[k] >= m || c[k] < 0 || c[k] >= n) {
      h->err = "set_jacobian_structure_coo: index out of range";
      return FPSQ_ERR_ARG;
    }
    cnt[r[k] + 1]++;
  }
  for (int64_t i = 0; i < m; ++i) cnt[i + 1] += cnt[i];
  // bucket by row (stable), then sort each row by column (stable: duplicates keep the caller's order)
  std::vector<int32_t> order(nnz);
  {
    std::vector<int32_t> next(cnt.begin(), cnt.end() - 1);
    for (int64_t k = 0; k < nnz; ++k) order[next[r[k]]++] = (int32_t)k;
  }
  for (int64_t i = 0; i < m; ++i)
    std::stable_sort(order.begin() + cnt[i], order.begin() + cnt[i + 1],
                     [&](int32_t a, int32_t b) { return c[a] < c[b]; });
  HostCsr HA;
  HA.nrows = m;
  HA.ncols = n;
  HA.rowptr.assign(m + 1, 0);
  std::vector<int32_t> slotptr;
  slotptr.push_back(0);
  for (int64_t i = 0; i < m; ++i) {
    for (int32_t k = cnt[i]; k < cnt[i + 1]; ++k) {
      const int64_t col = c[order[k]];
      if (k > cnt[i] && col == c[order[k - 1]]) {
        slotptr.back() = k + 1;  // duplicate: extend the current slot
      } else {
        HA.colind.push_back((int32_t)col);
        slotptr.push_back(k + 1);
      }
    }
    HA.rowptr[i + 1] = (int32_t)HA.colind.size();
  }
  const bool dup = (int64_t)HA.colind.size() != nnz;
  h->nnz_in = nnz;
  if (int rc = dalloc(h, &h->in_perm, (size_t)nnz)) return rc;
  if (int rc = dalloc(h, &h->in_vals, (size_t)nnz)) return rc;
  if (nnz) HIPCHK(h, hipMemcpy(h->in_perm, order.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
  if (dup) {
    if (int rc = dalloc(h, &h->in_slotptr, slotptr.size())) return rc;
    HIPCHK(h, hipMemcpy(h->in_slotptr, slotptr.data(), slotptr.size() * 4, hipMemcpyHostToDevice));
  }
  return finish_structure(h, HA);
}

int fpsq_set_jacobian_values(fpsq_handle h, const double* vals) {
  if (!h || !h->have_structure || (!vals && h->nnz_in > 0)) {
    if (h) h->err = "set_jacobian_values: structure not set or null values";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  hipStream_t s = h->stream;
  order_inputs(h);
  const bool on_dev = h->nnz_in > 0 && on_this_device(h, vals);
  if (h->nnz_in > 0 && h->refresh_3pass) {
    // rounds 1-3: a copy into the staging array, then one grid-stride gather per stored copy
    if (h->in_perm) {
      HIPCHK(h, hipMemcpyAsync(h->in_vals, vals, (size_t)h->nnz_in * 8, hipMemcpyDefault, s));
      if (h->in_slotptr)
        hipLaunchKernelGGL(k_gather_sum, dim3(ew_grid(h->nnz)), dim3(kBlock), 0, s, h->in_vals, h->in_perm,
                           h->in_slotptr, h->A.vals, h->nnz);
      else
        hipLaunchKernelGGL(k_gather, dim3(ew_grid(h->nnz)), dim3(kBlock), 0, s, h->in_vals, h->in_perm, h->A.vals,
                           h->nnz);
    } else {
      HIPCHK(h, hipMemcpyAsync(h->A.vals, vals, (size_t)h->nnz * 8, hipMemcpyDefault, s));
    }
    if (h->AT.nstore > 0)
      hipLaunchKernelGGL(k_gather, dim3(ew_grid(h->AT.nstore)), dim3(kBlock), 0, s, h->A.vals, h->permT, h->AT.vals,
                         h->AT.nstore);
    if (h->RA.ok)
      hipLaunchKernelGGL(k_gather, dim3(ew_grid(h->RA.nstore)), dim3(kBlock), 0, s, h->A.vals, h->RA.vperm, h->RA.vals,
                         h->RA.nstore);
  } else if (h->nnz_in > 0) {
    // ONE launch writes every stored copy (k_refresh).  Where the gathers read from:
    //   CSR input              the caller's array in place when it lives on this GPU, else its copy in the CSR array
    //   COO, no duplicates     the caller's array in place / its staged copy, through permutations composed at set-up
    //   COO with duplicates    the CSR array, after the slots were summed into it (one more pass; fixed order)
    const double* src = vals;
    bool csr_is_src = false;  // (the CSR array already holds the values the gathers read)
    if (h->in_perm && h->in_slotptr) {
      const double* coo = vals;
      if (!on_dev) {
        HIPCHK(h, hipMemcpyAsync(h->in_vals, vals, (size_t)h->nnz_in * 8, hipMemcpyDefault, s));
        coo = h->in_vals;
      }
      hipLaunchKernelGGL(k_gather_sum, dim3(ew_grid(h->nnz)), dim3(kBlock), 0, s, coo, h->in_perm, h->in_slotptr, h->A.vals,
                         h->nnz);
      src = h->A.vals;
      csr_is_src = true;
    } else if (!on_dev) {
      double* stage = h->in_perm ? h->in_vals : h->A.vals;
      HIPCHK(h, hipMemcpyAsync(stage, vals, (size_t)h->nnz_in * 8, hipMemcpyDefault, s));
      src = stage;
      csr_is_src = !h->in_perm;
    }
    const RefreshSeg none{nullptr, nullptr, 0, 0, 0};
    auto seg = [](double* out, const int32_t* perm, int64_t n) {
      return RefreshSeg{out, perm, n, (int32_t)((n + kRefreshChunk - 1) / kRefreshChunk), 0};
    };
    const RefreshSeg sT = h->AT.nstore > 0 ? seg(h->AT.vals, h->permT, h->AT.nstore) : none;
    const RefreshSeg sR = h->RA.ok ? seg(h->RA.vals, h->RA.vperm, h->RA.nstore) : none;
    // the CSR array itself: only when a product reads it (no row-group copy of A) and it is not the source already
    const RefreshSeg sC = (!h->RA.ok && !csr_is_src) ? seg(h->A.vals, h->perms_to_input ? h->in_perm : nullptr, h->nnz) : none;
    if (std::getenv("FPSQ_REFRESH_SPLIT")) {  // (developer: one launch per segment, to time them apart)
      for (const RefreshSeg* q : {&sT, &sR, &sC})
        if (q->nchunk) hipLaunchKernelGGL(k_refresh, dim3((q->nchunk + 7) / 8 * 8), dim3(kBlock), 0, s, src, *q, none, none, (q->nchunk + 7) / 8);
    } else {
      const int per_xcd = (sT.nchunk + sR.nchunk + sC.nchunk + 7) / 8;
      hipLaunchKernelGGL(k_refresh, dim3(per_xcd * 8), dim3(kBlock), 0, s, src, sT, sR, sC, per_xcd);
    }
  }
  if (on_dev && h->adopted) {
    // (the gathers were enqueued on the caller's own stream)
  } else if (on_dev && h->in_stream_on) {
    // device-resident values on a registered stream: no host synchronisation -- the caller's stream is made to wait for
    // the gathers that read its array (it may overwrite the array with the next Jacobian), the solves that follow run on
    // the library's stream behind them
    if (!h->ev_out) HIPCHK(h, hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->ev_out, s));
    HIPCHK(h, hipStreamWaitEvent(h->in_stream, h->ev_out, 0));
  } else {
    HIPCHK(h, hipStreamSynchronize(s));
  }
  h->have_values = true;
  return FPSQ_OK;
}

int fpsq_set_input_stream(fpsq_handle h, int32_t enabled, void* hip_stream) {
  if (!h) return FPSQ_ERR_ARG;
  hipSetDevice(h->opt.device);
  if (enabled && !h->ev_in) HIPCHK(h, hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
  if (h->adopt_streams && (!h->comm || h->comm->nranks == 1)) {  // (a communicator of one rank has no peers)
    if (enabled && h->adopted && h->stream == (hipStream_t)hip_stream) return FPSQ_OK;  // (registered again: nothing to do)
    // everything enqueued so far is on the stream in use: finish it, then move
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!h->own_stream) h->own_stream = h->stream;
    h->adopted = enabled != 0;
    h->stream = h->adopted ? (hipStream_t)hip_stream : h->own_stream;
  }
  h->in_stream_on = enabled != 0;
  h->in_stream = (hipStream_t)hip_stream;
  return FPSQ_OK;
}

int fpsq_set_output_ordering(fpsq_handle h, int32_t stream_ordered) {
  if (!h) return FPSQ_ERR_ARG;
  hipSetDevice(h->opt.device);
  if (stream_ordered && !h->ev_out) HIPCHK(h, hipEventCreateWithFlags(&h->ev_out, hipEventDisableTiming));
  h->out_ordered = stream_ordered != 0;
  return FPSQ_OK;
}

int fpsq_set_delta(fpsq_handle h, double delta) {
  if (!h || !(delta >= 0.0)) {
    if (h) h->err = "set_delta: delta must be >= 0";
    return FPSQ_ERR_ARG;
  }
  h->delta = delta;
  return FPSQ_OK;
}

static int impl_solve_two_mixed(fpsq_handle h, const double* rhs1, const double* rhs2, double* p1, double* q1, double* p2,
                         double* q2, fpsq_stats st[2]) {
  if (int rc = check_ready(h)) return rc;
  if (!rhs1 || !rhs2 || !p1 || !q1 || !p2 || !q2 || !st) {
    h->err = "solve_two_mixed: null argument";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  hipStream_t s = h->stream;
  order_inputs(h);
  const size_t nb = (size_t)h->n * 8, mb = (size_t)h->m * 8;
  HIPCHK(h, hipMemcpyAsync(h->in_n1, rhs1, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(h->in_m, rhs2, mb, hipMemcpyDefault, s));
  call_begin(h);
  if (int rc = two_mixed_device(h, h->in_n1, h->in_m)) return rc;
  HIPCHK(h, hipMemcpyAsync(p1, h->p1, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(q1, h->Lx[0], mb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(p2, h->Cx, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(q2, h->Cy, mb, hipMemcpyDefault, s));
  if (int rc = call_end(h)) return rc;
  st[0] = h->hstats[0];
  st[1] = h->hstats[1];
  return soft_rc(st);
}

static int impl_solve_two_least_squares(fpsq_handle h, const double* rhs1, const double* rhs2, double* p1, double* q1,
                                 double* p2, double* q2, fpsq_stats st[2]) {
  if (int rc = check_ready(h)) return rc;
  if (!rhs1 || !rhs2 || !p1 || !q1 || !p2 || !q2 || !st) {
    h->err = "solve_two_least_squares: null argument";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  hipStream_t s = h->stream;
  order_inputs(h);
  const size_t nb = (size_t)h->n * 8, mb = (size_t)h->m * 8;
  HIPCHK(h, hipMemcpyAsync(h->in_n1, rhs1, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(h->in_n2, rhs2, nb, hipMemcpyDefault, s));
  call_begin(h);
  if (int rc = two_least_squares_device(h, h->in_n1, h->in_n2)) return rc;
  HIPCHK(h, hipMemcpyAsync(p1, h->p1, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(q1, h->Lx[0], mb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(p2, h->p2b, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(q2, h->Lx[1], mb, hipMemcpyDefault, s));
  if (int rc = call_end(h)) return rc;
  st[0] = h->hstats[0];
  st[1] = h->hstats[1];
  return soft_rc(st);
}

static int impl_solve_two_extras(fpsq_handle h, const double* rhs1, const double* rhs2, double* out1, double* out2,
                          fpsq_stats st[2]) {
  if (int rc = check_ready(h)) return rc;
  if (!rhs1 || !rhs2 || !out1 || !out2 || !st) {
    h->err = "solve_two_extras: null argument";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  hipStream_t s = h->stream;
  order_inputs(h);
  const size_t nb = (size_t)h->n * 8, mb = (size_t)h->m * 8;
  HIPCHK(h, hipMemcpyAsync(h->in_n1, rhs1, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(h->in_m, rhs2, mb, hipMemcpyDefault, s));
  call_begin(h);
  const double tau = std::max(h->delta, 1e-14);  // src/solve_linear_system.jl:51
  // (invJtJJv, stats) = solve_least_square(qds, Aop', rhs1, sqrt(tau))          :53
  Lane lanes[2];
  lanes[0].kind = LANE_LSQR;
  lanes[0].rhs = h->in_n1;
  lanes[0].lambda = std::sqrt(tau);
  lanes[0].x = h->Lx[0];
  lanes[0].st = &h->hstats[0];
  // minres(JtJ, rhs2, lambda = tau)                                              :58-72
  // (fused: the MINRES recurrence shares the two products of every LSQR iteration, see run_krylov)
  lanes[1].kind = LANE_MINRES;
  lanes[1].rhs = h->in_m;
  lanes[1].lambda = tau;
  lanes[1].x = h->Mx;
  lanes[1].st = &h->hstats[1];
  if (int rc = run_lanes(h, lanes, 2)) return rc;
  HIPCHK(h, hipMemcpyAsync(out1, h->Lx[0], mb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(out2, h->Mx, mb, hipMemcpyDefault, s));
  if (int rc = call_end(h)) return rc;
  st[0] = h->hstats[0];
  st[1] = h->hstats[1];
  return soft_rc(st);
}

int fpsq_ys_gs(fpsq_handle h, const double* g, const double* c, double sigma, double* gs, double* ys, double* v,
               double* w, fpsq_stats st[2]) {
  if (int rc = check_ready(h)) return rc;
  if (!g || !c || !gs || !ys || !v || !w || !st) {
    h->err = "ys_gs: null argument";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  hipStream_t s = h->stream;
  order_inputs(h);
  const size_t nb = (size_t)h->n * 8, mb = (size_t)h->m * 8;
  HIPCHK(h, hipMemcpyAsync(h->in_n1, g, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(h->in_m, c, mb, hipMemcpyDefault, s));
  call_begin(h);
  if (int rc = two_mixed_device(h, h->in_n1, h->in_m)) return rc;
  // src/model-Fletcherpenaltynlp.jl:244-248
  hipLaunchKernelGGL(k_gs, dim3(ew_grid(h->n)), dim3(kBlock), 0, s, h->p1, h->Cx, sigma, h->gs, h->n);
  hipLaunchKernelGGL(k_ys, dim3(ew_grid(h->m)), dim3(kBlock), 0, s, h->Lx[0], h->Cy, (const double*)nullptr, sigma, h->ys,
                     h->m, (double*)nullptr, (double*)nullptr, (double*)nullptr, (const LaneCtl*)nullptr,
                     (const LaneCtl*)nullptr, seg_none());
  h->launches += 2;
  HIPCHK(h, hipMemcpyAsync(gs, h->gs, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(ys, h->ys, mb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(v, h->Cx, nb, hipMemcpyDefault, s));
  HIPCHK(h, hipMemcpyAsync(w, h->Cy, mb, hipMemcpyDefault, s));
  if (int rc = call_end(h)) return rc;
  st[0] = h->hstats[0];
  st[1] = h->hstats[1];
  return soft_rc(st);
}

int fpsq_jac_mul(fpsq_handle h, int32_t trans, double alpha, const double* x, double beta, double* y) {
  if (int rc = check_ready(h)) return rc;
  if (!x || !y) {
    h->err = "jac_mul: null argument";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  hipStream_t s = h->stream;
  order_inputs(h);
  const size_t xb = (size_t)(trans ? h->m : h->n) * 8, yb = (size_t)(trans ? h->n : h->m) * 8;
  double* dx = trans ? h->in_m : h->in_n1;
  double* dy = trans ? h->in_n2 : h->c;
  HIPCHK(h, hipMemcpyAsync(dx, x, xb, hipMemcpyDefault, s));
  if (beta != 0.0) HIPCHK(h, hipMemcpyAsync(dy, y, yb, hipMemcpyDefault, s));
  call_begin(h);
  if (trans) {
    if (int rc = at_product_const(h, alpha, dx, beta, dy, dy)) return rc;
  } else {
    spmv_const(h, TAG_A, alpha, dx, beta, dy, dy);
  }
  HIPCHK(h, hipMemcpyAsync(y, dy, yb, hipMemcpyDefault, s));
  return call_end(h);
}

// ---------------------------------------------------------------- device-resident equality-QP model

int fpsq_qp_create(fpsq_handle h, const double* qdiag, const double* d, const double* b, fpsq_qp* out) {
  if (!h || !qdiag || !d || !b || !out) return FPSQ_ERR_ARG;
  hipSetDevice(h->opt.device);
  fpsq_qp qp = new fpsq_qp_s();
  qp->h = h;
  if (dalloc(h, &qp->q, (size_t)h->n) || dalloc(h, &qp->d, (size_t)h->n) || dalloc(h, &qp->b, (size_t)h->m)) {
    delete qp;
    return FPSQ_ERR_HIP;
  }
  HIPCHK(h, hipMemcpy(qp->q, qdiag, (size_t)h->n * 8, hipMemcpyDefault));
  HIPCHK(h, hipMemcpy(qp->d, d, (size_t)h->n * 8, hipMemcpyDefault));
  HIPCHK(h, hipMemcpy(qp->b, b, (size_t)h->m * 8, hipMemcpyDefault));
  *out = qp;
  return FPSQ_OK;
}

int fpsq_qp_destroy(fpsq_qp qp) {
  if (!qp) return FPSQ_ERR_ARG;
  delete qp;  // device arrays are owned by the solver handle's arena
  return FPSQ_OK;
}

}  // extern "C"

namespace {
// stand-alone form of qp_fx (sharded runs: the m-vector sums pass through an all-reduce first)
__global__ __launch_bounds__(kBlock) void k_qp_fx(const FxArgs a, const LaneCtl* gate0, const LaneCtl* gate1) {
  if (gate0 != nullptr && !(gate0->done && gate1->done)) return;
  __shared__ double red[kFxRed];
  qp_fx(a, red);
}
}  // namespace

extern "C" {

static int impl_qp_objgrad(fpsq_handle h, fpsq_qp qp, const double* x, double sigma, double rho, double eta,
                    const double* xk, double* fx, double* gx, double* ys, double* gs, fpsq_stats st[2]) {
  if (h && h->ab_dynamic)
    if (const char* ev = std::getenv("FPSQ_AB_MASK")) h->ab_mask = std::atoi(ev);
  if (h && h->host_trace) {
    const auto now = std::chrono::steady_clock::now();
    if (h->ht_have_exit) h->ht_sum[0] += std::chrono::duration<double>(now - h->ht_exit).count();
    h->ht_last = now;
    h->ht_calls++;
  }
  if (int rc = check_ready(h)) return rc;
  if (!qp || qp->h != h || !x || !st || (eta > 0.0 && !xk)) {
    h->err = "qp_objgrad: bad argument";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  hipStream_t s = h->stream;
  order_inputs(h);
  ht_mark(h, 1);
  const int64_t n = h->n, m = h->m;
  const size_t nb = (size_t)n * 8, mb = (size_t)m * 8;
  const int gn = ew_grid(n), gm = ew_grid(m);
  // x and gx resident on this GPU are used in place (no staging copy); host buffers go through h->xin / h->gx
  const double* dx = x;
  if (!on_this_device(h, x)) {
    HIPCHK(h, hipMemcpyAsync(h->xin, x, nb, hipMemcpyDefault, s));
    dx = h->xin;
  }
  double* dgx = gx && on_this_device(h, gx) ? gx : h->gx;
  const double* dxk = nullptr;
  if (eta > 0.0) {
    HIPCHK(h, hipMemcpyAsync(h->xk, xk, nb, hipMemcpyDefault, s));
    dxk = h->xk;
  }
  call_begin(h);
  // user-model evaluations of _compute_ys_gs!  (src/model-Fletcherpenaltynlp.jl:238-240)
  // Fast start (single GPU, fused recurrences): the gradient kernel writes the long pair {g, x} itself and the LSQR
  // start-up product A u~_1, in which the CRAIG lane is otherwise parked, forms c = A x - b on the side: the separate
  // c = A x - b product and the right-hand-side loads of the start-up are not launched.
  const bool local_vec = !h->comm || h->halo;  // single GPU, or row-sharded with column windows (halo mode)
  const bool fast = local_vec && h->opt.fuse_two_rhs != 0 && h->opt.kkt_method == FPSQ_KKT_LSQR_CRAIG;
  {
    QpGradArgs qg{qp->q, qp->d, dx, dxk, h->g, n, h->pQ[0], h->pQ[1], fast ? h->LP : (double*)nullptr,
                  fast ? h->pE : (double*)nullptr, n_owned(h), gn};
    if (fast && !(h->ab_mask & 1)) {
      h->startup_qg = qg;  // evaluated by the start-up launch of the recurrences (k_startup): no launch of its own
    } else {
      hipLaunchKernelGGL(k_qp_grad, dim3(gn), dim3(kBlock), 0, s, qg);
      h->launches++;
    }
  }
  ht_mark(h, 2);
  if (!fast) spmv_const(h, TAG_A, 1.0, dx, -1.0, qp->b, h->c);  // c = A x - b
  // Single GPU with rho > 0: p1 = g - A'q1 and J'c (:424-428) share ONE two-right-hand-side product A'[q1, c], and
  // phi is reduced by an extra workgroup of the gradient kernel: 2 launches fewer at the end of every evaluation.
  // (halo mode: the same product, its overlap rows completed after the neighbour exchange; phi needs its all-reduce)
  const bool paired = local_vec && rho > 0.0;
  // the epilogue then starts with k_ys, which also applies the final LSQR x update (no launch of its own for it)
  h->absorb_flush = paired && fast && !(h->ab_mask & 2);
  const double seq = (h->call_seq += 1.0);
  // everything behind the two solves: enqueued speculatively (gated on the recurrences' `done` flags) by run_krylov when
  // the iteration count of the previous evaluation is known, else here
  TailFn epi = [&]() -> int {
    // ys = q1 + sigma q2 and the dots of objgrad!
    hipLaunchKernelGGL(k_ys, dim3(gm), dim3(kBlock), 0, s, h->Lx[0], h->Cy, h->c, sigma, h->ys, m, h->pC[0], h->pC[1],
                       paired ? h->SP : (double*)nullptr, h->gate0, h->gate1, h->pending_flush);
    h->pending_flush.kind = UPD_NONE;
    h->launches++;
    FxArgs fa{};
    fa.seq = seq;
    fa.pf = h->pQ[0];
    fa.pdx = h->pQ[1];
    fa.np_n = gn;
    fa.pcy = h->pC[0];
    fa.pcc = h->pC[1];
    fa.np_m = gm;
    fa.rho = rho;
    fa.eta = eta;
    fa.out = h->hscal_dev;
    fa.stride = 1;
    // (row-sharded with in-launch sums: the reduction adds the ranks' four local sums up itself, in rank order -- xch_sum)
    const bool in_launch = insum(h);
    if (const XchTable* xt = in_launch ? insum_table(h) : nullptr) {
      fa.xt = xt;
      fa.xseq = ++h->xch_seq;
    }
    FxArgs none = fa;
    none.out = nullptr;
    if (paired) {
      // phi: by default the extra workgroup of the gradient kernel below.  FPSQ_AB_MASK & 4: a one-workgroup launch of its
      // own right behind k_ys (everything it sums is complete there), which puts the scalar on the host a whole product +
      // gradient kernel earlier for the stream-ordered return -- measured on one handle: 814.7 against 818.8 evals/s, the
      // extra launch costs more than the earlier return gains.  (Round 3 first let it ride in the product launch as its
      // first workgroup: the extra case in the product kernel's update switch cost the A' kernel 11 VGPRs and ~5 % of its
      // time -- 2 % of an evaluation.)
      const bool early_fx = in_launch && (h->ab_mask & 4);
      if (early_fx) {
        hipLaunchKernelGGL(k_qp_fx, dim3(1), dim3(kBlock), 0, s, fa, h->gate0, h->gate1);
        h->launches++;
      }
      const bool grad_fx = in_launch && !early_fx;
      // one GPU: the rows of the raw product go straight into grad(phi) -- one launch, and the 16 MB product is neither written nor
      // re-read (k_spmv<.., GRAD>; bitwise the two launches below, FPSQ_FUSE_TAIL=0)
      // (a communicator of ONE rank has no overlap rows and no peers: the single-GPU tail)
      const bool alone = !h->comm || (h->comm->nranks == 1 && in_launch && h->ovl + h->ovr == 0);
      const bool one_launch = h->fuse_tail && alone && h->AT.sorted && h->AT.padded;
      if (one_launch) {
        GradEpi ge{};
        ge.g = h->g;
        ge.v = h->Cx;
        ge.q = qp->q;
        ge.x = dx;
        ge.xk = dxk;
        ge.sigma = sigma;
        ge.rho = rho;
        ge.eta = eta;
        ge.gs = h->gs;
        ge.gx = dgx;
        ge.fx = grad_fx ? fa : none;
        h->tail_grad = &ge;
        h->tail_grad_used = false;
        launch_spmv<2>(h, TAG_AT, h->SP, nullptr, h->LP, h->ctl_raw, h->ctl_raw, nullptr, seg_none(), seg_none(), false);
        h->tail_grad = nullptr;
        if (!h->tail_grad_used) {  // (the product wrote its rows as ever: combine them in a launch of their own)
          hipLaunchKernelGGL(k_qp_penalty_grad, dim3(grad_fx ? gn + 1 : gn), dim3(kBlock), 0, s, (const double*)nullptr, h->g, h->LP,
                             h->Cx, qp->q, (const double*)nullptr, dx, dxk, sigma, rho, eta, h->gs, dgx, n, grad_fx ? fa : none,
                             h->gate0, h->gate1);
          h->launches++;
        }
      } else {
        launch_spmv<2>(h, TAG_AT, h->SP, nullptr, h->LP, h->ctl_raw, h->ctl_raw, nullptr, seg_none(), seg_none(), h->halo);
        if (h->halo)
          if (int rc = halo_finish<2>(h, nullptr, h->LP, h->ctl_raw, h->ctl_raw, nullptr)) return rc;
        hipLaunchKernelGGL(k_qp_penalty_grad, dim3(grad_fx ? gn + 1 : gn), dim3(kBlock), 0, s, (const double*)nullptr, h->g, h->LP,
                           h->Cx, qp->q, (const double*)nullptr, dx, dxk, sigma, rho, eta, h->gs, dgx, n, grad_fx ? fa : none,
                           h->gate0, h->gate1);
        h->launches++;
      }
    } else {
      if (rho > 0.0)
        if (int rc = at_product_const(h, 1.0, h->c, 0.0, nullptr, h->jc)) return rc;  // J'c   (:424-428)
      hipLaunchKernelGGL(k_qp_penalty_grad, dim3(in_launch ? gn + 1 : gn), dim3(kBlock), 0, s, h->p1, h->g,
                         (const double*)nullptr, h->Cx, qp->q, h->jc, dx, dxk, sigma, rho, eta, h->gs, dgx, n,
                         in_launch ? fa : none, h->gate0, h->gate1);
      h->launches++;
    }
    if (!in_launch) {  // phi: c'ys and c'c are sums over the rank's rows only
      PresumArgs P{};
      P.p[0] = fa.pcy;
      P.n[0] = gm;
      P.p[1] = fa.pcc;
      P.n[1] = gm;
      if (h->halo) {  // f and ||x - xk||^2 are sums over the owned part of the rank's column window
        P.p[2] = fa.pf;
        P.n[2] = gn;
        P.p[3] = fa.pdx;
        P.n[3] = gn;
      }
      hipLaunchKernelGGL(k_presum, dim3(1), dim3(kBlock), 0, s, P, h->comm_scal);
      if (h->halo) {
        // the ranks' four local sums are ALL-GATHERED like the norm partials of the loop (through the same two buffers:
        // on the peer-to-peer route no collective call here either) and summed by every rank in rank order: phi is
        // bitwise the same on every rank whatever a reduction algorithm would do
        const int P_ = h->comm->nranks;
        double* gbuf = h->gath + (size_t)(h->gather_calls++ & 1) * (size_t)h->seg_len * P_;
        if (int rc = h->comm->allgather(h->comm_scal, gbuf, 4, s)) {
          h->err = h->comm->err;
          return rc;
        }
        fa.pcy = gbuf;
        fa.pcc = gbuf + 1;
        fa.pf = gbuf + 2;
        fa.pdx = gbuf + 3;
        fa.np_m = fa.np_n = P_;
        fa.stride = 4;
      } else {
        if (int rc = comm_allreduce(h, h->comm_scal, 4)) return rc;
        fa.pcy = h->comm_scal;
        fa.pcc = h->comm_scal + 1;
        fa.np_m = 1;
      }
      hipLaunchKernelGGL(k_qp_fx, dim3(1), dim3(kBlock), 0, s, fa, h->gate0, h->gate1);
      h->launches += 2;
    }
    return 0;
  };
  if (int rc = two_mixed_device(h, h->g, h->c, paired, fast ? qp->b : nullptr, local_vec ? &epi : nullptr)) return rc;
  if (!local_vec)
    if (int rc = epi()) return rc;
  h->absorb_flush = false;
  if (gx && dgx != gx) HIPCHK(h, hipMemcpyAsync(gx, h->gx, nb, hipMemcpyDefault, s));
  if (ys) HIPCHK(h, hipMemcpyAsync(ys, h->ys, mb, hipMemcpyDefault, s));
  if (gs) HIPCHK(h, hipMemcpyAsync(gs, h->gs, nb, hipMemcpyDefault, s));
  ht_mark(h, 5);
  // Stream-ordered outputs (fpsq_set_output_ordering): with every vector argument resident on this GPU the call returns
  // once phi and the statistics are on the host; the registered stream is made to wait for the rest of the epilogue.
  const bool ordered = h->out_ordered && !(h->ab_mask & 8) && h->in_stream_on && !h->profile && insum(h) && paired && dx == x &&
                       (!gx || dgx == gx) && (!ys || on_this_device(h, ys)) && (!gs || on_this_device(h, gs));
  if (ordered) {
    if (int rc = call_end_ordered(h, seq)) return rc;
  } else {
    if (int rc = call_end(h)) return rc;
  }
  ht_mark(h, 6);
  if (fx) *fx = h->hscal[0];
  st[0] = h->hstats[0];
  st[1] = h->hstats[1];
  if (h->host_trace) {
    ht_mark(h, 7);
    h->ht_exit = h->ht_last;
    h->ht_have_exit = true;
  }
  return soft_rc(st);
}

static int impl_qp_hprod(fpsq_handle h, fpsq_qp qp, const double* v, double sigma, double rho, double eta,
                  int32_t hessian_approx, double* Hv, fpsq_stats* st) {
  if (int rc = check_ready(h)) return rc;
  if (!qp || qp->h != h || !v || !Hv || !st || (hessian_approx != 1 && hessian_approx != 2)) {
    h->err = "qp_hprod: bad argument (hessian_approx is 1 or 2)";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  hipStream_t s = h->stream;
  order_inputs(h);
  const int64_t n = h->n;
  const size_t nb = (size_t)n * 8;
  const int gn = ew_grid(n);
  const double* dv = v;
  if (!on_this_device(h, v)) {
    HIPCHK(h, hipMemcpyAsync(h->in_n1, v, nb, hipMemcpyDefault, s));
    dv = h->in_n1;
  }
  double* dhv = on_this_device(h, Hv) ? Hv : h->gx;
  bool lsq_repeats = false;
  call_begin(h);
  hipLaunchKernelGGL(k_qp_hsv, dim3(gn), dim3(kBlock), 0, s, qp->q, dv, h->in_n2, n);                    // :537
  TailFn epi = [&]() -> int {
    bool fin_done = false;
    if (rho > 0.0) {                                                                                      // :557-558
      spmv_const(h, TAG_A, 1.0, dv, 0.0, nullptr, h->in_m);
      // one GPU: the rows of A'(A v) go straight into Hv (k_spmv<1, .., GRAD>; bitwise the product + k_qp_hprod_fin, FPSQ_FUSE_TAIL=0)
      const bool alone = !h->comm || (h->comm->nranks == 1 && h->ovl + h->ovr == 0);
      GradEpi ge{};
      if (h->fuse_tail && alone && h->AT.sorted && h->AT.padded) {
        ge.p1 = h->p1;
        ge.p2 = h->p2b;
        ge.v = dv;
        ge.q = qp->q;
        ge.sigma = sigma;
        ge.rho = rho;
        ge.eta = eta;
        ge.hv = dhv;
        h->tail_grad = &ge;
        h->tail_grad_used = false;
      }
      const int rc = at_product_const(h, 1.0, h->in_m, 0.0, nullptr, h->jc);
      fin_done = h->tail_grad != nullptr && h->tail_grad_used;
      h->tail_grad = nullptr;
      if (rc) return rc;
    }
    if (!fin_done) {
      hipLaunchKernelGGL(k_qp_hprod_fin, dim3(gn), dim3(kBlock), 0, s, h->p1, h->p2b, qp->q, dv, h->jc, sigma, rho, eta, dhv,
                         n, h->gate0, h->gate1);                                                          // :543-562
      h->launches++;
    }
    h->launches++;
    return 0;
  };
  const bool local_vec = !h->comm || h->halo;
  if (int rc = two_least_squares_device(h, dv, h->in_n2, local_vec ? &epi : nullptr)) return rc;         // :542
  if (!local_vec)
    if (int rc = epi()) return rc;
  if (hessian_approx == 1) {
    // Val(1) (src/model-Fletcherpenaltynlp.jl:572-634) adds, on top of everything above:
    //   Ssv = ghjvprod(x, gs, v) = 0 (linear constraints);  (invJtJJv, invJtJSsv) = solve_two_extras(v, Ssv)   :601-602
    //   Hv -= J' invJtJSsv                                                                                  :603-611
    //   Hv -= hprod_nln(x, invJtJJv, gs; obj_weight = 0) = 0                                                :613-614
    // i.e. the LSQR + MINRES lanes of solve_two_extras (tau = max(delta, 1e-14)) on the right-hand sides (v, 0) and one
    // more A' product; the two recurrences' statistics go to st[2], st[3].
    // For this model two of those pieces are known in advance (FPSQ_AB_MASK bit 16 computes them all the same, for the test
    // that holds the results bitwise equal): the MINRES lane's right-hand side is zero, so its solution is zero and so is
    // J' invJtJSsv -- no A' product, nothing to subtract; and when tau == delta (delta >= 1e-14) the LSQR lane repeats, bit
    // for bit, the first solve of solve_two_least_squares above (solve_least_square on the same operator, right-hand side v
    // and damping: linear_system.jl:53 against :87) -- its statistics are that solve's.
    const double tau = std::max(h->delta, 1e-14);
    const bool full = (h->ab_mask & 16) != 0;
    lsq_repeats = !full && tau == h->delta;
    HIPCHK(h, hipMemsetAsync(h->in_m, 0, (size_t)h->m * 8, s));
    Lane lanes[2];
    lanes[0].kind = LANE_LSQR;
    lanes[0].rhs = dv;
    lanes[0].lambda = std::sqrt(tau);
    lanes[0].x = h->Lx[0];
    lanes[0].st = &h->hstats[2];
    lanes[1].kind = LANE_MINRES;
    lanes[1].rhs = h->in_m;
    lanes[1].lambda = tau;
    lanes[1].x = h->Mx;
    lanes[1].st = &h->hstats[3];
    if (lsq_repeats) {
      if (int rc = run_lanes(h, lanes + 1, 1)) return rc;
    } else {
      if (int rc = run_lanes(h, lanes, 2)) return rc;
    }
    if (full) {
      if (int rc = at_product_const(h, 1.0, h->Mx, 0.0, nullptr, h->jc)) return rc;  // J' invJtJSsv
      hipLaunchKernelGGL(k_axpby_plain, dim3(gn), dim3(kBlock), 0, s, h->jc, -1.0, dhv, 1.0, dhv, n);
      h->launches++;
    }
  }
  if (dhv != Hv) HIPCHK(h, hipMemcpyAsync(Hv, dhv, nb, hipMemcpyDefault, s));
  if (int rc = call_end(h)) return rc;
  st[0] = h->hstats[0];
  st[1] = h->hstats[1];
  int rc = soft_rc(st);
  if (hessian_approx == 1) {
    st[2] = lsq_repeats ? h->hstats[0] : h->hstats[2];
    st[3] = h->hstats[3];
    rc |= soft_rc(st + 2) << 2;
  }
  return rc;
}

int fpsq_comm_unique_id(uint8_t id[128]) {
  std::string err;
  if (!id || !g_rccl.load(err)) {
    g_create_error = err.empty() ? "comm_unique_id: null argument" : err;
    return FPSQ_ERR_COMM;
  }
  ncclUniqueId u;
  ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) {
    g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r);
    return FPSQ_ERR_COMM;
  }
  static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(id, &u, 128);
  return FPSQ_OK;
}

int fpsq_comm_init(fpsq_handle h, int32_t nranks, int32_t rank, const uint8_t id[128]) {
  if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks || h->comm) {
    if (h) h->err = "comm_init: bad arguments or communicator already set";
    return FPSQ_ERR_ARG;
  }
  if (!g_rccl.load(h->err)) return FPSQ_ERR_COMM;
  hipSetDevice(h->opt.device);
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  IpcComm* c = new IpcComm();
  c->nranks = nranks;
  c->rank = rank;
  if (const char* ev = std::getenv("FPSQ_COMM_ROUTE"))  // rccl | p2p | auto (developer A/B; fpsq_comm_set_route is the API)
    c->want = !std::strcmp(ev, "rccl") ? FPSQ_ROUTE_RCCL : !std::strcmp(ev, "p2p") ? FPSQ_ROUTE_P2P : FPSQ_ROUTE_AUTO;
  ncclResult_t r = g_rccl.CommInitRank(&c->c, nranks, u, rank);
  if (r != ncclSuccess) {
    h->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r);
    c->c = nullptr;
    delete c;
    return FPSQ_ERR_COMM;
  }
  if (c->nranks > 1) unadopt_stream(h);  // (a sharded handle keeps a stream of its own: its peers' launches must not queue behind the caller's work)
  h->comm = c;
  h->info.comm_route = c->route();
  return FPSQ_OK;
}

int fpsq_comm_set_route(fpsq_handle h, int32_t route) {
  IpcComm* c = h ? dynamic_cast<IpcComm*>(h->comm) : nullptr;
  if (!c || (route != FPSQ_ROUTE_AUTO && route != FPSQ_ROUTE_RCCL && route != FPSQ_ROUTE_P2P)) {
    if (h) h->err = "comm_set_route: needs the communicator of fpsq_comm_init; route is FPSQ_ROUTE_AUTO / _RCCL / _P2P";
    return FPSQ_ERR_ARG;
  }
  if (h->halo || h->gather_ready) {
    h->err = "comm_set_route: call before fpsq_comm_set_halo (the exchange buffers are allocated for the route)";
    return FPSQ_ERR_STATE;
  }
  c->want = route;
  return FPSQ_OK;
}

int fpsq_local_group_create(int32_t nshards, void** out) {
  if (!out || nshards < 1 || nshards > 8) return FPSQ_ERR_ARG;
  LocalGroup* g = new LocalGroup();
  g->n = nshards;
  for (int r = 0; r < nshards; ++r) {
    hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming);
    hipEventCreateWithFlags(&g->copied[r], hipEventDisableTiming);
  }
  hipEventCreateWithFlags(&g->done, hipEventDisableTiming);
  *out = g;
  return FPSQ_OK;
}

int fpsq_local_group_destroy(void* group) {
  LocalGroup* g = (LocalGroup*)group;
  if (!g) return FPSQ_ERR_ARG;
  for (int r = 0; r < g->n; ++r) {
    hipEventDestroy(g->ready[r]);
    hipEventDestroy(g->copied[r]);
  }
  hipEventDestroy(g->done);
  delete g;
  return FPSQ_OK;
}

int fpsq_local_group_set_p2p(void* group, int32_t on) {
  LocalGroup* g = (LocalGroup*)group;
  if (!g) return FPSQ_ERR_ARG;
  g->p2p = on != 0;
  return FPSQ_OK;
}

int fpsq_comm_init_local(fpsq_handle h, void* group, int32_t shard) {
  LocalGroup* g = (LocalGroup*)group;
  if (!h || !g || shard < 0 || shard >= g->n || h->comm) {
    if (h) h->err = "comm_init_local: bad arguments or communicator already set";
    return FPSQ_ERR_ARG;
  }
  LocalComm* c = g->p2p ? new P2PLocalComm() : new LocalComm();
  c->nranks = g->n;
  c->rank = shard;
  c->g = g;
  if (c->nranks > 1) unadopt_stream(h);  // (a sharded handle keeps a stream of its own: its peers' launches must not queue behind the caller's work)
  h->comm = c;
  h->info.comm_route = c->route();
  return FPSQ_OK;
}

int fpsq_comm_set_halo(fpsq_handle h, int64_t overlap_left, int64_t overlap_right) {
  if (!h || !h->comm || overlap_left < 0 || overlap_right < 0 || overlap_left + overlap_right > h->n ||
      (h->comm->rank == 0 && overlap_left != 0) || (h->comm->rank == h->comm->nranks - 1 && overlap_right != 0)) {
    if (h) h->err = "comm_set_halo: needs a communicator; overlaps must fit the window and vanish at the outer ends";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(h->opt.device);
  if (h->gather_ready) {
    h->err = "comm_set_halo: call before the first solve";
    return FPSQ_ERR_STATE;
  }
  if (h->halo_recv) dfree(h, &h->halo_recv);
  if (h->halo_raw) dfree(h, &h->halo_raw);
  if (int rc = xalloc(h, &h->halo_recv, (size_t)(overlap_left + overlap_right) * 2 * 2)) return rc;
  if (int rc = dalloc(h, &h->halo_raw, (size_t)(overlap_left + overlap_right) * 2)) return rc;
  h->halo_gf = overlap_left + overlap_right > 0 ? ew_grid(overlap_left + overlap_right) : 0;
  h->halo = true;
  h->ovl = overlap_left;
  h->ovr = overlap_right;
  if (h->have_structure)
    if (int rc = setup_fused_halo(h)) return rc;
  return FPSQ_OK;
}

int fpsq_get_info(fpsq_handle h, fpsq_info* info) {
  if (!h || !info) return FPSQ_ERR_ARG;
  *info = h->info;
  return FPSQ_OK;
}

int fpsq_debug_expect_iterations(fpsq_handle h, int64_t expect) {
  if (!h) return FPSQ_ERR_ARG;
  h->force_expect = expect;
  return FPSQ_OK;
}

int fpsq_set_profiling(fpsq_handle h, int32_t on) {
  if (!h) return FPSQ_ERR_ARG;
  h->profile = on != 0;
  return FPSQ_OK;
}

// the solve entries, each repeated once on two launches per iteration when a one-launch iteration ran into a bounded wait
int fpsq_solve_two_mixed(fpsq_handle h, const double* rhs1, const double* rhs2, double* p1, double* q1, double* p2,
                         double* q2, fpsq_stats st[2]) {
  return with_fuse_fallback(h, [&]() { return impl_solve_two_mixed(h, rhs1, rhs2, p1, q1, p2, q2, st); });
}
int fpsq_solve_two_least_squares(fpsq_handle h, const double* rhs1, const double* rhs2, double* p1, double* q1,
                                 double* p2, double* q2, fpsq_stats st[2]) {
  return with_fuse_fallback(h, [&]() { return impl_solve_two_least_squares(h, rhs1, rhs2, p1, q1, p2, q2, st); });
}
int fpsq_solve_two_extras(fpsq_handle h, const double* rhs1, const double* rhs2, double* out1, double* out2,
                          fpsq_stats st[2]) {
  return with_fuse_fallback(h, [&]() { return impl_solve_two_extras(h, rhs1, rhs2, out1, out2, st); });
}
int fpsq_qp_objgrad(fpsq_handle h, fpsq_qp qp, const double* x, double sigma, double rho, double eta,
                    const double* xk, double* fx, double* gx, double* ys, double* gs, fpsq_stats st[2]) {
  return with_fuse_fallback(h, [&]() { return impl_qp_objgrad(h, qp, x, sigma, rho, eta, xk, fx, gx, ys, gs, st); });
}
int fpsq_qp_hprod(fpsq_handle h, fpsq_qp qp, const double* v, double sigma, double rho, double eta,
                  int32_t hessian_approx, double* Hv, fpsq_stats* st) {
  return with_fuse_fallback(h, [&]() { return impl_qp_hprod(h, qp, v, sigma, rho, eta, hessian_approx, Hv, st); });
}
}  // extern "C"

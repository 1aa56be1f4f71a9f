// loopback_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl that carries the ten entry
// points libfpsq.so binds (fpsq.hip: RcclApi) between PROCESSES OF ONE HOST through a shared file mapping (/tmp), so that the
// multi-rank code path (bench.py --gpus N: unique-id broadcast, fpsq_comm_init, all-gathers, grouped send/recv halo
// exchanges, all-reduces) can be executed for real on a box with ONE GPU -- RCCL itself refuses two ranks on one device.
// libfpsq.so loads it when FPSQ_RCCL_LIB names it (tests/test_gpu_bench.py).  Semantics: every call synchronises the stream
// it is given and moves the data with blocking copies through host memory (results identical to a real collective, timing
// meaningless); grouped sends / receives are queued and executed at ncclGroupEnd, all sends first.  Every wait is bounded
// (FPSQ_SHIM_TIMEOUT seconds, default 60): a missing peer is an error return, not a hang.
//   hipcc -O2 -fPIC -shared -o tests/shim/libloopback_rccl.so tests/shim/loopback_rccl.cpp
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;  // (the product's peer-to-peer tables hold 8: a 9-rank communicator must stay on the collectives)
constexpr size_t kCtlBytes = 8192;
constexpr size_t kSlotBytes = (size_t)20 << 20;  // per rank (collectives): 2 n doubles of the largest test problem; sparse until touched
constexpr size_t kMailBytes = (size_t)1 << 20;   // per ordered pair (send / recv): a halo is <= 2 x 8192 x 2 doubles

struct Control {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
  std::atomic<unsigned long long> mail_seq[kMaxRanks][kMaxRanks];  // [src][dst]: messages written
  std::atomic<unsigned long long> mail_ack[kMaxRanks][kMaxRanks];  // [src][dst]: messages consumed
  size_t mail_bytes[kMaxRanks][kMaxRanks];
};

struct ShimComm {
  int nranks = 0, rank = 0;
  int real = 0;  // processes that take part in the barriers: nranks, or 1 with FPSQ_SHIM_PHANTOM=1 (the other ranks are phantoms
                 // whose slots stay zero -- lets ONE process hold a communicator of 9+ ranks for the set-up decisions)
  std::string name;
  size_t total = 0;
  char* base = nullptr;
  Control* ctl = nullptr;
  char* slot(int r) { return base + kCtlBytes + (size_t)r * kSlotBytes; }
  char* mail(int src, int dst) { return base + kCtlBytes + (size_t)kMaxRanks * kSlotBytes + (size_t)(src * kMaxRanks + dst) * kMailBytes; }
};

struct Pending {
  bool send;
  const void* sbuf;
  void* rbuf;
  size_t bytes;
  int peer;
  ShimComm* c;
  hipStream_t s;
};
thread_local int g_group_depth = 0;
thread_local std::vector<Pending> g_pending;

double timeout_s() {
  static const double t = std::getenv("FPSQ_SHIM_TIMEOUT") ? std::atof(std::getenv("FPSQ_SHIM_TIMEOUT")) : 60.0;
  return t;
}

template <class F>
bool wait_until(F cond) {
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (!cond()) {
    if (++spins > 200) {
      sched_yield();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) return false;
    }
  }
  return true;
}

bool barrier(ShimComm* c) {
  Control* k = c->ctl;
  const int gen = k->generation.load(std::memory_order_acquire);
  if (k->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->real) {
    k->arrived.store(0, std::memory_order_relaxed);
    k->generation.store(gen + 1, std::memory_order_release);
    return true;
  }
  return wait_until([&] { return k->generation.load(std::memory_order_acquire) != gen; });
}

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

ncclResult_t do_send(const Pending& p) {
  ShimComm* c = p.c;
  auto& seq = c->ctl->mail_seq[c->rank][p.peer];
  auto& ack = c->ctl->mail_ack[c->rank][p.peer];
  if (p.bytes > kMailBytes) return ncclInvalidArgument;
  if (!wait_until([&] { return ack.load(std::memory_order_acquire) == seq.load(std::memory_order_relaxed); })) return ncclSystemError;
  if (hipMemcpy(c->mail(c->rank, p.peer), p.sbuf, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  c->ctl->mail_bytes[c->rank][p.peer] = p.bytes;
  seq.fetch_add(1, std::memory_order_release);
  return ncclSuccess;
}

ncclResult_t do_recv(const Pending& p) {
  ShimComm* c = p.c;
  auto& seq = c->ctl->mail_seq[p.peer][c->rank];
  auto& ack = c->ctl->mail_ack[p.peer][c->rank];
  if (!wait_until([&] { return seq.load(std::memory_order_acquire) > ack.load(std::memory_order_relaxed); })) return ncclSystemError;
  if (c->ctl->mail_bytes[p.peer][c->rank] != p.bytes) return ncclInvalidArgument;  // mismatched send / recv sizes
  if (hipMemcpy(p.rbuf, c->mail(p.peer, c->rank), p.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  ack.fetch_add(1, std::memory_order_release);
  return ncclSuccess;
}

ncclResult_t flush_group() {
  ncclResult_t r = ncclSuccess;
  for (const Pending& p : g_pending)
    if (hipStreamSynchronize(p.s) != hipSuccess) r = ncclUnhandledCudaError;
  for (const Pending& p : g_pending)
    if (r == ncclSuccess && p.send) r = do_send(p);
  for (const Pending& p : g_pending)
    if (r == ncclSuccess && !p.send) r = do_recv(p);
  g_pending.clear();
  return r;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof(*id));
  const unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof(id->internal), "/tmp/fpsq_shim_%d_%llx", (int)getpid(), t);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ShimComm* c = new ShimComm();
  c->nranks = nranks;
  c->rank = rank;
  c->real = std::getenv("FPSQ_SHIM_PHANTOM") && std::atoi(std::getenv("FPSQ_SHIM_PHANTOM")) != 0 ? 1 : nranks;
  c->name = std::string(id.internal, strnlen(id.internal, sizeof(id.internal)));
  c->total = kCtlBytes + (size_t)kMaxRanks * kSlotBytes + (size_t)kMaxRanks * kMaxRanks * kMailBytes;
  const int fd = open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->total) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    return ncclSystemError;
  }
  void* p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->base = (char*)p;
  c->ctl = (Control*)p;  // (a fresh file is zero-filled: all counters start at 0)
  static_assert(sizeof(Control) <= kCtlBytes, "control block");
  c->ctl->attached.fetch_add(1, std::memory_order_acq_rel);
  if (!wait_until([&] { return c->ctl->attached.load(std::memory_order_acquire) >= c->real; }) || !barrier(c)) {
    munmap(c->base, c->total);
    unlink(c->name.c_str());
    delete c;
    return ncclSystemError;
  }
  if (rank == 0) unlink(c->name.c_str());  // every rank holds its mapping: the name can go
  *comm = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  ShimComm* c = (ShimComm*)comm;
  if (!c) return ncclInvalidArgument;
  munmap(c->base, c->total);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
  ShimComm* c = (ShimComm*)comm;
  const size_t bytes = sendcount * type_bytes(datatype);
  if (!c || bytes == 0 || bytes > kSlotBytes) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slot(c->rank), sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  for (int r = 0; r < c->nranks; ++r)
    if (hipMemcpy((char*)recvbuff + (size_t)r * bytes, c->slot(r), bytes, hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
  return barrier(c) ? ncclSuccess : ncclSystemError;  // (the slots may be rewritten after this)
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
  ShimComm* c = (ShimComm*)comm;
  if (!c || datatype != ncclFloat64 || op != ncclSum || count * 8 > kSlotBytes) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slot(c->rank), sendbuff, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  std::vector<double> sum(count, 0.0);
  for (int r = 0; r < c->nranks; ++r) {  // rank order: the same bits on every rank
    const double* s = (const double*)c->slot(r);
    for (size_t i = 0; i < count; ++i) sum[i] += s[i];
  }
  if (hipMemcpy(recvbuff, sum.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  ShimComm* c = (ShimComm*)comm;
  if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
  g_pending.push_back(Pending{true, sendbuff, nullptr, count * type_bytes(datatype), peer, c, stream});
  return g_group_depth > 0 ? ncclSuccess : flush_group();
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  ShimComm* c = (ShimComm*)comm;
  if (!c || peer < 0 || peer >= c->nranks || peer == c->rank) return ncclInvalidArgument;
  g_pending.push_back(Pending{false, nullptr, recvbuff, count * type_bytes(datatype), peer, c, stream});
  return g_group_depth > 0 ? ncclSuccess : flush_group();
}

ncclResult_t ncclGroupStart() {
  ++g_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  return --g_group_depth == 0 ? flush_group() : ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success (loopback shim)";
    case ncclUnhandledCudaError: return "HIP error (loopback shim)";
    case ncclSystemError: return "peer did not arrive within the shim's timeout, or shared memory failed (loopback shim)";
    case ncclInvalidArgument: return "invalid argument (loopback shim)";
    case ncclInvalidUsage: return "invalid usage (loopback shim)";
    default: return "error (loopback shim)";
  }
}

}  // extern "C"

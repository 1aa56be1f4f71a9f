// fpsq_dense.hip -- host side of the dense-block direct back-end (C ABI: include/fpsq.h, "dense" section).
#include "../../include/fpsq.h"
#include "fpsq_dense.hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace fpsq;

struct fpsq_dense_s {
  int64_t n = 0, m = 0, npad = 0, mpad = 0, nb = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  bool have_jac = false, factored = false;
  double* A = nullptr;     // mpad x npad, row-major, zero padded
  double* M = nullptr;     // mpad x mpad: lower triangle holds the Cholesky factor after factorize
  double* invs = nullptr;  // nb inverses of the diagonal 128 x 128 blocks of L
  double* invsT = nullptr; // ... and their transposes (k_potrf_inv128m, k_trsv_step3)
  int64_t regularized = 0; // pivots replaced by the dynamic regularisation in the last factorisation
  double *r2 = nullptr, *y2 = nullptr, *x2 = nullptr, *part = nullptr;  // [mpad][2], [mpad][2], [npad][2], gemvt partials
  double *in_a = nullptr, *in_b = nullptr, *o_p1 = nullptr, *o_p2 = nullptr, *o_q1 = nullptr, *o_q2 = nullptr;
  int* info_dev = nullptr;
  int nchunk = 16;
  // (one generation of every kernel is left in the source: the sixteen-wave Gram product k_gemm_nt_f64_w16, the diagonal-block
  // kernel k_potrf_inv128m, the single-round-trip step products k_gemm128_lds and the latency-organised solve step
  // k_trsv_step3.  Their predecessors, the look-ahead and split-K variants -- all measured slower, DESIGN.md section 7 --
  // and the environment switches that selected them were removed in round 3.)
  double piv_tol = 0.0, piv_reg = 0.0;  // dynamic regularisation (fpsq_dense_set_regularization); reg <= 0: off
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  // the triangular sweeps in one launch each (k_trsv_chain): publication buffer, launch number, host-mapped error word
  unsigned long long* chain_pub = nullptr;
  unsigned long long* chain_err = nullptr;
  unsigned int chain_seq = 0;
  bool chain = true;  // FPSQ_TRSV_CHAIN=0: one launch per step (k_trsv_step3)
  bool chain_break = false;  // FPSQ_DEBUG_CHAIN_BREAK=1 (tests): the workgroups publish a wrong launch number
  // jac_coord! hand-over (fpsq_dense_set_structure_coo): the caller's COO entries sorted by target, duplicates grouped
  int64_t coo_nnz = -1, coo_slots = 0;
  int32_t *coo_perm = nullptr, *coo_slotptr = nullptr;
  int64_t* coo_target = nullptr;
  double* coo_in = nullptr;
  fpsq_dense_info info{};
  std::vector<void*> allocs;
};

namespace {
thread_local std::string g_dense_create_error;

// COO triplets (any order, duplicates allowed, `base`-based) -> row-major sorted slots.  order[k]: the caller's index of the
// k-th sorted entry (stable: duplicates keep the caller's order); slotptr: one range of sorted entries per distinct (row,
// col); srow / scol: the slots' coordinates.  Returns an error text, empty on success.
std::string coo_sort(int64_t m, int64_t n, int64_t nnz, const int64_t* rows, const int64_t* cols, int32_t base,
                     std::vector<int32_t>& order, std::vector<int32_t>& slotptr, std::vector<int32_t>& srow,
                     std::vector<int32_t>& scol) {
  std::vector<int32_t> cnt(m + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t r = rows[k] - base, c = cols[k] - base;
    if (r < 0 || r >= m || c < 0 || c >= n) return "COO index out of range";
    cnt[r + 1]++;
  }
  for (int64_t i = 0; i < m; ++i) cnt[i + 1] += cnt[i];
  order.resize(nnz);
  {
    std::vector<int32_t> next(cnt.begin(), cnt.end() - 1);
    for (int64_t k = 0; k < nnz; ++k) order[next[rows[k] - base]++] = (int32_t)k;
  }
  for (int64_t i = 0; i < m; ++i)
    std::stable_sort(order.begin() + cnt[i], order.begin() + cnt[i + 1],
                     [&](int32_t a, int32_t b) { return cols[a] < cols[b]; });
  slotptr.assign(1, 0);
  srow.clear();
  scol.clear();
  for (int64_t i = 0; i < m; ++i)
    for (int32_t k = cnt[i]; k < cnt[i + 1]; ++k) {
      const int64_t c = cols[order[k]] - base;
      if (k > cnt[i] && c == cols[order[k - 1]] - base) {
        slotptr.back() = k + 1;
      } else {
        srow.push_back((int32_t)i);
        scol.push_back((int32_t)c);
        slotptr.push_back(k + 1);
      }
    }
  return "";
}

#define DCHK(d, call)                                                          \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      (d)->err = std::string(#call) + ": " + hipGetErrorString(e_);            \
      return FPSQ_ERR_HIP;                                                     \
    }                                                                          \
  } while (0)

template <class T>
int dmalloc(fpsq_dense d, T** p, size_t count) {
  void* q = nullptr;
  DCHK(d, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  d->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// q (m x 2 in d->r2, overwritten) <- M^-1 r2 via L y = r, L' q = y; result in d->y2 after the backward sweep
void solve_two_rhs(fpsq_dense d) {
  hipStream_t s = d->stream;
  const int nb = (int)d->nb, ld = (int)d->mpad;
  if (d->chain) {
    // (tickets: word 1 behind the publication buffer counts every workgroup of every sweep of this handle, nb per launch)
    ChainArgs c{d->chain_pub, ++d->chain_seq, 0, nb, 0, 0, 0, d->chain_err, d->chain_pub + (size_t)nb * 512 + 1, 0};
    c.pubseq = d->chain_break ? ~c.seq : c.seq;
    c.ticket_base = (unsigned long long)(d->chain_seq - 1) * nb;
    hipLaunchKernelGGL(k_trsv_chain<true>, dim3(nb), dim3(256), 0, s, d->M, ld, d->invs, d->invsT, d->r2, d->y2, c);
    c.seq = ++d->chain_seq;
    c.pubseq = d->chain_break ? ~c.seq : c.seq;
    c.ticket_base = (unsigned long long)(d->chain_seq - 1) * nb;
    hipLaunchKernelGGL(k_trsv_chain<false>, dim3(nb), dim3(256), 0, s, d->M, ld, d->invs, d->invsT, d->y2, d->r2, c);
    return;
  }
  for (int k = 0; k < nb; ++k)
    hipLaunchKernelGGL(k_trsv_step3<true>, dim3(nb - k), dim3(256), 0, s, d->M, ld, d->invs, d->invsT, d->r2, d->y2, k, 0);
  for (int k = nb - 1; k >= 0; --k)
    hipLaunchKernelGGL(k_trsv_step3<false>, dim3(k + 1), dim3(256), 0, s, d->M, ld, d->invs, d->invsT, d->y2, d->r2, k, 0);
  // solution now in d->r2
}

// common tail: Q in d->r2 ([mpad][2]); P = [a0, a1] - A' Q
int finish(fpsq_dense d, const double* a0, const double* a1, double* p1, double* q1, double* p2, double* q2) {
  hipStream_t s = d->stream;
  const int rows_per_chunk = (int)((d->mpad + d->nchunk - 1) / d->nchunk);
  hipLaunchKernelGGL(k_dense_gemvt_part<2>, dim3((unsigned)((d->npad + 255) / 256), d->nchunk), dim3(256), 0, s, d->A,
                     (int)d->npad, (int)d->mpad, (int)d->npad, d->r2, d->part, rows_per_chunk);
  hipLaunchKernelGGL(k_dense_finish_p, dim3((unsigned)((d->n + 255) / 256)), dim3(256), 0, s, d->part, d->nchunk,
                     (int)d->npad, (int)d->n, a0, a1, d->o_p1, d->o_p2);
  hipLaunchKernelGGL(k_dense_unpack2, dim3((unsigned)((d->m + 255) / 256)), dim3(256), 0, s, d->r2, d->o_q1, d->o_q2,
                     (int)d->m);
  hipEventRecord(d->e1, s);
  DCHK(d, hipMemcpyAsync(p1, d->o_p1, (size_t)d->n * 8, hipMemcpyDefault, s));
  DCHK(d, hipMemcpyAsync(p2, d->o_p2, (size_t)d->n * 8, hipMemcpyDefault, s));
  DCHK(d, hipMemcpyAsync(q1, d->o_q1, (size_t)d->m * 8, hipMemcpyDefault, s));
  DCHK(d, hipMemcpyAsync(q2, d->o_q2, (size_t)d->m * 8, hipMemcpyDefault, s));
  DCHK(d, hipStreamSynchronize(s));
  if (d->chain_err && *d->chain_err) {
    *d->chain_err = 0;
    d->err = "triangular sweep: a block's solution did not arrive (bounded wait expired); FPSQ_TRSV_CHAIN=0 avoids the path";
    return FPSQ_ERR_TIMEOUT;
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, d->e0, d->e1);
  d->info.last_solve_ms = ms;
  return FPSQ_OK;
}
}  // namespace

extern "C" {

const char* fpsq_dense_last_error(fpsq_dense d) { return d ? d->err.c_str() : g_dense_create_error.c_str(); }

int fpsq_dense_create(fpsq_dense* out, int64_t n, int64_t m, int32_t device) {
  if (!out || n <= 0 || m <= 0 || n > (1 << 20) || m > (1 << 16)) {
    g_dense_create_error = "fpsq_dense_create: bad arguments (n <= 2^20, m <= 2^16)";
    return FPSQ_ERR_ARG;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    g_dense_create_error = std::string("fpsq_dense_create: no HIP device (") + hipGetErrorString(e) +
                           "); libfpsq has no CPU fallback";
    return FPSQ_ERR_HIP;
  }
  fpsq_dense d = new fpsq_dense_s();
  d->n = n;
  d->m = m;
  d->device = device;
  d->mpad = (m + kDB - 1) / kDB * kDB;
  d->npad = (n + kW16Kd - 1) / kW16Kd * kW16Kd;  // whole k-stages of the Gram product
  d->nb = d->mpad / kDB;
  d->nchunk = (int)std::min<int64_t>(32, d->nb * 4);
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
    g_dense_create_error = "fpsq_dense_create: cannot initialise device";
    delete d;
    return FPSQ_ERR_HIP;
  }
  hipEventCreate(&d->e0);
  hipEventCreate(&d->e1);
  hipEventCreate(&d->e2);
  int rc = 0;
  rc |= dmalloc(d, &d->A, (size_t)d->mpad * d->npad);
  rc |= dmalloc(d, &d->M, (size_t)d->mpad * d->mpad);
  rc |= dmalloc(d, &d->invs, (size_t)d->nb * kDB * kDB);
  rc |= dmalloc(d, &d->invsT, (size_t)d->nb * kDB * kDB);
  if (!rc) {  // k_potrf_inv128m writes the non-zero triangles only
    hipMemset(d->invs, 0, (size_t)d->nb * kDB * kDB * 8);
    hipMemset(d->invsT, 0, (size_t)d->nb * kDB * kDB * 8);
  }
  rc |= dmalloc(d, &d->r2, (size_t)d->mpad * 2);
  rc |= dmalloc(d, &d->y2, (size_t)d->mpad * 2);
  rc |= dmalloc(d, &d->x2, (size_t)d->npad * 2);
  rc |= dmalloc(d, &d->part, (size_t)d->nchunk * d->npad * 2);
  rc |= dmalloc(d, &d->in_a, (size_t)d->npad);
  rc |= dmalloc(d, &d->in_b, (size_t)std::max(d->npad, d->mpad));
  rc |= dmalloc(d, &d->o_p1, (size_t)d->npad);
  rc |= dmalloc(d, &d->o_p2, (size_t)d->npad);
  rc |= dmalloc(d, &d->o_q1, (size_t)d->mpad);
  rc |= dmalloc(d, &d->o_q2, (size_t)d->mpad);
  rc |= dmalloc(d, &d->info_dev, 4);
  rc |= dmalloc(d, &d->chain_pub, (size_t)d->nb * 512 + 8);  // (+ the abort word)
  if (!rc) hipMemset(d->chain_pub, 0, ((size_t)d->nb * 512 + 8) * 8);
  if (hipHostMalloc((void**)&d->chain_err, 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) rc = 1;
  else *d->chain_err = 0;
  if (const char* e = getenv("FPSQ_TRSV_CHAIN")) d->chain = atoi(e) != 0;
  if (const char* e = getenv("FPSQ_DEBUG_CHAIN_BREAK")) d->chain_break = atoi(e) != 0;
  if (rc) {
    g_dense_create_error = d->err;
    fpsq_dense_destroy(d);
    return FPSQ_ERR_HIP;
  }
  // on the solver's own (non-blocking) stream: a null-stream memset is not ordered against it
  hipMemsetAsync(d->A, 0, (size_t)d->mpad * d->npad * 8, d->stream);
  hipStreamSynchronize(d->stream);
  hipFuncSetAttribute((const void*)k_potrf_inv128m, hipFuncAttributeMaxDynamicSharedMemorySize, kPotrfLds5);
  hipFuncSetAttribute((const void*)k_gemm128_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kG128Lds0);
  hipFuncSetAttribute((const void*)k_gemm128_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kG128Lds1);
  hipFuncSetAttribute((const void*)k_gemm_nt_f64_w16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kW16Lds);
  d->info.n = n;
  d->info.m = m;
  *out = d;
  return FPSQ_OK;
}

int fpsq_dense_destroy(fpsq_dense d) {
  if (!d) return FPSQ_ERR_ARG;
  hipSetDevice(d->device);
  if (d->stream) hipStreamSynchronize(d->stream);
  for (void* p : d->allocs) hipFree(p);
  if (d->chain_err) hipHostFree(d->chain_err);
  if (d->e0) hipEventDestroy(d->e0);
  if (d->e1) hipEventDestroy(d->e1);
  if (d->e2) hipEventDestroy(d->e2);
  if (d->stream) hipStreamDestroy(d->stream);
  delete d;
  return FPSQ_OK;
}

int fpsq_dense_set_jacobian(fpsq_dense d, const double* a_rowmajor) {
  if (!d || !a_rowmajor) return FPSQ_ERR_ARG;
  hipSetDevice(d->device);
  DCHK(d, hipMemcpy2DAsync(d->A, (size_t)d->npad * 8, a_rowmajor, (size_t)d->n * 8, (size_t)d->n * 8, (size_t)d->m,
                           hipMemcpyDefault, d->stream));
  DCHK(d, hipStreamSynchronize(d->stream));
  d->have_jac = true;
  d->factored = false;
  return FPSQ_OK;
}

int fpsq_dense_set_structure_coo(fpsq_dense d, int64_t nnz, const int64_t* rows, const int64_t* cols, int32_t index_base) {
  if (!d || nnz < 0 || nnz >= INT32_MAX || (nnz > 0 && (!rows || !cols))) return FPSQ_ERR_ARG;
  hipSetDevice(d->device);
  std::vector<int64_t> r(nnz), c(nnz);
  if (nnz) {
    DCHK(d, hipMemcpy(r.data(), rows, (size_t)nnz * 8, hipMemcpyDefault));
    DCHK(d, hipMemcpy(c.data(), cols, (size_t)nnz * 8, hipMemcpyDefault));
  }
  std::vector<int32_t> order, slotptr, srow, scol;
  const std::string msg = coo_sort(d->m, d->n, nnz, r.data(), c.data(), index_base, order, slotptr, srow, scol);
  if (!msg.empty()) {
    d->err = "dense_set_structure_coo: " + msg;
    return FPSQ_ERR_ARG;
  }
  const int64_t ns = (int64_t)srow.size();
  std::vector<int64_t> target(std::max<int64_t>(ns, 1));
  for (int64_t i = 0; i < ns; ++i) target[i] = (int64_t)srow[i] * d->npad + scol[i];
  const bool dup = ns != nnz;
  // a second structure call replaces the first one's buffers (and must not keep its slot table when the new pattern has no
  // duplicates: k_coo_to_slots takes a non-null table for one)
  for (void** q : {(void**)&d->coo_perm, (void**)&d->coo_in, (void**)&d->coo_target, (void**)&d->coo_slotptr}) {
    if (!*q) continue;
    auto it = std::find(d->allocs.begin(), d->allocs.end(), *q);
    if (it != d->allocs.end()) d->allocs.erase(it);
    hipFree(*q);
    *q = nullptr;
  }
  d->coo_nnz = -1;
  if (dmalloc(d, &d->coo_perm, (size_t)std::max<int64_t>(nnz, 1)) || dmalloc(d, &d->coo_in, (size_t)std::max<int64_t>(nnz, 1)) ||
      dmalloc(d, &d->coo_target, target.size()) || (dup && dmalloc(d, &d->coo_slotptr, slotptr.size())))
    return FPSQ_ERR_HIP;
  if (nnz) DCHK(d, hipMemcpy(d->coo_perm, order.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
  DCHK(d, hipMemcpy(d->coo_target, target.data(), target.size() * 8, hipMemcpyHostToDevice));
  if (dup) DCHK(d, hipMemcpy(d->coo_slotptr, slotptr.data(), slotptr.size() * 4, hipMemcpyHostToDevice));
  // entries outside the pattern are zero for good: the value hand-over only rewrites the pattern's slots
  DCHK(d, hipMemsetAsync(d->A, 0, (size_t)d->mpad * d->npad * 8, d->stream));
  DCHK(d, hipStreamSynchronize(d->stream));
  d->coo_nnz = nnz;
  d->coo_slots = ns;
  d->have_jac = false;
  return FPSQ_OK;
}

int fpsq_dense_set_jacobian_coo(fpsq_dense d, const double* vals) {
  if (!d || d->coo_nnz < 0 || (!vals && d->coo_nnz > 0)) {
    if (d) d->err = "dense_set_jacobian_coo: structure not set (fpsq_dense_set_structure_coo) or null values";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(d->device);
  if (d->coo_nnz > 0) {
    DCHK(d, hipMemcpyAsync(d->coo_in, vals, (size_t)d->coo_nnz * 8, hipMemcpyDefault, d->stream));
    hipLaunchKernelGGL(k_coo_to_slots, dim3((unsigned)std::min<int64_t>((d->coo_slots + 255) / 256, 4096)), dim3(256), 0,
                       d->stream, d->coo_in, d->coo_perm, d->coo_slotptr, d->coo_target, d->A, d->coo_slots);
  }
  DCHK(d, hipStreamSynchronize(d->stream));
  d->have_jac = true;
  d->factored = false;
  return FPSQ_OK;
}

int fpsq_dense_factorize(fpsq_dense d, double delta, int32_t* info) {
  if (!d || !(delta >= 0.0)) return FPSQ_ERR_ARG;
  if (!d->have_jac) {
    d->err = "dense_factorize: Jacobian not set";
    return FPSQ_ERR_STATE;
  }
  hipSetDevice(d->device);
  hipStream_t s = d->stream;
  const int nb = (int)d->nb, ld = (int)d->mpad;
  DCHK(d, hipMemsetAsync(d->info_dev, 0, 8, s));
  hipEventRecord(d->e0, s);
  // M = A A' (lower tiles) on the fp64 matrix cores, then + delta I
  hipLaunchKernelGGL(k_gemm_nt_f64_w16<true>, dim3(nb, nb), dim3(1024), kW16Lds, s, d->M, ld, d->A, (int)d->npad, d->A,
                     (int)d->npad, (int)d->npad, 1.0, 0.0, 0, (size_t)0);
  hipLaunchKernelGGL(k_dense_diag, dim3((unsigned)((d->mpad + 255) / 256)), dim3(256), 0, s, d->M, ld, (int)d->m, (int)d->mpad,
                     delta);
  hipEventRecord(d->e1, s);
  // right-looking blocked Cholesky, block 128: potrf + inverse of the diagonal block (one workgroup), panel
  // L_ik = M_ik Linv_kk' and trailing update M_ij -= L_ik L_jk' on the matrix cores
  for (int k = 0; k < nb; ++k) {
    double* Mkk = d->M + (size_t)k * kDB * ld + (size_t)k * kDB;
    double* inv = d->invs + (size_t)k * kDB * kDB;
    hipLaunchKernelGGL(k_potrf_inv128m, dim3(1), dim3(kPotrfThreads5), kPotrfLds5, s, Mkk, ld, inv,
                       d->invsT + (size_t)k * kDB * kDB, k * kDB, d->info_dev, d->piv_tol, d->piv_reg);
    const int rem = nb - k - 1;
    if (rem > 0) {  // the K = 128 products of the step, each in one memory round trip (k_gemm128_lds)
      double* panel = d->M + (size_t)(k + 1) * kDB * ld + (size_t)k * kDB;
      double* trail = d->M + (size_t)(k + 1) * kDB * ld + (size_t)(k + 1) * kDB;
      hipLaunchKernelGGL(k_gemm128_lds<1>, dim3(1, 4 * rem), dim3(1024), kG128Lds1, s, panel, ld, panel, ld, inv, kDB,
                         BlockStrides{});
      hipLaunchKernelGGL(k_gemm128_lds<0>, dim3(2 * rem, 2 * rem), dim3(1024), kG128Lds0, s, trail, ld, panel, ld, panel, ld,
                         BlockStrides{});
    }
  }
  hipEventRecord(d->e2, s);
  int32_t hinfo2[2] = {0, 0};
  DCHK(d, hipMemcpyAsync(hinfo2, d->info_dev, 8, hipMemcpyDeviceToHost, s));
  DCHK(d, hipStreamSynchronize(s));
  float a = 0.f, b = 0.f;
  hipEventElapsedTime(&a, d->e0, d->e1);
  hipEventElapsedTime(&b, d->e1, d->e2);
  d->info.last_syrk_ms = a;
  d->info.last_chol_ms = b;
  const int32_t hinfo = hinfo2[0];
  d->regularized = hinfo2[1];
  d->info.regularized_pivots = hinfo2[1];
  if (info) *info = hinfo;
  d->factored = hinfo == 0;
  return hinfo == 0 ? FPSQ_OK : 1;  // soft failure: M not positive definite (the reference warns and goes on, :244-246)
}

int fpsq_dense_set_regularization(fpsq_dense d, double tol, double reg) {
  if (!d || !(tol >= 0.0)) return FPSQ_ERR_ARG;
  d->piv_tol = tol;
  d->piv_reg = reg;
  return FPSQ_OK;
}

int fpsq_dense_solve_two_mixed(fpsq_dense d, const double* rhs1, const double* rhs2, double* p1, double* q1, double* p2,
                               double* q2) {
  if (!d || !rhs1 || !rhs2 || !p1 || !q1 || !p2 || !q2) return FPSQ_ERR_ARG;
  if (!d->factored) {
    d->err = "dense_solve: no valid factorisation";
    return FPSQ_ERR_STATE;
  }
  hipSetDevice(d->device);
  hipStream_t s = d->stream;
  DCHK(d, hipMemcpyAsync(d->in_a, rhs1, (size_t)d->n * 8, hipMemcpyDefault, s));
  DCHK(d, hipMemcpyAsync(d->in_b, rhs2, (size_t)d->m * 8, hipMemcpyDefault, s));
  hipEventRecord(d->e0, s);
  // r = [A g, -c]:  q1 = M^-1 A g,  q2 = -M^-1 c   
  hipLaunchKernelGGL(k_dense_pack2, dim3((unsigned)((d->npad + 255) / 256)), dim3(256), 0, s, d->in_a, 1.0,
                     (const double*)nullptr, 0.0, d->x2, (int)d->n, (int)d->npad);
  hipLaunchKernelGGL(k_dense_gemv<2>, dim3((unsigned)((d->mpad + 3) / 4)), dim3(256), 0, s, d->A, (int)d->npad,
                     (int)d->mpad, (int)d->npad, d->x2, 1.0, (const double*)nullptr, 0.0, d->y2);
  hipLaunchKernelGGL(k_dense_unpack2, dim3((unsigned)((d->mpad + 255) / 256)), dim3(256), 0, s, d->y2, d->o_q1, d->o_q2,
                     (int)d->mpad);
  hipLaunchKernelGGL(k_dense_pack2, dim3((unsigned)((d->mpad + 255) / 256)), dim3(256), 0, s, d->o_q1, 1.0, d->in_b, -1.0,
                     d->r2, (int)d->m, (int)d->mpad);
  solve_two_rhs(d);
  return finish(d, d->in_a, nullptr, p1, q1, p2, q2);
}

int fpsq_dense_solve_two_least_squares(fpsq_dense d, const double* rhs1, const double* rhs2, double* p1, double* q1,
                                       double* p2, double* q2) {
  if (!d || !rhs1 || !rhs2 || !p1 || !q1 || !p2 || !q2) return FPSQ_ERR_ARG;
  if (!d->factored) {
    d->err = "dense_solve: no valid factorisation";
    return FPSQ_ERR_STATE;
  }
  hipSetDevice(d->device);
  hipStream_t s = d->stream;
  DCHK(d, hipMemcpyAsync(d->in_a, rhs1, (size_t)d->n * 8, hipMemcpyDefault, s));
  DCHK(d, hipMemcpyAsync(d->in_b, rhs2, (size_t)d->n * 8, hipMemcpyDefault, s));
  hipEventRecord(d->e0, s);
  hipLaunchKernelGGL(k_dense_pack2, dim3((unsigned)((d->npad + 255) / 256)), dim3(256), 0, s, d->in_a, 1.0, d->in_b, 1.0,
                     d->x2, (int)d->n, (int)d->npad);
  hipLaunchKernelGGL(k_dense_gemv<2>, dim3((unsigned)((d->mpad + 3) / 4)), dim3(256), 0, s, d->A, (int)d->npad,
                     (int)d->mpad, (int)d->npad, d->x2, 1.0, (const double*)nullptr, 0.0, d->r2);
  solve_two_rhs(d);
  return finish(d, d->in_a, d->in_b, p1, q1, p2, q2);
}

int fpsq_dense_get_factor(fpsq_dense d, double* l_out) {
  if (!d || !l_out) return FPSQ_ERR_ARG;
  hipSetDevice(d->device);
  DCHK(d, hipMemcpy2D(l_out, (size_t)d->m * 8, d->M, (size_t)d->mpad * 8, (size_t)d->m * 8, (size_t)d->m, hipMemcpyDefault));
  return FPSQ_OK;
}

// ===================================================================================== sparse direct path (block band)

}  // extern "C"

struct fpsq_band_s {
  int64_t n = 0, m = 0, nnz = 0, mpad = 0, nb = 0;
  int band_w = 1;  // blocks per block row of the band storage = half bandwidth (in blocks) + 1
  int span = 0;    // widest column span of a row (LDS window of k_band_form)
  // row reordering chosen by the symbolic phase (reverse Cuthill-McKee on the rows of A, adjacent = sharing a column):
  // row p of the stored structure is row rperm[p] of the caller's; vperm maps stored entries to the caller's
  bool reordered = false;
  std::vector<int32_t> rperm_host;
  // two elimination chains (see fpsq_band_create): blocks 2 c / 2 c + 1, c < chain_safe, are eliminated side by side on
  // two streams; their couplings reach chain_bw blocks of the same chain (stride 2 in the stored order)
  int chain_safe = 0, chain_bw = 0;
  int32_t *rperm = nullptr, *vperm = nullptr;
  double *vals_in = nullptr, *in_bp = nullptr;
  int form_gen = 2, form_R = 1;  // 2: k_band_form_t (by columns of A, form_R rows per pass); 1: k_band_form (row pairs)
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  bool factored = false;
  int32_t *rowptr = nullptr, *colind = nullptr, *t_rowptr = nullptr, *t_colind = nullptr, *t_perm = nullptr;
  int2* rowspan = nullptr;
  double *vals = nullptr, *t_vals = nullptr;
  double* Mb = nullptr;    // nb x band_w blocks of 128 x 128
  double *invs = nullptr, *invsT = nullptr;
  double *xn = nullptr, *ym = nullptr, *r2 = nullptr, *y2 = nullptr, *atq = nullptr;  // [n][2], [mpad][2] x 3, [n][2]
  double *in_a = nullptr, *in_b = nullptr, *o_p1 = nullptr, *o_p2 = nullptr, *o_q1 = nullptr, *o_q2 = nullptr;
  int* info_dev = nullptr;
  double piv_tol = 0.0, piv_reg = 0.0;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  hipStream_t stream2 = nullptr;  // the second elimination chain
  unsigned long long* chain_pub = nullptr;  // k_trsv_chain, as in fpsq_dense_s
  unsigned long long* chain_err = nullptr;
  unsigned int chain_seq = 0;
  bool chain = true, chain_break = false;
  hipEvent_t evA = nullptr, evB = nullptr;
  // jac_coord! hand-over (fpsq_band_create_coo): the caller's COO entries sorted into the CSR slots
  int64_t coo_nnz = -1;
  int32_t *coo_perm = nullptr, *coo_slotptr = nullptr;
  double *coo_in = nullptr, *csr_in = nullptr;
  fpsq_band_info info{};
  std::vector<void*> allocs;
};

namespace {
thread_local std::string g_band_create_error;

#define BCHK(b, call)                                                          \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      (b)->err = std::string(#call) + ": " + hipGetErrorString(e_);            \
      return FPSQ_ERR_HIP;                                                     \
    }                                                                          \
  } while (0)

template <class T>
int bmalloc(fpsq_band b, T** p, size_t count) {
  void* q = nullptr;
  BCHK(b, hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  b->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// Reverse Cuthill-McKee on the rows of A (two rows adjacent when they share a column: the graph of A A').  Returns the new
// order (position -> caller's row) or an empty vector when the adjacency is too large to walk (sum over the columns of
// length^2 > 4e8).  Start nodes: minimum degree, moved to a pseudo-peripheral node by two breadth-first sweeps.
std::vector<int32_t> rcm_rows(int64_t m, int64_t n, const std::vector<int32_t>& rp, const std::vector<int32_t>& ci) {
  std::vector<int32_t> cp(n + 1, 0);
  for (int64_t i = 0; i < m; ++i)
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) cp[ci[k] + 1]++;
  double work = 0.0;
  for (int64_t c = 0; c < n; ++c) {
    work += (double)cp[c + 1] * cp[c + 1];
    cp[c + 1] += cp[c];
  }
  if (work > 4e8) return {};
  std::vector<int32_t> cr(std::max<int64_t>(rp[m], 1)), nxt(cp.begin(), cp.end() - 1);
  for (int64_t i = 0; i < m; ++i)
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) cr[nxt[ci[k]]++] = (int32_t)i;
  std::vector<int64_t> deg(m, 0);
  for (int64_t i = 0; i < m; ++i)
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) deg[i] += cp[ci[k] + 1] - cp[ci[k]] - 1;
  std::vector<int32_t> order;
  order.reserve(m);
  std::vector<int32_t> mark(m, -1);  // mark[i] = id of the sweep that reached row i
  std::vector<char> placed(m, 0);
  std::vector<int32_t> level, nbr;
  int sweep = 0;
  // breadth-first sweep from `root` over the not yet placed rows; returns the visiting order (neighbours by degree)
  auto bfs = [&](int32_t root, std::vector<int32_t>& out) {
    out.clear();
    ++sweep;
    mark[root] = sweep;
    out.push_back(root);
    for (size_t h = 0; h < out.size(); ++h) {
      const int32_t u = out[h];
      nbr.clear();
      for (int32_t k = rp[u]; k < rp[u + 1]; ++k)
        for (int32_t t = cp[ci[k]]; t < cp[ci[k] + 1]; ++t) {
          const int32_t v = cr[t];
          if (!placed[v] && mark[v] != sweep) {
            mark[v] = sweep;
            nbr.push_back(v);
          }
        }
      std::sort(nbr.begin(), nbr.end(), [&](int32_t a, int32_t b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
      out.insert(out.end(), nbr.begin(), nbr.end());
    }
  };
  std::vector<int32_t> byd(m);
  for (int64_t i = 0; i < m; ++i) byd[i] = (int32_t)i;
  std::sort(byd.begin(), byd.end(), [&](int32_t a, int32_t b) { return deg[a] != deg[b] ? deg[a] < deg[b] : a < b; });
  size_t cursor = 0;
  while ((int64_t)order.size() < m) {
    while (placed[byd[cursor]]) ++cursor;
    int32_t root = byd[cursor];
    for (int pass = 0; pass < 2; ++pass) {  // towards a pseudo-peripheral node: restart from the last node reached
      bfs(root, level);
      root = level.back();
    }
    bfs(root, level);
    for (int32_t v : level) placed[v] = 1;
    order.insert(order.end(), level.begin(), level.end());
  }
  std::reverse(order.begin(), order.end());
  return order;
}

// The ordering part of the symbolic phase, host only (also behind fpsq_band_analyze, which needs no device): validates
// the pattern, reorders the rows when that pays (rp / ci are replaced by the reordered structure; rperm_h / vperm_h map
// stored rows / entries to the caller's, empty = identity) and decides on the two elimination chains.  Returns an error
// text, empty on success.
std::string band_order(int64_t n, int64_t m, std::vector<int32_t>& rp, std::vector<int32_t>& ci, std::vector<int32_t>& rperm_h,
                       std::vector<int32_t>& vperm_h, int& chain_safe, int& chain_bw) {
  const int64_t nnz = rp[m];
  chain_safe = chain_bw = 0;
  // validate, then the natural half bandwidth (rows): if the band is wide, try a reverse Cuthill-McKee ordering of the rows
  // (LDLFactorizations' ldl_analyze computes a fill-reducing ordering at this point; for a band factorisation the
  // ordering to look for is the bandwidth-reducing one).  FPSQ_BAND_REORDER = 0 never, 1 always tries.
  for (int64_t i = 0; i < m; ++i) {
    if (rp[i + 1] < rp[i] || rp[i + 1] > nnz) {
      return "fpsq_band_create: rowptr not monotone";
    }
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k)
      if (ci[k] < 0 || ci[k] >= n) {
        return "fpsq_band_create: column index out of range";
      }
  }
  {
    auto bandwidth_rows = [&](const std::vector<int32_t>& pos) {  // pos[row] = position; empty = identity
      std::vector<int32_t> lo(n, INT32_MAX), hi(n, -1);
      for (int64_t i = 0; i < m; ++i) {
        const int32_t p = pos.empty() ? (int32_t)i : pos[i];
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
          lo[ci[k]] = std::min(lo[ci[k]], p);
          hi[ci[k]] = std::max(hi[ci[k]], p);
        }
      }
      int64_t w = 0;
      for (int64_t c = 0; c < n; ++c)
        if (hi[c] >= 0) w = std::max<int64_t>(w, hi[c] - lo[c]);
      return w;
    };
    // row `ord[p]` of the current structure becomes row p; the maps to the caller's numbering are composed
    auto apply_order = [&](const std::vector<int32_t>& ord) {
      std::vector<int32_t> rp2(m + 1, 0), ci2(std::max<int64_t>(nnz, 1)), vp2(std::max<int64_t>(nnz, 1)), rr2(m);
      for (int64_t p = 0; p < m; ++p) {
        const int32_t r = ord[p];
        rr2[p] = rperm_h.empty() ? r : rperm_h[r];
        rp2[p + 1] = rp2[p] + (rp[r + 1] - rp[r]);
        for (int32_t k = rp[r], t = rp2[p]; k < rp[r + 1]; ++k, ++t) {
          ci2[t] = ci[k];
          vp2[t] = vperm_h.empty() ? k : vperm_h[k];
        }
      }
      rp.swap(rp2);
      ci.swap(ci2);
      rperm_h.swap(rr2);
      vperm_h.swap(vp2);
    };
    int mode = -1;  // auto
    if (const char* ev = std::getenv("FPSQ_BAND_REORDER")) mode = std::atoi(ev);
    const int64_t nbk = (m + kDB - 1) / kDB;
    int64_t bw_rows = bandwidth_rows({});
    if (mode != 0 && (mode == 1 || bw_rows / kDB > std::max<int64_t>(nbk / 8, 2))) {
      std::vector<int32_t> ord = rcm_rows(m, n, rp, ci);
      if (!ord.empty()) {
        std::vector<int32_t> pos(m);
        for (int64_t p = 0; p < m; ++p) pos[ord[p]] = (int32_t)p;
        const int64_t bw_new = bandwidth_rows(pos);
        if (bw_new / kDB < bw_rows / kDB) {  // fewer blocks in the band: take it
          apply_order(ord);
          bw_rows = bw_new;
        }
      }
    }
    // TWO ELIMINATION CHAINS.  A banded Cholesky is a chain of m / 128 dependent block steps, each a few latency-bound
    // launches.  Ordering the blocks from BOTH ends towards the middle -- stored block 2 c = block c from the top, stored
    // block 2 c + 1 = the c-th block of 128 rows from the bottom (rows descending) -- keeps the matrix banded (twice as
    // wide) and makes the even and the odd blocks two independent chains until they meet: their steps run side by side
    // on two streams, the chain is half as long.  Only the last 2 (chain_bw + 1) blocks and the rows left in the middle
    // are eliminated one after the other.  FPSQ_BAND_TWOCHAIN=0 turns it off.
    int two = 1;
    if (const char* ev = std::getenv("FPSQ_BAND_TWOCHAIN")) two = std::atoi(ev);
    const int64_t C = m / (2 * kDB);
    const int64_t bwc = (bw_rows + kDB - 1) / kDB;  // block distance two coupled rows of one chain can have
    if (two && bwc >= 1 && C - bwc - 1 >= 4 * (bwc + 1)) {
      std::vector<int32_t> ord(m);
      int64_t p = 0;
      for (int64_t c = 0; c < C; ++c) {
        for (int64_t t = 0; t < kDB; ++t) ord[p++] = (int32_t)(c * kDB + t);
        for (int64_t t = 0; t < kDB; ++t) ord[p++] = (int32_t)(m - 1 - c * kDB - t);
      }
      for (int64_t r = C * kDB; r < m - C * kDB; ++r) ord[p++] = (int32_t)r;
      apply_order(ord);
      chain_safe = (int)(C - bwc - 1);
      chain_bw = (int)bwc;
    }
  }
  return std::string();
}

inline size_t blk_off(const fpsq_band b, int64_t i, int64_t j) {  // block (i, j), i - (band_w - 1) <= j <= i
  return ((size_t)i * b->band_w + (size_t)(j - i + b->band_w - 1)) * kDB * kDB;
}

// q (in b->r2, [mpad][2]) <- M^-1 r2 with the banded factor; result in b->r2
void band_solve(fpsq_band b) {
  hipStream_t s = b->stream;
  const int nb = (int)b->nb, bw = b->band_w - 1;
  if (b->chain) {  // (both elimination chains advance side by side inside the one launch)
    ChainArgs c{b->chain_pub, ++b->chain_seq, 0, nb, b->band_w, b->chain_safe, b->chain_bw, b->chain_err,
                b->chain_pub + (size_t)nb * 512 + 1, 0};
    c.pubseq = b->chain_break ? ~c.seq : c.seq;
    c.ticket_base = (unsigned long long)(b->chain_seq - 1) * nb;
    hipLaunchKernelGGL(k_trsv_chain<true>, dim3(nb), dim3(256), 0, s, b->Mb, kDB, b->invs, b->invsT, b->r2, b->y2, c);
    c.seq = ++b->chain_seq;
    c.pubseq = b->chain_break ? ~c.seq : c.seq;
    c.ticket_base = (unsigned long long)(b->chain_seq - 1) * nb;
    hipLaunchKernelGGL(k_trsv_chain<false>, dim3(nb), dim3(256), 0, s, b->Mb, kDB, b->invs, b->invsT, b->y2, b->r2, c);
    return;
  }
  {
    int k0 = 0;
    const int cs = b->chain_safe, cb = b->chain_bw;
    hipStream_t s2 = b->stream2;
    if (cs > 0) {  // forward: the two chains side by side (each touches the blocks of its own parity only), then the rest
      hipEventRecord(b->evA, s);
      hipStreamWaitEvent(s2, b->evA, 0);
      for (int c = 0; c < cs; ++c) {
        hipLaunchKernelGGL(k_trsv_step3<true>, dim3(cb + 1), dim3(256), 0, s, b->Mb, kDB, b->invs, b->invsT, b->r2, b->y2, 2 * c,
                           b->band_w, 2);
        hipLaunchKernelGGL(k_trsv_step3<true>, dim3(cb + 1), dim3(256), 0, s2, b->Mb, kDB, b->invs, b->invsT, b->r2, b->y2,
                           2 * c + 1, b->band_w, 2);
      }
      hipEventRecord(b->evB, s2);
      hipStreamWaitEvent(s, b->evB, 0);
      k0 = 2 * cs;
    }
    for (int k = k0; k < nb; ++k)
      hipLaunchKernelGGL(k_trsv_step3<true>, dim3(std::min(bw, nb - 1 - k) + 1), dim3(256), 0, s, b->Mb, kDB, b->invs,
                         b->invsT, b->r2, b->y2, k, b->band_w, 1);
    for (int k = nb - 1; k >= k0; --k)
      hipLaunchKernelGGL(k_trsv_step3<false>, dim3(std::min(bw, k) + 1), dim3(256), 0, s, b->Mb, kDB, b->invs, b->invsT,
                         b->y2, b->r2, k, b->band_w, 1);
    if (cs > 0) {
      hipEventRecord(b->evA, s);
      hipStreamWaitEvent(s2, b->evA, 0);
      for (int c = cs - 1; c >= 0; --c) {
        hipLaunchKernelGGL(k_trsv_step3<false>, dim3(std::min(cb, c) + 1), dim3(256), 0, s, b->Mb, kDB, b->invs, b->invsT,
                           b->y2, b->r2, 2 * c, b->band_w, 2);
        hipLaunchKernelGGL(k_trsv_step3<false>, dim3(std::min(cb, c) + 1), dim3(256), 0, s2, b->Mb, kDB, b->invs, b->invsT,
                           b->y2, b->r2, 2 * c + 1, b->band_w, 2);
      }
      hipEventRecord(b->evB, s2);
      hipStreamWaitEvent(s, b->evB, 0);
    }
  }
}

// shared tail of the two solve entry points: right-hand sides of the M-solves are in b->r2
int band_finish(fpsq_band b, const double* a1, double* p1, double* q1, double* p2, double* q2) {
  hipStream_t s = b->stream;
  band_solve(b);
  // P = [a0, a1] - A' Q
  hipLaunchKernelGGL(k_csr_mv2, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, s, b->t_rowptr, b->t_colind, b->t_vals,
                     b->r2, b->atq, (int)b->n);
  hipLaunchKernelGGL(k_band_finish, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, s, b->atq, b->in_a, a1, b->o_p1,
                     b->o_p2, (int)b->n);
  if (b->reordered)  // back to the caller's row order
    hipLaunchKernelGGL(k_unpack2_scatter, dim3((unsigned)((b->m + 255) / 256)), dim3(256), 0, s, b->r2, b->rperm, b->o_q1,
                       b->o_q2, (int)b->m);
  else
    hipLaunchKernelGGL(k_dense_unpack2, dim3((unsigned)((b->m + 255) / 256)), dim3(256), 0, s, b->r2, b->o_q1, b->o_q2,
                       (int)b->m);
  hipEventRecord(b->e1, s);
  BCHK(b, hipMemcpyAsync(p1, b->o_p1, (size_t)b->n * 8, hipMemcpyDefault, s));
  BCHK(b, hipMemcpyAsync(p2, b->o_p2, (size_t)b->n * 8, hipMemcpyDefault, s));
  BCHK(b, hipMemcpyAsync(q1, b->o_q1, (size_t)b->m * 8, hipMemcpyDefault, s));
  BCHK(b, hipMemcpyAsync(q2, b->o_q2, (size_t)b->m * 8, hipMemcpyDefault, s));
  BCHK(b, hipStreamSynchronize(s));
  if (b->chain_err && *b->chain_err) {
    *b->chain_err = 0;
    b->err = "triangular sweep: a block's solution did not arrive (bounded wait expired); FPSQ_TRSV_CHAIN=0 avoids the path";
    return FPSQ_ERR_TIMEOUT;
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, b->e0, b->e1);
  b->info.last_solve_ms = ms;
  return FPSQ_OK;
}
}  // namespace

extern "C" {

const char* fpsq_band_last_error(fpsq_band b) { return b ? b->err.c_str() : g_band_create_error.c_str(); }

int fpsq_band_analyze(int64_t n, int64_t m, const int32_t* rowptr, const int32_t* colind, int32_t* row_perm,
                      fpsq_band_info* info) {
  if (n <= 0 || m <= 0 || !rowptr || n >= INT32_MAX || m >= INT32_MAX - 256 || rowptr[0] != 0) {
    g_band_create_error = "fpsq_band_analyze: bad arguments (0-based CSR in HOST memory expected)";
    return FPSQ_ERR_ARG;
  }
  std::vector<int32_t> rp(rowptr, rowptr + m + 1);
  for (int64_t i = 0; i < m; ++i)
    if (rp[i + 1] < rp[i]) {
      g_band_create_error = "fpsq_band_analyze: rowptr not monotone";
      return FPSQ_ERR_ARG;
    }
  const int64_t nnz = rp[m];
  if (nnz > 0 && !colind) {
    g_band_create_error = "fpsq_band_analyze: colind missing";
    return FPSQ_ERR_ARG;
  }
  std::vector<int32_t> ci(colind, colind + nnz), rperm_h, vperm_h;
  ci.resize(std::max<int64_t>(nnz, 1));
  int chain_safe = 0, chain_bw = 0;
  const std::string msg = band_order(n, m, rp, ci, rperm_h, vperm_h, chain_safe, chain_bw);
  if (!msg.empty()) {
    g_band_create_error = msg;
    return FPSQ_ERR_ARG;
  }
  if (row_perm)
    for (int64_t p = 0; p < m; ++p) row_perm[p] = rperm_h.empty() ? (int32_t)p : rperm_h[p];
  if (info) {
    std::vector<int32_t> lo(n, INT32_MAX), hi(n, -1);
    for (int64_t i = 0; i < m; ++i)
      for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
        lo[ci[k]] = std::min(lo[ci[k]], (int32_t)(i / kDB));
        hi[ci[k]] = std::max(hi[ci[k]], (int32_t)(i / kDB));
      }
    int64_t bwb = 0;
    for (int64_t c = 0; c < n; ++c)
      if (hi[c] >= 0) bwb = std::max<int64_t>(bwb, hi[c] - lo[c]);
    const int64_t nb = (m + kDB - 1) / kDB;
    bwb = std::min(bwb, nb - 1);
    *info = fpsq_band_info{};
    info->n = n;
    info->m = m;
    info->nnz = nnz;
    info->nblocks = nb;
    info->bandwidth_blocks = bwb;
    info->factor_bytes = nb * (bwb + 1) * (int64_t)kDB * kDB * 8;
    info->reordered = rperm_h.empty() ? 0 : 1;
    info->chains = chain_safe > 0 ? 2 : 1;
  }
  return FPSQ_OK;
}

int fpsq_band_destroy(fpsq_band b) {
  if (!b) return FPSQ_ERR_ARG;
  hipSetDevice(b->device);
  if (b->stream) hipStreamSynchronize(b->stream);
  for (void* p : b->allocs) hipFree(p);
  if (b->chain_err) hipHostFree(b->chain_err);
  if (b->e0) hipEventDestroy(b->e0);
  if (b->e1) hipEventDestroy(b->e1);
  if (b->e2) hipEventDestroy(b->e2);
  if (b->evA) hipEventDestroy(b->evA);
  if (b->evB) hipEventDestroy(b->evB);
  if (b->stream2) {
    hipStreamSynchronize(b->stream2);
    hipStreamDestroy(b->stream2);
  }
  if (b->stream) hipStreamDestroy(b->stream);
  delete b;
  return FPSQ_OK;
}

int fpsq_band_create(fpsq_band* out, int64_t n, int64_t m, const int32_t* rowptr, const int32_t* colind, int32_t device) {
  if (!out || n <= 0 || m <= 0 || !rowptr || n >= INT32_MAX || m >= INT32_MAX - 256) {
    g_band_create_error = "fpsq_band_create: bad arguments";
    return FPSQ_ERR_ARG;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    g_band_create_error = std::string("fpsq_band_create: no HIP device (") + hipGetErrorString(e) +
                          "); libfpsq has no CPU fallback";
    return FPSQ_ERR_HIP;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_band_create_error = "fpsq_band_create: cannot select the device";
    return FPSQ_ERR_HIP;
  }
  // ---- symbolic analysis on the host (the role of ldl_analyze, src/solve_two_systems_struct.jl:344): the structure of
  // A A' + delta I is a band whose half width is the largest row distance of two entries of one column of A
  std::vector<int32_t> rp(m + 1);
  if (hipMemcpy(rp.data(), rowptr, (size_t)(m + 1) * 4, hipMemcpyDefault) != hipSuccess || rp[0] != 0) {
    g_band_create_error = "fpsq_band_create: cannot read rowptr (0-based CSR expected)";
    return FPSQ_ERR_ARG;
  }
  const int64_t nnz = rp[m];
  std::vector<int32_t> ci(std::max<int64_t>(nnz, 1));
  if (nnz > 0 && (!colind || hipMemcpy(ci.data(), colind, (size_t)nnz * 4, hipMemcpyDefault) != hipSuccess)) {
    g_band_create_error = "fpsq_band_create: cannot read colind";
    return FPSQ_ERR_ARG;
  }
  std::vector<int32_t> rperm_h, vperm_h;  // stored row / entry -> the caller's (empty: identity)
  int chain_safe = 0, chain_bw = 0;
  {
    const std::string msg = band_order(n, m, rp, ci, rperm_h, vperm_h, chain_safe, chain_bw);
    if (!msg.empty()) {
      g_band_create_error = msg;
      return FPSQ_ERR_ARG;
    }
  }
  std::vector<int32_t> cfirst(n, INT32_MAX), clast(n, -1), tcnt(n + 1, 0);
  std::vector<int2> span(m);
  std::vector<int32_t> seen(n, -1);
  bool has_dup = false;
  int maxspan = 1;
  for (int64_t i = 0; i < m; ++i) {
    if (rp[i + 1] < rp[i]) {
      g_band_create_error = "fpsq_band_create: rowptr not monotone";
      return FPSQ_ERR_ARG;
    }
    int lo = INT32_MAX, hi = -1;
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int32_t c = ci[k];
      if (c < 0 || c >= n) {
        g_band_create_error = "fpsq_band_create: column index out of range";
        return FPSQ_ERR_ARG;
      }
      lo = std::min(lo, c);
      hi = std::max(hi, c);
      has_dup |= seen[c] == (int32_t)i;
      seen[c] = (int32_t)i;
      cfirst[c] = std::min<int32_t>(cfirst[c], (int32_t)i);
      clast[c] = std::max<int32_t>(clast[c], (int32_t)i);
      tcnt[c + 1]++;
    }
    if (hi < 0) lo = hi = 0;
    span[i] = int2{lo, hi};
    maxspan = std::max(maxspan, hi - lo + 1);
  }
  int64_t bwb = 0;
  for (int64_t c = 0; c < n; ++c)
    if (clast[c] >= 0) bwb = std::max<int64_t>(bwb, clast[c] / kDB - cfirst[c] / kDB);
  fpsq_band b = new fpsq_band_s();
  b->n = n;
  b->m = m;
  b->nnz = nnz;
  b->device = device;
  b->mpad = (m + kDB - 1) / kDB * kDB;
  b->nb = b->mpad / kDB;
  b->band_w = (int)std::min<int64_t>(bwb, b->nb - 1) + 1;
  b->span = maxspan;
  b->chain_safe = chain_safe;
  b->chain_bw = std::min(chain_bw, (b->band_w - 1) / 2);
  if (has_dup) {
    g_band_create_error = "fpsq_band_create: the CSR pattern has duplicate entries (sum them first)";
    delete b;
    return FPSQ_ERR_ARG;
  }
  // M is formed by columns of A (k_band_form_t) when its accumulator rows fit in LDS; otherwise by row pairs
  // (k_band_form), which needs the widest row span in LDS twice
  b->form_R = 16;
  while (b->form_R > 1 && b->form_R * b->band_w > 144) b->form_R /= 2;
  b->form_gen = b->band_w > 144 ? 1 : 2;
  if (const char* ev = std::getenv("FPSQ_BAND_FORM")) {
    const int want = std::atoi(ev);
    if (want == 1 || (want == 2 && b->band_w <= 144)) b->form_gen = want;
  }
  const size_t fbytes = (size_t)b->nb * b->band_w * kDB * kDB * 8;
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  if ((b->form_gen == 1 && (size_t)maxspan * 16 > 150 * 1024) || fbytes + 3 * ((size_t)b->nb * kDB * kDB * 8) > free_b / 10 * 9) {
    char msg[256];
    snprintf(msg, sizeof msg, "fpsq_band_create: the banded direct path does not fit this Jacobian (half bandwidth %d "
             "blocks > 143 and a row span of %d columns > 9600, or factor storage %.1f GB of %.1f GB "
             "free): use the iterative back-end", b->band_w - 1, maxspan, fbytes / 1e9, free_b / 1e9);
    g_band_create_error = msg;
    delete b;
    return FPSQ_ERR_STATE;
  }
  // transposed structure (for P = rhs - A' Q) with the value permutation
  for (int64_t c = 0; c < n; ++c) tcnt[c + 1] += tcnt[c];
  std::vector<int32_t> trow(std::max<int64_t>(nnz, 1)), tperm(std::max<int64_t>(nnz, 1)), nxt(tcnt.begin(), tcnt.end() - 1);
  for (int64_t i = 0; i < m; ++i)
    for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int32_t t = nxt[ci[k]]++;
      trow[t] = (int32_t)i;
      tperm[t] = k;
    }
  if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
    g_band_create_error = "fpsq_band_create: cannot create a stream";
    delete b;
    return FPSQ_ERR_HIP;
  }
  hipEventCreate(&b->e0);
  hipEventCreate(&b->e1);
  hipEventCreate(&b->e2);
  hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking);
  hipEventCreateWithFlags(&b->evA, hipEventDisableTiming);
  hipEventCreateWithFlags(&b->evB, hipEventDisableTiming);
  int rc = 0;
  const size_t nz = (size_t)std::max<int64_t>(nnz, 1);
  rc |= bmalloc(b, &b->rowptr, (size_t)m + 1) | bmalloc(b, &b->colind, nz) | bmalloc(b, &b->vals, nz);
  rc |= bmalloc(b, &b->t_rowptr, (size_t)n + 1) | bmalloc(b, &b->t_colind, nz) | bmalloc(b, &b->t_vals, nz);
  rc |= bmalloc(b, &b->t_perm, nz) | bmalloc(b, &b->rowspan, (size_t)m);
  rc |= bmalloc(b, &b->Mb, (size_t)b->nb * b->band_w * kDB * kDB);
  rc |= bmalloc(b, &b->invs, (size_t)b->nb * kDB * kDB) | bmalloc(b, &b->invsT, (size_t)b->nb * kDB * kDB);
  rc |= bmalloc(b, &b->xn, (size_t)n * 2) | bmalloc(b, &b->atq, (size_t)n * 2);
  rc |= bmalloc(b, &b->ym, (size_t)b->mpad * 2) | bmalloc(b, &b->r2, (size_t)b->mpad * 2) | bmalloc(b, &b->y2, (size_t)b->mpad * 2);
  rc |= bmalloc(b, &b->in_a, (size_t)n) | bmalloc(b, &b->in_b, (size_t)std::max(n, b->mpad));
  rc |= bmalloc(b, &b->o_p1, (size_t)n) | bmalloc(b, &b->o_p2, (size_t)n);
  rc |= bmalloc(b, &b->o_q1, (size_t)b->mpad) | bmalloc(b, &b->o_q2, (size_t)b->mpad) | bmalloc(b, &b->info_dev, 4);
  rc |= bmalloc(b, &b->chain_pub, (size_t)b->nb * 512 + 8);  // (+ the abort word)
  if (!rc) hipMemset(b->chain_pub, 0, ((size_t)b->nb * 512 + 8) * 8);
  if (hipHostMalloc((void**)&b->chain_err, 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) rc = 1;
  else *b->chain_err = 0;
  if (const char* e = getenv("FPSQ_TRSV_CHAIN")) b->chain = atoi(e) != 0;
  if (const char* e = getenv("FPSQ_DEBUG_CHAIN_BREAK")) b->chain_break = atoi(e) != 0;
  b->reordered = !rperm_h.empty();
  b->rperm_host = rperm_h;
  if (b->reordered)
    rc |= bmalloc(b, &b->rperm, (size_t)m) | bmalloc(b, &b->vperm, nz) | bmalloc(b, &b->vals_in, nz) |
          bmalloc(b, &b->in_bp, (size_t)b->mpad);
  if (rc) {
    g_band_create_error = b->err;
    fpsq_band_destroy(b);
    return FPSQ_ERR_HIP;
  }
  if (b->reordered) {
    hipMemcpy(b->rperm, rperm_h.data(), (size_t)m * 4, hipMemcpyHostToDevice);
    if (nnz > 0) hipMemcpy(b->vperm, vperm_h.data(), (size_t)nnz * 4, hipMemcpyHostToDevice);
  }
  hipMemset(b->invs, 0, (size_t)b->nb * kDB * kDB * 8);  // k_potrf_inv128m writes the non-zero triangles only
  hipMemset(b->invsT, 0, (size_t)b->nb * kDB * kDB * 8);
  hipMemcpy(b->rowptr, rp.data(), (size_t)(m + 1) * 4, hipMemcpyHostToDevice);
  hipMemcpy(b->t_rowptr, tcnt.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice);
  hipMemcpy(b->rowspan, span.data(), (size_t)m * sizeof(int2), hipMemcpyHostToDevice);
  if (nnz > 0) {
    hipMemcpy(b->colind, ci.data(), (size_t)nnz * 4, hipMemcpyHostToDevice);
    hipMemcpy(b->t_colind, trow.data(), (size_t)nnz * 4, hipMemcpyHostToDevice);
    hipMemcpy(b->t_perm, tperm.data(), (size_t)nnz * 4, hipMemcpyHostToDevice);
  }
  hipDeviceSynchronize();
  hipFuncSetAttribute((const void*)k_potrf_inv128m, hipFuncAttributeMaxDynamicSharedMemorySize, kPotrfLds5);
  hipFuncSetAttribute((const void*)k_gemm128_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kG128Lds0);
  hipFuncSetAttribute((const void*)k_gemm128_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kG128Lds1);
  if (b->form_gen == 1)
    hipFuncSetAttribute((const void*)k_band_form, hipFuncAttributeMaxDynamicSharedMemorySize, maxspan * 16);
  else
    hipFuncSetAttribute((const void*)k_band_form_t, hipFuncAttributeMaxDynamicSharedMemorySize,
                        b->form_R * b->band_w * kDB * 8);
  b->info.n = n;
  b->info.m = m;
  b->info.nnz = nnz;
  b->info.nblocks = b->nb;
  b->info.bandwidth_blocks = b->band_w - 1;
  b->info.reordered = b->reordered ? 1 : 0;
  b->info.chains = b->chain_safe > 0 ? 2 : 1;
  b->info.factor_bytes = (int64_t)fbytes;
  *out = b;
  return FPSQ_OK;
}

int fpsq_band_create_coo(fpsq_band* out, int64_t n, int64_t m, int64_t nnz, const int64_t* rows, const int64_t* cols,
                         int32_t index_base, int32_t device) {
  if (!out || n <= 0 || m <= 0 || nnz < 0 || nnz >= INT32_MAX || (nnz > 0 && (!rows || !cols))) {
    g_band_create_error = "fpsq_band_create_coo: bad arguments";
    return FPSQ_ERR_ARG;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_band_create_error = "fpsq_band_create_coo: cannot select the device";
    return FPSQ_ERR_HIP;
  }
  std::vector<int64_t> r(nnz), c(nnz);
  if (nnz && (hipMemcpy(r.data(), rows, (size_t)nnz * 8, hipMemcpyDefault) != hipSuccess ||
              hipMemcpy(c.data(), cols, (size_t)nnz * 8, hipMemcpyDefault) != hipSuccess)) {
    g_band_create_error = "fpsq_band_create_coo: cannot read the triplets";
    return FPSQ_ERR_ARG;
  }
  std::vector<int32_t> order, slotptr, srow, scol;
  const std::string msg = coo_sort(m, n, nnz, r.data(), c.data(), index_base, order, slotptr, srow, scol);
  if (!msg.empty()) {
    g_band_create_error = "fpsq_band_create_coo: " + msg;
    return FPSQ_ERR_ARG;
  }
  const int64_t ns = (int64_t)srow.size();
  std::vector<int32_t> rp(m + 1, 0);
  for (int64_t i = 0; i < ns; ++i) rp[srow[i] + 1]++;
  for (int64_t i = 0; i < m; ++i) rp[i + 1] += rp[i];
  if (int rc = fpsq_band_create(out, n, m, rp.data(), scol.data(), device)) return rc;
  fpsq_band b = *out;
  const bool dup = ns != nnz;
  if (bmalloc(b, &b->coo_perm, (size_t)std::max<int64_t>(nnz, 1)) || bmalloc(b, &b->coo_in, (size_t)std::max<int64_t>(nnz, 1)) ||
      bmalloc(b, &b->csr_in, (size_t)std::max<int64_t>(ns, 1)) || (dup && bmalloc(b, &b->coo_slotptr, slotptr.size()))) {
    g_band_create_error = b->err;
    fpsq_band_destroy(b);
    *out = nullptr;
    return FPSQ_ERR_HIP;
  }
  if (nnz) hipMemcpy(b->coo_perm, order.data(), (size_t)nnz * 4, hipMemcpyHostToDevice);
  if (dup) hipMemcpy(b->coo_slotptr, slotptr.data(), slotptr.size() * 4, hipMemcpyHostToDevice);
  hipDeviceSynchronize();
  b->coo_nnz = nnz;
  return FPSQ_OK;
}

int fpsq_band_factorize_coo(fpsq_band b, const double* vals, double delta, int32_t* info) {
  if (!b || b->coo_nnz < 0 || (!vals && b->coo_nnz > 0)) {
    if (b) b->err = "band_factorize_coo: the handle was not created with fpsq_band_create_coo, or null values";
    return FPSQ_ERR_ARG;
  }
  hipSetDevice(b->device);
  if (b->coo_nnz > 0) {
    BCHK(b, hipMemcpyAsync(b->coo_in, vals, (size_t)b->coo_nnz * 8, hipMemcpyDefault, b->stream));
    hipLaunchKernelGGL(k_coo_to_slots, dim3((unsigned)std::min<int64_t>((b->nnz + 255) / 256, 4096)), dim3(256), 0, b->stream,
                       b->coo_in, b->coo_perm, b->coo_slotptr, (const int64_t*)nullptr, b->csr_in, b->nnz);
  }
  return fpsq_band_factorize(b, b->csr_in, delta, info);  // (same stream: the slots are complete when it reads them)
}

int fpsq_band_set_regularization(fpsq_band b, double tol, double reg) {
  if (!b || !(tol >= 0.0)) return FPSQ_ERR_ARG;
  b->piv_tol = tol;
  b->piv_reg = reg;
  return FPSQ_OK;
}

int fpsq_band_factorize(fpsq_band b, const double* vals, double delta, int32_t* info) {
  if (!b || (!vals && b->nnz > 0) || !(delta >= 0.0)) return FPSQ_ERR_ARG;
  hipSetDevice(b->device);
  hipStream_t s = b->stream;
  const int nb = (int)b->nb, W = b->band_w, bw = W - 1;
  b->factored = false;
  if (b->nnz > 0) {
    if (b->reordered) {
      BCHK(b, hipMemcpyAsync(b->vals_in, vals, (size_t)b->nnz * 8, hipMemcpyDefault, s));
      hipLaunchKernelGGL(k_gather_d, dim3((unsigned)std::min<int64_t>((b->nnz + 255) / 256, 4096)), dim3(256), 0, s, b->vals_in,
                         b->vperm, b->vals, b->nnz);
    } else {
      BCHK(b, hipMemcpyAsync(b->vals, vals, (size_t)b->nnz * 8, hipMemcpyDefault, s));
    }
    hipLaunchKernelGGL(k_gather_d, dim3((unsigned)std::min<int64_t>((b->nnz + 255) / 256, 4096)), dim3(256), 0, s, b->vals,
                       b->t_perm, b->t_vals, b->nnz);
  }
  BCHK(b, hipMemsetAsync(b->info_dev, 0, 8, s));
  BCHK(b, hipMemsetAsync(b->Mb, 0, (size_t)nb * W * kDB * kDB * 8, s));
  hipEventRecord(b->e0, s);
  // numeric phase 1: M = A A' + delta I into the band (jac_coord! + sparse(...) of src/solve_linear_system.jl:223-233)
  if (b->form_gen == 1)
    hipLaunchKernelGGL(k_band_form, dim3(nb), dim3(256), (size_t)b->span * 16, s, b->rowptr, b->colind, b->vals, b->rowspan,
                       (int)b->m, (int)b->mpad, W, delta, b->Mb, b->span);
  else
    hipLaunchKernelGGL(k_band_form_t, dim3(nb), dim3(256), (size_t)b->form_R * W * kDB * 8, s, b->rowptr, b->colind, b->vals,
                       b->t_rowptr, b->t_colind, b->t_vals, (int)b->m, (int)b->mpad, W, delta, b->Mb, b->form_R);
  hipEventRecord(b->e1, s);
  // numeric phase 2: right-looking block-banded Cholesky (ldl_factorize!, :234), the dense back-end's block kernels.
  // One step: diagonal block k, panel blocks (k + st j, k) and trailing blocks (k + st i, k + st j), 1 <= j <= i <= rem
  // (st = 1: the whole band below k; st = 2: the blocks of k's own chain)
  auto potrf = [&](hipStream_t q, int k) {
    double* Mkk = b->Mb + blk_off(b, k, k);
    double* inv = b->invs + (size_t)k * kDB * kDB;
    hipLaunchKernelGGL(k_potrf_inv128m, dim3(1), dim3(kPotrfThreads5), kPotrfLds5, q, Mkk, kDB, inv, b->invsT + (size_t)k * kDB * kDB,
                       k * kDB, b->info_dev, b->piv_tol, b->piv_reg);
    return inv;
  };
  auto step = [&](hipStream_t q, int k, int st, int rem) {
    double* inv = potrf(q, k);
    if (rem <= 0) return;
    BlockStrides ps, ts;
    ps.on = ts.on = 1;
    ps.a = ps.ci = (size_t)st * bw * kDB * kDB;  // block (k + st (1 + bi), k): st block rows down, st columns of the band left
    ps.b = ps.cj = 0;
    ts.a = ts.b = ts.ci = ps.a;
    ts.cj = (size_t)st * kDB * kDB;
    double* panel = b->Mb + blk_off(b, k + st, k);
    double* trail = b->Mb + blk_off(b, k + st, k + st);
    hipLaunchKernelGGL(k_gemm128_lds<1>, dim3(1, 4 * rem), dim3(1024), kG128Lds1, q, panel, kDB, panel, kDB, inv, kDB, ps);
    hipLaunchKernelGGL(k_gemm128_lds<0>, dim3(2 * rem, 2 * rem), dim3(1024), kG128Lds0, q, trail, kDB, panel, kDB, panel, kDB, ts);
  };
  int k0 = 0;
  if (b->chain_safe > 0) {  // the two chains side by side
    hipStream_t s2 = b->stream2;
    hipEventRecord(b->evA, s);
    hipStreamWaitEvent(s2, b->evA, 0);
    for (int c = 0; c < b->chain_safe; ++c) {
      step(s, 2 * c, 2, b->chain_bw);
      step(s2, 2 * c + 1, 2, b->chain_bw);
    }
    hipEventRecord(b->evB, s2);
    hipStreamWaitEvent(s, b->evB, 0);
    k0 = 2 * b->chain_safe;
  }
  for (int k = k0; k < nb; ++k) step(s, k, 1, std::min(bw, nb - 1 - k));
  hipEventRecord(b->e2, s);
  int32_t hinfo[2] = {0, 0};
  BCHK(b, hipMemcpyAsync(hinfo, b->info_dev, 8, hipMemcpyDeviceToHost, s));
  BCHK(b, hipStreamSynchronize(s));
  float t0 = 0.f, t1 = 0.f;
  hipEventElapsedTime(&t0, b->e0, b->e1);
  hipEventElapsedTime(&t1, b->e1, b->e2);
  b->info.last_form_ms = t0;
  b->info.last_chol_ms = t1;
  b->info.regularized_pivots = hinfo[1];
  if (info)  // (first non-positive pivot, 1-based, in the CALLER's row numbering)
    *info = hinfo[0] > 0 && b->reordered && hinfo[0] <= (int32_t)b->m ? b->rperm_host[hinfo[0] - 1] + 1 : hinfo[0];
  b->factored = hinfo[0] == 0;
  return hinfo[0] == 0 ? FPSQ_OK : 1;  // soft: not positive definite (factorized(str) == false, :242-246)
}

int fpsq_band_solve_two_mixed(fpsq_band b, const double* rhs1, const double* rhs2, double* p1, double* q1, double* p2,
                              double* q2) {
  if (!b || !rhs1 || !rhs2 || !p1 || !q1 || !p2 || !q2) return FPSQ_ERR_ARG;
  if (!b->factored) {
    b->err = "band_solve: no valid factorisation";
    return FPSQ_ERR_STATE;
  }
  hipSetDevice(b->device);
  hipStream_t s = b->stream;
  BCHK(b, hipMemcpyAsync(b->in_a, rhs1, (size_t)b->n * 8, hipMemcpyDefault, s));
  BCHK(b, hipMemcpyAsync(b->in_b, rhs2, (size_t)b->m * 8, hipMemcpyDefault, s));
  hipEventRecord(b->e0, s);
  // r = [A g, -c]:  q1 = M^-1 A g,  q2 = -M^-1 c;  then p1 = g - A'q1, p2 = -A'q2   
  hipLaunchKernelGGL(k_dense_pack2, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, s, b->in_a, 1.0,
                     (const double*)nullptr, 0.0, b->xn, (int)b->n, (int)b->n);
  hipLaunchKernelGGL(k_csr_mv2, dim3((unsigned)((b->m + 255) / 256)), dim3(256), 0, s, b->rowptr, b->colind, b->vals, b->xn,
                     b->ym, (int)b->m);
  const double* cperm = b->in_b;
  if (b->reordered) {
    hipLaunchKernelGGL(k_gather_d, dim3((unsigned)((b->m + 255) / 256)), dim3(256), 0, s, b->in_b, b->rperm, b->in_bp, b->m);
    cperm = b->in_bp;
  }
  hipLaunchKernelGGL(k_band_rhs, dim3((unsigned)((b->mpad + 255) / 256)), dim3(256), 0, s, b->ym, 0, cperm, -1.0, b->r2,
                     (int)b->m, (int)b->mpad, 0);
  return band_finish(b, nullptr, p1, q1, p2, q2);
}

int fpsq_band_solve_two_least_squares(fpsq_band b, const double* rhs1, const double* rhs2, double* p1, double* q1,
                                      double* p2, double* q2) {
  if (!b || !rhs1 || !rhs2 || !p1 || !q1 || !p2 || !q2) return FPSQ_ERR_ARG;
  if (!b->factored) {
    b->err = "band_solve: no valid factorisation";
    return FPSQ_ERR_STATE;
  }
  hipSetDevice(b->device);
  hipStream_t s = b->stream;
  BCHK(b, hipMemcpyAsync(b->in_a, rhs1, (size_t)b->n * 8, hipMemcpyDefault, s));
  BCHK(b, hipMemcpyAsync(b->in_b, rhs2, (size_t)b->n * 8, hipMemcpyDefault, s));
  hipEventRecord(b->e0, s);
  hipLaunchKernelGGL(k_dense_pack2, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, s, b->in_a, 1.0, b->in_b, 1.0, b->xn,
                     (int)b->n, (int)b->n);
  hipLaunchKernelGGL(k_csr_mv2, dim3((unsigned)((b->m + 255) / 256)), dim3(256), 0, s, b->rowptr, b->colind, b->vals, b->xn,
                     b->ym, (int)b->m);
  hipLaunchKernelGGL(k_band_rhs, dim3((unsigned)((b->mpad + 255) / 256)), dim3(256), 0, s, b->ym, 0, (const double*)nullptr,
                     0.0, b->r2, (int)b->m, (int)b->mpad, 1);
  return band_finish(b, b->in_b, p1, q1, p2, q2);
}

int fpsq_band_get_info(fpsq_band b, fpsq_band_info* info) {
  if (!b || !info) return FPSQ_ERR_ARG;
  *info = b->info;
  return FPSQ_OK;
}

int fpsq_dense_get_info(fpsq_dense d, fpsq_dense_info* info) {
  if (!d || !info) return FPSQ_ERR_ARG;
  *info = d->info;
  return FPSQ_OK;
}

}  // extern "C"

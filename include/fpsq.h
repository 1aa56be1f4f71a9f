/*
 * fpsq.h -- C ABI of libfpsq.so: the MI355X (gfx950) back-end for the two regularised KKT saddle-point
 * solves  [I A'; A -delta I]  that FletcherPenaltySolver.jl performs on every obj/grad of FletcherPenaltyNLP.
 *
 * This is the drop-in boundary: what a `struct HIPQDSolver <: QDSolver` on the Julia side binds with `ccall`
 * (see INTEGRATION.md).  Plain C, opaque handle, plain pointers and sizes; no C++/torch types.
 * Reference = JuliaSmoothOptimizers/FletcherPenaltySolver.jl v0.3.0; citations are file:line in that repo.
 *
 * Conventions
 *   n = nvar, m = number of penalised constraints, A = constraint Jacobian (m x n), fp64 throughout.
 *   Every `double*` / index pointer argument may point to HOST or DEVICE memory; the library inspects it
 *   (hipPointerGetAttributes) and stages host data itself.  Output buffers are CALLER-owned (the reference hands
 *   out aliases of solver-owned buffers, src/solve_linear_system.jl:139; callers copy out immediately,
 *   src/model-Fletcherpenaltynlp.jl:244-248, so caller-owned outputs are equivalent and remove the aliasing).
 *   All calls are synchronous: on return the outputs are complete (the solver's stream has been synchronised).
 *   INPUT READINESS: the library works on its own non-blocking HIP stream and reads DEVICE-resident arguments in place.
 *   Device data handed to a call must therefore be complete when the call is made -- either the caller has
 *   synchronised the stream that produced it, or it has registered that stream once with fpsq_set_input_stream(): every
 *   call then first makes the solver's stream wait (event, no host block) for all work queued on the registered
 *   stream so far, which orders both the reads of device inputs and the overwriting of device output buffers that
 *   earlier kernels of that stream may still be reading.  Host-resident arguments need nothing.
 *   Since round 5 a single-GPU handle (or one whose communicator has ONE rank) does more with a registered stream: it ENQUEUES
 *   ON IT, instead of on a stream of its own (FPSQ_ADOPT_STREAM=0 keeps the own stream and the event pair) -- the inputs and the
 *   outputs are then ordered by the stream itself, and the two hops between queues that lay between the last kernel of an
 *   evaluation and the first of the next (the caller's stream waits for the tail, the library's for the caller's) are gone:
 *   ~28 us per evaluation at the headline size.  The registered stream must outlive its registration; registering another
 *   stream (or none: enabled = 0) first drains the one in use.  Handles sharded over several ranks keep their own stream.
 *   One handle is non-re-entrant, exactly like one reference QDSolver (shared mutable workspaces).
 *
 * Return codes
 *   0            success
 *   > 0          soft numerical failure, bit 0: first system `solved == false`, bit 1: second system.  The
 *                reference only `@warn`s in that case and returns what was computed
 *                (src/solve_linear_system.jl:54-56,73-75,92-94,101-103,128-130,136-138); so does this library.
 *   < 0          hard error (bad handle/argument, HIP/RCCL failure); message via fpsq_last_error().
 */
#ifndef FPSQ_H
#define FPSQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fpsq_solver_s *fpsq_handle;

enum {
  FPSQ_OK = 0,
  FPSQ_ERR_ARG = -1,
  FPSQ_ERR_HIP = -2,
  FPSQ_ERR_STATE = -3, /* e.g. solve before the Jacobian was set */
  FPSQ_ERR_COMM = -4,
  FPSQ_ERR_TIMEOUT = -5
};

/* Krylov termination status (Krylov.jl `stats.status` strings, as an enum) */
enum {
  FPSQ_ST_UNKNOWN = 0,
  FPSQ_ST_ZERO_RHS = 1,     /* "x = 0 is a zero-residual solution" */
  FPSQ_ST_ZERO_ATB = 2,     /* "x = 0 is a minimum least-squares solution" */
  FPSQ_ST_SOLVED = 3,       /* tolerances met */
  FPSQ_ST_ZERO_RESID = 4,   /* "found approximate zero-residual solution" */
  FPSQ_ST_FWD_ERR = 5,      /* "truncated forward error small enough" */
  FPSQ_ST_ILL_COND = 6,     /* condition number limit */
  FPSQ_ST_MAXITER = 7,      /* "maximum number of iterations exceeded" */
  FPSQ_ST_INCONSISTENT = 8, /* "system may be inconsistent" (CRAIG) */
  FPSQ_ST_SOLVED_LQ = 9     /* LNLQ: "solutions xL and yL good enough" (FPSQ_ST_SOLVED: the CRAIG point xC, yC) */
};

/* how the two saddle-point systems of a call are solved (fpsq_options.kkt_method) */
enum {
  FPSQ_KKT_LSQR_CRAIG = 0, /* the reference's iterative path: LSQR on A' + CRAIG / LNLQ on A (solve_linear_system.jl:107-140)
                              for solve_two_mixed, two LSQR for solve_two_least_squares (:79-105).  Default. */
  FPSQ_KKT_MINRES_K = 1    /* MINRES on K = [I A'; A -delta I] itself, order n + m, both systems in lock-step.  NOT a path
                              of the reference: BASELINE.json north_star / configs[1] ("MINRES matrix-free") and SURVEY
                              8(b)'s method enum name it.  Stops on the reference's MINRES tolerances (ne_atol, ne_rtol,
                              ne_etol, ne_conlim; ne_itmax = 0 -> 2 (n + m)); stats are MINRES's.  Serves
                              fpsq_solve_two_mixed, fpsq_solve_two_least_squares and fpsq_ys_gs on a single-GPU handle;
                              the fused fpsq_qp_* entries and sharded handles return FPSQ_ERR_STATE with it. */
};

/* least-norm method of solve_two_mixed / ys_gs / qp_objgrad (fpsq_options.ln_method) */
enum {
  FPSQ_LN_CRAIG = 0, /* craig! with sqd and M = (1/delta) I: the reference's default workspace (struct.jl:121, :210-244) */
  FPSQ_LN_LNLQ = 1   /* lnlq! through the generic solve_least_norm (struct.jl:251-281; the commented alternative of
                        :121): M = (1/delta) I WITHOUT sqd, i.e. the minimum-norm solution of A x = -c for every delta */
};

/* The fields of Krylov.jl's `stats` that the reference reads
 * (src/solve_two_systems_struct.jl:183,242,279; src/solve_linear_system.jl:71). */
typedef struct {
  int32_t solved;
  int32_t inconsistent;
  int32_t niter;
  int32_t status;
  double rnorm;  /* last residual-norm estimate */
  double arnorm; /* last ||A'r|| estimate (LSQR, MINRES), 0 for CRAIG */
} fpsq_stats;

/* Replaces the keyword arguments of `IterativeSolver(nlp, ::T; ...)`, src/solve_two_systems_struct.jl:94-131.
 * Same names, same defaults (fpsq_default_options). */
typedef struct {
  double ls_atol, ls_rtol; /* LSQR, :99-100 */
  int64_t ls_itmax;        /* 5 (m + n), :101-103 */
  double ln_atol, ln_rtol, ln_btol, ln_conlim; /* CRAIG, :104-107 */
  int64_t ln_itmax;                            /* 5 (m + n), :108-110 */
  double ne_atol, ne_rtol, ne_etol;            /* MINRES on A A' + tau I, :111-113 */
  int64_t ne_itmax;                            /* 0 => 2 m, :114 */
  double ne_conlim;                            /* :115 */
  /* Krylov.jl lsqr! keywords the reference leaves at their defaults (sqrt(eps), 1/sqrt(eps)) */
  double ls_axtol, ls_btol, ls_etol, ls_conlim;
  /* execution knobs (no reference counterpart) */
  int32_t fuse_two_rhs;  /* 1: run the two recurrences of a solve_two_* call in lock-step, one SpMM(k=2) per
                            product instead of two SpMVs; results per system are identical to the unfused run */
  int32_t lookahead;     /* Krylov iterations the host may enqueue ahead of the device's progress word */
  int32_t device;        /* HIP device ordinal */
  int32_t jac_format;    /* storage of A for the A product: 0 = auto (column-sorted row groups when every group spans
                            < 2^21 columns, else CSR), 1 = CSR only */
  int32_t ln_method;     /* FPSQ_LN_CRAIG (default) or FPSQ_LN_LNLQ: which Krylov.jl workspace `solver_struct_least_norm`
                            would be (src/solve_two_systems_struct.jl:121) */
  int32_t kkt_method;    /* FPSQ_KKT_LSQR_CRAIG (default, the reference's path) or FPSQ_KKT_MINRES_K */
} fpsq_options;

/* defaults of src/solve_two_systems_struct.jl:99-115 for an (n, m) problem; fuse_two_rhs = 1 */
void fpsq_default_options(int64_t n, int64_t m, fpsq_options *opts);

/* Replaces the constructor contract `QDS(nlp, ::T; kwargs...)` (src/solve_two_systems_struct.jl:94-98;
 * call site src/parameters.jl:299).  `opts == NULL` => defaults. */
int fpsq_create(fpsq_handle *out, int64_t n, int64_t m, const fpsq_options *opts);
int fpsq_destroy(fpsq_handle h);
const char *fpsq_last_error(fpsq_handle h); /* h may be NULL: error of the last failed fpsq_create */

/* Jacobian sparsity, once per model.  Replaces `jac_structure!(nlp, rows, cols)` at
 * src/solve_two_systems_struct.jl:331-337: COO triplets in the model's fixed order, `index_base` 1 for
 * Julia.  Duplicates are allowed and are summed, as in SparseArrays.sparse (src/solve_linear_system.jl:233).
 * Builds CSR(A), CSR(A') and the value permutations on the device. */
int fpsq_set_jacobian_structure_coo(fpsq_handle h, int64_t nnz, const int64_t *rows, const int64_t *cols,
                                    int32_t index_base);
/* Same for a caller that already holds 0-based CSR (rowptr: m+1, colind: rowptr[m]); values then come in
 * CSR order. */
int fpsq_set_jacobian_structure_csr(fpsq_handle h, const int32_t *rowptr, const int32_t *colind);

/* Jacobian values at the current x, in the order of the structure call.  Replaces `jac_coord!(nlp, x, vals)`
 * at src/solve_linear_system.jl:223-228 and the per-x operator rebuild `jac_op!` at :118-122. */
int fpsq_set_jacobian_values(fpsq_handle h, const double *vals);

/* Registers (enabled != 0) or clears the caller's producer stream for device-resident arguments, see "INPUT
 * READINESS" above.  `hip_stream` is a hipStream_t (NULL = the legacy default stream). */
int fpsq_set_input_stream(fpsq_handle h, int32_t enabled, void *hip_stream);

/* STREAM-ORDERED OUTPUTS (optional; needs a stream registered with fpsq_set_input_stream).  By default every call returns
 * with all outputs complete.  With stream_ordered != 0, fpsq_qp_objgrad on a single-GPU handle whose vector arguments
 * (x, gx, ys, gs) all reside on the handle's GPU returns as soon as its HOST-visible results -- the return code, *fx and
 * the statistics -- are final, while the last kernels that write the DEVICE-resident outputs may still be running: the
 * registered stream has been made to wait for them (event, no host block), so work queued there afterwards -- the caller's
 * next kernels reading gx, its updates of x -- is ordered behind the evaluation, exactly as a stream-ordered library call.
 * A caller that touches the outputs from the host or from another stream must synchronise the registered stream first.
 * The host turn-around between dependent evaluations (return, the caller's decision, the next call's set-up) then overlaps
 * the GPU's tail instead of leaving it idle.  Every other case (host-resident arguments, sharded handles, profiling) keeps
 * the synchronous behaviour. */
int fpsq_set_output_ordering(fpsq_handle h, int32_t stream_ordered);

/* `nlp.delta`, mutated by the outer loop (src/algo.jl:389-393). */
int fpsq_set_delta(fpsq_handle h, double delta);

/* solve_two_mixed (src/solve_linear_system.jl:28-43, iterative method :107-140):
 *   K [p1; q1] = [rhs1; 0],  K [p2; q2] = [0; rhs2];   rhs1: n, rhs2: m;  p*: n, q*: m.
 * st[0] = LSQR stats, st[1] = CRAIG stats. */
int fpsq_solve_two_mixed(fpsq_handle h, const double *rhs1, const double *rhs2, double *p1, double *q1,
                         double *p2, double *q2, fpsq_stats st[2]);

/* solve_two_least_squares (src/solve_linear_system.jl:12-25, :79-105):
 *   K [p1; q1] = [rhs1; 0],  K [p2; q2] = [rhs2; 0];   rhs1, rhs2: n.  Uses the Jacobian of the last
 *   fpsq_set_jacobian_values (the reference does not refresh `nlp.Aop` either, :85-86). */
int fpsq_solve_two_least_squares(fpsq_handle h, const double *rhs1, const double *rhs2, double *p1, double *q1,
                                 double *p2, double *q2, fpsq_stats st[2]);

/* solve_two_extras (src/solve_linear_system.jl:2-9, :45-77), tau = max(delta, 1e-14):
 *   out1 = argmin ||A'q - rhs1||^2 + tau ||q||^2  (LSQR, lambda = sqrt(tau)),  out2 = (A A' + tau I)^-1 rhs2 (MINRES).
 *   rhs1: n, rhs2: m; out1, out2: m. */
int fpsq_solve_two_extras(fpsq_handle h, const double *rhs1, const double *rhs2, double *out1, double *out2,
                          fpsq_stats st[2]);

/* Fused convenience for `_compute_ys_gs!` after the user-model evaluations
 * (src/model-Fletcherpenaltynlp.jl:242-248): solve_two_mixed(g, c) then
 *   gs = p1 + sigma p2, ys = q1 + sigma q2, v = p2, w = q2  in one epilogue kernel. */
int fpsq_ys_gs(fpsq_handle h, const double *g, const double *c, double sigma, double *gs, double *ys, double *v,
               double *w, fpsq_stats st[2]);

/* y = alpha * op(A) x + beta * y with the handle's Jacobian; trans = 0: A (x: n, y: m), 1: A' (x: m, y: n).
 * The device `jprod!` / `jtprod!` (e.g. the rho A'c term, src/model-Fletcherpenaltynlp.jl:388-395). */
int fpsq_jac_mul(fpsq_handle h, int32_t trans, double alpha, const double *x, double beta, double *y);

/* ---- device-resident equality-QP user model (the synthetic workloads of BASELINE.json):
 *        f(x) = 1/2 x' diag(q) x + d'x,   c(x) = A x - b,   A = the handle's Jacobian.
 * fpsq_qp_objgrad is one `objgrad!(::FletcherPenaltyNLP, x, gx)` at a fresh x
 * (src/model-Fletcherpenaltynlp.jl:403-437) run entirely on the device: g, c, the two solves, the ys/gs
 * epilogue, Hsv = q .* v (constraints are linear, so S(x,w) gs = 0), + rho A'c, + eta (x - xk).
 * Any of gx, ys, gs may be NULL.  xk may be NULL when eta == 0. */
typedef struct fpsq_qp_s *fpsq_qp;
int fpsq_qp_create(fpsq_handle h, const double *qdiag, const double *d, const double *b, fpsq_qp *out);
int fpsq_qp_destroy(fpsq_qp qp);
int fpsq_qp_objgrad(fpsq_handle h, fpsq_qp qp, const double *x, double sigma, double rho, double eta,
                    const double *xk, double *fx, double *gx, double *ys, double *gs, fpsq_stats st[2]);
/* One `hprod!(::FletcherPenaltyNLP, x, v, Hv)` on the same model, entirely on the device.
 * hessian_approx = 2, Val(2) (src/model-Fletcherpenaltynlp.jl:521-570):
 *   Hsv = q .* v;  (p1, _, p2, _) = solve_two_least_squares(v, Hsv);  Ptv = v - p1;
 *   Hv = p2 - q .* Ptv + 2 sigma Ptv (+ rho A'(A v)) (+ eta v)      (the constraint Hessians vanish: c is linear).
 *   st[0], st[1] = the two LSQR recurrences.
 * hessian_approx = 1, Val(1) (:572-634): the same, then Ssv = ghjvprod(x, gs, v) (= 0 for linear constraints),
 *   (invJtJJv, invJtJSsv) = solve_two_extras(v, Ssv) and Hv -= A' invJtJSsv (- hprod_nln(x, invJtJJv, gs; obj_weight = 0) = 0):
 *   the LSQR + MINRES lanes of fpsq_solve_two_extras and one more A' product; st must then hold FOUR entries, st[2], st[3] =
 *   the statistics of those two recurrences, and bits 2, 3 of a positive return code flag their `solved == false`.
 * The Hessian-free sub-solvers call this once per inner CG iteration. */
int fpsq_qp_hprod(fpsq_handle h, fpsq_qp qp, const double *v, double sigma, double rho, double eta, int32_t hessian_approx,
                  double *Hv, fpsq_stats *st);

/* ---- multi-GPU: 1-D row sharding of A across ranks.  Each rank creates its handle with
 * the GLOBAL n and its LOCAL m (its block of constraints), passes its local rows to set_jacobian_*, and
 * m-vectors (rhs2, q1, q2, ys, w, c, b) are the rank's slices.  n-vectors are replicated.  Partial A'u
 * products and m-vector dot products are all-reduced with RCCL on the solver's stream.
 * fpsq_comm_unique_id fills a 128-byte id on rank 0; the caller broadcasts it and every rank calls
 * fpsq_comm_init.  Without fpsq_comm_init the handle is single-GPU. */
int fpsq_comm_unique_id(uint8_t id[128]);
int fpsq_comm_init(fpsq_handle h, int32_t nranks, int32_t rank, const uint8_t id[128]);
/* How the exchanges of the halo-sharded Krylov loop travel between the ranks of a node (fpsq_info.comm_route reports what
 * a handle ended up with):
 *   FPSQ_ROUTE_P2P   peer to peer, no collective call inside the loop: every rank exports its gather buffers, halo slots
 *                    and flag words (hipIpcGetMemHandle), maps its peers' (hipIpcOpenMemHandle, peer access over xGMI) and
 *                    from then on WRITES its records -- the norm partials of a product, the raw A'u sums of its two overlap
 *                    regions, the four sums of phi -- straight into the peers' buffers, announces them with sequence
 *                    numbers and waits (a bounded number of polls: FPSQ_ERR_TIMEOUT, never a hang) for theirs, one
 *                    one-workgroup kernel per exchange.  Per joint Krylov iteration: 2 product launches + 3 exchange kernels --
 *                    or, when every rank has a device of its own (round 5), NO exchange kernel for the sums: the leader
 *                    workgroups of the product launches write their rank's local sums into the peers' mapped receive
 *                    areas and add the ranks' rows up in rank order themselves (fpsq_info.comm_in_launch_sums = 1:
 *                    2 product launches + the halo launch; one launch per iteration when no row is shared with a neighbour).
 *   FPSQ_ROUTE_RCCL  ncclAllGather / grouped ncclSend + ncclRecv on the solver's stream (3 RCCL operations per iteration:
 *                    latency-bound at the headline size); also what the replicated (non-halo) layout always uses.
 * fpsq_comm_set_route(h, route), after fpsq_comm_init and BEFORE fpsq_comm_set_halo: FPSQ_ROUTE_AUTO (default) = P2P when
 * every rank can export and map, else RCCL -- decided unanimously at the first solve (the handles travel through one RCCL
 * all-gather); FPSQ_ROUTE_RCCL = never try; FPSQ_ROUTE_P2P = fail the first solve with FPSQ_ERR_COMM when it cannot be
 * set up.  The reference has nothing to mirror here (no parallelism at all). */
enum { FPSQ_ROUTE_AUTO = 0, FPSQ_ROUTE_RCCL = 1, FPSQ_ROUTE_P2P = 2, FPSQ_ROUTE_LOCAL = 3, FPSQ_ROUTE_LOCAL_P2P = 4 };
int fpsq_comm_set_route(fpsq_handle h, int32_t route);
/* HALO MODE (banded Jacobians; SURVEY.md 8e "contract path").  When the rows of a rank touch only a column window
 * [w_lo(r), w_hi(r)) of the n columns, windows tile [0, n) and only overlap between neighbouring ranks, the n-vectors
 * need not be replicated: the rank's handle is created with n = its WINDOW length (column indices relative to
 * w_lo(r)), every n-vector argument is the rank's window of the global vector (identical on overlaps), and per
 * Krylov iteration the ranks exchange the partial A'u products of the overlap regions with their neighbours
 * (overlap_left = w_hi(r-1) - w_lo(r) entries at the head of the window, overlap_right = w_hi(r) - w_lo(r+1) at its
 * tail; <= 2 x 8192 x 2 doubles at the headline size) instead of all-reducing an n x 2 vector, plus one 4-double
 * all-reduce per reduction (sums over n-vectors run over the owned prefix [0, n - overlap_right)).  Call after
 * fpsq_comm_init / fpsq_comm_init_local; overlaps must be 0 at the outer ends (rank 0 left, last rank right). */
int fpsq_comm_set_halo(fpsq_handle h, int64_t overlap_left, int64_t overlap_right);
/* In-process stand-in for RCCL: `nshards` (<= 8) row-shard handles living in ONE process on ONE GPU, each driven
 * by its own host thread; the all-reduce is a summation kernel.  Lets the sharded numerics be parity-tested on a
 * one-GPU box.  Create the group, attach every shard handle, run the same call on all shards concurrently. */
int fpsq_local_group_create(int32_t nshards, void **group);
int fpsq_local_group_destroy(void *group);
int fpsq_comm_init_local(fpsq_handle h, void *group, int32_t shard);
/* on != 0 (before the shards attach): the shards of the group exchange their halo records and norm partials PEER TO PEER --
 * every rank writes its record straight into its peers' buffers, announces it with a sequence number and waits (bounded)
 * for theirs, one small kernel per exchange and no collective call inside the Krylov loop.  This is the protocol of the
 * xGMI route between the GPUs of a node (peers' buffers mapped with hipIpcOpenMemHandle instead of living on the same
 * device); it is exercised here between logical shards of one GPU (<= 3: every shard needs a hardware queue of its own while
 * it waits), not measured on a multi-GPU node.  Halo mode only. */
int fpsq_local_group_set_p2p(void *group, int32_t on);


/* ---- dense-block Jacobian variant (BASELINE configs[2]; the DIRECT back-end of the seam for small / dense problems).
 * Replaces the reference's factorisation path: LDLtSolver (src/solve_two_systems_struct.jl:308-353),
 * solve_two_mixed / solve_two_least_squares with `ldl_factorize!` + `ldiv!` (src/solve_linear_system.jl:161-252) and
 * the dense A A' + tau I contraction of src/model-Fletcherpenaltynlp.jl:478-484.  Here: M = A A' + delta I on the fp64
 * matrix cores (v_mfma_f64_16x16x4_f64), blocked Cholesky M = L L', two right-hand sides per solve:
 *   q1 = M^-1 A rhs1, p1 = rhs1 - A'q1;   mixed: q2 = -M^-1 rhs2, p2 = -A'q2;   least squares: q2 = M^-1 A rhs2, p2 = rhs2 - A'q2.
 * fpsq_dense_factorize returns 1 (soft) with *info = first non-positive pivot row (1-based) when M is not positive
 * definite -- the counterpart of `factorized(str) == false` (src/solve_linear_system.jl:242-246). */
typedef struct fpsq_dense_s *fpsq_dense;
typedef struct {
  int64_t n, m;
  double last_syrk_ms; /* device time of M = A A' + delta I */
  double last_chol_ms; /* device time of the blocked Cholesky */
  double last_solve_ms;
  int64_t regularized_pivots; /* pivots replaced by the dynamic regularisation in the last factorisation */
} fpsq_dense_info;
int fpsq_dense_create(fpsq_dense *out, int64_t n, int64_t m, int32_t device);
int fpsq_dense_destroy(fpsq_dense d);
const char *fpsq_dense_last_error(fpsq_dense d);
int fpsq_dense_set_jacobian(fpsq_dense d, const double *a_rowmajor); /* m x n, the Jacobian at the current x */
/* The reference's hand-over instead of a dense array: `jac_structure!` once (src/solve_two_systems_struct.jl:331-337: COO
 * triplets in the model's order, index_base 1 for Julia, duplicates allowed), then per x the output of `jac_coord!`
 * (src/solve_linear_system.jl:223-228) -- nnz values in that order, HOST or DEVICE memory; one scatter kernel puts them into
 * the dense storage, duplicates summed in a fixed order like `sparse(rows, cols, vals)` (:233). */
int fpsq_dense_set_structure_coo(fpsq_dense d, int64_t nnz, const int64_t *rows, const int64_t *cols, int32_t index_base);
int fpsq_dense_set_jacobian_coo(fpsq_dense d, const double *vals);
int fpsq_dense_factorize(fpsq_dense d, double delta, int32_t *info);
/* Dynamic regularisation of LDLFactorizations.jl as the reference's LDLtSolver configures it
 * (src/solve_two_systems_struct.jl:345-348: tol = r1 = sqrt(eps), r2 = -sqrt(eps)).  A pivot d of M = A A' + delta I with
 * d <= tol -- minus d is the pivot of the (2,2) block of K = [I A'; A -delta I] once the identity block is eliminated
 * (its pivots are 1: r1 never fires) -- is replaced by `reg` = -r2 and counted (fpsq_dense_info.regularized_pivots); the
 * factorisation then succeeds on rank-deficient Jacobians (test/rank-deficient.jl) instead of reporting a non-positive
 * pivot.  reg <= 0 switches it off (the default).  `tol` is absolute, like the scaled threshold the reference uses.
 * reg = FPSQ_REG_DROP drops the pivot instead: the row counts as linearly dependent on the earlier ones, its multiplier comes
 * out as (numerically) zero and the other rows solve the consistent part of the normal equations -- multiplier estimates
 * stay bounded on rank-deficient Jacobians (test/test-2.jl:264-287 FLT), where a pivot of sqrt(eps) makes them ~ 1/sqrt(eps).
 * The host bindings default to the reference's ldlt_r2 = -sqrt(eps) (reg = sqrt(eps)); the drop rule is their explicit option. */
#define FPSQ_REG_DROP 1e200
int fpsq_dense_set_regularization(fpsq_dense d, double tol, double reg);
int fpsq_dense_solve_two_mixed(fpsq_dense d, const double *rhs1, const double *rhs2, double *p1, double *q1, double *p2,
                               double *q2);
int fpsq_dense_solve_two_least_squares(fpsq_dense d, const double *rhs1, const double *rhs2, double *p1, double *q1,
                                       double *p2, double *q2);
/* m x m row-major copy of the factor storage: lower triangle = L with M = L L' (upper triangle unspecified) */
int fpsq_dense_get_factor(fpsq_dense d, double *l_out);
int fpsq_dense_get_info(fpsq_dense d, fpsq_dense_info *info);

/* ---- sparse direct back-end: the reference's LDLtSolver path for SPARSE Jacobians whose
 * normal-equations matrix is banded (PDE-like Jacobians: row i only touches a column window that moves with i).
 *   fpsq_band_create      = the constructor's symbolic work (src/solve_two_systems_struct.jl:326-344: the COO pattern of
 *                           triu(K) and `ldl_analyze`): here the block band structure of M = A A' + delta I -- the Schur
 *                           complement of the identity block of K = [I A'; A -delta I] -- from the CSR pattern of A
 *                           (half bandwidth = the largest row distance of two entries of one column).  When the
 *                           natural band is wide (> 1/8 of the matrix) the rows are first reordered by reverse
 *                           Cuthill-McKee on the graph of A A' -- the bandwidth-reducing counterpart of the fill-reducing
 *                           ordering `ldl_analyze` computes -- and the ordering is kept if it narrows the band.  A long
 *                           narrow band is further ordered from BOTH ends towards the middle (blocks alternately from
 *                           the top and from the bottom), which makes its elimination two independent chains that
 *                           fpsq_band_factorize and the solves run side by side on two streams.  All permutations are
 *                           internal (right-hand sides, solutions and reported pivot rows stay in the caller's order).
 *                           Environment switches for A/B runs: FPSQ_BAND_REORDER, FPSQ_BAND_TWOCHAIN (0 = off).
 *   fpsq_band_factorize   = `jac_coord!` + `sparse(...)` + `ldl_factorize!` (src/solve_linear_system.jl:223-234): forms M
 *                           into 128 x 128 blocks of the band on the device and factors it with a right-looking
 *                           block-banded Cholesky (the dense back-end's MFMA block kernels); returns 1 (soft) with *info
 *                           = first non-positive pivot row when M is not positive definite and no regularisation is set.
 *   fpsq_band_set_regularization = the dynamic regularisation of :345-348, as for the dense back-end.
 *   fpsq_band_solve_two_*  = `ldiv!` with two right-hand sides (:189-203, :236-251) on the cached factor.
 * Storage is (m / 128) x (half bandwidth in blocks + 1) blocks; create fails with FPSQ_ERR_STATE when that does not fit
 * the device (or the half bandwidth exceeds 143 blocks AND a row spans more than 9600 columns), with FPSQ_ERR_ARG when
 * the pattern holds duplicate entries.  Arguments may be host or device pointers; calls are synchronous. */
typedef struct fpsq_band_s *fpsq_band;
typedef struct {
  int64_t n, m, nnz;
  int64_t nblocks;           /* 128-row blocks of M */
  int64_t bandwidth_blocks;  /* half bandwidth of M in blocks */
  int64_t factor_bytes;      /* storage of the banded factor */
  double last_form_ms, last_chol_ms, last_solve_ms;
  int64_t regularized_pivots;
  int64_t reordered;         /* 1: the symbolic phase reordered the rows of A (reverse Cuthill-McKee and / or the two-ended
                                order of the two elimination chains) */
  int64_t chains;            /* 2: the band is eliminated from both ends at once (two streams), 1: one chain */
} fpsq_band_info;
int fpsq_band_create(fpsq_band *out, int64_t n, int64_t m, const int32_t *rowptr, const int32_t *colind, int32_t device);
/* The same from the model's COO structure (`jac_structure!`, struct.jl:331-337; index_base 1 for Julia; duplicates allowed
 * and summed) -- with fpsq_band_factorize_coo taking the output of `jac_coord!` (nnz values in that order, HOST or DEVICE
 * memory): the sorted order is kept on the device and one gather(-sum) kernel fills the CSR slots, so neither the caller nor
 * the binding re-orders anything per x (src/solve_linear_system.jl:223-234). */
int fpsq_band_create_coo(fpsq_band *out, int64_t n, int64_t m, int64_t nnz, const int64_t *rows, const int64_t *cols,
                         int32_t index_base, int32_t device);
int fpsq_band_factorize_coo(fpsq_band b, const double *vals, double delta, int32_t *info);
/* the ordering decisions of fpsq_band_create alone, on the host (no device needed; rowptr / colind in HOST memory): row_perm
 * (m entries, may be null) = the caller's row stored at each position, info = blocks / half bandwidth / factor bytes /
 * reordered / chains of the structure fpsq_band_create would set up. */
int fpsq_band_analyze(int64_t n, int64_t m, const int32_t *rowptr, const int32_t *colind, int32_t *row_perm,
                      fpsq_band_info *info);
int fpsq_band_destroy(fpsq_band b);
const char *fpsq_band_last_error(fpsq_band b);
int fpsq_band_set_regularization(fpsq_band b, double tol, double reg);
int fpsq_band_factorize(fpsq_band b, const double *vals, double delta, int32_t *info);
int fpsq_band_solve_two_mixed(fpsq_band b, const double *rhs1, const double *rhs2, double *p1, double *q1, double *p2,
                              double *q2);
int fpsq_band_solve_two_least_squares(fpsq_band b, const double *rhs1, const double *rhs2, double *p1, double *q1,
                                      double *p2, double *q2);
int fpsq_band_get_info(fpsq_band b, fpsq_band_info *info);

/* ---- introspection for benchmarks / profiling */
typedef struct {
  int64_t n, m, nnz;
  int64_t spmv_a_blocks, spmv_at_blocks; /* workgroups per A / A' product */
  double last_solve_ms;                  /* device time of the last solve_two_* / qp_objgrad (HIP events) */
  double last_spmv_ms;                   /* device time inside SpMV/SpMM kernels during that call */
  int64_t last_spmv_launches;
  int64_t last_kernel_launches;
  int64_t last_prod_a[2];  /* launches of the A  product during that call with 1 and with 2 right-hand sides */
  int64_t last_prod_at[2]; /* same for the A' product */
  int64_t at_sorted;       /* 1: the row blocks of A' are stored column-sorted (coalesced gathers, see k_spmv<.., CSORT>) */
  int64_t comm_route;      /* 0: single GPU; else FPSQ_ROUTE_RCCL / _P2P (decided at the first solve of a halo-sharded handle)
                              / _LOCAL / _LOCAL_P2P (in-process groups) */
  int64_t last_fused_launches; /* of the launches counted in last_prod_a[1] AND last_prod_at[1]: those that carried both products of a
                                  joint iteration in one grid (k_iter_fused; FPSQ_FUSE_ITER=0 disables) */
  /* CUMULATIVE over the life of the handle -- all three are 0 on a healthy run; a test or a benchmark asserts so:
   *   fuse_fallbacks  calls that ran into a bounded wait of a one-launch iteration and were REPEATED on two launches per
   *                   iteration (the caller saw a delay and rc = the repeated call's; the handle stays on two launches);
   *   wait_timeouts   calls in which a bounded wait inside a product launch expired (leaders' record, block flags, tagged
   *                   partials) -- every fuse_fallback is one of them, the others ended in FPSQ_ERR_TIMEOUT;
   *   p2p_timeouts    calls in which a bounded wait for a PEER's record expired (peer-to-peer route): FPSQ_ERR_TIMEOUT. */
  int64_t fuse_fallbacks, wait_timeouts, p2p_timeouts;
  /* the Krylov loop of the last call: joint iterations enqueued and kernel launches enqueued for them (exchanges and stand-alone
   * steps included; start-up and epilogue excluded): launches / iterations = launches per joint iteration */
  int64_t last_loop_iterations, last_loop_launches;
  int64_t last_multi_launches, last_multi_iterations; /* of last_fused_launches: joint iterations that shared a launch with others
                                  (k_iter_multi: several iterations per launch, FPSQ_MULTI_ITER=1 disables), and how many launches
                                  carried them: launches of the call = last_prod_a + last_prod_at - last_fused_launches
                                  - (last_multi_iterations - last_multi_launches) */
  int64_t comm_in_launch_sums; /* 1: a sharded handle whose sums over the ranks need no launch of their own -- formed inside the
                                  launches that need them (peer-to-peer route, every rank on a device of its own: csrc
                                  fpsq_krylov.hip.h xch_sum), or a communicator of one rank; 0: gather / collective launches */
} fpsq_info;
int fpsq_get_info(fpsq_handle h, fpsq_info *info);
/* on != 0: bracket every SpMV/SpMM launch with HIP events on the solver's stream so that last_spmv_ms is filled
 * (adds event records between kernels; leave off when timing whole evaluations) */
int fpsq_set_profiling(fpsq_handle h, int32_t on);

/* Test hook for the run-ahead heuristics of the Krylov loop.  A solve normally enqueues the iterations of the PREVIOUS call
 * of the same kind without looking at the device and, right behind them, its final vector update and the caller's epilogue
 * kernels gated on the recurrences' `done` flags (csrc/fpsq.hip run_krylov).  This call overrides that expected count for
 * the NEXT solve call of the handle only (expect >= 0; 0 = "unknown": no speculation), so that tests can place the
 * speculation before, at and behind the true iteration count deterministically.  Results never depend on it. */
int fpsq_debug_expect_iterations(fpsq_handle h, int64_t expect);

const char *fpsq_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FPSQ_H */

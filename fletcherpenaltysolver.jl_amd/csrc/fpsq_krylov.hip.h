// fpsq_krylov.hip.h -- device-resident scalar recurrences of LSQR / CRAIG / MINRES.
//
// Each Krylov recurrence ("lane") keeps ALL of its scalars (Golub-Kahan alpha/beta, Givens state, norm estimates,
// stopping tests) in one device struct.  After every product kernel a one-workgroup "scalar step" kernel reduces
// that product's norm partials in a fixed order, advances the recurrences, writes the coefficients the next
// vector kernels will read (LaneCtl) and, once a stopping test fires, raises `done`: every later kernel of the lane
// then exits at its first instruction.  The host never reads a scalar inside the loop; it only watches a progress
// word in host-mapped memory to bound how far ahead it enqueues.
//
// Algorithms: Paige & Saunders LSQR (1982) and MINRES (1975), Craig's method in the Golub-Kahan form with the SQD
// extension of Arioli & Orban; stopping rules and variable names follow Krylov.jl 0.10 (lsqr!, craig!, minres!),
// which is what the reference calls (src/solve_two_systems_struct.jl:173-181, :217-239;
// src/solve_linear_system.jl:60-70).  Vector normalisations are DEFERRED: the stored Golub-Kahan vectors are
// beta*u / alpha*v and the 1/alpha, 1/beta factors ride in the coefficients of the next fused kernel.
#pragma once
#include "fpsq_kernels.hip.h"
#include "../../include/fpsq.h"

namespace fpsq {

struct alignas(8) Progress {  // host-mapped, written by the device, polled by the host: ONE 8-byte word, see publish()
  int32_t iter;
  int32_t done;
};

__device__ __forceinline__ void publish(Progress* p, int iter, int done) {
  // relaxed system-scope stores, no fence: the host only needs to see them eventually (it bounds its run-ahead with
  // them and re-reads the device state itself if the stream drains first); a fence here would hold the kernel for a
  // PCIe round trip on every step.  Even so the END of the kernel waits for these stores to be acknowledged over the
  // host link (measured: the next kernel starts ~3.6 us later than after a step that published nothing), so the steps
  // only publish when the host is going to look: at the end of a recurrence and from iteration `pub_from` on.
  // {iter, done} go out as ONE naturally aligned 8-byte store and the host reads them with one 8-byte load
  // (load_progress).  As two stores (rounds 1-2) the host could see iter = k with done still 0 for an instant and take a
  // recurrence that HAD ended at iteration k for one that had not: it then went on for an iteration whose (otherwise
  // no-op) A' launch carried the final LSQR update a second time -- the gated speculative flush, whose gates the device
  // rightly found open, had already applied it.  That was the intermittent last-digits mismatch of hprod (round 2's logs).
  if (p == nullptr) return;  // (a workgroup that only needs the step's RESULT: see step_run)
  const unsigned long long v = (unsigned long long)(unsigned)iter | ((unsigned long long)(unsigned)done << 32);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ double dsign(double a) { return (double)((a > 0.0) - (a < 0.0)); }

// Symmetric Givens reflection [c s; s -c][a; b] = [rho; 0] (Choi's SymOrtho, as Krylov.jl's sym_givens)
__device__ __forceinline__ void sym_givens(double a, double b, double& c, double& s, double& rho) {
  if (b == 0.0) {
    c = (a == 0.0) ? 1.0 : dsign(a);
    s = 0.0;
    rho = fabs(a);
  } else if (a == 0.0) {
    c = 0.0;
    s = dsign(b);
    rho = fabs(b);
  } else if (fabs(b) > fabs(a)) {
    const double t = a / b;
    s = dsign(b) / sqrt(1.0 + t * t);
    c = s * t;
    rho = b / s;
  } else {
    const double t = b / a;
    c = dsign(a) / sqrt(1.0 + t * t);
    s = c * t;
    rho = a / c;
  }
}

// =============================================================================================== LSQR
// min ||b - Bx||^2 + lambda^2 ||x||^2.  In the reference B = A' (n x m): "long" u in R^n, "short" v, w, x in R^m.
struct LsqrState {
  LaneCtl ctl;
  // parameters
  double lambda, atol, rtol, axtol, btol, etol, ctol;
  int64_t itmax;
  // recurrences
  double alpha, beta, beta1, Anorm2, dNorm2, xNorm2, c2, s2, z, xENorm2, phibar, rhobar, res2, ArNorm0;
  double err_vec[5];
  int32_t iter;
  int32_t pub_from;  // progress is published to the host from this iteration on (and whenever the recurrence ends)
  fpsq_stats stats;
};

__device__ __forceinline__ void lane_finish(LaneCtl& c, int it) {
  c.done = 1;
  c.upd_iter = it;
}

// after u~ = b: beta1 = ||b||
__device__ __forceinline__ void lsqr_begin_step(LsqrState* S, double bb, Progress* prog) {
  const double beta1 = sqrt(bb);
  S->beta1 = beta1;
  S->beta = beta1;
  S->iter = 0;
  S->ctl.skip = 0;
  S->ctl.upd_iter = -1;
  S->stats = fpsq_stats{0, 0, 0, FPSQ_ST_UNKNOWN, beta1, 0.0};
  if (beta1 == 0.0) {
    S->stats.solved = 1;
    S->stats.status = FPSQ_ST_ZERO_RHS;
    S->ctl.done = 1;
    publish(prog, 0, 1);
    return;
  }
  S->ctl.done = 0;
  S->ctl.ca = 1.0 / beta1;  // v~_1 = B' u~_1 / beta1
  S->ctl.cb = 0.0;
  // (nothing to publish: the host zeroed its progress word before the launch)
}

// after v~_1 = B'u_1: alpha_1 = ||v~_1||; start-up tests of lsqr!
__device__ __forceinline__ void lsqr_begin2_step(LsqrState* S, double aa, Progress* prog) {
  const double alpha = sqrt(aa), beta1 = S->beta1;
  S->alpha = alpha;
  S->Anorm2 = aa;
  S->dNorm2 = 0.0;
  S->xNorm2 = 0.0;
  S->c2 = -1.0;
  S->s2 = 0.0;
  S->z = 0.0;
  S->xENorm2 = 0.0;
  S->res2 = 0.0;
  for (int i = 0; i < 5; ++i) S->err_vec[i] = 0.0;
  const double ArNorm = alpha * beta1;
  S->ArNorm0 = ArNorm;
  S->stats.arnorm = ArNorm;
  if (alpha == 0.0) {
    S->stats.solved = 1;
    S->stats.status = FPSQ_ST_ZERO_ATB;
    S->stats.arnorm = 0.0;
    S->ctl.done = 1;
    publish(prog, 0, 1);
    return;
  }
  S->phibar = beta1;
  S->rhobar = alpha;
  const double Anorm = alpha, rNorm = beta1;
  const bool solved = (ArNorm / (Anorm * rNorm) <= S->axtol) | (1.0 + ArNorm / (Anorm * rNorm) <= 1.0);
  const bool tired = 0 >= S->itmax;
  if (solved | tired) {
    S->stats.solved = solved;
    S->stats.inconsistent = 1;
    S->stats.status = solved ? FPSQ_ST_SOLVED : FPSQ_ST_MAXITER;
    S->ctl.done = 1;
    publish(prog, 0, 1);
    return;
  }
  S->ctl.e[2] = 1.0 / alpha;     // w_1 = v_1 = v~_1 / alpha
  S->ctl.ca = 1.0 / alpha;       // u~_2 = B v~_1 / alpha - (alpha / beta) u~_1
  S->ctl.cb = -alpha / beta1;
}

// after u~ <- B v - alpha u: beta = ||u~||
__device__ __forceinline__ void lsqr_sa_step(LsqrState* S, double bb) {
  const double beta = sqrt(bb), alpha = S->alpha;
  S->beta = beta;
  if (beta != 0.0) {
    S->Anorm2 += alpha * alpha + beta * beta;
    if (S->lambda > 0.0) S->Anorm2 += S->lambda * S->lambda;
    S->ctl.ca = 1.0 / beta;  // v~ <- B'u~ / beta - (beta / alpha) v~
    S->ctl.cb = -beta / alpha;
    S->ctl.skip = 0;
  } else {
    S->ctl.skip = 1;  // u = 0: lsqr! skips the second half-step, alpha and v stay
  }
}

// after v~ <- B'u - beta v: alpha = ||v~||, then the QR update, norm estimates and stopping tests of lsqr!
__device__ __forceinline__ void lsqr_sb_step(LsqrState* S, double aa, double ww, int it, Progress* prog) {
  const double beta = S->beta, lambda = S->lambda;
  double alpha = S->alpha;
  if (!S->ctl.skip) alpha = sqrt(aa);
  S->alpha = alpha;
  S->ctl.skip = 0;
  S->iter = it;

  double c1, s1, rhobar1;
  sym_givens(S->rhobar, lambda, c1, s1, rhobar1);
  const double psi = s1 * S->phibar;
  double phibar = c1 * S->phibar;
  double c, s, rho;
  sym_givens(rhobar1, beta, c, s, rho);
  const double phi = c * phibar;
  phibar = s * phibar;
  S->phibar = phibar;

  S->xENorm2 += phi * phi;
  S->err_vec[it % 5] = phi;
  double err_lbnd = 0.0;
  if (it >= 5) {
    double t = 0.0;
    for (int i = 0; i < 5; ++i) t += S->err_vec[i] * S->err_vec[i];
    err_lbnd = sqrt(t);
  }
  const double tau = s * phi;
  const double theta = s * alpha;
  S->rhobar = -c * alpha;
  S->dNorm2 += ww / (rho * rho);

  // coefficients of the update kernel and of the next first half-step
  S->ctl.e[0] = phi / rho;
  S->ctl.e[1] = theta / rho;
  S->ctl.e[2] = (alpha != 0.0) ? 1.0 / alpha : 0.0;
  S->ctl.ca = (alpha != 0.0) ? 1.0 / alpha : 0.0;
  S->ctl.cb = (beta != 0.0) ? -alpha / beta : 0.0;

  const double delta = S->s2 * rho;
  const double gammabar = -S->c2 * rho;
  const double rhs = phi - delta * S->z;
  const double zbar = rhs / gammabar;
  const double xNorm = sqrt(S->xNorm2 + zbar * zbar);
  double c2, s2, gamma;
  sym_givens(gammabar, theta, c2, s2, gamma);
  S->c2 = c2;
  S->s2 = s2;
  S->z = rhs / gamma;
  S->xNorm2 += S->z * S->z;

  const double Anorm = sqrt(S->Anorm2);
  const double Acond = Anorm * sqrt(S->dNorm2);
  const double res1 = phibar * phibar;
  S->res2 += psi * psi;
  const double rNorm = sqrt(res1 + S->res2);
  const double ArNorm = alpha * fabs(tau);
  const double beta1 = S->beta1;

  const double test1 = rNorm / beta1;
  const double test2 = ArNorm / (Anorm * rNorm);
  const double test3 = 1.0 / Acond;
  const double t1 = test1 / (1.0 + Anorm * xNorm / beta1);
  const double rNormtol = S->btol + S->axtol * Anorm * xNorm / beta1;

  const bool ill_cond_mach = (1.0 + test3 <= 1.0);
  const bool solved_mach = (1.0 + test2 <= 1.0);
  const bool zero_resid_mach = (1.0 + t1 <= 1.0);
  const bool tired = it >= S->itmax;
  const bool ill_cond_lim = (test3 <= S->ctol);
  const bool solved_lim = (test2 <= S->axtol);
  const bool solved_opt = ArNorm <= S->atol + S->rtol * S->ArNorm0;
  const bool zero_resid_lim = (test1 <= rNormtol);
  const bool fwd_err = (it >= 5) && (err_lbnd <= S->etol * sqrt(S->xENorm2));
  const bool ill_cond = ill_cond_mach | ill_cond_lim;
  const bool zero_resid = zero_resid_mach | zero_resid_lim;
  const bool solved = solved_mach | solved_lim | solved_opt | zero_resid | fwd_err;

  S->stats.niter = it;
  S->stats.rnorm = rNorm;
  S->stats.arnorm = ArNorm;
  if (solved | tired | ill_cond) {
    int status = FPSQ_ST_UNKNOWN;
    if (tired) status = FPSQ_ST_MAXITER;
    if (ill_cond) status = FPSQ_ST_ILL_COND;
    if (solved) status = FPSQ_ST_SOLVED;
    if (zero_resid) status = FPSQ_ST_ZERO_RESID;
    if (fwd_err) status = FPSQ_ST_FWD_ERR;
    S->stats.status = status;
    S->stats.solved = solved;
    S->stats.inconsistent = !zero_resid;
    lane_finish(S->ctl, it);
    publish(prog, it, 1);
  } else if (it >= S->pub_from) {
    publish(prog, it, 0);
  }
}

// =============================================================================================== CRAIG
// min ||x|| s.t. Bx = b; with delta != 0 the reference passes M = (1/delta) I and sqd = true
// (src/solve_two_systems_struct.jl:216-228): [-I B'; B delta I][x; y] = [0; b].  B = A (m x n):
// "short" Mu~, w, y in R^m, "long" v~, x, w2 in R^n.  mu = 1/delta (or 1), lambda = 1 (or 0).
struct CraigState {
  LaneCtl ctl;
  double mu, lambda, atol, rtol, btol, ctol, xsign;
  int64_t itmax;
  double alpha, beta, beta1, theta, xi, deltag, rho_prev, omega, c1, s1, rho;
  double Anorm2, Dnorm2, xNorm2, eps_c;
  int32_t iter;
  int32_t pub_from;
  fpsq_stats stats;
};

// after Mu~ = b: beta1 = sqrt(mu) ||b||
__device__ __forceinline__ void craig_begin_step(CraigState* S, double bb, Progress* prog) {
  const double beta1 = sqrt(S->mu * bb);
  S->beta1 = beta1;
  S->beta = beta1;
  S->theta = beta1;
  S->xi = -1.0;
  S->deltag = S->lambda;
  S->rho_prev = 1.0;
  S->omega = 1.0;
  S->c1 = 1.0;
  S->s1 = 0.0;
  S->Anorm2 = 0.0;
  S->Dnorm2 = 0.0;
  S->xNorm2 = 0.0;
  S->iter = 0;
  S->ctl.skip = 0;
  S->ctl.upd_iter = -1;
  S->stats = fpsq_stats{0, 0, 0, FPSQ_ST_UNKNOWN, beta1, 0.0};
  if (beta1 == 0.0) {
    S->stats.solved = 1;
    S->stats.status = FPSQ_ST_ZERO_RHS;
    S->ctl.done = 1;
    publish(prog, 0, 1);
    return;
  }
  S->eps_c = S->atol + S->rtol * beta1;
  // start-up tests of craig! (bkwerr = 1, Anorm = xNorm = 0)
  const bool solved = (1.0 <= S->btol) | (beta1 <= S->eps_c) | (beta1 <= S->btol);
  const bool tired = 0 >= S->itmax;
  if (solved | tired) {
    S->stats.solved = solved;
    S->stats.status = solved ? FPSQ_ST_SOLVED : FPSQ_ST_MAXITER;
    S->ctl.done = 1;
    publish(prog, 0, 1);
    return;
  }
  S->ctl.done = 0;
  S->ctl.ca = S->mu / beta1;  // v~_1 = B'u_1 = (mu / beta1) B' Mu~
  S->ctl.cb = 0.0;            // Nv_0 = 0
}

// after v~ <- B'u - beta v: alpha = ||v~||, first Givens, xi; coefficients of the x / w2 / w / y updates
__device__ __forceinline__ void craig_sa_step(CraigState* S, double aa, int it, Progress* prog) {
  const double alpha = sqrt(aa);
  if (alpha == 0.0) {  // craig!: inconsistent = true; leave the loop without touching x, y
    S->stats.inconsistent = 1;
    S->stats.solved = 0;
    S->stats.status = FPSQ_ST_INCONSISTENT;
    S->stats.niter = S->iter;
    S->ctl.done = 1;
    S->ctl.upd_iter = -1;
    publish(prog, it, 1);
    return;
  }
  S->alpha = alpha;
  const double lambda = S->lambda;
  S->Anorm2 += alpha * alpha + lambda * lambda;
  double c1 = 1.0, s1 = 0.0, rho = alpha;
  if (lambda > 0.0) sym_givens(alpha, S->deltag, c1, s1, rho);
  S->c1 = c1;
  S->s1 = s1;
  S->rho = rho;
  const double xi = -S->theta / rho * S->xi;
  S->xi = xi;
  const double sx = S->xsign;
  if (lambda > 0.0) {
    S->ctl.e[0] = sx * xi * c1 / alpha;
    S->ctl.e[1] = sx * xi * s1 * S->omega;
    S->ctl.e[2] = s1 / alpha;
    S->ctl.e[3] = -c1 * S->omega;
  } else {
    S->ctl.e[0] = sx * xi / alpha;
    S->ctl.e[1] = S->ctl.e[2] = S->ctl.e[3] = 0.0;
  }
  S->ctl.e[4] = S->mu / S->beta;          // u = mu Mu~ / beta
  S->ctl.e[5] = S->theta / S->rho_prev;
  S->ctl.e[6] = xi / rho;
  S->ctl.upd_iter = it;                    // the updates of this iteration always run
  S->ctl.ca = 1.0 / alpha;                 // Mu~ <- B v~ / alpha - (alpha / beta) Mu~
  S->ctl.cb = -alpha / S->beta;
}

// after Mu~ <- B v - alpha Mu: beta, second Givens, estimates and the stopping tests of craig!
__device__ __forceinline__ void craig_sb_step(CraigState* S, double bb, double ww, int it, Progress* prog) {
  const double lambda = S->lambda, alpha = S->alpha;
  S->Dnorm2 += sqrt(ww);  // craig! accumulates ||w||, not ||w||^2
  const double beta = sqrt(S->mu * bb);
  S->beta = beta;
  double theta = beta;
  if (lambda > 0.0) {
    theta = beta * S->c1;
    const double gamma = beta * S->s1;
    double c2, s2, dg;
    sym_givens(lambda, gamma, c2, s2, dg);
    S->deltag = dg;
    S->omega = s2;  // w2 <- s2 w2, applied lazily by the next update
  }
  S->theta = theta;
  S->Anorm2 += beta * beta;
  const double Anorm = sqrt(S->Anorm2);
  const double Acond = Anorm * sqrt(S->Dnorm2);
  S->xNorm2 += S->xi * S->xi;
  const double xNorm = sqrt(S->xNorm2);
  double rNorm = beta * fabs(S->xi);
  if (lambda > 0.0) rNorm *= fabs(S->c1);
  S->iter = it;
  const double beta1 = S->beta1;
  const double bkwerr = rNorm / sqrt(beta1 * beta1 + S->Anorm2 * S->xNorm2);
  S->rho_prev = S->rho;
  (void)alpha;

  const bool solved_lim = bkwerr <= S->btol;
  const bool solved_mach = 1.0 + bkwerr <= 1.0;
  const bool solved_resid_tol = rNorm <= S->eps_c;
  const bool solved_resid_lim = rNorm <= S->btol + S->atol * Anorm * xNorm / beta1;
  const bool solved = solved_mach | solved_lim | solved_resid_tol | solved_resid_lim;
  const bool ill_cond_mach = 1.0 + 1.0 / Acond <= 1.0;
  const bool ill_cond_lim = 1.0 / Acond <= S->ctol;
  const bool ill_cond = ill_cond_mach | ill_cond_lim;
  const bool tired = it >= S->itmax;

  S->stats.niter = it;
  S->stats.rnorm = rNorm;
  // next first half-step: v~ <- (mu / beta) B' Mu~ - (beta / alpha) v~
  S->ctl.ca = (beta != 0.0) ? S->mu / beta : 0.0;
  S->ctl.cb = -beta / alpha;
  if (solved | ill_cond | tired) {
    int status = FPSQ_ST_UNKNOWN;
    if (tired) status = FPSQ_ST_MAXITER;
    if (solved) status = FPSQ_ST_SOLVED;
    if (ill_cond) status = FPSQ_ST_ILL_COND;
    S->stats.status = status;
    S->stats.solved = solved;
    S->stats.inconsistent = 0;
    S->ctl.done = 1;
    publish(prog, it, 1);
  } else if (it >= S->pub_from) {
    publish(prog, it, 0);
  }
}

// =============================================================================================== LNLQ
// Estrin, Orban & Saunders (2019); Krylov.jl lnlq! with N = I, lambda = 0, sigma = 0, transfer_to_craig = true, as
// the reference's GENERIC solve_least_norm calls it (src/solve_two_systems_struct.jl:251-281; the commented default
// workspace of :121): M = (1/delta) I when delta != 0 -- WITHOUT sqd, so M only preconditions (mu = 1/delta or 1).
// Same Golub-Kahan vectors as CRAIG ("short" Mu~, wbar, y in R^m, "long" v~, x in R^n); pass k of lnlq!'s loop needs
// beta_{k+1} (A product of iteration k) AND alpha_{k+1} (A' product of iteration k + 1), so its scalars, its stopping
// tests and the coefficients of its vector updates are computed by the step that follows the A' product of iteration
// k + 1, together with the x update that opens pass k + 1 (or the transfer to the final point).
struct LnlqState {
  LaneCtl ctl;
  double mu, atol, rtol, xsign;
  int64_t itmax;
  double alpha, beta, bNorm, eps_l, ahat, epsbar, tau, zetabar, zeta_km1, eta, ck, sk;
  int32_t iter;      // passes of lnlq!'s loop completed
  int32_t pub_from;
  fpsq_stats stats;
};

// after Mu~ = b: beta_1 = sqrt(mu) ||b||, ||b||_2 for the tolerance
__device__ __forceinline__ void lnlq_begin_step(LnlqState* S, double bb, Progress* prog) {
  const double bNorm = sqrt(bb);
  S->bNorm = bNorm;
  S->iter = 0;
  S->ctl.skip = 0;
  S->ctl.upd_iter = -1;
  S->stats = fpsq_stats{0, 0, 0, FPSQ_ST_UNKNOWN, bNorm, 0.0};
  if (bNorm == 0.0) {
    S->stats.solved = 1;
    S->stats.status = FPSQ_ST_ZERO_RHS;
    S->ctl.done = 1;
    publish(prog, 0, 1);
    return;
  }
  S->eps_l = S->atol + S->rtol * bNorm;
  const double beta1 = sqrt(S->mu * bb);
  S->beta = beta1;
  S->ctl.done = 0;
  S->ctl.ca = S->mu / beta1;  // v~_1 = B'u_1 = (mu / beta1) B' Mu~
  S->ctl.cb = 0.0;
}

// after v~ <- B'u - beta v (A' product of iteration it): alpha_it; it = 1: the initialisation of lnlq!; it >= 2:
// pass k = it - 1 of its loop (scalars, stopping tests), then the coefficients of the updates riding in the A product.
__device__ __forceinline__ void lnlq_sa_step(LnlqState* S, double aa, int it, Progress* prog) {
  const double alpha_n = sqrt(aa);
  const double sx = S->xsign;
  S->ctl.upd_iter = it;  // the updates of this iteration always run (they carry the transfer to the final point)
  S->ctl.e[4] = (S->beta != 0.0) ? S->mu / S->beta : 0.0;  // u = mu Mu~ / beta
  if (it == 1) {
    S->alpha = alpha_n;
    S->ahat = alpha_n;
    S->epsbar = alpha_n;
    S->tau = S->beta / alpha_n;
    S->zetabar = S->tau / S->epsbar;
    S->zeta_km1 = 0.0;
    S->eta = 0.0;
    S->ck = 0.0;
    S->sk = 0.0;
    // wbar_1 = u_1 (wbar starts at 0: s = 0, c = -1), nothing for y yet; x_aux += tau_1 v_1
    S->ctl.e[2] = 0.0;
    S->ctl.e[3] = 1.0;
    S->ctl.e[5] = 0.0;
    S->ctl.e[6] = 0.0;
    S->ctl.e[7] = 0.0;
    S->ctl.e[0] = (alpha_n != 0.0) ? sx * S->tau / alpha_n : 0.0;
    S->ctl.ca = (alpha_n != 0.0) ? 1.0 / alpha_n : 0.0;  // Mu~ <- B v~ / alpha - (alpha / beta) Mu~
    S->ctl.cb = -alpha_n / S->beta;
    return;
  }
  const int k = it - 1;  // `iter` of lnlq! during this pass
  const double beta_n = S->beta, bhat_n = beta_n, ahat_n = alpha_n;
  double c_n, s_n, eps_k;
  sym_givens(S->epsbar, bhat_n, c_n, s_n, eps_k);
  const double eta_n = ahat_n * s_n;
  const double epsbar_n = -ahat_n * c_n;
  const double tau_n = -bhat_n * S->tau / ahat_n;
  const double zeta_k = c_n * S->zetabar;
  const double zetabar_n = (tau_n - eta_n * zeta_k) / epsbar_n;
  double rNorm_lq = S->bNorm;
  if (k != 1) {
    const double t1 = S->epsbar * S->zetabar, t2 = bhat_n * S->sk * S->zeta_km1;
    rNorm_lq = fabs(S->ahat) * sqrt(t1 * t1 + t2 * t2);
  }
  const double rNorm_cg = fabs(bhat_n * S->tau);
  S->ck = c_n;
  S->sk = s_n;
  S->alpha = alpha_n;
  S->ahat = ahat_n;
  S->eta = eta_n;
  S->epsbar = epsbar_n;
  S->tau = tau_n;
  S->zeta_km1 = zeta_k;
  S->zetabar = zetabar_n;
  const bool tired = k >= S->itmax;
  const bool solved_lq = rNorm_lq <= S->eps_l;
  const bool solved_cg = rNorm_cg <= S->eps_l;
  S->iter = k;
  // y += zeta_k (c wbar + s u);  wbar <- s wbar - c u
  S->ctl.e[2] = s_n;
  S->ctl.e[3] = -c_n;
  S->ctl.e[5] = zeta_k * c_n;
  S->ctl.e[6] = zeta_k * s_n;
  S->ctl.e[7] = 0.0;
  S->stats.niter = k + 1;  // lnlq! advances its counter at the end of every pass, including the last
  if (solved_lq | solved_cg | tired) {
    int status = FPSQ_ST_UNKNOWN;
    if (tired) status = FPSQ_ST_MAXITER;
    if (solved_lq) status = FPSQ_ST_SOLVED_LQ;
    if (solved_cg) status = FPSQ_ST_SOLVED;
    S->stats.status = status;
    S->stats.solved = solved_lq | solved_cg;
    S->stats.inconsistent = 0;
    S->stats.rnorm = solved_cg ? rNorm_cg : rNorm_lq;
    // transfer: CRAIG point x += tau v, y += zetabar wbar;  else LQ point x += eta zeta v
    const double cx = solved_cg ? tau_n : eta_n * zeta_k;
    S->ctl.e[0] = (alpha_n != 0.0) ? sx * cx / alpha_n : 0.0;
    S->ctl.e[7] = solved_cg ? zetabar_n : 0.0;
    S->ctl.done = 1;
    publish(prog, k, 1);
    return;
  }
  S->stats.rnorm = rNorm_lq;
  S->ctl.e[0] = (alpha_n != 0.0) ? sx * tau_n / alpha_n : 0.0;  // x_aux += tau_{k+1} v_{k+1}: opens pass k + 1
  S->ctl.ca = (alpha_n != 0.0) ? 1.0 / alpha_n : 0.0;
  S->ctl.cb = -alpha_n / beta_n;
  if (k >= S->pub_from) publish(prog, k, 0);
}

// after Mu~ <- B v - alpha Mu (A product of iteration it): beta_{it+1}; coefficients of the next A' product
__device__ __forceinline__ void lnlq_sb_step(LnlqState* S, double bb) {
  const double beta = sqrt(S->mu * bb);
  S->beta = beta;
  S->ctl.ca = (beta != 0.0) ? S->mu / beta : 0.0;  // v~ <- (mu / beta) B' Mu~ - (beta / alpha) v~
  S->ctl.cb = -beta / S->alpha;
}

// =============================================================================================== MINRES
// (A A' + lambda I) x = b, all vectors in R^m; Krylov.jl minres! with M = I (src/solve_linear_system.jl:58-70).
// Per iteration: tmp = A' r2 (n), q = (A tmp + lambda r2) / beta, then three short element-wise stages:
//   E1: y0 = q - (beta/oldbeta) r1            partial <r2, y0>      -> alpha
//   E2: y = y0 - (alpha/beta) r2; w~ = r2/beta - delta w2 - eps w1  partial ||y||^2 -> beta_new, rotation
//   E3: w = w~ / gamma; x += phi w                                  partial ||x||^2 -> tests
struct MinresState {
  LaneCtl ctl;   // ca/cb: coefficients of the A product (q = ca A tmp + cb r2); e[0..6]: stage coefficients
  LaneCtl ctlT;  // the A' product tmp = A' r2 (ca = 1, cb = 0); `done` mirrors ctl.done
  double lambda, atol, rtol, etol, ctol;
  int64_t itmax;
  double beta1, beta, oldbeta, alpha, deltabar, epsln, delta, phibar, rhs1, rhs2, gmax, gmin, cs, sn;
  double ANorm2, xENorm2, root, gammabar, phi, gamma;
  double err_vec[5];
  int32_t iter;
  int32_t pub_from;
  fpsq_stats stats;
  // kmode = 1: the operator is K = [I A'; A -kdelta I] itself on stacked (n + m)-vectors (fpsq_options.kkt_method =
  // FPSQ_KKT_MINRES_K): y_p = (r2_p + A' r2_q) / beta through ctlT, y_q = (A r2_p - kdelta r2_q) / beta through ctl
  int32_t kmode, pad_;
  double kdelta;
};

__device__ __forceinline__ void minres_begin_step(MinresState* S, double bb, Progress* prog) {
  S->iter = 0;
  S->ctl.skip = 0;
  S->ctl.upd_iter = -1;
  S->stats = fpsq_stats{0, 0, 0, FPSQ_ST_UNKNOWN, 0.0, 0.0};
  if (bb == 0.0) {
    S->stats.solved = 1;
    S->stats.status = FPSQ_ST_ZERO_RHS;
    S->ctl.done = 1;
    S->ctlT.done = 1;
    publish(prog, 0, 1);
    return;
  }
  const double beta1 = sqrt(bb);
  S->beta1 = beta1;
  S->beta = beta1;
  S->oldbeta = 0.0;
  S->deltabar = 0.0;
  S->epsln = 0.0;
  S->phibar = beta1;
  S->rhs1 = beta1;
  S->rhs2 = 0.0;
  S->gmax = 0.0;
  S->gmin = INFINITY;
  S->cs = -1.0;
  S->sn = 0.0;
  S->ANorm2 = 0.0;
  S->xENorm2 = 0.0;
  for (int i = 0; i < 5; ++i) S->err_vec[i] = 0.0;
  S->stats.rnorm = beta1;
  const double eps_tol = S->atol + S->rtol * beta1;
  const bool solved = beta1 <= S->rtol;
  const bool tired = 0 >= S->itmax;
  if (solved | tired) {
    S->stats.solved = solved;
    S->stats.inconsistent = !(beta1 <= eps_tol);
    S->stats.status = solved ? FPSQ_ST_SOLVED : FPSQ_ST_MAXITER;
    S->ctl.done = 1;
    S->ctlT.done = 1;
    publish(prog, 0, 1);
    return;
  }
  S->ctl.done = 0;
  S->ctlT.done = 0;
  S->ctlT.skip = 0;
  S->ctlT.ca = 1.0;
  S->ctlT.cb = 0.0;
  S->ctl.ca = 1.0 / beta1;         // q = (A tmp + lambda r2) / beta
  S->ctl.cb = S->lambda / beta1;
  if (S->kmode) {
    S->ctlT.ca = S->ctlT.cb = 1.0 / beta1;
    S->ctl.cb = -S->kdelta / beta1;
  }
  S->ctl.e[0] = 0.0;               // E1: beta/oldbeta (no r1 term at iteration 1)
}

// after E1: alpha = <r2, y0> / beta
__device__ __forceinline__ void minres_a_step(MinresState* S, double dot) {
  const double beta = S->beta;
  const double alpha = dot / beta;
  S->alpha = alpha;
  const double delta = S->cs * S->deltabar + S->sn * alpha;
  S->delta = delta;
  S->ctl.e[1] = alpha / beta;  // E2: y = y0 - e1 r2
  S->ctl.e[2] = 1.0 / beta;    //     w~ = e2 r2 - e3 w2 - e4 w1
  S->ctl.e[3] = delta;
  S->ctl.e[4] = S->epsln;
}

// after E2: beta_new = ||y||, plane rotation, phi
__device__ __forceinline__ void minres_b_step(MinresState* S, double yy, int it) {
  const double epsM = 2.220446049250313e-16;
  const double alpha = S->alpha;
  S->oldbeta = S->beta;
  const double beta = sqrt(yy);
  S->beta = beta;
  S->ANorm2 += alpha * alpha + S->oldbeta * S->oldbeta + beta * beta;
  const double gammabar = S->sn * S->deltabar - S->cs * alpha;
  S->epsln = S->sn * beta;
  S->deltabar = -S->cs * beta;
  S->root = sqrt(gammabar * gammabar + S->deltabar * S->deltabar);
  S->gammabar = gammabar;
  double gamma = sqrt(gammabar * gammabar + beta * beta);
  gamma = fmax(gamma, epsM);
  S->gamma = gamma;
  S->stats.arnorm = S->phibar * S->root;
  S->cs = gammabar / gamma;
  S->sn = beta / gamma;
  const double phi = S->cs * S->phibar;
  S->phibar = S->sn * S->phibar;
  S->phi = phi;
  S->xENorm2 += phi * phi;
  S->err_vec[it % 5] = phi;
  S->ctl.e[5] = 1.0 / gamma;  // E3: w = e5 w~; x += e6 w
  S->ctl.e[6] = phi;
  S->ctl.upd_iter = it;
  // The next iteration's product coefficients only depend on beta: set here, so the next A' product (in which E3 of
  // THIS iteration rides) and A product need not wait for the stopping tests of stage C.
  S->ctl.ca = (beta != 0.0) ? 1.0 / beta : 0.0;
  S->ctl.cb = (beta != 0.0) ? S->lambda / beta : 0.0;
  if (S->kmode) {
    S->ctlT.ca = S->ctlT.cb = S->ctl.ca;
    S->ctl.cb = -S->kdelta * S->ctl.ca;
  }
  S->ctl.e[0] = (S->oldbeta != 0.0) ? beta / S->oldbeta : 0.0;
}

// after E3: ||x||, estimates and the stopping tests of minres!
__device__ __forceinline__ void minres_c_step(MinresState* S, double xx, int it, Progress* prog) {
  const double epsM = 2.220446049250313e-16;
  S->iter = it;
  double err_lbnd = 0.0;
  if (it >= 5) {
    double t = 0.0;
    for (int i = 0; i < 5; ++i) t += S->err_vec[i] * S->err_vec[i];
    err_lbnd = sqrt(t);
  }
  const double gamma = S->gamma;
  S->gmax = fmax(S->gmax, gamma);
  S->gmin = fmin(S->gmin, gamma);
  const double zeta = S->rhs1 / gamma;
  S->rhs1 = S->rhs2 - S->delta * zeta;
  S->rhs2 = -S->epsln * zeta;
  const double ANorm = sqrt(S->ANorm2);
  const double xNorm = sqrt(xx);
  const double rNorm = S->phibar;
  const double test1 = rNorm / (ANorm * xNorm);
  const double test2 = S->root / ANorm;
  const double Acond = S->gmax / S->gmin;
  const double beta = S->beta, beta1 = S->beta1;
  S->stats.niter = it;
  S->stats.rnorm = rNorm;
  if (it == 1 && beta / beta1 <= 10 * epsM) {
    S->stats.solved = 1;
    S->stats.inconsistent = 1;
    S->stats.status = FPSQ_ST_ZERO_ATB;
    S->ctl.done = 1;
    S->ctlT.done = 1;
    publish(prog, it, 1);
    return;
  }
  const double eps_tol = S->atol + S->rtol * beta1;
  const bool ill_cond_mach = (1.0 + 1.0 / Acond <= 1.0);
  const bool solved_mach = (1.0 + test2 <= 1.0);
  const bool zero_resid_mach = (1.0 + test1 <= 1.0);
  const bool resid_decrease_mach = (rNorm + 1.0 <= 1.0);
  const bool tired = it >= S->itmax;
  const bool ill_cond_lim = (1.0 / Acond <= S->ctol);
  const bool solved_lim = (test2 <= eps_tol);
  const bool zero_resid_lim = (test1 <= epsM);
  const bool resid_decrease_lim = (rNorm <= eps_tol);
  const bool fwd_err = (it >= 5) && (err_lbnd <= S->etol * sqrt(S->xENorm2));
  const bool zero_resid = zero_resid_mach | zero_resid_lim;
  const bool resid_decrease = resid_decrease_mach | resid_decrease_lim;
  const bool ill_cond = ill_cond_mach | ill_cond_lim;
  const bool solved = solved_mach | solved_lim | zero_resid | fwd_err | resid_decrease;
  if (solved | tired | ill_cond) {
    int status = FPSQ_ST_UNKNOWN;
    if (tired) status = FPSQ_ST_MAXITER;
    if (ill_cond) status = FPSQ_ST_ILL_COND;
    if (solved) status = FPSQ_ST_SOLVED;
    if (zero_resid) status = FPSQ_ST_ZERO_RESID;
    if (fwd_err) status = FPSQ_ST_FWD_ERR;
    S->stats.status = status;
    S->stats.solved = solved;
    S->stats.inconsistent = !zero_resid;
    S->ctl.done = 1;
    S->ctlT.done = 1;
    publish(prog, it, 1);
  } else if (it >= S->pub_from) {
    publish(prog, it, 0);
  }
}

// =============================================================================================== sums over the ranks, inside a launch
// Row-sharded runs in halo mode on the peer-to-peer route: a sum over the ranks is formed BY THE WORKGROUP THAT NEEDS IT, inside the
// launch it belongs to -- no gather kernel, no collective call between two product launches (round 5; the all-gathered arrays of
// round 4, StepArgs::nseg, remain for the RCCL route and for ranks sharing one device).  Every rank owns a small receive area
//     rx[kXchRing][nranks][8 words]          (fine-grained memory, mapped by the peers: hipIpcOpenMemHandle / the same device)
// of self-validating 8-byte words -- the exchange's sequence number in the low half, half a double in the high half, like the
// leaders' records: a word is written and read in one piece, so no flag, no fence, no ordering between the words.  A workgroup that
// holds its rank's local sums (a scalar-step leader: two sums of its lane; the phi reduction: four) WRITES them into every
// peer's area (row = its own rank), then polls its own area until every peer's row carries this exchange's number, and adds the
// P rows up IN RANK ORDER: the same bits on every rank whatever arrives first, so the replicated recurrence scalars -- and with
// them iteration counts, loop exits and the host's launch sequence -- stay identical by construction.
//  * Redundant writers are harmless: the eight leaders of a lane (one per XCD) all hold the same bits and all push them.
//  * The ring: exchange k uses slot k mod 4.  A launch carries at most two exchanges per lane (the head and the mid step of a
//    one-launch iteration) and a rank cannot START exchange k + 2 before every workgroup of the launch that held exchange k has
//    ended (stream order); a peer can be at most one exchange ahead of the slowest rank (it needs that rank's row to go on).  So
//    slot k mod 4 is rewritten (by exchange k + 4) only after every reader of exchange k is gone -- including a mid leader
//    that repeats the head step's exchange, and a leader that another kernel held up for a whole launch.
//  * Every wait is bounded (max_polls looks with a sleep in between); an expired wait raises the communicator's host-mapped
//    failure word, the sums come out as NaN, and every later exchange of the call leaves at once when it finds the word set:
//    the call returns FPSQ_ERR_TIMEOUT (fpsq_info.p2p_timeouts), never hangs.
// ONE PROCESS PER GPU is assumed here: the waiting leaders keep the other workgroups of their launch waiting for the record, and
// those hold workgroup slots -- ranks that SHARE a device (the one-GPU rehearsals of tests/) could starve each other of the
// slots their own leaders need.  The communicator therefore switches this on only when every rank has a device of its own
// (IpcComm::arm), or when a test with small grids forces it (FPSQ_LX=2).
constexpr int kXchRanks = 8;
constexpr int kXchRing = 4;
constexpr int kXchWords = 8;  // per (slot, sender): lane 0's two sums (4 words), lane 1's (4 words); phi uses all eight for its four
struct XchTable {
  unsigned long long* peer[kXchRanks];  // rank p's receive area; peer[rank] is this rank's own
  int32_t nranks, rank;
  int32_t max_polls;
  int32_t delay_rank;                   // tests (FPSQ_DEBUG_XCH_DELAY = r + 1): rank r holds every push back by ~100 us
  int* fail;                            // host-mapped: a bounded wait for a peer expired
  unsigned int long_delay_ticks;        // tests (FPSQ_DEBUG_XCH_LONG_DELAY_MS): instead, that rank holds the pushes of every 128th exchange
  unsigned int pad_;                    // back by this long (100 MHz ticks) -- a host that was descheduled in the middle of a solve
};
__device__ __forceinline__ unsigned long long xch_hi(double v, unsigned int seq) {
  return ((unsigned long long)__double_as_longlong(v) & 0xffffffff00000000ull) | seq;
}
__device__ __forceinline__ unsigned long long xch_lo(double v, unsigned int seq) {
  return ((unsigned long long)__double_as_longlong(v) << 32) | seq;
}
// Every thread of the workgroup calls this (>= 64 threads; workgroup barriers inside).  In: v[0 .. NV) valid in thread 0 = this
// rank's local sums.  Out: v[] in thread 0 = the sums over the ranks.  `half`: which half of the sender's row (NV = 2: the lane
// of the step; NV = 4: 0).  red: LDS scratch, >= 9 * NV doubles.
template <int NV>
__device__ __forceinline__ void xch_sum(const XchTable* xt, unsigned int seq, int half, double (&v)[NV], double* red) {
  static_assert(NV == 2 || NV == 4, "a step's two sums or phi's four");
  const int t = threadIdx.x;
  const int P = xt->nranks, me = xt->rank;
  __syncthreads();  // (red may still be read by the reduction that produced v)
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[k] = v[k];
  }
  __syncthreads();
  const int woff = NV == 2 ? 4 * half : 0;
  const size_t slot = (size_t)(seq & (kXchRing - 1)) * (size_t)P * kXchWords;
  if (t < P) {
    double got[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) got[k] = red[k];
    if (t != me) {
      const bool dead = __hip_atomic_load(xt->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
      if (!dead) {
        if (xt->delay_rank == me + 1) {  // (tests: what a rank that is late with its push does to the others)
          const unsigned long long ticks = xt->long_delay_ticks == 0 ? 10000ull : (seq & 127u) == 100u ? xt->long_delay_ticks : 0ull;
          const unsigned long long t0 = wall_clock64();
          while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
        }
        unsigned long long* dst = xt->peer[t] + slot + (size_t)me * kXchWords + woff;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
          __hip_atomic_store(dst + 2 * k, xch_hi(got[k], seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(dst + 2 * k + 1, xch_lo(got[k], seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      const unsigned long long* src = xt->peer[me] + slot + (size_t)t * kXchWords + woff;
      unsigned long long w[2 * NV];
      bool ok = false;
      for (int look = 0; look < xt->max_polls && !dead && !ok; ++look) {
        if (look) {
          __builtin_amdgcn_s_sleep(8);
          if ((look & 255) == 0 && __hip_atomic_load(xt->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        }
        ok = true;
#pragma unroll
        for (int k = 0; k < 2 * NV; ++k) {
          w[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          ok = ok && (unsigned int)(w[k] & 0xffffffffull) == seq;
        }
      }
      if (ok) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
          got[k] = __longlong_as_double((long long)((w[2 * k] & 0xffffffff00000000ull) | (w[2 * k + 1] >> 32)));
      } else {
        if (!dead) __hip_atomic_store(xt->fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int k = 0; k < NV; ++k) got[k] = __builtin_nan("");
      }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) red[NV + t * NV + k] = got[k];
  }
  __syncthreads();
  if (t == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      double a = 0.0;
      for (int r = 0; r < P; ++r) a += red[NV + r * NV + k];  // rank order
      v[k] = a;
    }
  }
}

// =============================================================================================== step kernel
// One launch advances up to two recurrences: workgroup b handles step[b]; the norm partials of a product are summed in a
// fixed order (reproducible).
enum StepKind : int32_t {
  STEP_NONE = 0,
  STEP_LSQR_BEGIN, STEP_LSQR_BEGIN2, STEP_LSQR_SA, STEP_LSQR_SB,
  STEP_CRAIG_BEGIN, STEP_CRAIG_SA, STEP_CRAIG_SB,
  STEP_MINRES_BEGIN, STEP_MINRES_A, STEP_MINRES_B, STEP_MINRES_C,
  STEP_LNLQ_BEGIN, STEP_LNLQ_SA, STEP_LNLQ_SB
};

struct StepArgs {
  int32_t kind;
  int32_t it;
  void* state;
  const double* p0;  // partials of the product that just ran
  const double* p1;  // second partial array (||w||^2 of the last update), may be null
  int32_t n0, n1;
  Progress* prog;
  fpsq_stats* host_stats;  // host-mapped: the step that ends the recurrence leaves the final stats there (no copy)
  // Row-sharded runs in halo mode: p0 / p1 point into an ALL-GATHERED buffer holding `nseg` ranks' copies of the array,
  // `seg_stride` doubles apart, n0 / n1 entries each (zero-padded to a count common to all ranks).  Every rank sums the
  // same numbers in the same (rank-major) order: the replicated scalars stay bitwise identical whatever reduction
  // algorithm the collective library would use.  nseg <= 1: one plain array.
  int32_t nseg, seg_stride;
  // where the advanced state goes; null = in place.  Steps that ride in a product launch (step_run with many readers)
  // alternate between two copies of the state: every workgroup of the launch reads `state`, only workgroup 0 writes
  // `state_out`, so no reader can ever see a half-written or an already advanced state.
  void* state_out;
  // riding steps: the 8-byte-word offset, inside the state, of the control block whose ca / cb / done / skip the NEXT product
  // reads (0: the state's first member; a MINRES lane has a second block, ctlT, for A' products)
  int32_t prod_ctl_off, pad_;
  // (Row-sharded runs with the sums over the ranks formed inside the launch, xch_sum above: p0 / p1 are the rank's LOCAL arrays;
  // the peer table and the exchange's number travel NEXT to the steps -- RideArgs, k_step's own arguments -- not in this struct:
  // four copies of it are kernel arguments of the one-launch iteration, and 24 more bytes in each cost that kernel 3 %, measured)
};

// 256 threads: the <= ~5000 norm partials are still summed with a few batches of independent loads per thread, and a
// 4-wave workgroup is dispatched (and synchronised) markedly faster than a 16-wave one: +6.5 % evaluations/s at the
// headline size against 1024 threads (128: +4 %, 512: +5 %).
constexpr int kStepThreads = 256;

// sums of two partial arrays at once; results valid in thread 0
// U loads per thread issued back to back and UNCONDITIONALLY (clamped index, value masked afterwards): a predicated
// load makes hipcc wait for each one, and a loop over batches costs one global round trip (~1.5 us) per batch.
template <int U>
__device__ __forceinline__ double partial_batch(const double* p, int base, int n, int t) {
  double v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = base + u * kStepThreads + t;
    v[u] = p[i < n ? i : n - 1];
  }
  double a = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) a += (base + u * kStepThreads + t < n) ? v[u] : 0.0;
  return a;
}

// the same for entries [base, base + U * kStepThreads) of a flattened [nseg][n] array whose segments lie `stride` doubles
// apart (one integer division per thread, then incremental index arithmetic)
template <int U>
__device__ __forceinline__ double seg_batch(const double* p, int n, int nseg, int stride, int base, int t) {
  const int total = nseg * n;
  int j = base + t;
  int sg = j / n, i = j - sg * n;
  const int q = kStepThreads / n, r = kStepThreads - q * n;
  double v[U];
  bool ok[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    ok[u] = j < total;
    v[u] = p[ok[u] ? (size_t)sg * stride + i : 0];
    j += kStepThreads;
    i += r;
    sg += q;
    if (i >= n) {
      i -= n;
      ++sg;
    }
  }
  double a = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) a += ok[u] ? v[u] : 0.0;
  return a;
}

__device__ __forceinline__ void block_reduce_two(double a, double b, double* red, double& s0, double& s1) {
  const int t = threadIdx.x;
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = t & 63, w = t >> 6;
  if (lane == 0) {
    red[w] = a;
    red[16 + w] = b;
  }
  __syncthreads();
  if (t == 0) {
    s0 = 0.0;
    s1 = 0.0;
    for (int k = 0; k < kStepThreads / 64; ++k) {
      s0 += red[k];
      s1 += red[16 + k];
    }
  }
}

// all-gathered arrays (StepArgs::nseg > 1)
__device__ __forceinline__ void reduce_two_seg(const double* p0, int n0, const double* p1, int n1, int nseg, int stride,
                                               double* red, double& s0, double& s1) {
  double a = 0.0, b = 0.0;
  const int t = threadIdx.x;
  const int t0 = nseg * n0, t1 = nseg * n1;
  if (t0 > 0 && t0 <= kStepThreads * 24 && t1 <= kStepThreads * 4) {  // one batch of loads
    const double x = t0 <= kStepThreads * 8 ? seg_batch<8>(p0, n0, nseg, stride, 0, t) : seg_batch<24>(p0, n0, nseg, stride, 0, t);
    const double y = t1 > 0 ? seg_batch<4>(p1, n1, nseg, stride, 0, t) : 0.0;
    a = x;
    b = y;
  } else {
    if (n0 > 0)
      for (int base = 0; base < t0; base += kStepThreads * 8) a += seg_batch<8>(p0, n0, nseg, stride, base, t);
    if (n1 > 0)
      for (int base = 0; base < t1; base += kStepThreads * 8) b += seg_batch<8>(p1, n1, nseg, stride, base, t);
  }
  block_reduce_two(a, b, red, s0, s1);
}

__device__ __forceinline__ void reduce_two(const double* p0, int n0, const double* p1, int n1, double* red,
                                           double& s0, double& s1) {
  double a = 0.0, b = 0.0;
  const int t = threadIdx.x;
  // the whole of both arrays in ONE batch of loads whenever they fit (<= 24 x 256 = 6144 and <= 4 x 256 entries:
  // 4883 + 391 at the headline size), in a fixed order
  constexpr int U0 = 24, U1 = 4;
  if (n0 > 0 && n0 <= kStepThreads * U0 && n1 <= kStepThreads * U1) {
    if (n0 <= kStepThreads * 4) {
      const double x = partial_batch<4>(p0, 0, n0, t);
      const double y = n1 > 0 ? partial_batch<U1>(p1, 0, n1, t) : 0.0;
      a = x;
      b = y;
    } else if (n0 <= kStepThreads * 8) {
      const double x = partial_batch<8>(p0, 0, n0, t);
      const double y = n1 > 0 ? partial_batch<U1>(p1, 0, n1, t) : 0.0;
      a = x;
      b = y;
    } else {
      const double x = partial_batch<U0>(p0, 0, n0, t);
      const double y = n1 > 0 ? partial_batch<U1>(p1, 0, n1, t) : 0.0;
      a = x;
      b = y;
    }
  } else {
    for (int base = 0; base < n0; base += kStepThreads * 8) a += partial_batch<8>(p0, base, n0, t);
    for (int base = 0; base < n1; base += kStepThreads * 8) b += partial_batch<8>(p1, base, n1, t);
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = t & 63, w = t >> 6;
  if (lane == 0) {
    red[w] = a;
    red[16 + w] = b;
  }
  __syncthreads();
  if (t == 0) {
    s0 = 0.0;
    s1 = 0.0;
    for (int k = 0; k < kStepThreads / 64; ++k) {
      s0 += red[k];
      s1 += red[16 + k];
    }
  }
}

__device__ __forceinline__ bool lane_done(const StepArgs& a) {
  // LaneCtl is the first member of every state struct
  return a.kind != STEP_LSQR_BEGIN && a.kind != STEP_CRAIG_BEGIN && a.kind != STEP_MINRES_BEGIN &&
         a.kind != STEP_LNLQ_BEGIN && reinterpret_cast<const LaneCtl*>(a.state)->done;
}

__device__ __forceinline__ int state_bytes(int kind) {
  switch (kind) {
    case STEP_LSQR_BEGIN: case STEP_LSQR_BEGIN2: case STEP_LSQR_SA: case STEP_LSQR_SB: return (int)sizeof(LsqrState);
    case STEP_CRAIG_BEGIN: case STEP_CRAIG_SA: case STEP_CRAIG_SB: return (int)sizeof(CraigState);
    case STEP_LNLQ_BEGIN: case STEP_LNLQ_SA: case STEP_LNLQ_SB: return (int)sizeof(LnlqState);
    default: return (int)sizeof(MinresState);
  }
}

// thread 0: one recurrence advanced in `S` (the staged state) from the sums of its partial arrays
__device__ __forceinline__ void step_advance(const StepArgs& a, void* S, double s0, double s1, Progress* prog) {
  switch (a.kind) {
    case STEP_LSQR_BEGIN: lsqr_begin_step((LsqrState*)S, s0, prog); break;
    case STEP_LSQR_BEGIN2: lsqr_begin2_step((LsqrState*)S, s0, prog); break;
    case STEP_LSQR_SA: lsqr_sa_step((LsqrState*)S, s0); break;
    case STEP_LSQR_SB: lsqr_sb_step((LsqrState*)S, s0, s1, a.it, prog); break;
    case STEP_CRAIG_BEGIN: craig_begin_step((CraigState*)S, s0, prog); break;
    case STEP_CRAIG_SA: craig_sa_step((CraigState*)S, s0, a.it, prog); break;
    case STEP_CRAIG_SB: craig_sb_step((CraigState*)S, s0, s1, a.it, prog); break;
    case STEP_MINRES_BEGIN: minres_begin_step((MinresState*)S, s0, prog); break;
    case STEP_MINRES_A: minres_a_step((MinresState*)S, s0); break;
    case STEP_MINRES_B: minres_b_step((MinresState*)S, s0, a.it); break;
    case STEP_MINRES_C: minres_c_step((MinresState*)S, s0, a.it, prog); break;
    case STEP_LNLQ_BEGIN: lnlq_begin_step((LnlqState*)S, s0, prog); break;
    case STEP_LNLQ_SA: lnlq_sa_step((LnlqState*)S, s0, a.it, prog); break;
    case STEP_LNLQ_SB: lnlq_sb_step((LnlqState*)S, s0); break;
    default: break;
  }
}

// thread 0 of a committing workgroup: the step that ends a recurrence leaves the final statistics in host-mapped memory
__device__ __forceinline__ void step_final_stats(const StepArgs& a, const void* S) {
  if (a.host_stats && reinterpret_cast<const LaneCtl*>(S)->done) {
    const fpsq_stats* fin = a.kind >= STEP_LNLQ_BEGIN ? &((const LnlqState*)S)->stats
                            : a.kind >= STEP_MINRES_BEGIN ? &((const MinresState*)S)->stats
                            : a.kind >= STEP_CRAIG_BEGIN ? &((const CraigState*)S)->stats
                                                         : &((const LsqrState*)S)->stats;
    *a.host_stats = *fin;
  }
}

// One scalar step by the calling workgroup (kStepThreads threads): the state is staged in `st` (LDS, 80 words), the partial
// sums are reduced in the fixed order, thread 0 advances the recurrence IN `st`.  On return (after the closing barrier) `st`
// holds the new state for every thread of the workgroup.  commit: this workgroup also performs the step's side effects --
// the progress word, the final statistics in host-mapped memory, the state written to a.state_out (or back in place).
// Without commit a workgroup only wants the result (the redundant leaders of a launch with riding steps).
// hook (optional; the riding leaders, several of which compute a lane's step): hook->advanced() is called by thread 0 as soon
// as the advanced state stands in `st`, before anything is written to global memory (the leaders publish the next product's
// coefficients there).
struct NoStepHook {
  __device__ void advanced() const {}
};
// xt != null (row-sharded, in-launch sums): the two sums travel as exchange `xseq`, in half `xlane` of the ranks' rows (xch_sum)
// XCH is a TEMPLATE parameter: the exchange's code merely being PRESENT in the leaders' path cost the one-launch iteration of a
// single-GPU handle 3 % (978 -> 950 evals/s against round 4's library on one box, bisected to this: the product workgroups
// of the same kernel pay for the leaders' registers) -- kernels of handles that never exchange are instantiated without it.
template <class Hook = NoStepHook, bool XCH = false>
__device__ __forceinline__ void step_run(const StepArgs& a, double* red /* 32 */, unsigned long long* st /* 80 */, bool commit,
                                         const Hook* hook = nullptr, const XchTable* xt = nullptr, unsigned int xseq = 0, int xlane = 0) {
  // The recurrence state (<= 0.5 KB) is staged in LDS with one coalesced read issued together with the partial-sum
  // loads, advanced there by thread 0, and written back with one coalesced store: the ~40 dependent scalar accesses of
  // a step then cost LDS latency instead of a global round trip each.
  const int nq = state_bytes(a.kind) / 8;
  const unsigned long long* gsrc = reinterpret_cast<const unsigned long long*>(a.state);
  if ((int)threadIdx.x < nq) st[threadIdx.x] = gsrc[threadIdx.x];
  const bool skip = lane_done(a);  // (uniform: read from global memory by every thread)
  double s0 = 0.0, s1 = 0.0;
  if (!skip) {
    // (contains the workgroup barrier that publishes `st`)
    if (a.nseg > 1) reduce_two_seg(a.p0, a.n0, a.p1, a.p1 ? a.n1 : 0, a.nseg, a.seg_stride, red, s0, s1);
    else reduce_two(a.p0, a.n0, a.p1, a.p1 ? a.n1 : 0, red, s0, s1);
    if constexpr (XCH) {
      if (xt != nullptr) {  // (uniform; `skip` is too, on every rank: the states are replicated)
        double v[2] = {s0, s1};
        xch_sum<2>(xt, xseq, xlane, v, red);
        s0 = v[0];
        s1 = v[1];
      }
    }
  } else {
    __syncthreads();
  }
  if (threadIdx.x == 0 && hook && skip) hook->advanced();  // (a finished lane: its state stands as it is)
  if (threadIdx.x == 0 && !skip) {
    void* S = st;
    Progress* prog = commit ? a.prog : nullptr;
    step_advance(a, S, s0, s1, prog);
    if (hook) hook->advanced();
    if (commit) step_final_stats(a, S);
  }
  __syncthreads();
  if (commit && (!skip || a.state_out != nullptr)) {
    unsigned long long* gdst = reinterpret_cast<unsigned long long*>(a.state_out ? a.state_out : a.state);
    if ((int)threadIdx.x < nq) gdst[threadIdx.x] = st[threadIdx.x];
  }
}

template <bool XCH = false>
__global__ __launch_bounds__(kStepThreads) void k_step(StepArgs a0, StepArgs a1, const XchTable* xt, unsigned int xseq) {
  const StepArgs& a = blockIdx.x == 0 ? a0 : a1;
  if (a.kind == STEP_NONE) return;
  if (a.state_out == nullptr && lane_done(a)) return;  // (in place and nothing to do)
  __shared__ double red[32];
  __shared__ __attribute__((aligned(16))) unsigned long long st[80];
  step_run<NoStepHook, XCH>(a, red, st, true, nullptr, xt, xseq, (int)blockIdx.x);
}

static_assert(sizeof(LsqrState) % 8 == 0 && sizeof(LsqrState) <= 640, "state staging");
static_assert(sizeof(CraigState) % 8 == 0 && sizeof(CraigState) <= 640, "state staging");
static_assert(sizeof(MinresState) % 8 == 0 && sizeof(MinresState) <= 640, "state staging");
static_assert(sizeof(LnlqState) % 8 == 0 && sizeof(LnlqState) <= 640, "state staging");

}  // namespace fpsq

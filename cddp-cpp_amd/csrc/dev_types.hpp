// Shared host/device POD types of the batched solver core.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include "../../include/cddp_hip.h"

namespace cddp_dev {

constexpr int kMaxCons = 4;     // path-constraint objects per problem
constexpr int kMaxTerms = 2;    // terminal-constraint objects per problem
constexpr int kPool = 1024;     // doubles of problem constants (Q, R, Qf, x_ref, bounds, ...)
constexpr int kFilterCap = 8;   // > ipddp.max_filter_size + 1
constexpr int kHistCols = 9;

// per-trajectory phases of the device state machine (cddp_solver_base.cpp:74-152)
enum Phase : int { PH_ACTIVE = 0, PH_FWD1 = 1, PH_FWD2 = 2, PH_DONE = 3 };

struct ConDev {
  int kind, dim, dual_dim, offset;               // offset = first row in the stacked dual vector
  int off_lower, off_upper, off_center, off_A, off_b;   // offsets into ProblemDev::pool
  int _pad;
  double scale, radius;
};

struct TermDev {
  int kind, dim, offset;     // offset into stacked terminal-ineq rows (or terminal-eq rows)
  int off_target, off_A, off_b;
};

struct ProblemDev {
  int solver, model, integrator, nx, nu, N, m, n_cons;
  int n_term, mT, pT, ls_rule, n_alphas, clddp_box;   // clddp_box: index of "ControlConstraint" box or -1
  int has_xref_traj, _pad;
  double dt;
  double mp[32];                 // model parameters (LTI: A | B | dt)
  cddp_hip_options opt;
  ConDev cons[kMaxCons];
  TermDev terms[kMaxTerms];
  int off_Qdt, off_Rdt, off_Qf, off_xref;
  double alphas[CDDP_HIP_MAX_ALPHAS];
  double pool[kPool];
};

// All device buffers of one handle.  Trajectory-like arrays are wave-tiled "(N x batch) stacks":
// element e of step t of trajectory b lives at (((t*NB + b/64)*E + e)*64 + b%64), NB = Bp/64.  The E elements
// of one (step, 64-trajectory tile) record are contiguous (E x 512 B): a wavefront reads its whole step record
// from ONE scalar base address with compile-time immediate offsets e*512 (no per-element address VGPRs), and
// the record sits in one or two DRAM / TLB pages instead of E pages Bp*8 bytes apart.
constexpr int kLS = 64;   // lane stride of the wave-tiled stacks (doubles)

constexpr int kSinkDoubles = 1024 * 64;   // 512 KB

struct DevBuf {
  int B, Bp, N, n_slots, n_alphas, hist_batch, hist_cap, NB;   // NB = Bp / 64 wave tiles
  int fail_costate_mask;                   // TEST HOOK (0 in production; environment variable CDDP_HIP_TEST_FAIL_COSTATE at create): the costate trial of
                                           // alpha index i counts as non-finite when bit i is set -- the trial "passed every other test, costate not finite"
                                           // (flag 2 of k_costate), i.e. the device's form of a forward pass the reference discards (cddp_solver_base.cpp:
                                           // 280-296, ipddp_solver.cpp:1613-1616); forces the candidate walk of k_update (tests/test_gpu_parity_r2.py)
  const ProblemDev *P;
  const double *xref_traj;                 // [(N+1)][nx] shared by the batch, or null
  // iterate + line-search trial slots: [n_slots] planes
  double *X, *U, *S, *Y, *G, *Lam;
  size_t planeX, planeU, planeM;           // plane strides in doubles
  // derivative / gain / value stacks
  double *A, *Bm, *K, *k, *Vx, *Vxx, *ks, *ky, *Ks, *Ky;
  // terminal-constraint state [mT or pT][Bp]
  double *ST, *YT, *GT, *dST, *dYT, *LamT, *dLamT;
  double *STt, *YTt, *GTt, *LamTt;        // trial copies [n_alphas][kMTMax|kPTMax][Bp]
  double *te_k, *te_p;                     // terminal-equality LQR variants: [(pT+1)][N][nu][Bp], [(pT+1)][N+1][nx][Bp]
                                           // (cooperative sweep, kernels_te.hpp: [N][Bp][nu][16], [N+1][Bp][nx][16])
  double *te_cst;                          // [N][REC][Bp] LQ-model terms of the terminal-equality sweep (k_te_condense), or null
  int *te_cnt;                             // [Bp] k_te_post: steps finished per trajectory (-1: sweep failed)
  // per-trajectory scalars [Bp]
  double *cost, *merit, *inf_pr, *inf_du, *inf_comp, *step_norm, *alpha_pr, *alpha_du, *reg, *mu;
  double *dV0, *dV1, *phi, *theta, *filter_theta, *apr_max, *adu_max;
  double *filt;                            // [2*kFilterCap][Bp]: merit then violation
  int *filt_n, *iter, *status, *phase, *cur, *n_bwd, *n_fwd, *bwd_ok;
  // trial records [n_alphas][Bp]
  double *t_cost, *t_merit, *t_theta, *t_inf_pr, *t_inf_comp, *t_apr, *t_adu;
  double *t_ysmin, *t_ysmax;               // extreme y*s products of the trial (complementarity residual under a new mu)
  int ddp;                                 // options.use_ilqr == 0: second-order dynamics terms (one-lane sweeps)
  double *sink;                            // [kSinkDoubles] write-only scratch: lanes without a real destination store here, so stores stay unconditional
  int *t_success;
  int *t_steps;                            // [n_alphas][Bp] rollout steps the trial completed before it was abandoned (N = ran through)
  int *n_fwd_steps;                        // [Bp] sum of t_steps over the trials the line-search rule walked (roofline accounting)
  double *cst;                             // [N][CST][Bp] V-independent condensed stage terms written by K1b (k_condense)
  double *ys;                              // [N][m][Bp] Y S^-1 ratios of the last sweep (K3 -> rollout consumer)
  double *Kt;                              // [N][nu nx + nu] sub-tile-minor copy of K | k: what the G = 16 sweeps' own linear rollouts re-read (kernels.hpp::G4); NULL for nx <= 8
  double *dX;                              // [N][nx][Bp] linear-policy rollout of the last sweep (read by K3 k_post)
  double *ev;                              // [n_alphas][N][2*NSEG][Bp]: per-step log-barrier / |g+s| terms parked by K4
  // history [hist_batch][hist_cap][9] (row-major) + counts
  double *hist;
  int *hist_n;
  int *n_active;                           // device counter of trajectories still running
  int ev_valid;                            // set per K5 launch: the trials' parked barrier terms (ev, t_ysmin/max) were written by the two-role rollout
  int *win_hist;                           // [n_alphas + 1] accepted-alpha histogram of the solve so far (host picks the ladder shape)
  unsigned long long *launched;            // rollouts actually executed (speculative alphas included)
  int *cand;                               // [Bp] best-merit rule: the trial whose costate K4b evaluates (k_pick_candidate), -1 = none
  int t4;                                  // 1: A / B / cst / te_cst stacks in the sub-tile-minor layout of the G = 16 cooperative sweeps (kernels.hpp::GT)
  int lg;                                  // 1: the handle runs LogDDP (kernels_logddp.hpp); ev = [n_slots][N][NSEG] parked barrier sums per trial slot
  // MSIPDDP resident (kernels_msipddp.hpp)
  int ms;                                  // 1: the handle runs MSIPDDP
  int ms_fresh;                            // set per init launch: first initialize of the handle (the per-step factor cache starts invalid)
  int filt_cap;                            // rows per half of d.filt when ms (max_iterations + 2: MSIPDDP never bounds its filter between barrier updates)
  double *F;                               // [n_slots] planes (stride planeX): dynamics values f(x_t, u_t) of every iterate / trial, rows 0 .. N-1
  double *kl;                              // [N][nx][Bp] costate feed-forward gains k_lambda of the last sweep (K_lambda = V_xx(t+1) is d.Vxx)
  double *fac;                             // [N][nu nu + nu + 1][Bp] unconstrained branch: cached LDLT of Q_uu per step (matrix | transpositions | valid)
  int ladder_sorted;                       // the line-search ladder is strictly decreasing (what cddp_hip_build_alphas makes): the MSIPDDP rollout's dual step search probes only the two ends of the accepted interval
  double *ms_res;                          // [3][Bp] of the CURRENT iterate (mu-independent, kept for resetBarrierFilter): max |g + s|, max |F_t - x_{t+1}|, the violation sum
  int xcd_map;                             // cooperative sweeps: groups of one 64-trajectory tile on one XCD (kernels_coop.hpp::coop_group); CDDP_HIP_XCD_MAP=0 turns it off
};

}  // namespace cddp_dev

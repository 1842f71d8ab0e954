/*
 * dqmc.h -- C ABI of libdqmc_hip.so, the MI355X (gfx950) local-energy / MCMC evaluator.
 *
 * The reference (deepqmc/deepqmc) has no FFI boundary: its hot path is a set of Python
 * callables.  Each entry point below replaces one of them; the
 * reference-side binding a maintainer would add is the ctypes stub in INTEGRATION.md.
 *
 *   dqmc_wf_eval        <- ansatz.apply(params, phys_conf) -> Psi(sign, log)
 *                          reference src/deepqmc/types.py:107-150,
 *                          wf/nn_wave_function.py:127-173 (vmapped over walkers at
 *                          sampling/electron_samplers.py:76-81)
 *   dqmc_local_energy   <- hamil.local_energy(ansatz)(rng, params, phys_conf)
 *                          hamil.py:156-184, vmapped over walkers at loss/energy.py:50-57
 *   dqmc_mcmc_steps     <- DecorrSampler.sample / MetropolisSampler.sample
 *                          sampling/electron_samplers.py:140-163,347-357
 *   dqmc_langevin_*     <- LangevinSampler._update / .sample, sampling/electron_samplers.py:176-232
 *   dqmc_exchange_step  <- OppositeSpinExchangeSampler.sample's exchange branch, electron_samplers.py:235-330
 *   dqmc_energy_stats   <- EnergyMonitor's mean/std/min/max, observable.py:474-479
 *                          (per-rank partial record; ranks merge them after one all-gather)
 *   dqmc_set_weights    <- a new `params` tree after an optimiser step
 *
 * Conventions: plain pointers and sizes only.  Unless marked "host", every pointer is a
 * DEVICE pointer.  Walker-major contiguous layouts: r[B][N][3], R[n_nuc][3].  `real` is
 * float (dtype 0) or double (dtype 1), fixed per context.  Every function returns 0 on
 * success or a negative DQMC_E_* code and never throws; dqmc_last_error() gives a message.
 * A context is bound to one device (every entry point makes it the calling thread's current device) and is
 * not thread-safe; use one context per GPU.
 * All work is enqueued on the stream given at creation (0 = the null stream); functions
 * that return host values synchronise that stream.
 */
#ifndef DQMC_H
#define DQMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQMC_OK 0
#define DQMC_E_ARG (-1)     /* invalid argument / malformed program */
#define DQMC_E_HIP (-2)     /* a HIP runtime call failed */
#define DQMC_E_NOMEM (-3)
#define DQMC_E_UNSUPPORTED (-4)

/* ---- the layer program ------------------------------------------------------------
 * The wave function is handed over as a flat list of ops acting on per-walker
 * activation buffers.  Activation buffer b has `rows` rows per walker and `width`
 * features per row (width % 4 == 0, zero padded); in memory it is
 * real[B][rows][TP][width], TP = number of forward-Laplacian lanes: lane 0 = value,
 * lanes 1..3N = d/dr_c, lane 3N+1 = Laplacian, further lanes zero padding.  TP = 1 for
 * value-only evaluation (dqmc_wf_eval, MCMC); TP = round_up(3N+2, 16) for the local
 * energy.  Weight offsets index the `real` weight buffer, table offsets the int32 table. */

typedef struct dqmc_buf {
  int32_t rows;
  int32_t width;
} dqmc_buf;

enum dqmc_op_kind {
  /* i: [0]=dst buf [1]=log_rescale [2]=use_spin. Electron-nucleus input features
   * (gnn/electron_gnn.py:596-625): row i = [|d|,dx,dy,dz] per nucleus (+ spin). */
  DQMC_OP_FEAT_EN = 1,
  /* i: [0]=dst buf [1]=table offset of (recv,send) int pairs [2]=n_edge_rows
   * [3]=log_rescale.  Edge features of d = r_recv - r_send (gnn/graph.py:23-31, gnn/edge_features.py:21-123);
   * send >= 0: an electron, send < 0: nucleus -1 - send (the 'ne' edges, d = r_recv - R). */
  DQMC_OP_FEAT_EE = 2,
  /* Forward-Laplacian linear layer y = act(concat(pieces) W + b) (+ residual).
   * i: [0]=n_pieces, then per piece p (p<4) at [1+4p..]: src buf, r0, K (unpadded
   * width used), bcast (1: every dst row of a walker reads src row r0);
   * [17]=dst buf [18]=dst r0 [19]=dst col0 [20]=nrows (per walker) [21]=Nout
   * [22]=W offset [23]=bias offset or -1 [24]=act (0 none, 1 tanh, 2 silu, 3 shifted softplus hkext.py:13-19,
   * 4 the multiplicative backflow activation 1 + 2 tanh(x/4) of wf/nn_wave_function.py:17)
   * [25]=residual buf or -1 [26]=residual r0 [27]=normalize (1: out = (res + y)/sqrt(2),
   * 0: out = res + y; hkext.py:130-137).  W is row-major [sum_p pad4(K_p)][pad4(Nout)]. */
  DQMC_OP_LINEAR = 3,
  /* i: [0]=src buf [1]=dst buf (rows = 2) [2]=n_up.  Mean over up / down electrons
   * (gnn/update_features.py:86-102). */
  DQMC_OP_SPIN_MEAN = 4,
  /* i: [0]=edge buf (we) [1]=node buf (hx; rows = electrons, or nuclei for 'ne' edges) [2]=dst buf [3]=dst col0
   * [4]=table offset of int [N][S][2] (edge row, sender; row -1 = none; sender < 0 = nucleus -1 - sender = row of hx)
   * [5]=S [6]=width.
   * out[i] = sum_s we[row(i,s)] * hx[send(i,s)] (gnn/graph.py:226-335). */
  DQMC_OP_CONV = 5,
  /* i: as CONV with [1]=divisor instead of a node buf.  out[i] = sum_s e[row(i,s)] / divisor
   * (EdgeSum with normalize, gnn/update_features.py:109-159). */
  DQMC_OP_EDGE_SUM = 6,
  /* i: [0]=src buf [1]=dst buf (rows = 1).  Sum over all rows (Jastrow sum_first,
   * wf/omni.py:35-40). */
  DQMC_OP_ROW_SUM = 7,
  /* i: [0]=backflow buf ([N][pad4(K*N)], column k*N+mu) [1]=dst buf (rows=K,
   * width=pad4(N*N), column i*N+mu) [2..5]=weight offsets of pi_up, pi_down, zetas_up,
   * zetas_down ([K*N][n_nuc*n_env]) [6]=n_env envelopes per nucleus (0 = 1).  Slater matrix
   * entries A = envelope * backflow (wf/env.py:57-75, wf/nn_wave_function.py:135-147;
   * n_env = 3 with pi = 1: the SimplifiedNucleusDependentEnvelopes of env.py:110-226). */
  DQMC_OP_ORBITALS = 8,
  /* i: [0]=orbital buf.  sign/log|det| of the K matrices and their forward-Laplacian
   * lanes, kept in double inside the context (wf/nn_wave_function.py:36-39).  At most 44 electrons (the
   * largest LDS-resident Gauss-Jordan instance); programs with more are rejected at creation (DQMC_E_ARG). */
  DQMC_OP_SLOGDET = 9,
  /* i: [0]=jastrow buf or -1 [1]=conf_coeff weight offset or -1 (SumPool) [2]=cusp kind
   * (0 none,1 deepqmc,2 psiformer) [3]=weight offset of {same_alpha, anti_alpha}.
   * f: [0]=same_scale [1]=anti_scale.  CI sum + cusps + Jastrow
   * (wf/nn_wave_function.py:152-171); in Laplacian mode also the potentials and E_loc
   * (hamil.py:160-182, physics.py:105-133). */
  DQMC_OP_FINAL = 10,
  /* Multi-head self attention over the electrons of a walker, forward-Laplacian form
   * (hk.MultiHeadAttention called at gnn/update_features.py:273-278).
   * i: [0]=q buf [1]=k buf [2]=v buf [3]=dst buf [4]=heads [5]=head_dim
   * [6]=n_const extra key/value rows that do not depend on the electrons (the nuclear tokens of
   * CombinedNodeAttention with elec_to_nuc = false, update_features.py:385-451; they come first
   * in the key order) [7],[8]=weight offsets of their keys / values ([n_const][heads*head_dim]). */
  DQMC_OP_ATTENTION = 11,
  /* i: [0]=dst buf [1]=weight offset of real[rows][width].  Rows that do not depend on the electron positions
   * (hk.Embed electron / nuclear embeddings, gnn/electron_gnn.py:497-503,596-625): value lane = the table,
   * derivative lanes zero. */
  DQMC_OP_CONST = 12
};

typedef struct dqmc_op {
  int32_t kind;
  int32_t i[28];
  float f[4];
} dqmc_op;

typedef struct dqmc_system {
  int32_t n_up, n_down, n_nuc, n_det;
  int32_t dtype;        /* 0 = float32, 1 = float64 */
  int32_t reserved;
  double norm_eps;      /* eps under the safe norm, utils.py:79-85 (finfo(dtype).eps) */
  double e_nuc;         /* unused (kept for layout): the nuclear repulsion is recomputed from the R of every call,
                           physics.py:112-116 */
} dqmc_system;

typedef struct dqmc_ctx dqmc_ctx;

/* Create a context on `device`.  charges: host double[n_nuc].  weights: host double[]
 * (converted to `real` on upload).  itable: host int32[].  stream: a hipStream_t (or 0). */
int dqmc_create(dqmc_ctx** out, int device, void* stream, const dqmc_system* sys,
                const double* charges_host, const dqmc_buf* bufs_host, int n_bufs,
                const dqmc_op* ops_host, int n_ops, const double* weights_host,
                size_t n_weights, const int32_t* itable_host, size_t n_itable);
void dqmc_destroy(dqmc_ctx* ctx);
const char* dqmc_last_error(void);

/* Replace the weight buffer (same length as at creation).  host double[]. */
int dqmc_set_weights(dqmc_ctx* ctx, const double* weights_host, size_t n_weights);

/* psi for B walkers.  r: real[B][N][3], R: real[n_nuc][3]; out logpsi: real[B],
 * sign: int32[B] (values -1, 0, +1; bit-exact item of the parity contract). */
int dqmc_wf_eval(dqmc_ctx* ctx, const void* r, const void* R, int B, void* logpsi, int32_t* sign);

/* Local energy for B walkers.  e_loc: real[B]; stats: real[6][B] in the order V_el, E_kin,
 * V_loc, V_nl, lap, quantum_force (hamil.py:173-180), may be NULL; grad: real[B][3N]
 * (d log|psi| / dr, the quantum force), may be NULL; logpsi/sign may be NULL. */
int dqmc_local_energy(dqmc_ctx* ctx, const void* r, const void* R, int B, void* e_loc,
                      void* stats, void* grad, void* logpsi, int32_t* sign);

/* sign, log|psi| and grad log|psi| (real[B][3N]) from the forward-Laplacian pass alone -- no potentials, no
 * non-local ECP quadrature: what jax.value_and_grad(psi) hands the reference's LangevinSampler
 * (sampling/electron_samplers.py:193-201).  logpsi / sign may be NULL. */
int dqmc_psi_grad(dqmc_ctx* ctx, const void* r, const void* R, int B, void* logpsi, int32_t* sign, void* grad);

/* Gaussian-type effective core potential (reference ecp/gaussian_type_ecp.py:32-93 table layout,
 * :127-159 local part, :161-255 non-local part; replaces GaussianTypeECP.__init__'s pyscf lookup by
 * caller-supplied tables).  The `charges_host` given to dqmc_create must then be the valence charges
 * (ns_valence) and dqmc_system.e_nuc the repulsion of those.
 *   loc_host: double[n_nuc][3][2][n_terms_loc] -- terms r^-1, r^0, r^1; [.,term,0,.] exponents alpha,
 *             [.,term,1,.] coefficients beta; zero padded (all-zero = no ECP on that nucleus); may be NULL.
 *   nl_host:  double[n_nuc][n_l][2][n_terms_nl] -- channel l: V_l(r) = sum_k beta_lk exp(-alpha_lk r^2);
 *             nuclei whose block is all zero have no non-local part; may be NULL.
 * Afterwards dqmc_local_energy adds the local terms to V_loc and V_nl (12-point icosahedron quadrature,
 * 12 N n_ecp value-only psi evaluations per walker) to E_loc and stats[3]. */
int dqmc_set_ecp(dqmc_ctx* ctx, int n_terms_loc, const double* loc_host, int n_l, int n_terms_nl,
                 const double* nl_host);
/* Random rotation of the quadrature about the electron-nucleus axis for the following
 * dqmc_local_energy calls (ecp_utils.py:55: uniform in [0, pi/5) per (nucleus, electron); the reference
 * keys it by fold_in(fold_in(rng, j), i)).  phi: device real[B][n_ecp_nl][N] angles, or NULL to draw them
 * from Philox4x32-10 keyed by (seed, nucleus, walker, electron). */
int dqmc_ecp_rotation(dqmc_ctx* ctx, uint64_t seed, const void* phi);

/* Pseudo-Hamiltonian (reference ecp/pseudo_hamiltonian.py: PseudoHamiltonian.local_potential :173-190,
 * compute_coefficients_of_differential_operators :192-234, kinetic_term :236-278; replaces load_PH_functions'
 * XML lookup :73-112 by caller-supplied tables).  `charges_host` of dqmc_create must be the valence charges.
 *   rv_loc_host, rv_l2_host: double[n_nuc][n_grid] -- r*V_loc(r) (+ Z_eff) and r*V_L2(r) on the regular grid
 *             linspace(0, r_max, n_grid) (what parse_xml :32-70 returns; linear interpolation, zero beyond r_max);
 *   mask_host: int32[n_nuc], non-zero = this nucleus carries a PH (rows of the others are ignored).
 * An all-zero mask or n_grid == 0 switches the PH off.  Afterwards dqmc_local_energy evaluates
 *   E_kin = sum_i b(r_i).grad_i log|psi| - sum_i tr(A(r_i) Hess_ii) - |Q^T grad|^2,   A = Q Q^T (3x3 per electron),
 * by seeding the forward-Laplacian lanes with the columns of the Cholesky factors Q (the reference's coordinate
 * change r = Q v with Q held fixed), adds sum rV_loc(rho)/rho to V_loc, and reports in stats[4], stats[5] and
 * `grad` the transformed Laplacian, |grad_v|^2 and grad_v = Q^T grad_r (what the reference's kinetic_term returns).
 * dqmc_psi_grad and the Langevin entry points keep the plain gradient.  Not combinable with a non-local dqmc_set_ecp. */
int dqmc_set_pseudo_hamiltonian(dqmc_ctx* ctx, int n_grid, double r_max, const double* rv_loc_host,
                                const double* rv_l2_host, const int32_t* mask_host);

/* n_sub Metropolis sub-steps, in place on the sampler state
 * (sampling/electron_samplers.py:102-138,347-357).
 *   r real[B][N][3], logpsi real[B], sign int32[B], age int32[B], tau real[1] (device).
 *   max_age < 0: off.  target_acceptance <= 0: tau not adapted.
 *   noise: real[n_sub][B][N][3] standard normals and unif: real[n_sub][B] in [0,1), or
 *   both NULL to draw them on the device from Philox4x32-10 keyed by (seed, sub-step).
 *   accept_out: uint8[n_sub][B] accept decisions or NULL.
 *   stats7_host: host double[7] = acceptance, tau, age mean, age max, log|psi| mean,
 *   log|psi| std, mean e-e distance of the LAST sub-step (electron_samplers.py:154-163),
 *   or NULL (then the call does not synchronise). */
int dqmc_mcmc_steps(dqmc_ctx* ctx, void* r, void* logpsi, int32_t* sign, int32_t* age,
                    void* tau, const void* R, int B, int n_sub, int max_age,
                    double target_acceptance, uint64_t seed, const void* noise,
                    const void* unif, uint8_t* accept_out, double* stats7_host);

/* Metropolis-adjusted Langevin sampler (reference sampling/electron_samplers.py:176-232; drift cleaning
 * sampling/sampling_utils.py:72-101).  The sampler state additionally carries the cleaned drift force real[B][N][3].
 * mol_charges_host: host double[n_nuc], the FULL nuclear charges (mol.charges; the crossover parameter of the
 * drift cleaning uses them even under an ECP).
 *   dqmc_langevin_update <- LangevinSampler._update: log|psi|, sign and the drift of the CURRENT positions, cleaned
 *                           with the step size `tau` (device real[1]) of the previous iteration.
 *   dqmc_langevin_steps  <- n_sub x LangevinSampler.sample, in place: r' = r + tau F + sqrt(tau) xi, forward-Laplacian
 *                           pass for psi' and its gradient (no ECP quadrature), Green's-function ratio in the acceptance,
 *                           age override and step-size adaptation as dqmc_mcmc_steps.  noise / unif / accept_out /
 *                           stats7_host as there. */
int dqmc_langevin_update(dqmc_ctx* ctx, const void* r, const void* R, const double* mol_charges_host, int B, const void* tau,
                         void* logpsi, int32_t* sign, void* force);
int dqmc_langevin_steps(dqmc_ctx* ctx, void* r, void* logpsi, int32_t* sign, int32_t* age, void* force, void* tau, const void* R,
                        const double* mol_charges_host, int B, int n_sub, int max_age, double target_acceptance, uint64_t seed,
                        const void* noise, const void* unif, uint8_t* accept_out, double* stats7_host);
/* One opposite-spin exchange step (sampling/electron_samplers.py:235-330): per walker the positions of spin-up electron
 * up_idx[b] and spin-down electron n_up + down_idx[b] (device int32[B] each) are swapped, psi re-evaluated (value path),
 * accepted with 2 (log|psi'| - log|psi|) > log unif[b]; no age override, step size untouched (:312-313).  stats7_host
 * (may be NULL): the sampler statistics after the step, [0] = acceptance of this step. */
int dqmc_exchange_step(dqmc_ctx* ctx, void* r, void* logpsi, int32_t* sign, int32_t* age, const void* tau, const void* R, int B,
                       const int32_t* up_idx, const int32_t* down_idx, const void* unif, uint8_t* accept_out, double* stats7_host);

/* Per-rank partial record of the energy reduction: host double[7] =
 * {n, sum_w, sum_wE, sum_E, M2 (sum of squared deviations from this rank's mean), min,
 * max}.  w may be NULL (all ones).  Ranks all-gather the 56-byte records (RCCL) and
 * merge them with dqmc_merge_energy_stats. */
int dqmc_energy_stats(dqmc_ctx* ctx, const void* e_loc, const void* w, int B, double* out7_host);
/* The whole cross-GPU energy reduction in one call: this rank's record, ONE ncclAllGather of the 56-byte records over
 * the caller's RCCL communicator (`rccl_comm` = an ncclComm_t of `n_ranks` ranks; the call is enqueued on the context's
 * stream), Chan merge -> host double[5] as dqmc_merge_energy_stats.  Replaces the reference's five pmean / pmin / pmax
 * calls (parallel.py:175-225).  librccl.so is loaded at the first call; DQMC_E_UNSUPPORTED if it is absent. */
int dqmc_energy_stats_allgather(dqmc_ctx* ctx, void* rccl_comm, int n_ranks, const void* e_loc, const void* w, int B,
                                double* out5_host);
/* Chan merge of n_ranks records -> host double[5] = mean, std (population), min, max,
 * weighted mean.  Pure host arithmetic. */
int dqmc_merge_energy_stats(const double* records_host, int n_ranks, double* out5_host);

/* Debug / test access: copy activation buffer `buf` of the last evaluation (layout
 * real[B][rows][TP][width]) to host as double[].  n must equal B*rows*TP*width.
 * buf = -1: log|det| lanes double[B][K][TP]; buf = -2: det signs as double[B][K]; buf = -4: the psi-weighted
 * conditioning record of the Slater matrices, double[B] (Laplacian-mode evaluations). */
int dqmc_debug_read(dqmc_ctx* ctx, int buf, double* out_host, size_t n);
/* Lanes (TP) used by the last evaluation. */
int dqmc_debug_lanes(dqmc_ctx* ctx);
/* Walkers the last dqmc_local_energy / dqmc_psi_grad call re-evaluated in float64 (float32 contexts: the local
 * energy near a node of psi is a difference of huge numbers, E_kin = -(lap + |grad|^2)/2 with both terms ~ 1/psi^2;
 * walkers whose error predictor score = (|lap| + |grad|^2) / max(1, |E_loc|) x max(1, sum_k |p_k| kappa_k) -- node
 * cancellation times the conditioning record of the Slater matrices that carry psi -- exceeds "refine_thresh", or whose
 * E_loc is not finite, are run again by a float64 twin of the context and their E_loc / stats / grad / log|psi| / sign
 * replaced.  The threshold calibrates itself: on the first and then every "refine_probe"-th call a strided sample of
 * <= "refine_sample" (256) further walkers is evaluated in float64 too.  Measured on the MI355X, the float32 error of a
 * walker is m x score x (an exponentially distributed factor): the sample gives the scale m, and the threshold is the largest
 * one for which the expected share of float32-kept walkers beyond the tolerance "refine_target_e7" x 1e-7 (default 1e-5
 * relative) -- the mean of exp(-tol / (m score_i)) over the kept walkers of the probed batch -- stays below
 * "refine_miss_e9" x 1e-9 (default 1e-8).  A batch with more than
 * "refine_direct_pct" (60 %) of its walkers above the threshold is evaluated in float64 whole, and so are the next
 * "refine_direct_calls" (15) calls -- a stay that doubles with every float32 look that confirms the mode, at most
 * "refine_direct_backoff" times --;
 * the context returns to the mixed mode only when a float32 pass then finds fewer than "refine_direct_exit_pct" (45 %)
 * above it -- hysteresis: one calibration draw near a single line used to flip the mode from run to run). */
int dqmc_last_refined(dqmc_ctx* ctx);
/* State of the refinement after the last local-energy call: out4 = {mode ("refine": 0 / 1 / 2; 0 in a float64
 * context), current score threshold, measured float32 error per unit of score (0 before the first probe), calls
 * that will still go to the direct float64 pass because most walkers were flagged}. */
int dqmc_refine_info(dqmc_ctx* ctx, double* out4);
/* Walker chunks of the last dqmc_local_energy call: out2 = {chunks of the context's own forward-Laplacian pass (0: it did
 * not run one, e.g. the whole batch went to float64), chunks of the largest pass of its float64 twin}.  An evaluation
 * whose activations exceed "ws_budget_mb" is split into chunks inside the library (the reference has no such limit to
 * replace: its vmap over the electron batch, loss/energy.py:50-57, simply needs the memory). */
int dqmc_last_chunks(dqmc_ctx* ctx, int* out2);
/* Running counts since the context was created: out4 = {dqmc_local_energy / dqmc_psi_grad calls, of them calls evaluated in
 * float64 whole ("direct" mode or "refine" 2), calibration probe calls, walkers re-evaluated in float64 (sum of
 * dqmc_last_refined)} -- what a caller needs to say which mode its steps actually ran in (bench.py: config.refine_engaged). */
int dqmc_refine_counters(dqmc_ctx* ctx, int64_t* out4);
/* Which kernel dqmc_mcmc_steps launches per Metropolis sub-step (sampling/electron_samplers.py:102-138): writes the name of the
 * PLAN-SPECIALISED kernel bound to this context's program (deepqmc_amd/csrc/gen/, generated by deepqmc_amd/codegen from the
 * layer program and matched by a hash of its structure; option "fused_spec" 0 switches it off) into name_out[n] and returns
 * 1; returns 0 with an empty string when the program runs on the descriptor-driven kernel or the one-launch-per-op path. */
int dqmc_substep_kernel(dqmc_ctx* ctx, char* name_out, size_t n);
/* The error-predictor scores (see dqmc_last_refined) of the first n walkers of the context's last float32 forward-Laplacian
 * pass, copied to the host: together with the threshold of dqmc_refine_info they say which walkers kept their float32
 * result.  DQMC_E_UNSUPPORTED if the last call ran no such pass (float64 context, "refine" 0 / 2, direct mode). */
int dqmc_refine_scores(dqmc_ctx* ctx, double* out, int n);
/* Non-local ECP term of a float32 context (replaces nonloc_potential, ecp/gaussian_type_ecp.py:161-255, for a whole
 * batch): with "refine" 1 the 12-point quadrature of every (walker, ECP nucleus, electron) triple runs in the precision its
 * weight w = max_l (2l+1)|V_l(|r_i - R_a|)| calls for -- float64 psi ratios above "ecp_heavy_e6" (default 10000 = 1e-2 Ha),
 * float32 below, none below "ecp_skip_e12" (default 100 = 1e-10 Ha: the contribution is below that times the mean ratio).
 * The bound is per walker: psi(r) of every walker is evaluated by both value paths first, and where the float32
 * log|psi(r)| is off by more than "ecp_dlog_floor_e6" (default 30 = 3e-5: half the median float32 error of these networks) the
 * walker's float64 bound tightens in proportion -- near a node psi(r), the denominator of all its ratios, is what float32
 * cannot resolve; a sign mismatch sends all its kept pairs to float64 (0: weights alone decide).  Without a float64 twin
 * (DQMC_E_UNSUPPORTED for this program) every kept pair runs in float32.
 * "ecp_mixed" 0 restores whole-walker float64 quadrature for flagged walkers only.  out3 = triples of the last call
 * {float32, float64, dropped}. */
int dqmc_ecp_counts(dqmc_ctx* ctx, int64_t* out3);
/* Tuning / debugging switches.  "fused" (default 1): evaluate value-only psi (dqmc_wf_eval,
 * MCMC) with the single LDS-resident kernel instead of one launch per op where that is the faster path
 * (N <= 4, or fewer than 1024 walkers; 2 = always, 0 = never, which also keeps every
 * activation buffer readable by dqmc_debug_read); "fused_substep" (1): fold propose / determinants /
 * accept of a Metropolis sub-step into that kernel when N <= 4; "fused_wt": walkers per workgroup tile (0 = automatic);
 * "fused_sched" (3: list scheduling under an LDS budget, 2: as late as possible, 1: full dependency
 * levels, 0: program order); "fused_occ": register budget as workgroups per CU (0 = from the LDS size);
 * "fused_lds_kb", "fused_sched_kb": LDS budgets; "fused_dbg": clock stamps readable through
 * dqmc_debug_read(buf = -3); "fused_print": plan summary on stderr; "ecp_max_cfg": quadrature walkers
 * per value-mode batch of the non-local ECP term; "ws_budget_mb": activation workspace per evaluation
 * (larger batches are split into walker chunks); "lane_compact" (1): 8-lane storage of the edge stream;
 * "attention_mfma", "slogdet_mfma" (1: MFMA kernels where profitable, 2: wherever supported, 0: never);
 * "fused_lean" (1): straight-line unit body for layers with one input piece and K <= 32; "fused_wg_per_cu" (4): LDS share a
 * tile is planned for;
 * "fused_prio" (1): the co-resident tiles of a CU take turns at the highest issue priority, level by level, instead of the
 * hardware's oldest-wave-first order (2: unit by unit, 0: off); "fused_bf" (float32 contexts; 1: the float32 layers of the
 * fused kernel whose pieces are whole octets wide run on the bf16 matrix pipe -- operands split into three bf16 pieces,
 * six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block, float32-class results; 2: only the layers deep enough to pay by
 * the stricter rule; 0: v_mfma_f32_16x16x4_f32 everywhere); "linear_bf" (2): the same split with nine
 * products for the per-op linear kernel -- 2: the Laplacian tiles of the 48-lane groups (11-15 electrons), where it is
 * faster, and of the 16-lane groups (up to 4 electrons), where it costs the same and rounds less often (fewer walkers reach the
 * float64 pass); 1: every layer deep enough (measured slower elsewhere); 0: never; "dual_stream" (1): edge stream of the Laplacian pass on a
 * companion HIP stream; options prefixed "twin." go to the float64 refinement twin;
 * "pass_graph" (1): a forward-Laplacian pass that fits one workspace chunk is captured into a hipGraph on its second call
 * with the same buffers and batch size and replayed afterwards (one hipGraphLaunch instead of ~40 launches and their
 * cross-stream events; contexts without a non-local ECP / pseudo-Hamiltonian, timing off); the float64 twin's batch is
 * rounded up to a multiple of 64 walkers so that a few graphs serve every call; a context whose captures outnumber its
 * replays (callers passing fresh buffers every time) returns to eager launches; 0: always eager;
 * "refine" (float32 contexts; 1: float64 re-evaluation of ill-conditioned walkers, 2: the whole local-energy pass in
 * float64 while sampling stays float32, 0: off), "refine_thresh" (200 until the first probe): score above which mode 1
 * refines a walker, "refine_probe" (32): calls between self-calibration probes (0: keep refine_thresh as set), "refine_sample"
 * (256; 64 until round 4): walkers of the calibration sample, "refine_direct_pct" (60) / "refine_direct_exit_pct" (45): share
 * of a batch above the threshold at which the context enters / leaves the whole-batch float64 mode, "refine_direct_calls" (15): calls it
 * stays there before a float32 pass looks again, "refine_direct_backoff" (4): every look that confirms the mode doubles the stay, at
 * most this many times (15, 30, ... 240 calls),
 * "refine_target_e7" (100): relative tolerance the float32-kept walkers are to meet, in units of 1e-7; "refine_miss_e9" (10):
 * accepted share of kept walkers beyond it, in units of 1e-9 (1e-8 costs LiH / PauliNet ~16 % and N2 / FermiNet ~12 % of their
 * walkers in float64 and leaves the largest float32 error of a 82 k-evaluation trajectory at 5e-6; 100 = 1e-7: ~13 % / ~9 %,
 * largest error 7.6e-6 / 1.01e-5; the 90th-percentile rule of rounds 3-4 refined 5 % and left ~30 of 82 k evaluations beyond
 * the tolerance).
 * "linear_bf", "linear_bkx" act on the calling context only;
 * "linear_f64_split" (float64 contexts, 1): layers over 96- / 128-lane groups with a PAIR of waves per group (two waves per
 * SIMD instead of one); "attention_split" (float64 contexts, 1): eight-wave attention kernel, a pair of waves per query row
 * block; "attention_ncb" (-1: kernel instance per number of key tiles in float32, four-tile instance in float64);
 * "mlp_fuse" (1): row-wise two-layer MLPs in one launch where the chained kernel has an instance; "tail_f64" (float32 contexts, 0):
 * the ops from the backflow head on -- the LINEAR ops that write what ORBITALS reads, ORBITALS, SLOGDET, FINAL -- of an unchunked
 * forward-Laplacian pass run on the float64 twin for every walker, reading the float32 head's activations in place (the float32 error
 * of E_loc x 0.81 LiH / x 0.63 N2 with the single accumulator chain of the linear kernels, x 0.87 / x 0.91 since their fresh per-chunk
 * accumulators -- less than the tail costs, hence off by default); "no_twin" (0; set before the first local-energy call): never create the float64 twin, i.e. plain
 * float32 everywhere, as for a program without a float64 kernel set; "ecp_mixed" (1),
 * "ecp_heavy_e6" (10000), "ecp_skip_e12" (100), "ecp_dlog_floor_e6" (30): mixed-precision non-local ECP quadrature
 * (dqmc_ecp_counts); passes of fewer than 64 walkers are never captured into graphs.
 * Removed in round 5 (each had been measured slower than the default or neutral on the MI355X and was never on; DESIGN.md
 * section 4 keeps the numbers): "refine_defer" + dqmc_refine_finish, "refine_ahead", "mlp_dual", "fused_chain",
 * "fused_stagger", "linear_bf" 3 and its host-split weight planes, "linear_bkx_big", "linear_bkx_val", "linear_f64_nr1".
 * Unknown names return DQMC_E_ARG. */
int dqmc_set_option(dqmc_ctx* ctx, const char* name, int value);

/* Per-kernel timing (HIP events on the context's stream).  enable != 0 starts recording
 * every launch; dqmc_timing_get returns total ms / launch count / algorithmic flops of the
 * kernel class `name` ("linear", "slogdet", ...) accumulated since the last reset. */
int dqmc_timing_enable(dqmc_ctx* ctx, int enable);
int dqmc_timing_reset(dqmc_ctx* ctx);
int dqmc_timing_get(dqmc_ctx* ctx, const char* name, double* ms, int64_t* launches, double* flops);
/* The flops the launches of kernel class `name` actually multiplied, beside dqmc_timing_get's ALGORITHMIC count (the dense
 * T = 3N + 2 lanes of SURVEY.md section 8d, which `roofline.achieved` is priced with): edge rows carry 8 pair-compact lanes,
 * per-walker (broadcast) pieces are multiplied once per walker, lanes are padded to TP and widths to multiples of 4.
 * executed / time / peak is the matrix-pipe utilisation; algorithmic / time / peak is the roofline fraction -- for
 * N2 / FermiNet the two differ by ~2x. */
int dqmc_timing_get_executed(dqmc_ctx* ctx, const char* name, double* flops_executed);
int dqmc_timing_names(dqmc_ctx* ctx, char* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* DQMC_H */

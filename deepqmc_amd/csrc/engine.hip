// engine.hip -- the C ABI of include/dqmc.h: context, layer-program executor, MCMC driver,
// per-kernel timing.  Host code only; every arithmetic step is one of the kernels in
// kernel_linear.hip / kernels_graph.hip / kernels_head.hip / kernels_mcmc.hip.
// This file: the context base, the state of Engine<real> (members, options' storage, captured-graph bookkeeping), and
// the extern "C" entry points.  The member functions live in engine_*.inl by concern -- program analyses
// (engine_program.inl), planner of the LDS-resident kernel (engine_fused_plan.inl), pass runner (engine_pass.inl),
// float64 refinement (engine_refine.inl), ECP / pseudo-Hamiltonian (engine_ecp.inl), samplers (engine_mcmc.inl) -- and
// are included inside the struct body: one translation unit, as before.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <algorithm>
#include <vector>

#include "../../include/dqmc.h"
#include "kernels.h"
#include "spec_device.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess)                                                                               \
      return fail(DQMC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ +   \
                                  ":" + std::to_string(__LINE__) + ")");                                \
  } while (0)

inline int pad4(int n) { return (n + 3) / 4 * 4; }

struct TimingRec {
  double ms = 0;
  int64_t launches = 0;
  double flops = 0;       // algorithmic: the dense-T count of SURVEY.md section 8d (what `roofline.achieved` is priced with)
  double executed = 0;    // what the launches actually multiply: 8 compact lanes on edge rows, per-walker pieces once per walker,
                          // lanes padded to TP, widths padded to 4 (matrix-pipe utilisation = this / time / peak)
};

}  // namespace

struct dqmc_ctx {
  virtual ~dqmc_ctx() { if (d_gather) (void)hipFree(d_gather); }
  virtual int set_weights(const double* w, size_t n) = 0;
  virtual int wf_eval(const void* r, const void* R, int B, void* logpsi, int32_t* sign) = 0;
  virtual int local_energy(const void* r, const void* R, int B, void* e_loc, void* stats, void* grad, void* logpsi,
                           int32_t* sign) = 0;
  virtual int psi_grad(const void* r, const void* R, int B, void* logpsi, int32_t* sign, void* grad) = 0;
  virtual int mcmc(void* r, void* logpsi, int32_t* sign, int32_t* age, void* tau, const void* R, int B, int n_sub,
                   int max_age, double target, uint64_t seed, const void* noise, const void* unif, uint8_t* accept_out,
                   double* stats7) = 0;
  virtual int langevin_update(const void* r, const void* R, const double* mol_charges, int B, const void* tau, void* logpsi,
                              int32_t* sign, void* force) = 0;
  virtual int langevin(void* r, void* logpsi, int32_t* sign, int32_t* age, void* force, void* tau, const void* R,
                       const double* mol_charges, int B, int n_sub, int max_age, double target, uint64_t seed, const void* noise,
                       const void* unif, uint8_t* accept_out, double* stats7) = 0;
  virtual int exchange(void* r, void* logpsi, int32_t* sign, int32_t* age, const void* tau, const void* R, int B,
                       const int32_t* up_idx, const int32_t* down_idx, const void* unif, uint8_t* accept_out, double* stats7) = 0;
  virtual int set_ecp(int n_t_loc, const double* loc, int n_l, int n_t_nl, const double* nl) = 0;
  virtual int ecp_rotation(uint64_t seed, const void* phi) = 0;
  virtual int set_ph(int n_grid, double r_max, const double* rv_loc, const double* rv_l2, const int32_t* mask) = 0;
  virtual int energy_stats(const void* e, const void* w, int B, double* out7) = 0;
  virtual int energy_stats_dev(const void* e, const void* w, int B, double** rec_dev) = 0;
  virtual int debug_read(int buf, double* out, size_t n) = 0;
  virtual int option(const char* name, int value) = 0;
  virtual int refine_scores(double* out, int n) { (void)out; (void)n; return DQMC_E_UNSUPPORTED; }
  virtual int substep_kernel(char* name_out, size_t n) { (void)n; name_out[0] = 0; return 0; }
  int64_t refine_counters[4] = {0, 0, 0, 0};   // local-energy / psi_grad calls; of them whole-batch float64; probe calls; walkers refined (sum)
  virtual dqmc_ctx* twin_ctx() { return nullptr; }
                                                         // (the float64 refinement twin of a float32 context, once it exists)
  int last_TP = 0;
  bool ph_skip = false;     // set on a float64 twin while it serves a plain-gradient call (no pseudo-Hamiltonian seeding)
  bool ecp_skip_nl = false; // ... and no non-local ECP quadrature
  int last_refined = 0;     // walkers re-evaluated in float64 by the last local-energy / psi_grad call
  double refine_info[4] = {0, 0, 0, 0};   // {mode, score threshold, measured error per unit of score, direct float64 calls left}
  int64_t ecp_last_counts[3] = {0, 0, 0};   // (nucleus, electron) pairs of the last mixed-precision quadrature: float32, float64, dropped
  int last_chunks[2] = {0, 0};   // walker chunks of the last Laplacian-mode evaluation: this context's own pass, its float64 twin's (max over its passes)
  int device = 0;           // every entry point makes this the calling thread's current device
  double* d_gather = nullptr;   // all-gathered energy records (dqmc_energy_stats_allgather)
  size_t gather_cap = 0;
  // timing
  bool timing = false;
  std::map<std::string, TimingRec> trec;
  struct Pending { std::string name; hipEvent_t a, b; double flops, executed; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> ev_pool;
  hipStream_t st = nullptr;

  hipEvent_t get_event() {
    if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
  }
  hipStream_t t_stream = nullptr;      // stream of the launch being timed (the context's, or its edge-stream companion)
  void t_begin(const char* name, double flops, hipStream_t s = nullptr, double executed = -1.0) {
    if (!timing) return;
    t_stream = s ? s : st;
    Pending p{name, get_event(), get_event(), flops, executed < 0 ? flops : executed};
    (void)hipEventRecord(p.a, t_stream);
    pending.push_back(p);
  }
  void t_end() {
    if (!timing) return;
    (void)hipEventRecord(pending.back().b, t_stream);
  }
  void t_collect() {
    if (pending.empty()) return;
    (void)hipStreamSynchronize(st);
    for (auto& p : pending) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, p.a, p.b);
      auto& r = trec[p.name];
      r.ms += ms; r.launches += 1; r.flops += p.flops; r.executed += p.executed;
      ev_pool.push_back(p.a); ev_pool.push_back(p.b);
    }
    pending.clear();
  }
};

namespace {

template <typename real>
struct Engine : dqmc_ctx {
  dqmc_system sys{};
  int N = 0;
  std::vector<dqmc_buf> bufs;
  std::vector<dqmc_op> ops;
  std::vector<double> charges_h;
  size_t n_weights = 0, n_itable = 0;
  std::vector<int32_t> h_itable;
  real* d_w = nullptr;
  int32_t* d_it = nullptr;
  double* d_charges = nullptr;
  // workspace
  char* d_ws = nullptr;
  size_t ws_bytes = 0;
  std::vector<size_t> buf_off;  // byte offsets for the current (B, TP)
  size_t off_logdet = 0, off_signk = 0, off_z = 0, off_cond = 0, off_kappa = 0;
  int split_bcast = 1;           // option "split_bcast": per-walker pieces of a linear layer multiplied once per walker
  int last_B = 0;
  // mcmc scratch
  char* d_mc = nullptr;
  size_t mc_bytes = 0;
  int32_t* d_nacc = nullptr;
  void* d_tau_ring = nullptr;   // real[2]: step sizes of the last two one-launch sub-steps
  double* d_acc = nullptr;     // [1] acceptance, then [7] stats, then [7] energy record
  std::vector<real> wtmp;
  // fused value-only plan
  int fused_enabled = 1;      // 0 off, 1 where it is the faster value path (fused_pays), 2 always
  int fused_n_ops = 0, fused_wt_req = 0, fused_dbg = 0, fused_sched_mode = 3, fused_occ_req = 0;
  size_t wpk_cap = 0;
  std::vector<int> f_order, f_level;   // fused schedule: op index and dependency level per slot
  dqmc_op* d_ops = nullptr;
  int32_t* d_wpk_off = nullptr;   // per scheduled op: {packed-weight offset, barrier-after flag}
  real* d_wpk = nullptr;
  long long* d_prof = nullptr;
  // pair-compact edge buffers (common.h: PAIR_LANES): which buffers carry 8 lanes in Laplacian mode, and the
  // (receiver, sender) of each of their rows (for the lane maps of debug_read)
  size_t ws_budget = (size_t)32 << 30;   // activation workspace per evaluation chunk (option "ws_budget_mb"); several contexts
                                         // (electronic states, float64 twins) share the 288 GB of one GPU
  bool lane_compact = true;
  int attention_mfma = 1;      // 1: where profitable (N > 16), 2: wherever supported, 0: never
  int attention_ncb = -1;      // 1: kernel instance for the exact number of key tiles, 0: the four-tile instance, -1: exact in float32 only (measured)
  int attention_split = -1;    // float64 contexts: eight-wave variant (a pair of waves per query row block) for Laplacian passes (1 / -1), 0: never
  int slogdet_mfma = 1;        // 1: where profitable (N > 16), 2: from N > 8 on, 0: never
  std::vector<char> compact;
  // two-layer row-wise MLPs run as ONE launch (kernel_linear.hip: CHAIN): mlp_child[k] = op index of the second layer
  // of the LINEAR op k, or -1; mlp_skip[k]: op k is such a second layer (executed with its parent)
  std::vector<int> mlp_child;
  std::vector<char> mlp_skip;
  int mlp_fuse = 1;
  int linear_bf = dqmc::LINEAR_BF_DEFAULT, linear_bkx = dqmc::LINEAR_BKX_DEFAULT, linear_f64_split = 1;   // LinArgs::cfg_*
  std::vector<std::vector<int>> pair_rs;   // per compact buffer: [2*row] = recv, [2*row+1] = send
  // descriptor-driven fused kernel (kernel_fused2.hip): the default when its plan exists
  int fused2_WT = 0, fused2_shift = 0, fused_sched_wt = 4, fused_substep = 1;
  size_t fused_sched_budget = 36 * 1024;   // bytes live per level the list scheduler (mode 3) aims for
  size_t fused2_lds_quarter = 160 * 1024 / 4;   // LDS per workgroup for 4 (option "fused_wg_per_cu": 5) workgroups per CU
  size_t fused2_lds = 0, fused2_lds_budget = 80 * 1024;
  std::vector<dqmc::FusedBuf> fbufs2_h;
  bool fused2_ma1 = false;       // every unit of the plan has ma == 1
  bool fbufs2_uploaded = false;
  // Laplacian pass on two HIP streams: the two-particle (edge) stream does not depend on the node stream (reference
  // gnn/electron_gnn.py:160-276: edges are updated from edges), its launches are HBM bound while the node layers are
  // MFMA bound, so they run on a companion stream and the node stream waits (event) only where a convolution or an
  // edge sum consumes an edge buffer.  Option "dual_stream" (1).
  int dual_stream = 1;
  // Laplacian pass of the small systems: its ~40 kernels are latency bound one by one (2-3 TB/s, 40 % of the MFMA peak at
  // best), and a layer's node-MLP + convolution of one edge type, that of the other, and the spin means + their product
  // are three INDEPENDENT branches that meet in the g layer.  Option "multi_stream" (1): those branches run on up to two
  // more HIP streams (op_sid, analyse_streams), ordered by events per produced buffer.
  int multi_stream = 1;
  std::vector<int> op_sid;             // stream slot per op: 0 main, 1 edge stream, 2 / 3 node branches
  hipStream_t st_extra[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> ms_events;   // event pool of one pass
  hipStream_t st2 = nullptr;
  std::vector<hipEvent_t> buf_ev;      // per buffer: last write on the companion stream (nullptr: none pending)
  hipEvent_t ev_fork = nullptr;
  int fused_stagger_div = 256;   // workgroups per dispatch wave (= CUs): workgroup b is the (b / 256)-th placed on its CU
  int fused_prio = 1;            // option "fused_prio": issue priority rotates among the tiles that share a CU (1: per level, 2: per unit, 0: off)
  int fused_lean = 1;            // option "fused_lean": lean unit body for small layers
  int fused_bf = 1;              // option "fused_bf": float32 layers of the value path on the bf16 matrix pipe (three-piece split, six products; float engines only)
  std::vector<char> op_bf;       // per scheduled op: packed in the bf16 plane layout
  std::vector<std::vector<dqmc::FDesc>> plan_lists;   // the four wave lists (kept for "fused_print")
  // plan-specialised sub-step kernel (csrc/gen/*.hip, spec_device.h): found by the hash of the program; its weight tape
  const dqmc::SpecKernel* spec_k = nullptr;
  uint32_t* d_tape = nullptr;
  int fused_spec = 1;            // option "fused_spec": 0 = always the descriptor-driven kernel
  dqmc::FDesc* d_descs = nullptr;
  int32_t* d_wave_begin = nullptr;
  dqmc::FusedBuf* d_fbufs2 = nullptr;
  // effective core potential (dqmc_set_ecp): local terms for k_final, non-local channels per nucleus
  double* d_ecp_loc = nullptr;
  double* d_ecp_nl = nullptr;
  int32_t* d_ecp_nuc = nullptr;
  int ecp_nt_loc = 0, ecp_n_nl = 0, ecp_L = 0, ecp_nt_nl = 0;
  uint64_t ecp_seed = 0;
  const void* ecp_phi = nullptr;
  const int32_t* ecp_idx = nullptr;   // twin serving a float32 context: global walker index per walker (EcpArgs::walker_idx)
  bool ecp_phi_f32 = false;
  std::vector<double> ecp_nl_h;       // host copy of the non-local tables for the float64 twin
  int ecp_nl_L_h = 0, ecp_nl_nt_h = 0;
  char* d_ecp = nullptr;        // quadrature walkers + their psi + psi of the walkers themselves
  size_t ecp_bytes = 0;
  // mixed-precision quadrature of a float32 context (ecp_mixed): "ecp_mixed" 1 (default) with "refine" 1; a triple whose
  // weight w = max_l (2l+1)|V_l| exceeds ecp_w_heavy ("ecp_heavy_e6", in 1e-6 Ha) gets float64 psi ratios, one below
  // ecp_w_skip ("ecp_skip_e12", in 1e-12 Ha) is dropped
  int ecp_mixed_on = 1;
  bool ecp_defer = false;       // set while lap_refined_core runs for a call whose quadrature follows in ecp_mixed
  double ecp_w_heavy = 1e-2, ecp_w_skip = 1e-10;
  // float32 error of log|psi(r)| up to which a walker counts as ordinary ("ecp_dlog_floor_e6", in 1e-6; 0: weights alone
  // decide, the round-4 rule): beyond it the walker's float64 bound tightens in proportion (kernels.h: EcpMixArgs::l32).
  // 3e-5 = half the median float32 error of log|psi| of these networks (6e-5).  Sweep on the MI355X (tools/ecp_sweep_b.py,
  // profiles/r05_ecp_threshold_sweep.txt), largest E_loc error of 256 set-B / 32 set-A walkers and time per call:
  // weights alone 1.2e-5 (one walker beyond the tolerance) / 1.7e-6; floor 1e-4: 9.8e-6 / 1.7e-6 at + 0.6 %; floor 3e-5:
  // 3.5e-6 / 5.9e-7 at + 3 % -- the same as tightening "ecp_heavy_e6" to 3e-3 at + 6 %.
  double ecp_dlog_floor = 3e-5;
  char* d_ecpm = nullptr;
  size_t ecpm_bytes = 0;
  size_t ecp_max_cfg = 1 << 16; // quadrature walkers per value-mode batch
  // float64 refinement (float32 build): walkers flagged by k_final as ill conditioned (near a node of psi the kinetic
  // energy is a difference of huge numbers and float32 round-off is amplified by the CI cancellation) are
  // re-evaluated by a float64 twin of this context and their results replace the float32 ones
  int refine = 1;
  // Flag rule: score > refine_thresh (kernels.h: FinalArgs).  The threshold is SELF-CALIBRATED: every refine_probe-th
  // local-energy call (and the first) a strided sample of <= refine_sample (256) walkers is evaluated by the float64 twin
  // as well.  What the MI355X shows (round 5: 82 k evaluations each along the bench trajectories of LiH / PauliNet and
  // N2 / FermiNet, 5 k of benzene / Psiformer; tools/calib_data.py, profiles/r05_calibration_model.txt): the float32 error
  // of a walker is  err = m x score x xi  with xi EXPONENTIALLY distributed -- every quantile of err / score is the same in
  // every score bin (p50 : p90 : p99 : p99.9 = ln 2 : ln 10 : ln 100 : ln 1000), m = 1.3e-8 (LiH), 0.8e-8 (N2, benzene),
  // 5.6e-8 (LiH / Psiformer).  A walker kept in float32 at score s therefore misses the tolerance with probability
  // exp(-tol / (m s)), and a score threshold can promise a RATE of misses, not their absence.  The probe measures m on
  // its sample (robustly: the larger of median / ln 2 and 90th percentile / ln 10) and sets the largest threshold for
  // which the expected share of kept walkers beyond refine_target (the tolerance, 1e-5) -- the mean of
  // exp(-tol / (m s_i)) over the kept walkers of the probed batch -- stays below refine_miss (1e-8).  (1e-7 until the linear kernels got their fresh
  // per-chunk accumulators, kernel_linear.hip: the typical error fell to 0.65-0.70 of what it was, the rare outliers whose error is
  // made elsewhere did not, and relative to the smaller m they reach further: 3 of 19 k N2 walkers at scores 100-400 beyond
  // 10 m s where the old kernels had none -- one of them at 1.008e-5 under the threshold of the 1e-7 rule.  1e-8 keeps the
  // largest float32 error of both 82 k-evaluation trajectories at 5e-6.)  Rounds 3-4 used threshold = 7e-6 / (90th percentile of err / score): along the same
  // trajectories that left 27-32 of 82 k evaluations beyond 1e-5 (max 2.2e-5).  Deep / ill-conditioned systems
  // (Psiformer, a random-init TransPsiformer) end up with most walkers above the threshold and fall into the direct
  // float64 pass by themselves.  refine_probe = 0 freezes the threshold at the option's value.
  double refine_thresh = 200.0;
  double refine_target = 1e-5;      // "refine_target_e7" (100): the relative tolerance the kept walkers are to meet
  double refine_miss = 1e-8;        // "refine_miss_e9" (10): accepted share of kept walkers beyond it
  int refine_probe = 32;
  int refine_sample = 256;       // walkers of the calibration sample (option "refine_sample"; 64 until round 4 -- six seeds on benzene
                                 // then drew thresholds that flagged 36-50 % of the batch)
  // Whole-batch float64 ("direct") mode, with hysteresis: a context ENTERS it when more than refine_direct_enter of a batch
  // lies above the threshold (the float32 pass would mostly be wasted: f32(B) + f64(p B) costs more than f64(B) from
  // p ~ 0.6 on, the float64 pass being ~2.5-3 x the float32 one per walker), runs 15 calls there, then looks again with a
  // float32 pass and LEAVES only if the flagged fraction has fallen below refine_direct_exit.  One draw near a single
  // 50 % line used to flip the mode -- and the cost of a benzene step between 166 and 207 ms -- from run to run.
  double refine_direct_enter = 0.60, refine_direct_exit = 0.45;
  int refine_direct_calls = 15;  // calls a context stays in the direct mode before it looks again (option "refine_direct_calls")
  int refine_direct_backoff = 4; // every look that CONFIRMS the direct mode doubles the stay, at most this many times (15, 30, ... 240 calls:
                                 // the float32 pass of such a look is thrown away -- 1/16 of ~0.4 of a benzene / C4H4 E_loc call at a fixed stay);
                                 // option "refine_direct_backoff"
  int direct_streak = 0;         // consecutive confirmations
  bool was_direct = false;       // the last mode decision was "direct"

  int twin_full_budget = 1;      // the twin's activation workspace may be as large as this context's (option "twin_full_budget")
  // option "pass_graph" (1): a forward-Laplacian pass that fits one workspace chunk is captured ONCE per (buffers, batch
  // size) into a hipGraph -- its ~40 launches, and the event records / waits that spread them over four streams -- and
  // replayed with one hipGraphLaunch per call.  The pass is launch-bound at the batch sizes it is used for (LiH, 4096
  // walkers: 140 us of a 1.7 ms float32 pass are gaps between kernels, 206 of the 690 us of the float64 twin's pass over
  // ~190 walkers: profiles/r03_eloc_pass_timeline.txt).  The first pass at a batch size runs eagerly (workspace, streams,
  // events, kernel attributes get created), the second one is captured on a stream of the context's own (the caller's
  // may be the legacy default stream, which cannot be captured) fenced by two events.  Everything a kernel of the pass
  // receives by value is a function of the key; what is not (the score threshold) is read from device memory.
  int pass_graph = 1;
  bool graph_broken = false;
  hipStream_t st_g = nullptr;
  hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;
  struct PassGraph { const void* p[7]; int B; bool flag; const void* ws; const void* flagp; const void* ws_tail[2]; uint64_t epoch, used; void* exec; };
  std::vector<PassGraph> pgraphs;
  uint64_t graph_epoch = 0, graph_clock = 0, graph_captures = 0, graph_hits = 0;
  std::vector<int> graph_warm;          // batch sizes that have run eagerly
  double* d_thresh = nullptr;           // device copy of refine_thresh (read by k_final: graphs must not bake it in)
  double thresh_uploaded = -1.0;
  bool graphs_active() const {
    return pass_graph && !graph_broken && !timing && !fused_dbg && ph_n == 0 && ecp_n_nl == 0;
  }
  // ... and only passes that are launch-bound: up to 2 GiB of activations per pass (LiH / PauliNet: 4096 walkers in
  // float32, the twin's few hundred in float64).  The passes of the larger systems run for milliseconds per kernel: a
  // replay gains nothing there, and rounding the twin's batch up would cost real work (benzene, 87 flagged walkers -> 128:
  // 266 -> 338 ms per step when it was tried)
  bool graph_fits(int B) const {
    const int TP = (3 * N + 2 + 15) / 16 * 16;
    // (batches below 64 walkers -- the twin's are multiples of 64 -- stay eager: probes, test and debugging calls; a capture
    // costs milliseconds and such calls rarely repeat)
    return graphs_active() && B >= 64 && (double)ws_bytes_per_walker(TP) * (double)B <= 2147483648.0;
  }
  void drop_graphs() {
    if (st_g && !pgraphs.empty()) (void)hipStreamSynchronize(st_g);      // (a replay may still be running)
    for (auto& g : pgraphs) if (g.exec) (void)hipGraphExecDestroy((hipGraphExec_t)g.exec);
    pgraphs.clear();
  }
  int calls_since_probe = -1;    // -1: never probed
  double probe_c = 0.0;          // last measured scale m of the float32 error per unit of score (0: none yet)
  bool flag_on = false;
  int refine_all_calls = 0;      // > 0: most walkers were flagged last time -> the next calls go to float64 directly
  std::vector<std::pair<std::string, int>> twin_opts;
  std::vector<double> w64_h, ecp_loc_h;
  int ecp_loc_nt_h = 0;
  // pseudo-Hamiltonian (dqmc_set_pseudo_hamiltonian): radial tables of the PH nuclei, compacted
  double* d_ph_loc = nullptr;
  double* d_ph_l2 = nullptr;
  int32_t* d_ph_nuc = nullptr;
  int ph_n = 0, ph_grid = 0;
  double ph_rmax = 0.0;
  size_t off_phq = 0;
  std::vector<double> ph_loc_h, ph_l2_h;      // host copies for the float64 twin
  std::vector<int32_t> ph_mask_h;
  dqmc_ctx* twin = nullptr;
  dqmc_ctx* twin_ctx() override { return twin; }
  int no_twin = 0;        // option "no_twin" 1: this float32 context never creates its float64 twin -- ensure_twin answers like a program
                          // without a float64 kernel set (DQMC_E_UNSUPPORTED): plain float32 everywhere (refinement off by itself,
                          // float32 tail, float32 ECP quadrature)
  // float64 TAIL of a float32 forward-Laplacian pass (option "tail_f64"; engine_refine.inl: run_tail).  DEFAULT 0 since the float32
  // linear kernels sum every k chunk into a fresh accumulator (kernel_linear.hip: FRESH): with the accumulator chain gone the head no longer
  // stands out (m x 0.87 / x 0.91 with the tail instead of x 0.81 / x 0.63), and the tail costs more than the walkers it saves the twin --
  // same call, tail on / off: LiH 5.54-5.70 / 5.51 ms per step (15.4 / 17.9 % refined), N2 54.3 / 49.2 ms (10.7 / 11.5 %).  What follows is
  // the reasoning it was built on, measured with the single accumulator chain:  The last
  // linear layer -- the backflow head whose output multiplies the envelopes into the Slater matrices -- is where a float32
  // rounding hurts most: everything after it is the ill-conditioned part of the path (inverse of A, CI cancellation), which
  // amplifies a relative error of the matrix entries by the walker's score, while the roundings of earlier layers average
  // out.  tests/f32_model.py (the float32 rounding model, which reproduces the MI355X's error percentiles) with that ONE
  // buffer kept in double: E_loc errors x 0.54 (LiH / PauliNet), x 0.39 (N2 / FermiNet) -- the scale m of the error model
  // drops by that factor, the score threshold rises by its inverse, and the share of walkers that need the float64 twin
  // falls.  Measured on the MI355X (tools/gpu_r05_d.sh): m x 0.81 on LiH (26.7 -> 20.8 % refined; the step is unchanged,
  // the tail's cost eats the saving), m x 0.63 on N2 (34.1 -> 16.2 % refined, 63.7 -> 58.4 ms per step with the first
  // version, which widened the 2.8 GB input of the head layer).  So the TAIL OPS -- the LINEAR ops that write the buffer
  // ORBITALS reads, then ORBITALS, SLOGDET and FINAL (envelopes x backflow, determinants, CI sum, E_loc) -- run on the
  // float64 twin for EVERY walker of an unchunked Laplacian pass: the twin's linear kernel reads the head's float32
  // activations where they lie (LinArgs::src_f32), the few other inputs (the Jastrow row) are widened into the twin's
  // workspace, the twin's k_final flags / scores for this context, the results are narrowed back.
  int tail_f64 = 0;
  int k_tail = -1;                   // first op of the tail; -1: this program has none (analyse_tail)
  std::vector<char> in_tail;         // per op: runs on the twin when a pass hands over its tail
  std::vector<int> tail_in;          // buffers the head writes and the tail reads ...
  std::vector<char> tail_direct;     // ... per buffer: every tail reader is a LINEAR op that can read it as float32 (not widened)
  std::vector<char> tail_written;    // buffers written by tail ops (dqmc_debug_read finds them in the twin's workspace)
  std::vector<char> tail_alloc;      // buffers the twin needs memory for while it runs a tail (written + widened)
  bool tail_only = false;            // this context (a twin) executes tail ops only ...
  std::vector<const float*> src32_of;   // ... and reads these buffers from the float32 context's workspace
  bool tail_now = false;             // the chunk being executed hands its tail to the twin
  bool last_tail = false;            // ... and so did the last Laplacian-mode evaluation
  char* d_tail = nullptr;            // widened positions / geometry and the tail's float64 results
  size_t tail_bytes = 0;
  int32_t* d_flag = nullptr;     // [0] = count, [1..] = walker indices
  double* d_score = nullptr;     // [B] error predictor of the last flagged pass
  size_t flag_cap = 0;
  char* d_ref = nullptr;
  size_t ref_bytes = 0;
  size_t score_cap = 0;
  const double* ref_e64 = nullptr;   // float64 local energies of the last refine_listed pass (device)
  std::vector<double> probe_sample_e;          // ... of the calibration sample of a probe call (host)
  std::function<void()> probe_rethreshold;     // set by the probe call: derives refine_thresh from probe_sample_e

  ~Engine() override {
    delete twin;
    drop_graphs();
    if (st_g) (void)hipStreamDestroy(st_g);
    if (ev_g0) (void)hipEventDestroy(ev_g0);
    if (ev_g1) (void)hipEventDestroy(ev_g1);
    if (d_thresh) (void)hipFree(d_thresh);
    if (st2) (void)hipStreamDestroy(st2);
    for (auto x : st_extra) if (x) (void)hipStreamDestroy(x);
    for (auto e : ms_events) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (auto e : buf_ev) if (e) (void)hipEventDestroy(e);
    if (d_molz) (void)hipFree(d_molz);
    if (d_flag) (void)hipFree(d_flag);
    if (d_score) (void)hipFree(d_score);
    if (d_tail) (void)hipFree(d_tail);
    if (d_ref) (void)hipFree(d_ref);
    if (d_descs) (void)hipFree(d_descs);
    if (d_wave_begin) (void)hipFree(d_wave_begin);
    if (d_fbufs2) (void)hipFree(d_fbufs2);
    if (d_ecp_loc) (void)hipFree(d_ecp_loc);
    if (d_ecp_nl) (void)hipFree(d_ecp_nl);
    if (d_ecp_nuc) (void)hipFree(d_ecp_nuc);
    if (d_ph_loc) (void)hipFree(d_ph_loc);
    if (d_ph_l2) (void)hipFree(d_ph_l2);
    if (d_ph_nuc) (void)hipFree(d_ph_nuc);
    if (d_ecp) (void)hipFree(d_ecp);
    if (d_ecpm) (void)hipFree(d_ecpm);
    for (auto e : ev_pool) (void)hipEventDestroy(e);
    if (d_w) (void)hipFree(d_w);
    if (d_it) (void)hipFree(d_it);
    if (d_charges) (void)hipFree(d_charges);
    if (d_ws) (void)hipFree(d_ws);
    if (d_mc) (void)hipFree(d_mc);
    if (d_nacc) (void)hipFree(d_nacc);
    if (d_tau_ring) (void)hipFree(d_tau_ring);
    if (d_acc) (void)hipFree(d_acc);
    if (d_ops) (void)hipFree(d_ops);
    if (d_wpk_off) (void)hipFree(d_wpk_off);
    if (d_wpk) (void)hipFree(d_wpk);
    if (d_tape) (void)hipFree(d_tape);
  }

  int init(const dqmc_system* s, const double* charges, const dqmc_buf* b, int nb, const dqmc_op* o, int no,
           const double* w, size_t nw, const int32_t* it, size_t nit) {
    sys = *s;
    N = sys.n_up + sys.n_down;
    if (N < 1 || sys.n_nuc < 1 || sys.n_det < 1) return fail(DQMC_E_ARG, "bad system sizes");
    bufs.assign(b, b + nb);
    ops.assign(o, o + no);
    charges_h.assign(charges, charges + sys.n_nuc);
    for (auto& bb : bufs)
      if (bb.rows < 1 || bb.width < 4 || bb.width % 4) return fail(DQMC_E_ARG, "buffer width must be a positive multiple of 4");
    n_weights = nw; n_itable = nit;
    HIP_TRY(hipMalloc((void**)&d_w, sizeof(real) * (nw ? nw : 1)));
    HIP_TRY(hipMalloc((void**)&d_it, sizeof(int32_t) * (nit ? nit : 1)));
    HIP_TRY(hipMalloc((void**)&d_charges, sizeof(double) * sys.n_nuc));
    HIP_TRY(hipMalloc((void**)&d_nacc, sizeof(int32_t) * 4));
    HIP_TRY(hipMalloc((void**)&d_tau_ring, sizeof(double) * 2));
    HIP_TRY(hipMalloc((void**)&d_acc, sizeof(double) * 16));
    HIP_TRY(hipMemsetAsync(d_nacc, 0, sizeof(int32_t) * 4, st));
    HIP_TRY(hipMemcpyAsync(d_charges, charges, sizeof(double) * sys.n_nuc, hipMemcpyHostToDevice, st));
    if (nit) HIP_TRY(hipMemcpyAsync(d_it, it, sizeof(int32_t) * nit, hipMemcpyHostToDevice, st));
    h_itable.assign(it, it + nit);
    HIP_TRY(hipStreamSynchronize(st));
    int rc = validate();
    if (rc) return rc;
    analyse_lanes();
    analyse_chains();
    analyse_streams();
    analyse_tail();
    if (sizeof(real) == 4) spec_k = dqmc::find_spec_kernel(program_hash());
    rc = set_weights(w, nw);
    if (rc) return rc;
    return build_fused_plan();
  }

  // ---- the rest of the engine, by concern (member functions; textual includes inside the struct body) ----
#include "engine_program.inl"
#include "engine_fused_plan.inl"
#include "engine_pass.inl"
#include "engine_refine.inl"
#include "engine_ecp.inl"
#include "engine_mcmc.inl"
};

}  // namespace

extern "C" {

const char* dqmc_last_error(void) { return g_err.c_str(); }

int dqmc_create(dqmc_ctx** out, int device, void* stream, const dqmc_system* sys, const double* charges_host,
                const dqmc_buf* bufs_host, int n_bufs, const dqmc_op* ops_host, int n_ops, const double* weights_host,
                size_t n_weights, const int32_t* itable_host, size_t n_itable) {
  if (!out || !sys || !charges_host || !bufs_host || !ops_host || !weights_host || n_bufs < 1 || n_ops < 1)
    return fail(DQMC_E_ARG, "null argument");
  *out = nullptr;
  HIP_TRY(hipSetDevice(device));
  dqmc_ctx* ctx = nullptr;
  int rc;
  if (sys->dtype == 0) {
    auto* e = new Engine<float>();
    e->st = (hipStream_t)stream; e->device = device;
    rc = e->init(sys, charges_host, bufs_host, n_bufs, ops_host, n_ops, weights_host, n_weights, itable_host, n_itable);
    ctx = e;
  } else if (sys->dtype == 1) {
    auto* e = new Engine<double>();
    e->st = (hipStream_t)stream; e->device = device;
    rc = e->init(sys, charges_host, bufs_host, n_bufs, ops_host, n_ops, weights_host, n_weights, itable_host, n_itable);
    ctx = e;
  } else {
    return fail(DQMC_E_ARG, "dtype must be 0 (float32) or 1 (float64)");
  }
  if (rc) { delete ctx; return rc; }
  *out = ctx;
  return DQMC_OK;
}

void dqmc_destroy(dqmc_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  delete ctx;
}

int dqmc_set_weights(dqmc_ctx* ctx, const double* w, size_t n) {
  if (!ctx || !w) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->set_weights(w, n);
}
int dqmc_wf_eval(dqmc_ctx* ctx, const void* r, const void* R, int B, void* logpsi, int32_t* sign) {
  if (!ctx || !r || !R || !logpsi || !sign) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->wf_eval(r, R, B, logpsi, sign);
}
int dqmc_local_energy(dqmc_ctx* ctx, const void* r, const void* R, int B, void* e_loc, void* stats, void* grad,
                      void* logpsi, int32_t* sign) {
  if (!ctx || !r || !R || !e_loc) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->local_energy(r, R, B, e_loc, stats, grad, logpsi, sign);
}
int dqmc_psi_grad(dqmc_ctx* ctx, const void* r, const void* R, int B, void* logpsi, int32_t* sign, void* grad) {
  if (!ctx || !r || !R || !grad) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->psi_grad(r, R, B, logpsi, sign, grad);
}
int dqmc_mcmc_steps(dqmc_ctx* ctx, void* r, void* logpsi, int32_t* sign, int32_t* age, void* tau, const void* R,
                    int B, int n_sub, int max_age, double target_acceptance, uint64_t seed, const void* noise,
                    const void* unif, uint8_t* accept_out, double* stats7_host) {
  if (!ctx || !r || !logpsi || !sign || !age || !tau || !R) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->mcmc(r, logpsi, sign, age, tau, R, B, n_sub, max_age, target_acceptance, seed, noise, unif, accept_out,
                   stats7_host);
}
int dqmc_langevin_update(dqmc_ctx* ctx, const void* r, const void* R, const double* mol_charges_host, int B, const void* tau,
                         void* logpsi, int32_t* sign, void* force) {
  if (!ctx || !r || !R || !mol_charges_host || !tau || !force) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->langevin_update(r, R, mol_charges_host, B, tau, logpsi, sign, force);
}
int dqmc_langevin_steps(dqmc_ctx* ctx, void* r, void* logpsi, int32_t* sign, int32_t* age, void* force, void* tau, const void* R,
                        const double* mol_charges_host, int B, int n_sub, int max_age, double target_acceptance, uint64_t seed,
                        const void* noise, const void* unif, uint8_t* accept_out, double* stats7_host) {
  if (!ctx || !r || !logpsi || !sign || !age || !force || !tau || !R || !mol_charges_host) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->langevin(r, logpsi, sign, age, force, tau, R, mol_charges_host, B, n_sub, max_age, target_acceptance, seed, noise, unif,
                       accept_out, stats7_host);
}
int dqmc_exchange_step(dqmc_ctx* ctx, void* r, void* logpsi, int32_t* sign, int32_t* age, const void* tau, const void* R, int B,
                       const int32_t* up_idx, const int32_t* down_idx, const void* unif, uint8_t* accept_out, double* stats7_host) {
  if (!ctx || !r || !logpsi || !sign || !age || !tau || !R || !up_idx || !down_idx || !unif) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->exchange(r, logpsi, sign, age, tau, R, B, up_idx, down_idx, unif, accept_out, stats7_host);
}
int dqmc_set_ecp(dqmc_ctx* ctx, int n_terms_loc, const double* loc_host, int n_l, int n_terms_nl, const double* nl_host) {
  if (!ctx) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->set_ecp(n_terms_loc, loc_host, n_l, n_terms_nl, nl_host);
}
int dqmc_set_pseudo_hamiltonian(dqmc_ctx* ctx, int n_grid, double r_max, const double* rv_loc_host, const double* rv_l2_host,
                                const int32_t* mask_host) {
  if (!ctx) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->set_ph(n_grid, r_max, rv_loc_host, rv_l2_host, mask_host);
}
int dqmc_ecp_rotation(dqmc_ctx* ctx, uint64_t seed, const void* phi) {
  if (!ctx) return fail(DQMC_E_ARG, "null argument");
  return ctx->ecp_rotation(seed, phi);
}
int dqmc_energy_stats(dqmc_ctx* ctx, const void* e_loc, const void* w, int B, double* out7_host) {
  if (!ctx || !e_loc || !out7_host) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->energy_stats(e_loc, w, B, out7_host);
}

// The ONE collective of a VMC step inside the library: per-rank record -> ncclAllGather over the caller's RCCL
// communicator on the context's stream -> Chan merge.  librccl is resolved at the first call (dlopen), so the
// library itself carries no link-time dependency on it.
int dqmc_energy_stats_allgather(dqmc_ctx* ctx, void* rccl_comm, int n_ranks, const void* e_loc, const void* w, int B,
                                double* out5_host) {
  if (!ctx || !rccl_comm || !e_loc || !out5_host || n_ranks < 1 || n_ranks > 4096) return fail(DQMC_E_ARG, "null / bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
  static allgather_fn nccl_all_gather = nullptr;
  if (!nccl_all_gather) {
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(DQMC_E_UNSUPPORTED, std::string("librccl.so not found: ") + dlerror());
    nccl_all_gather = (allgather_fn)dlsym(h, "ncclAllGather");
    if (!nccl_all_gather) return fail(DQMC_E_UNSUPPORTED, "ncclAllGather not found in librccl.so");
  }
  double* rec = nullptr;
  int rc = ctx->energy_stats_dev(e_loc, w, B, &rec);
  if (rc) return rc;
  if ((size_t)n_ranks * 7 > ctx->gather_cap) {
    if (ctx->d_gather) HIP_TRY(hipFree(ctx->d_gather));
    HIP_TRY(hipMalloc((void**)&ctx->d_gather, sizeof(double) * 7 * (size_t)n_ranks));
    ctx->gather_cap = (size_t)n_ranks * 7;
  }
  const int nrc = nccl_all_gather(rec, ctx->d_gather, 7, /* ncclFloat64 */ 8, rccl_comm, ctx->st);
  if (nrc != 0) return fail(DQMC_E_HIP, "ncclAllGather failed with code " + std::to_string(nrc));
  std::vector<double> host((size_t)n_ranks * 7);
  HIP_TRY(hipMemcpyAsync(host.data(), ctx->d_gather, sizeof(double) * host.size(), hipMemcpyDeviceToHost, ctx->st));
  HIP_TRY(hipStreamSynchronize(ctx->st));
  return dqmc_merge_energy_stats(host.data(), n_ranks, out5_host);
}

int dqmc_merge_energy_stats(const double* rec, int n_ranks, double* out5) {
  if (!rec || !out5 || n_ranks < 1) return fail(DQMC_E_ARG, "null argument");
  // Chan et al. pairwise merge of (n, mean, M2); min/max/weighted sums are plain reductions.
  double n = 0, mean = 0, m2 = 0, sw = 0, swe = 0, mn = INFINITY, mx = -INFINITY;
  for (int k = 0; k < n_ranks; ++k) {
    const double* r = rec + 7 * k;
    const double nb = r[0], mb = r[3] / r[0];
    const double d = mb - mean, nt = n + nb;
    m2 += r[4] + d * d * n * nb / nt;
    mean += d * nb / nt;
    n = nt;
    sw += r[1]; swe += r[2];
    mn = std::fmin(mn, r[5]); mx = std::fmax(mx, r[6]);
  }
  out5[0] = mean; out5[1] = std::sqrt(m2 / n); out5[2] = mn; out5[3] = mx; out5[4] = swe / sw;
  return DQMC_OK;
}

int dqmc_debug_read(dqmc_ctx* ctx, int buf, double* out, size_t n) {
  if (!ctx || !out) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->debug_read(buf, out, n);
}
int dqmc_debug_lanes(dqmc_ctx* ctx) { return ctx ? ctx->last_TP : 0; }
int dqmc_last_refined(dqmc_ctx* ctx) { return ctx ? ctx->last_refined : 0; }
int dqmc_ecp_counts(dqmc_ctx* ctx, int64_t* out3) {
  if (!ctx || !out3) return DQMC_E_ARG;
  for (int k = 0; k < 3; ++k) out3[k] = ctx->ecp_last_counts[k];
  return DQMC_OK;
}
int dqmc_last_chunks(dqmc_ctx* ctx, int* out2) {
  if (!ctx || !out2) return DQMC_E_ARG;
  out2[0] = ctx->last_chunks[0]; out2[1] = ctx->last_chunks[1];
  return DQMC_OK;
}
int dqmc_refine_info(dqmc_ctx* ctx, double* out4) {
  if (!ctx || !out4) return fail(DQMC_E_ARG, "null argument");
  for (int k = 0; k < 4; ++k) out4[k] = ctx->refine_info[k];
  return DQMC_OK;
}
int dqmc_substep_kernel(dqmc_ctx* ctx, char* name_out, size_t n) {
  if (!ctx || !name_out || n < 1) return fail(DQMC_E_ARG, "null argument");
  return ctx->substep_kernel(name_out, n);
}
int dqmc_refine_counters(dqmc_ctx* ctx, int64_t* out4) {
  if (!ctx || !out4) return fail(DQMC_E_ARG, "null argument");
  for (int k = 0; k < 4; ++k) out4[k] = ctx->refine_counters[k];
  return DQMC_OK;
}
int dqmc_refine_scores(dqmc_ctx* ctx, double* out, int n) {
  if (!ctx || !out || n < 1) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->refine_scores(out, n);
}
int dqmc_set_option(dqmc_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return fail(DQMC_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  return ctx->option(name, value);
}

// Per-launch HIP-event timing.  The records of a float32 context's float64 refinement twin are reported under the same
// names with the prefix "f64." ("f64.linear", "f64.attention", ...).
int dqmc_timing_enable(dqmc_ctx* ctx, int enable) {
  if (!ctx) return fail(DQMC_E_ARG, "null argument");
  ctx->t_collect();
  ctx->timing = enable != 0;
  if (dqmc_ctx* t = ctx->twin_ctx()) { t->t_collect(); t->timing = enable != 0; }
  return DQMC_OK;
}
int dqmc_timing_reset(dqmc_ctx* ctx) {
  if (!ctx) return fail(DQMC_E_ARG, "null argument");
  ctx->t_collect();
  ctx->trec.clear();
  if (dqmc_ctx* t = ctx->twin_ctx()) { t->t_collect(); t->trec.clear(); }
  return DQMC_OK;
}
int dqmc_timing_get(dqmc_ctx* ctx, const char* name, double* ms, int64_t* launches, double* flops) {
  if (!ctx || !name) return fail(DQMC_E_ARG, "null argument");
  if (!std::strncmp(name, "f64.", 4)) {
    dqmc_ctx* t = ctx->twin_ctx();
    if (t) return dqmc_timing_get(t, name + 4, ms, launches, flops);
    if (ms) *ms = 0; if (launches) *launches = 0; if (flops) *flops = 0;
    return DQMC_OK;
  }
  ctx->t_collect();
  auto it = ctx->trec.find(name);
  TimingRec r = it == ctx->trec.end() ? TimingRec{} : it->second;
  if (ms) *ms = r.ms;
  if (launches) *launches = r.launches;
  if (flops) *flops = r.flops;
  return DQMC_OK;
}
int dqmc_timing_get_executed(dqmc_ctx* ctx, const char* name, double* flops_executed) {
  if (!ctx || !name || !flops_executed) return fail(DQMC_E_ARG, "null argument");
  if (!std::strncmp(name, "f64.", 4)) {
    dqmc_ctx* t = ctx->twin_ctx();
    if (t) return dqmc_timing_get_executed(t, name + 4, flops_executed);
    *flops_executed = 0;
    return DQMC_OK;
  }
  ctx->t_collect();
  auto it = ctx->trec.find(name);
  *flops_executed = it == ctx->trec.end() ? 0.0 : it->second.executed;
  return DQMC_OK;
}
int dqmc_timing_names(dqmc_ctx* ctx, char* out, size_t n) {
  if (!ctx || !out || n == 0) return fail(DQMC_E_ARG, "null argument");
  ctx->t_collect();
  std::string s;
  for (auto& kv : ctx->trec) { if (!s.empty()) s += ","; s += kv.first; }
  if (dqmc_ctx* t = ctx->twin_ctx()) {
    t->t_collect();
    for (auto& kv : t->trec) { if (!s.empty()) s += ","; s += "f64." + kv.first; }
  }
  std::snprintf(out, n, "%s", s.c_str());
  return DQMC_OK;
}

}  // extern "C"

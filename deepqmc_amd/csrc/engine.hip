// engine.hip -- the C ABI of include/dqmc.h: context, layer-program executor, MCMC driver,
// per-kernel timing.  Host code only; every arithmetic step is one of the kernels in
// kernel_linear.hip / kernels_graph.hip / kernels_head.hip / kernels_mcmc.hip.
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
  double flops = 0;
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
  struct Pending { std::string name; hipEvent_t a, b; double flops; };
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
  void t_begin(const char* name, double flops, hipStream_t s = nullptr) {
    if (!timing) return;
    t_stream = s ? s : st;
    Pending p{name, get_event(), get_event(), flops};
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
      r.ms += ms; r.launches += 1; r.flops += p.flops;
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
  // decide, the round-4 rule): beyond it the walker's float64 bound tightens in proportion (kernels.h: EcpMixArgs::l32)
  double ecp_dlog_floor = 1e-4;
  char* d_ecpm = nullptr;
  size_t ecpm_bytes = 0;
  size_t ecp_max_cfg = 1 << 16; // quadrature walkers per value-mode batch
  // float64 refinement (float32 build): walkers flagged by k_final as ill conditioned (near a node of psi the kinetic
  // energy is a difference of huge numbers and float32 round-off is amplified by the CI cancellation) are
  // re-evaluated by a float64 twin of this context and their results replace the float32 ones
  int refine = 1;
  // Flag rule: score > refine_thresh (kernels.h: FinalArgs).  The threshold is SELF-CALIBRATED: every refine_probe-th
  // local-energy call (and the first) a strided sample of <= refine_sample (256) walkers is evaluated by the float64 twin
  // as well, the measured float32 error per unit of score -- its 90th percentile c over the sample -- sets
  // refine_thresh = refine_target / c, i.e. the score at which the expected error reaches refine_target (7e-6
  // relative, 0.7 of the tolerance of the north star).  Deep / ill-conditioned systems (Psiformer, a random-init
  // TransPsiformer) measure a large c, flag most walkers and fall into the direct float64 pass by themselves; a
  // small system keeps a few per cent.  refine_probe = 0 freezes the threshold at the option's value.
  double refine_thresh = 200.0;
  double refine_target = 7e-6;
  int refine_probe = 32;
  int refine_sample = 256;       // walkers of the calibration sample (option "refine_sample": c is the 90th percentile of a sample this
                                 // large; 64 until round 4 -- six seeds on benzene then drew thresholds that flagged 36-50 % of the batch)
  // Whole-batch float64 ("direct") mode, with hysteresis: a context ENTERS it when more than refine_direct_enter of a batch
  // lies above the threshold (the float32 pass would mostly be wasted: f32(B) + f64(p B) costs more than f64(B) from
  // p ~ 0.6 on, the float64 pass being ~2.5-3 x the float32 one per walker), runs 15 calls there, then looks again with a
  // float32 pass and LEAVES only if the flagged fraction has fallen below refine_direct_exit.  One draw near a single
  // 50 % line used to flip the mode -- and the cost of a benzene step between 166 and 207 ms -- from run to run.
  double refine_direct_enter = 0.60, refine_direct_exit = 0.45;
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
  struct PassGraph { const void* p[7]; int B; bool flag; const void* ws; const void* flagp; uint64_t epoch, used; void* exec; };
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
  double probe_c = 0.0;          // last measured error per unit of score (0: none yet)
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
    rc = set_weights(w, nw);
    if (rc) return rc;
    return build_fused_plan();
  }

  int validate() {
    const int nb = (int)bufs.size();
    auto okb = [&](int b) { return b >= 0 && b < nb; };
    for (size_t k = 0; k < ops.size(); ++k) {
      const dqmc_op& op = ops[k];
      const int32_t* i = op.i;
      bool ok = true;
      switch (op.kind) {
        case DQMC_OP_FEAT_EN: ok = okb(i[0]) && bufs[i[0]].rows == N && bufs[i[0]].width >= 4 * sys.n_nuc + (i[2] ? 1 : 0); break;
        case DQMC_OP_FEAT_EE: ok = okb(i[0]) && bufs[i[0]].rows == i[2] && bufs[i[0]].width == 4 && i[1] >= 0 && (size_t)(i[1] + 2 * i[2]) <= n_itable; break;
        case DQMC_OP_LINEAR: {
          ok = i[0] >= 1 && i[0] <= 4 && okb(i[17]);
          size_t wrows = 0;
          for (int p = 0; ok && p < i[0]; ++p) {
            const int sb = i[1 + 4 * p], r0 = i[2 + 4 * p], K = i[3 + 4 * p], bc = i[4 + 4 * p];
            ok = okb(sb) && K >= 1 && pad4(K) <= bufs[sb].width && r0 >= 0 && r0 + (bc ? 1 : i[20]) <= bufs[sb].rows;
            wrows += pad4(K);
          }
          if (ok) {
            const dqmc_buf& d = bufs[i[17]];
            const int ldw = pad4(i[21]);
            ok = i[18] >= 0 && i[18] + i[20] <= d.rows && i[19] >= 0 && i[19] % 4 == 0 && i[19] + ldw <= d.width &&
                 i[22] >= 0 && i[22] % 4 == 0 && (size_t)i[22] + wrows * ldw <= n_weights &&
                 (i[23] < 0 || (i[23] % 4 == 0 && (size_t)(i[23] + ldw) <= n_weights)) && i[24] >= 0 && i[24] <= 4;
            if (ok && i[25] >= 0)
              ok = okb(i[25]) && bufs[i[25]].width >= i[19] + ldw && i[26] >= 0 && i[26] + i[20] <= bufs[i[25]].rows;
          }
          break;
        }
        case DQMC_OP_SPIN_MEAN: ok = okb(i[0]) && okb(i[1]) && bufs[i[0]].rows == N && bufs[i[1]].rows == 2 && bufs[i[0]].width == bufs[i[1]].width; break;
        case DQMC_OP_CONST: ok = okb(i[0]) && i[1] >= 0 && (size_t)i[1] + (size_t)bufs[i[0]].rows * bufs[i[0]].width <= n_weights; break;
        case DQMC_OP_CONV: ok = okb(i[0]) && okb(i[1]) && okb(i[2]) && bufs[i[2]].rows == N && i[6] <= bufs[i[0]].width && i[6] <= bufs[i[1]].width && i[3] + i[6] <= bufs[i[2]].width && (size_t)(i[4] + 2 * N * i[5]) <= n_itable; break;
        case DQMC_OP_EDGE_SUM: ok = okb(i[0]) && okb(i[2]) && bufs[i[2]].rows == N && i[6] <= bufs[i[0]].width && i[3] + i[6] <= bufs[i[2]].width && (size_t)(i[4] + 2 * N * i[5]) <= n_itable; break;
        case DQMC_OP_ROW_SUM: ok = okb(i[0]) && okb(i[1]) && bufs[i[1]].rows == 1 && bufs[i[0]].width == bufs[i[1]].width; break;
        case DQMC_OP_ORBITALS: {
          if (i[6] < 0) { ok = false; break; }
          const size_t ne = (size_t)sys.n_det * N * sys.n_nuc * (i[6] > 0 ? i[6] : 1);
          ok = okb(i[0]) && okb(i[1]) && bufs[i[0]].rows == N && bufs[i[0]].width >= sys.n_det * N && bufs[i[1]].rows == sys.n_det && bufs[i[1]].width >= N * N;
          for (int q = 2; ok && q < 6; ++q) ok = i[q] >= 0 && (size_t)i[q] + ne <= n_weights;
          break;
        }
        case DQMC_OP_SLOGDET: ok = okb(i[0]) && bufs[i[0]].rows == sys.n_det && bufs[i[0]].width >= N * N && N <= 44; break;
        case DQMC_OP_FINAL: ok = (i[0] < 0 || (okb(i[0]) && bufs[i[0]].rows == 1)) && (i[1] < 0 || (size_t)(i[1] + sys.n_det) <= n_weights) && i[3] >= 0 && (size_t)(i[3] + 2) <= n_weights && i[2] >= 0 && i[2] <= 2; break;
        case DQMC_OP_ATTENTION:
          ok = i[4] >= 1 && i[5] >= 1;
          for (int q = 0; ok && q < 4; ++q) ok = okb(i[q]) && bufs[i[q]].rows == N && bufs[i[q]].width >= i[4] * i[5];
          if (ok) ok = bufs[i[0]].width == bufs[i[1]].width && bufs[i[0]].width == bufs[i[2]].width && bufs[i[0]].width == bufs[i[3]].width;
          if (ok) ok = i[6] >= 0 && (i[6] == 0 || (i[7] >= 0 && i[8] >= 0 && (size_t)i[7] + (size_t)i[6] * i[4] * i[5] <= n_weights &&
                                                   (size_t)i[8] + (size_t)i[6] * i[4] * i[5] <= n_weights));
          if (ok && dqmc::attention_lds_bytes<real>(N, i[5], i[6]) > (size_t)160 * 1024)
            return fail(DQMC_E_UNSUPPORTED, "attention tile set (N, head_dim) exceeds the 160 KiB LDS");
          break;
        default: ok = false;
      }
      if (ok && op.kind == DQMC_OP_FEAT_EE)          // senders: electron s >= 0, or nucleus -1 - s
        for (int r = 0; r < i[2] && ok; ++r) {
          const int rc = h_itable[i[1] + 2 * r], sd = h_itable[i[1] + 2 * r + 1];
          ok = rc >= 0 && rc < N && sd < N && -1 - sd < sys.n_nuc;
        }
      if (ok && op.kind == DQMC_OP_CONV)
        for (int q = 0; q < N * i[5] && ok; ++q) {
          const int row = h_itable[i[4] + 2 * q], sd = h_itable[i[4] + 2 * q + 1];
          if (row < 0) continue;
          ok = row < bufs[i[0]].rows && (sd >= 0 ? sd : -1 - sd) < bufs[i[1]].rows;
        }
      if (!ok) return fail(DQMC_E_ARG, "malformed op #" + std::to_string(k) + " kind " + std::to_string(op.kind));
    }
    return DQMC_OK;
  }

  // Row-wise two-layer MLPs (hkext.MLP with one hidden layer: the edge MLPs w / u and the node MLP h of a message-passing
  // layer): LINEAR op k writes a private hidden buffer that exactly one later LINEAR op reads as its single piece, same
  // rows, whole width.  The pair runs as one launch of the chained kernel at k's position, which is legal when nothing
  // between the two ops touches the second layer's output and its residual input is complete before k.
  void analyse_chains() {
    const int no = (int)ops.size(), nb = (int)bufs.size();
    mlp_child.assign(no, -1);
    mlp_skip.assign(no, 0);
    if (!mlp_fuse) return;
    std::vector<int> rd, wr;
    std::vector<std::vector<int>> writers(nb), readers(nb);
    for (int k = 0; k < no; ++k) { op_io(ops[k], rd, wr); for (int b : wr) writers[b].push_back(k); for (int b : rd) readers[b].push_back(k); }
    for (int c = 0; c < no; ++c) {
      const int32_t* ci = ops[c].i;
      if (ops[c].kind != DQMC_OP_LINEAR || ci[0] != 1 || ci[4]) continue;                 // one non-broadcast piece
      const int hb = ci[1];
      if (writers[hb].size() != 1 || readers[hb].size() != 1) continue;                   // a private hidden buffer
      const int p = writers[hb][0];
      if (p >= c || ops[p].kind != DQMC_OP_LINEAR || mlp_child[p] >= 0 || mlp_skip[p]) continue;
      const int32_t* pi = ops[p].i;
      bool bc = false;
      for (int q = 0; q < pi[0]; ++q) bc = bc || pi[4 + 4 * q];
      if (bc || pi[25] >= 0) continue;                                                    // no broadcast pieces, no residual on the hidden layer
      if (pi[18] != ci[2] || pi[20] != ci[20] || pi[19] != 0 || pad4(ci[3]) != pad4(pi[21])) continue;   // same rows, whole width
      if (pad4(pi[21]) > 64 || pad4(ci[21]) > 32) continue;
      if ((pi[24] & 7) > 2 || (ci[24] & 7) > 4) continue;
      bool ok = true;
      if (ci[25] >= 0) for (int w : writers[ci[25]]) ok = ok && w < p;                    // residual input complete before the pair runs
      for (int m = p + 1; m < c && ok; ++m) {                                             // nobody in between reads or writes the output buffer
        op_io(ops[m], rd, wr);
        for (int b : rd) ok = ok && b != ci[17];
        for (int b : wr) ok = ok && b != ci[17];
      }
      if (!ok) continue;
      mlp_child[p] = c;
      mlp_skip[c] = 1;
    }
  }

  // Stream slots of the Laplacian pass.  Edge-stream ops (pair-compact destination) keep slot 1.  Every other op goes, in
  // program order, to a slot whose last op it depends on anyway (directly or through other ops) -- placing it there costs
  // no concurrency -- preferring the main slot, then the slot of its most recent producer, then a free one; only if
  // there is none does it queue behind unrelated work on the main slot.  A chained MLP pair counts as one op.
  void analyse_streams() {
    const int no = (int)ops.size(), nb = (int)bufs.size();
    op_sid.assign(no, 0);
    std::vector<int> rd, wr, rd2, wr2;
    std::vector<std::vector<int>> writers(nb);
    std::vector<std::vector<char>> dep(no, std::vector<char>(no, 0));
    auto io = [&](int k) {
      op_io(ops[k], rd, wr);
      if (mlp_child.size() == (size_t)no && mlp_child[k] >= 0) {
        op_io(ops[mlp_child[k]], rd2, wr2);
        for (int b : rd2) if (b != ops[k].i[17]) rd.push_back(b);
        for (int b : wr2) wr.push_back(b);
      }
    };
    int last[4] = {-1, -1, -1, -1};
    for (int k = 0; k < no; ++k) {
      if (mlp_skip.size() == (size_t)no && mlp_skip[k]) {     // rides with its parent (same slot if it has to run on its own)
        op_sid[k] = 0;
        for (int p = 0; p < k; ++p) if (mlp_child[p] == k) op_sid[k] = op_sid[p];
        continue;
      }
      io(k);
      int producer = -1;
      for (int b : rd)
        for (int w : writers[b]) {
          if (w >= k) continue;
          dep[k][w] = 1;
          for (int x = 0; x < w; ++x) if (dep[w][x]) dep[k][x] = 1;
          if (w > producer) producer = w;
        }
      const dqmc_op& o = ops[k];
      const bool edge = (o.kind == DQMC_OP_FEAT_EE && compact[o.i[0]]) || (o.kind == DQMC_OP_LINEAR && compact[o.i[17]]);
      int sid = 0;
      if (edge) sid = 1;
      else if (o.kind == DQMC_OP_SLOGDET || o.kind == DQMC_OP_FINAL || o.kind == DQMC_OP_ATTENTION) sid = 0;
      else {
        auto eligible = [&](int s_) { return last[s_] < 0 || dep[k][last[s_]]; };
        if (eligible(0)) sid = 0;
        else if (producer >= 0 && op_sid[producer] >= 2 && eligible(op_sid[producer])) sid = op_sid[producer];
        else if (eligible(2)) sid = 2;
        else if (eligible(3)) sid = 3;
        else sid = 0;
      }
      op_sid[k] = sid;
      last[sid] = k;
      for (int b : wr) writers[b].push_back(k);
    }
  }

  // Which buffers can carry pair-compact lanes: outputs of FEAT_EE and of row-wise LINEAR ops on them (all
  // pieces compact, none broadcast; residual compact with the same row pairs).  Consumers that understand the
  // compact layout: LINEAR, CONV (edge operand), EDGE_SUM.  Anything else turns the optimisation off.
  void analyse_lanes() {
    const int nb = (int)bufs.size();
    compact.assign(nb, 0);
    pair_rs.assign(nb, std::vector<int>());
    if (!lane_compact) return;
    std::vector<char> full_written(nb, 0);
    bool ok = true;
    auto set_pairs = [&](int b, int row, int rc, int sd) {
      if (pair_rs[b].empty()) pair_rs[b].assign(2 * (size_t)bufs[b].rows, -2);
      int& pr = pair_rs[b][2 * row];
      int& ps = pair_rs[b][2 * row + 1];
      if (pr != -2 && (pr != rc || ps != sd)) ok = false;
      pr = rc; ps = sd;
    };
    for (size_t k = 0; k < ops.size() && ok; ++k) {
      const int32_t* i = ops[k].i;
      switch (ops[k].kind) {
        case DQMC_OP_FEAT_EE:
          if (full_written[i[0]]) { ok = false; break; }
          compact[i[0]] = 1;
          for (int r = 0; r < i[2]; ++r) set_pairs(i[0], r, h_itable[i[1] + 2 * r], h_itable[i[1] + 2 * r + 1]);
          break;
        case DQMC_OP_LINEAR: {
          int n_c = 0;
          for (int p = 0; p < i[0]; ++p) n_c += compact[i[1 + 4 * p]] ? 1 : 0;
          if (n_c == 0) {
            if (compact[i[17]] || (i[25] >= 0 && compact[i[25]])) ok = false;
            full_written[i[17]] = 1;
            break;
          }
          if (n_c != i[0] || full_written[i[17]]) { ok = false; break; }
          compact[i[17]] = 1;
          for (int rr = 0; rr < i[20] && ok; ++rr) {
            const int sb0 = i[1], r00 = i[2];
            if (i[4] || pair_rs[sb0].empty()) { ok = false; break; }
            const int rc = pair_rs[sb0][2 * (r00 + rr)], sd = pair_rs[sb0][2 * (r00 + rr) + 1];
            for (int p = 1; p < i[0]; ++p) {
              const int sb = i[1 + 4 * p], r0 = i[2 + 4 * p];
              if (i[4 + 4 * p] || pair_rs[sb].empty() || pair_rs[sb][2 * (r0 + rr)] != rc || pair_rs[sb][2 * (r0 + rr) + 1] != sd) ok = false;
            }
            if (i[25] >= 0) {
              const int rb = i[25];
              if (!compact[rb] || pair_rs[rb].empty() || pair_rs[rb][2 * (i[26] + rr)] != rc || pair_rs[rb][2 * (i[26] + rr) + 1] != sd) ok = false;
            }
            set_pairs(i[17], i[18] + rr, rc, sd);
          }
          break;
        }
        case DQMC_OP_CONV:
          if (compact[i[1]] || compact[i[2]]) ok = false;
          full_written[i[2]] = 1;
          if (compact[i[0]])      // the table's (row, sender) of receiver el must be the row's own pair
            for (int el = 0; el < N && ok; ++el)
              for (int sdx = 0; sdx < i[5]; ++sdx) {
                const int row = h_itable[i[4] + 2 * (el * i[5] + sdx)], snd = h_itable[i[4] + 2 * (el * i[5] + sdx) + 1];
                if (row >= 0 && (pair_rs[i[0]][2 * row] != el || pair_rs[i[0]][2 * row + 1] != snd)) ok = false;
              }
          break;
        case DQMC_OP_EDGE_SUM:
          if (compact[i[2]]) ok = false;
          full_written[i[2]] = 1;
          if (compact[i[0]])
            for (int el = 0; el < N && ok; ++el)
              for (int sdx = 0; sdx < i[5]; ++sdx) {
                const int row = h_itable[i[4] + 2 * (el * i[5] + sdx)], snd = h_itable[i[4] + 2 * (el * i[5] + sdx) + 1];
                if (row >= 0 && (pair_rs[i[0]][2 * row] != el || pair_rs[i[0]][2 * row + 1] != snd)) ok = false;
              }
          break;
        case DQMC_OP_FEAT_EN: case DQMC_OP_CONST: full_written[i[0]] = 1; break;
        case DQMC_OP_SPIN_MEAN: case DQMC_OP_ROW_SUM: if (compact[i[0]]) ok = false; full_written[i[1]] = 1; break;
        case DQMC_OP_ORBITALS: if (compact[i[0]]) ok = false; full_written[i[1]] = 1; break;
        case DQMC_OP_SLOGDET: if (compact[i[0]]) ok = false; break;
        case DQMC_OP_FINAL: if (i[0] >= 0 && compact[i[0]]) ok = false; break;
        case DQMC_OP_ATTENTION:
          for (int q = 0; q < 3; ++q) if (compact[i[q]]) ok = false;
          full_written[i[3]] = 1;
          break;
        default: break;
      }
    }
    for (int b = 0; b < nb && ok; ++b)
      if (compact[b]) for (int v : pair_rs[b]) if (v == -2) ok = false;   // every row of a compact buffer has a pair
    if (!ok) { compact.assign(nb, 0); pair_rs.assign(nb, std::vector<int>()); }
  }
  // lanes of buffer b in an evaluation with TP lanes
  int lanes_of(int b, int TP) const { return (TP > 1 && compact[b]) ? dqmc::PAIR_LANES : TP; }

  int set_weights(const double* w, size_t n) override {
    if (n != n_weights) return fail(DQMC_E_ARG, "weight buffer length differs from the one given at creation");
    ++graph_epoch;
    wtmp.resize(n);
    for (size_t k = 0; k < n; ++k) wtmp[k] = (real)w[k];
    if (sizeof(real) == 4) {
      w64_h.assign(w, w + n);
      if (twin) { const int rc = twin->set_weights(w, n); if (rc) return rc; }
    }
    HIP_TRY(hipMemcpyAsync(d_w, wtmp.data(), sizeof(real) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (fused_n_ops > 0) return pack_fused_weights();
    return DQMC_OK;
  }
  // ---- fused value-only evaluation (kernel_fused2.hip) ----------------------------------
  // Ops [0, fused_n_ops) (everything up to and including ORBITALS) run in one kernel on a tile
  // of WT walkers with LDS-resident buffers; buffers read by later ops stay in the workspace.
  int option(const char* name, int value) override {
    const std::string s(name);
    if (s.rfind("twin.", 0) == 0) {            // an option of the float64 refinement twin (applied when it is created, too)
      twin_opts.emplace_back(s.substr(5), value);
      return twin ? twin->option(s.c_str() + 5, value) : DQMC_OK;
    }
    ++graph_epoch;                             // (any switch may change what a captured pass would launch)
    if (s == "pass_graph") { pass_graph = value; if (!value) drop_graphs(); if (twin) twin->option("pass_graph", value); twin_opts.emplace_back(s, value); return DQMC_OK; }
    if (s == "fused") { fused_enabled = value; return DQMC_OK; }
    if (s == "fused_wt") { fused_wt_req = value; return build_fused_plan(); }
    if (s == "fused_occ") { fused_occ_req = value; return DQMC_OK; }
    if (s == "attention_mfma") { attention_mfma = value; return DQMC_OK; }
    if (s == "attention_ncb") { attention_ncb = value; return DQMC_OK; }
    if (s == "attention_split") { attention_split = value; return DQMC_OK; }
    if (s == "ws_budget_mb") { if (value < 1) return fail(DQMC_E_ARG, "ws_budget_mb must be positive"); ws_budget = (size_t)value << 20; return DQMC_OK; }
    if (s == "slogdet_mfma") { slogdet_mfma = value; return DQMC_OK; }
    if (s == "lane_compact") { lane_compact = value != 0; analyse_lanes(); last_B = 0; return DQMC_OK; }
    if (s == "linear_f64_split") { linear_f64_split = value; return DQMC_OK; }
    if (s == "linear_bkx") { linear_bkx = value; return DQMC_OK; }      // (kernel selection of kernel_linear.hip, this context only)
    if (s == "mlp_fuse") { mlp_fuse = value; analyse_chains(); return DQMC_OK; }
    if (s == "fused_substep") { fused_substep = value; return DQMC_OK; }
    if (s == "split_bcast") { split_bcast = value; return DQMC_OK; }
    if (s == "dual_stream") { dual_stream = value; return DQMC_OK; }
    if (s == "multi_stream") { multi_stream = value; return DQMC_OK; }
    if (s == "fused_prio") { fused_prio = value; return DQMC_OK; }
    if (s == "fused_lean") { fused_lean = value; return build_fused_plan(); }
    if (s == "fused_bf") { fused_bf = value; return build_fused_plan(); }
    if (s == "linear_bf") { linear_bf = value; return DQMC_OK; }      // (kernel selection of kernel_linear.hip, this context only)
    if (s == "fused_wg_per_cu") {
      if (value < 4 || value > 6) return fail(DQMC_E_ARG, "fused_wg_per_cu must be 4, 5 or 6");
      fused2_lds_quarter = (size_t)160 * 1024 / value;
      return build_fused_plan();
    }
    if (s == "refine") { refine = value; return DQMC_OK; }
    if (s == "twin_full_budget") { twin_full_budget = value; if (twin) twin->option("ws_budget_mb", (int)((value ? ws_budget : ws_budget / 2) >> 20)); return DQMC_OK; }
    if (s == "refine_sample") { if (value < 2) return fail(DQMC_E_ARG, "refine_sample must be >= 2"); refine_sample = value; calls_since_probe = -1; return DQMC_OK; }
    if (s == "refine_probe") { if (value < 0) return fail(DQMC_E_ARG, "refine_probe must be >= 0"); refine_probe = value; calls_since_probe = -1; return DQMC_OK; }
    if (s == "refine_target_e7") { if (value < 1) return fail(DQMC_E_ARG, "refine_target_e7 must be >= 1"); refine_target = 1e-7 * value; calls_since_probe = -1; return DQMC_OK; }
    if (s == "refine_direct_pct") { if (value < 1 || value > 100) return fail(DQMC_E_ARG, "refine_direct_pct must be 1..100"); refine_direct_enter = 0.01 * value; if (refine_direct_exit > refine_direct_enter) refine_direct_exit = refine_direct_enter; return DQMC_OK; }
    if (s == "refine_direct_exit_pct") { if (value < 0 || value > 100) return fail(DQMC_E_ARG, "refine_direct_exit_pct must be 0..100"); refine_direct_exit = 0.01 * value; if (refine_direct_exit > refine_direct_enter) refine_direct_enter = refine_direct_exit; return DQMC_OK; }
    if (s == "refine_thresh") { if (value < 0) return fail(DQMC_E_ARG, "refine_thresh must be >= 0"); refine_thresh = (double)value; return DQMC_OK; }
    if (s == "fused_sched_kb") { fused_sched_budget = (size_t)value * 1024; return build_fused_plan(); }
    if (s == "fused_print") {   // plan summary on stderr (tuning aid)
      fprintf(stderr, "[dqmc] fused plan: WT=%d lds=%zu B; %d fused ops, %d levels\n", fused2_WT, fused2_lds, fused_n_ops,
              fused_n_ops ? f_level[fused_n_ops - 1] + 1 : 0);
      if (value >= 2)        // one line per descriptor of every wave list: kind, scheduled op, its op kind, row blocks, quads of k-steps
        for (size_t w = 0; w < plan_lists.size(); ++w)
          for (size_t k = 0; k < plan_lists[w].size(); ++k) {
            const dqmc::FDesc& d = plan_lists[w][k];
            int nq = 0;
            for (int p = 0; p < d.n_pieces; ++p) nq += d.a_nq[p];
            fprintf(stderr, "[dqmc] wave %zu desc %zu kind %d op %d opkind %d ma %d row0 %d col0 %d quads %d ldw %d\n", w, k, d.kind, d.op,
                    d.kind == 2 ? 0 : ops[f_order[d.op]].kind, d.ma, d.row0, d.col0, nq, d.ldw);
          }
      return DQMC_OK;
    }
    if (s == "ecp_mixed") { ecp_mixed_on = value; return DQMC_OK; }
    if (s == "ecp_heavy_e6") { if (value < 0) return fail(DQMC_E_ARG, "ecp_heavy_e6 must be >= 0"); ecp_w_heavy = 1e-6 * value; return DQMC_OK; }
    if (s == "ecp_dlog_floor_e6") { if (value < 0) return fail(DQMC_E_ARG, "ecp_dlog_floor_e6 must be >= 0"); ecp_dlog_floor = 1e-6 * value; return DQMC_OK; }
    if (s == "ecp_skip_e12") { if (value < 0) return fail(DQMC_E_ARG, "ecp_skip_e12 must be >= 0"); ecp_w_skip = 1e-12 * value; return DQMC_OK; }
    if (s == "ecp_max_cfg") { if (value < 1) return fail(DQMC_E_ARG, "ecp_max_cfg must be positive"); ecp_max_cfg = (size_t)value; return DQMC_OK; }
    if (s == "fused_sched") { fused_sched_mode = value; return build_fused_plan(); }
    if (s == "fused_dbg") {
      fused_dbg = value;
      if (value && !d_prof) { HIP_TRY(hipMalloc((void**)&d_prof, sizeof(long long) * (9 * ops.size() + 80 + 1024 + 2 * 8192))); HIP_TRY(hipMemsetAsync(d_prof, 0, sizeof(long long) * (9 * ops.size() + 80 + 1024 + 2 * 8192), st)); }
      return DQMC_OK;
    }
    if (s == "fused_lds_kb") { fused2_lds_budget = (size_t)value * 1024; return build_fused_plan(); }
    return fail(DQMC_E_ARG, "unknown option " + s);
  }

  // LDS placement of the buffers for a tile of WT walkers: first-fit over live intervals.
  // Buffers read / written by an op (buffer granularity; partial writers of one buffer touch
  // disjoint row or column ranges by construction of the program).
  static void op_io(const dqmc_op& op, std::vector<int>& rd, std::vector<int>& wr) {
    const int32_t* i = op.i;
    rd.clear(); wr.clear();
    switch (op.kind) {
      case DQMC_OP_FEAT_EN: case DQMC_OP_FEAT_EE: case DQMC_OP_CONST: wr.push_back(i[0]); break;
      case DQMC_OP_LINEAR:
        for (int p = 0; p < i[0]; ++p) rd.push_back(i[1 + 4 * p]);
        if (i[25] >= 0) rd.push_back(i[25]);
        wr.push_back(i[17]); break;
      case DQMC_OP_SPIN_MEAN: case DQMC_OP_ROW_SUM: rd.push_back(i[0]); wr.push_back(i[1]); break;
      case DQMC_OP_CONV: rd.push_back(i[0]); rd.push_back(i[1]); wr.push_back(i[2]); break;
      case DQMC_OP_EDGE_SUM: rd.push_back(i[0]); wr.push_back(i[2]); break;
      case DQMC_OP_ORBITALS: rd.push_back(i[0]); wr.push_back(i[1]); break;
      case DQMC_OP_SLOGDET: rd.push_back(i[0]); break;
      case DQMC_OP_FINAL: if (i[0] >= 0) rd.push_back(i[0]); break;
      default: break;
    }
  }

  // Dependency levels of ops [0, fused_n_ops): ops of one level are independent and run
  // without a workgroup barrier between them inside the fused kernel.
  void fused_schedule() {
    const int no = fused_n_ops, nb = (int)bufs.size();
    std::vector<int> lvl(no, 0), wlevel(nb, -1);
    std::vector<int> rd, wr;
    if (fused_sched_mode == 0) {
      // Program order kept (so LDS liveness is what the program compiler laid out); a new level
      // starts whenever an op reads a buffer written inside the current level.
      int cur = 0;
      std::vector<char> written(nb, 0);
      f_order.resize(no);
      f_level.assign(no, 0);
      for (int k = 0; k < no; ++k) {
        op_io(ops[k], rd, wr);
        bool dep = false;
        for (int b : rd) dep = dep || written[b];
        if (dep) { ++cur; std::fill(written.begin(), written.end(), 0); }
        for (int b : wr) written[b] = 1;
        f_order[k] = k;
        f_level[k] = cur;
      }
      return;
    }
    for (int k = 0; k < no; ++k) {
      op_io(ops[k], rd, wr);
      int l = 0;
      for (int b : rd) if (wlevel[b] + 1 > l) l = wlevel[b] + 1;
      for (int b : wr) if (wlevel[b] >= 0 && wlevel[b] > l) l = wlevel[b];   // co-writers share a level or later
      lvl[k] = l;
      for (int b : wr) if (l > wlevel[b]) wlevel[b] = l;
    }
    // a buffer's readers must come after ALL its writers: raise readers to max writer level + 1
    for (bool changed = true; changed;) {
      changed = false;
      std::fill(wlevel.begin(), wlevel.end(), -1);
      for (int k = 0; k < no; ++k) { op_io(ops[k], rd, wr); for (int b : wr) if (lvl[k] > wlevel[b]) wlevel[b] = lvl[k]; }
      for (int k = 0; k < no; ++k) {
        op_io(ops[k], rd, wr);
        for (int b : rd) if (wlevel[b] >= 0 && lvl[k] <= wlevel[b]) { lvl[k] = wlevel[b] + 1; changed = true; }
      }
    }
    if (fused_sched_mode == 3) {
      // List scheduling under an LDS budget: level by level, ops on the critical path (no slack against the
      // as-late-as-possible levels) are placed unconditionally, the others -- in program order -- only while
      // the bytes live in the level stay under the budget.  Gives (nearly) the short critical path of the full
      // levels with (nearly) the footprint of program order, i.e. one more co-resident workgroup per CU.
      int L = 0;
      for (int k = 0; k < no; ++k) L = lvl[k] + 1 > L ? lvl[k] + 1 : L;
      std::vector<int> alap(no, L - 1), rlevel(nb, L);
      for (int k = no - 1; k >= 0; --k) {
        op_io(ops[k], rd, wr);
        int l = L - 1;
        for (int b : wr) if (rlevel[b] - 1 < l) l = rlevel[b] - 1;
        if (l < lvl[k]) l = lvl[k];
        alap[k] = l;
        for (int b : rd) if (l < rlevel[b]) rlevel[b] = l;
      }
      // per buffer: writers, readers; global buffers (read after the fused range) cost no LDS
      std::vector<std::vector<int>> writers(nb), readers(nb);
      for (int k = 0; k < no; ++k) { op_io(ops[k], rd, wr); for (int b : wr) writers[b].push_back(k); for (int b : rd) readers[b].push_back(k); }
      std::vector<char> is_glob(nb, 0);
      for (int k = no; k < (int)ops.size(); ++k) { op_io(ops[k], rd, wr); for (int b : rd) is_glob[b] = 1; }
      const int WTl = fused_sched_wt > 0 ? fused_sched_wt : 4;
      auto blen = [&](int b) { return is_glob[b] ? (size_t)0 : sizeof(real) * (((size_t)WTl * bufs[b].rows * (bufs[b].width + 2) + 3) / 4 * 4); };
      std::vector<int> sched(no, -1);
      int n_done = 0, delay = 0;
      for (int l = 0; n_done < no; ++l) {
        auto ready = [&](int k) {
          op_io(ops[k], rd, wr);
          for (int b : rd) for (int w : writers[b]) if (sched[w] < 0 || sched[w] >= l) return false;
          return true;
        };
        auto live_bytes = [&]() {       // buffers with a scheduled writer and a reader not scheduled before level l
          size_t tot = 0;
          for (int b = 0; b < nb; ++b) {
            bool written = false, needed = false;
            for (int w : writers[b]) written = written || sched[w] >= 0;
            if (!written) continue;
            for (int rr : readers[b]) needed = needed || sched[rr] < 0 || sched[rr] >= l;
            for (int w : writers[b]) needed = needed || sched[w] == l || sched[w] < 0;
            if (needed) tot += blen(b);
          }
          return tot;
        };
        std::vector<int> cand;
        for (int k = 0; k < no; ++k) if (sched[k] < 0 && ready(k)) cand.push_back(k);
        int placed_now = 0;
        std::stable_sort(cand.begin(), cand.end(), [&](int x, int y) { return alap[x] < alap[y]; });   // least slack first
        for (int k : cand) {
          sched[k] = l;
          if (live_bytes() <= fused_sched_budget) { ++n_done; ++placed_now; }
          else sched[k] = -1;
        }
        if (placed_now == 0 && !cand.empty()) { sched[cand[0]] = l; ++n_done; }   // budget too small: make progress
        (void)delay;
      }
      lvl = sched;
    }
    if (fused_sched_mode == 2) {
      // As late as possible within the same number of levels: ops with slack (the edge-stream MLPs, which do
      // not depend on the node stream) move next to their consumers, which shortens buffer live ranges and
      // so the LDS footprint of a tile (what decides how many workgroups share a CU).
      int L = 0;
      for (int k = 0; k < no; ++k) L = lvl[k] + 1 > L ? lvl[k] + 1 : L;
      std::vector<int> alap(no, L - 1), rlevel(nb, L);     // rlevel[b]: earliest level of a reader of b
      for (int k = no - 1; k >= 0; --k) {
        op_io(ops[k], rd, wr);
        int l = L - 1;
        for (int b : wr) if (rlevel[b] - 1 < l) l = rlevel[b] - 1;
        if (l < lvl[k]) l = lvl[k];
        alap[k] = l;
        for (int b : rd) if (l < rlevel[b]) rlevel[b] = l;
      }
      lvl = alap;
    }
    f_order.resize(no);
    for (int k = 0; k < no; ++k) f_order[k] = k;
    // (within a level the program order is kept)
    std::stable_sort(f_order.begin(), f_order.end(), [&](int x, int y) { return lvl[x] < lvl[y]; });
    f_level.assign(no, 0);
    for (int j = 0; j < no; ++j) f_level[j] = lvl[f_order[j]];
  }

  // LDS placement of the buffers for a tile of WT walkers: interval colouring over live level ranges.
  size_t fused_layout(int WT, std::vector<dqmc::FusedBuf>& fb) const {
    const int nb = (int)bufs.size(), no = fused_n_ops;
    const int BIG = 1 << 30;
    std::vector<int> first(nb, BIG), last(nb, -1);
    std::vector<int> rd, wr;
    for (int j = 0; j < no; ++j) {
      op_io(ops[f_order[j]], rd, wr);
      for (int b : wr) { if (f_level[j] < first[b]) first[b] = f_level[j]; if (f_level[j] > last[b]) last[b] = f_level[j]; }
      for (int b : rd) { if (f_level[j] < first[b]) first[b] = f_level[j]; if (f_level[j] > last[b]) last[b] = f_level[j]; }
    }
    for (int k = no; k < (int)ops.size(); ++k) {      // consumers after the fused range: keep in HBM
      op_io(ops[k], rd, wr);
      for (int b : rd) last[b] = BIG;
    }
    fb.assign(nb, dqmc::FusedBuf{});
    // Placement = interval colouring: buffers in decreasing size, each at the lowest offset that does not
    // overlap an already placed buffer whose live range [first, last] intersects its own (largest-first beats
    // first-fit-in-time by ~10 % here, which decides how many workgroups share a CU).
    struct Seg { size_t off, len; int a, b; };
    std::vector<Seg> placed;
    const size_t base = 0;
    size_t peak = base;
    std::vector<int> order;
    for (int b = 0; b < nb; ++b) {
      if (first[b] == BIG) continue;            // not touched by the fused range
      dqmc::FusedBuf& f = fb[b];
      f.rows = bufs[b].rows; f.width = bufs[b].width;
      if (last[b] == BIG) { f.is_global = 1; continue; }
      f.is_global = 0;
      f.stride = bufs[b].width + 2;
      order.push_back(b);
    }
    auto len_of = [&](int b) { return ((size_t)WT * fb[b].rows * fb[b].stride + 3) / 4 * 4; };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return len_of(x) > len_of(y); });
    if (getenv("DQMC_FUSED_VERBOSE")) {
      const int n_levels = no ? f_level[no - 1] + 1 : 0;
      for (int l = 0; l < n_levels; ++l) {
        size_t live = 0;
        std::string who;
        for (int b : order) if (first[b] <= l && l <= last[b]) { live += len_of(b); who += " " + std::to_string(b) + ":" + std::to_string(len_of(b) * sizeof(real)); }
        fprintf(stderr, "[dqmc] WT=%d level %d live %zu B:%s\n", WT, l, live * sizeof(real), who.c_str());
      }
    }
    for (int b : order) {
      const size_t len = len_of(b);
      size_t off = base;
      for (bool moved = true; moved;) {
        moved = false;
        for (const Seg& s : placed)
          if (s.a <= last[b] && first[b] <= s.b && off < s.off + s.len && s.off < off + len) { off = s.off + s.len; moved = true; }
      }
      fb[b].off = (int)off;
      placed.push_back(Seg{off, len, first[b], last[b]});
      if (off + len > peak) peak = off + len;
    }
    return peak * sizeof(real);
  }

  int build_fused_plan() {
    fused_n_ops = 0;
    int n_f = -1;
    for (int k = 0; k < (int)ops.size(); ++k) {
      if (ops[k].kind == DQMC_OP_ORBITALS) { n_f = k + 1; break; }
      if (ops[k].kind == DQMC_OP_ATTENTION || ops[k].kind == DQMC_OP_SLOGDET || ops[k].kind == DQMC_OP_FINAL) return DQMC_OK;
    }
    if (n_f < 0) return DQMC_OK;
    for (int k = 0; k < n_f; ++k) {       // the LDS-resident kernel implements tanh / silu layers over electron senders only
      if (ops[k].kind == DQMC_OP_CONST) return DQMC_OK;
      if (ops[k].kind == DQMC_OP_LINEAR && ops[k].i[24] > 2) return DQMC_OK;
      if (ops[k].kind == DQMC_OP_FEAT_EE)
        for (int r = 0; r < ops[k].i[2]; ++r) if (h_itable[ops[k].i[1] + 2 * r + 1] < 0) return DQMC_OK;
    }
    fused_n_ops = n_f;
    if (fused_sched_mode == 3) {
      // largest per-level budget (= fewest levels) whose packed 4-walker tile leaves room for 4 workgroups per CU
      bool fit = false;
      for (int kb = 44; kb >= 24 && !fit; --kb) {
        fused_sched_budget = (size_t)kb * 1024;
        fused_schedule();
        std::vector<dqmc::FusedBuf> fbt;
        fit = fused_layout(4, fbt) + 16 + dqmc::fused2_scratch_bytes(4, N, sys.n_det, (int)sizeof(real), (int)n_itable) <= fused2_lds_quarter;
      }
      if (!fit) { fused_sched_mode = 1; fused_schedule(); fused_sched_mode = 3; }   // too big for that: full levels
    } else {
      fused_schedule();
    }
    if (!d_ops) {
      HIP_TRY(hipMalloc((void**)&d_ops, sizeof(dqmc_op) * ops.size()));
      HIP_TRY(hipMalloc((void**)&d_wpk_off, 2 * sizeof(int32_t) * ops.size()));
    }
    std::vector<dqmc_op> sched(fused_n_ops);
    for (int j = 0; j < fused_n_ops; ++j) sched[j] = ops[f_order[j]];
    HIP_TRY(hipMemcpy(d_ops, sched.data(), sizeof(dqmc_op) * fused_n_ops, hipMemcpyHostToDevice));
    const int rc = pack_fused_weights();
    if (rc) return rc;
    return build_fused2_plan();
  }

  // Work lists of the descriptor-driven fused kernel (kernel_fused2.hip): the units of every linear layer,
  // dealt to the 4 waves level by level (longest first onto the least loaded wave), structured ops for all
  // waves, one barrier per dependency level.  Tile layout [row][WT] needs a power-of-two tile.
  int build_fused2_plan() {
    fused2_WT = 0;
    if (fused_n_ops == 0) return DQMC_OK;
    const int cand[] = {16, 8, 4, 2, 1};
    std::vector<dqmc::FusedBuf> fb;
    auto with_scratch = [&](size_t act, int WT) { return (act + 15) / 16 * 16 + (size_t)dqmc::fused2_scratch_bytes(WT, N, sys.n_det, (int)sizeof(real), (int)n_itable); };
    if (fused_wt_req <= 0 && with_scratch(fused_layout(4, fb), 4) <= fused2_lds_quarter) {
      // 4 walkers per tile and 4 tiles per CU: for the batch sizes of the north star (4096 walkers = 1024 tiles =
      // 256 CUs x 4) the whole batch is ONE round of co-resident workgroups (measured fastest)
      fused2_WT = 4; fused2_lds = with_scratch(fused_layout(4, fb), 4);
    } else {
      for (int WT : cand) {
        if (fused_wt_req > 0 && WT != fused_wt_req) continue;
        const size_t bytes = with_scratch(fused_layout(WT, fb), WT);
        if (bytes <= (fused_wt_req > 0 ? (size_t)160 * 1024 : fused2_lds_budget)) { fused2_WT = WT; fused2_lds = bytes; break; }
      }
    }
    if (fused2_WT == 0) return DQMC_OK;
    const int WT = fused2_WT, n_waves = 4;
    fused2_shift = 0;
    while ((1 << fused2_shift) < WT) ++fused2_shift;
    std::vector<int32_t> words(2 * (size_t)fused_n_ops);
    HIP_TRY(hipMemcpy(words.data(), d_wpk_off, sizeof(int32_t) * words.size(), hipMemcpyDeviceToHost));
    std::vector<std::vector<dqmc::FDesc>> lists(n_waves);
    struct Unit { dqmc::FDesc d; long cost; int opidx; };
    std::vector<Unit> level_units;
    std::vector<int> level_generic;
    auto flush_level = [&]() {
      // units go longest first onto the least loaded wave
      std::vector<Unit> jobs(level_units);
      std::stable_sort(jobs.begin(), jobs.end(), [](const Unit& x, const Unit& y) { return x.cost > y.cost; });
      long load[4] = {0, 0, 0, 0};
      for (const Unit& jb : jobs) {
        int best = 0;
        for (int w = 1; w < n_waves; ++w) if (load[w] < load[best]) best = w;
        lists[best].push_back(jb.d);
        load[best] += jb.cost;
      }
      for (int j : level_generic)
        for (int w = 0; w < n_waves; ++w) { dqmc::FDesc g{}; g.kind = 3; g.op = j; lists[w].push_back(g); }
      for (int w = 0; w < n_waves; ++w) { dqmc::FDesc b{}; b.kind = 2; lists[w].push_back(b); }
      level_units.clear(); level_generic.clear();
    };
    for (int j = 0; j < fused_n_ops; ++j) {
      const dqmc_op& op = ops[f_order[j]];
      const int32_t* i = op.i;
      if (op.kind != DQMC_OP_LINEAR) {
        level_generic.push_back(j);
      } else {
        const int ldw = pad4(i[21]), Rtot = WT * i[20];
        const int NRB = (Rtot + 15) / 16, NCB = (ldw + 15) / 16, n_cg = (NCB + 1) / 2;
        int rpu = NRB * n_cg / n_waves;
        rpu = rpu < 1 ? 1 : (rpu > 4 ? 4 : rpu);
        dqmc::FDesc t{};
        t.kind = 1; t.op = j; t.n_pieces = i[0]; t.rtot = Rtot; t.ldw = ldw;
        long kq = 0;
        bool bf = op_bf[j] != 0;
        for (int p = 0; p < i[0]; ++p) {
          const dqmc::FusedBuf& sb = fb[i[1 + 4 * p]];
          if (sb.is_global) { fused2_WT = 0; return DQMC_OK; }
          t.a_base[p] = sb.off + i[2 + 4 * p] * WT * sb.stride;
          t.a_stride[p] = sb.stride;
          if ((t.a_base[p] | t.a_stride[p]) & 1) bf = false;        // (8-byte LDS reads of the A octets)
        }
        if (op_bf[j] && !bf) return fail(DQMC_E_UNSUPPORTED, "fused plan: odd LDS offset under a bf16-packed layer");
        for (int p = 0; p < i[0]; ++p) {
          if (bf) {       // bf16 matrix pipe: octets of k, chunks of 32 k
            t.a_ks[p] = i[3 + 4 * p] / 8;
            t.a_nq[p] = (i[3 + 4 * p] + 31) / 32;
          } else {
            t.a_ks[p] = pad4(i[3 + 4 * p]) / 4;
            t.a_nq[p] = (t.a_ks[p] + 3) / 4;
          }
          if (i[4 + 4 * p]) t.bcast |= 1 << p;
          kq += t.a_nq[p];
        }
        t.qstride = bf ? NCB * 192 : NCB * 64;
        t.bias_off = i[23];
        const dqmc::FusedBuf& db = fb[i[17]];
        t.flags = (i[24] & 3) | (i[27] ? 4 : 0);
        if (db.is_global) { t.flags |= 8; t.dst_base = i[17]; t.g_r0 = i[18]; t.g_col0 = i[19]; }
        else { t.dst_base = db.off + i[18] * WT * db.stride + i[19]; t.dst_stride = db.stride; }
        t.res_base = -1;
        if (i[25] >= 0) {
          const dqmc::FusedBuf& rb = fb[i[25]];
          if (rb.is_global) { fused2_WT = 0; return DQMC_OK; }
          t.res_base = rb.off + i[26] * WT * rb.stride + i[19];
          t.res_stride = rb.stride;
        }
        for (int rb0 = 0; rb0 < NRB; rb0 += rpu)
          for (int cg = 0; cg < n_cg; ++cg) {
            Unit u{t, 0, f_order[j]};
            u.d.ma = (NRB - rb0) < rpu ? (NRB - rb0) : rpu;
            u.d.row0 = rb0 * 16;
            u.d.col0 = cg * 32;
            u.d.w_off = words[2 * j] / 4 + (cg * 2) * (bf ? 192 : 64);
            u.d.w_cb1 = (cg * 2 + 1 < NCB) ? (bf ? 192 : 64) : 0;
            if ((rb0 + u.d.ma) * 16 <= Rtot && cg * 32 + 32 <= ldw) u.d.flags |= 16;
            if (bf) u.d.kind = 6;
            // small layers take the lean unit body (kernel_fused2.hip: fused2_unit_lean / FusedBfUnit::lean)
            if (fused_lean && u.d.ma == 1 && i[0] == 1 && !u.d.bcast && t.a_nq[0] <= (bf ? 1 : dqmc::FusedGroup<real>::P) && !(t.flags & 8) && (i[24] & 3) <= 1) u.d.kind = bf ? 7 : 5;
            u.cost = 12 + (long)u.d.ma * ((bf ? 3 : 4) * kq + 6);     // ~ fixed setup + MFMA quads / chunks + epilogue, in 100-cycle units
            level_units.push_back(u);
          }
      }
      if (j + 1 == fused_n_ops || f_level[j + 1] > f_level[j]) flush_level();
    }
    fused2_ma1 = true;
    auto is_unit = [](int k) { return k == 1 || k == 5 || k == 6 || k == 7; };
    for (auto& l : lists) for (auto& dd : l) if (is_unit(dd.kind) && dd.ma != 1) fused2_ma1 = false;
    std::vector<dqmc::FDesc> flat;
    int32_t begin[8];
    // (plan_lists is recorded after the chaining below)
    for (int w = 0; w < n_waves; ++w) {
      begin[w] = (int32_t)flat.size();
      begin[4 + w] = -1;
      int last_unit = -1;                          // chain the units of the list: each prefetches the next one's first weights
      for (size_t k = 0; k < lists[w].size(); ++k) {
        if (!is_unit(lists[w][k].kind)) continue;
        if (last_unit < 0) begin[4 + w] = begin[w] + (int32_t)k;
        last_unit = (int)k;
      }
      last_unit = -1;
      for (size_t k = lists[w].size(); k-- > 0;) {     // every unit carries the first-group parameters of the unit after it
        dqmc::FDesc& u = lists[w][k];
        if (!is_unit(u.kind)) continue;
        const dqmc::FDesc& nx = last_unit < 0 ? u : lists[w][last_unit];
        const bool nx_bf = nx.kind == 6 || nx.kind == 7;          // (its first group: the three planes of its first chunk)
        u.nx_w_off = nx.w_off; u.nx_cb1 = nx.w_cb1; u.nx_qstride = nx_bf ? 64 : nx.qstride; u.nx_nq = nx_bf ? 3 : nx.a_nq[0];
        last_unit = (int)k;
      }
      flat.insert(flat.end(), lists[w].begin(), lists[w].end());
      dqmc::FDesc e{}; e.kind = 0; flat.push_back(e);
    }
    plan_lists = lists;
    if (d_descs) { HIP_TRY(hipFree(d_descs)); d_descs = nullptr; }
    HIP_TRY(hipMalloc((void**)&d_descs, sizeof(dqmc::FDesc) * flat.size()));
    if (!d_wave_begin) HIP_TRY(hipMalloc((void**)&d_wave_begin, sizeof(int32_t) * 8));
    if (!d_fbufs2) HIP_TRY(hipMalloc((void**)&d_fbufs2, sizeof(dqmc::FusedBuf) * bufs.size()));
    HIP_TRY(hipMemcpy(d_descs, flat.data(), sizeof(dqmc::FDesc) * flat.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_wave_begin, begin, sizeof(begin), hipMemcpyHostToDevice));
    fbufs2_h = fb;
    fbufs2_uploaded = false;
    if (dqmc::fused2_set_lds_limit<real>(fused2_lds) != 0) { fused2_WT = 0; return DQMC_OK; }
    return DQMC_OK;
  }

  // LDS offset for the Slater matrices of the sub-step tail: dead activation space clear of the backflow
  // buffer, which the tail still reads; -1 if there is none (then the staged path runs).
  int substep_mat_off() const {
    int orb = -1;
    for (int j = 0; j < fused_n_ops; ++j) if (ops[f_order[j]].kind == DQMC_OP_ORBITALS) orb = f_order[j];
    if (orb < 0 || fused2_WT == 0) return -1;
    const dqmc::FusedBuf& bfb = fbufs2_h[ops[orb].i[0]];
    const size_t need = (size_t)fused2_WT * sys.n_det * N * N;
    const size_t bf_len = (size_t)fused2_WT * bfb.rows * bfb.stride;
    const size_t act_end = (fused2_lds - (size_t)dqmc::fused2_scratch_bytes(fused2_WT, N, sys.n_det, (int)sizeof(real), (int)n_itable)) / sizeof(real);
    if (bfb.is_global) return -1;
    if ((size_t)bfb.off >= need) return 0;
    if ((size_t)bfb.off + bf_len + need <= act_end) return (int)((size_t)bfb.off + bf_len);
    return -1;
  }

  int run_fused2(const real* r, const real* R, int B, dqmc::LaneInfo li, const dqmc::FusedMc* mc = nullptr) {
    bool changed = !fbufs2_uploaded;       // the buffer table goes to the device only when an offset moved (not once per sub-step)
    for (size_t b = 0; b < bufs.size(); ++b) {
      if (fbufs2_h[b].goff != (long)buf_off[b]) changed = true;
      fbufs2_h[b].goff = (long)buf_off[b];
    }
    if (changed) {
      HIP_TRY(hipMemcpyAsync(d_fbufs2, fbufs2_h.data(), sizeof(dqmc::FusedBuf) * bufs.size(), hipMemcpyHostToDevice, st));
      if (!fbufs2_uploaded) HIP_TRY(hipStreamSynchronize(st));   // (the host vector may change before an asynchronous copy from pageable memory ran)
      fbufs2_uploaded = true;
    }
    dqmc::Fused2Args<real> a{};
    a.descs = d_descs; a.wave_begin = d_wave_begin; a.ops = d_ops; a.fbufs = d_fbufs2;
    a.w = d_w; a.wpk = d_wpk; a.itable = d_it; a.ws = d_ws; a.r = r; a.R = R;
    a.B = B; a.WT = fused2_WT; a.wt_shift = fused2_shift; a.n_up = sys.n_up; a.n_nuc = sys.n_nuc; a.K = sys.n_det;
    a.li = li; a.eps = sys.norm_eps; a.prof = fused_dbg ? d_prof : nullptr;
    a.prof_wg = (fused_dbg & 2) ? d_prof + 9 * ops.size() + 80 + 1024 : nullptr;
    a.scratch_off = (int)((fused2_lds - (size_t)dqmc::fused2_scratch_bytes(fused2_WT, N, sys.n_det, (int)sizeof(real), (int)n_itable)) / sizeof(real));
    a.it_off = (int)(fused2_lds - (size_t)((4 * n_itable + 15) / 16 * 16));
    a.n_it = (int)n_itable;
    a.ma1 = fused2_ma1 ? 1 : 0;
    a.stagger_div = fused_stagger_div; a.prio_mode = fused_prio;
    if (mc) a.mc = *mc;
    double flops = 0;
    for (int k = 0; k < fused_n_ops; ++k)
      if (ops[k].kind == DQMC_OP_LINEAR) {
        int ktot = 0;
        for (int p = 0; p < ops[k].i[0]; ++p) ktot += ops[k].i[3 + 4 * p];
        flops += 2.0 * B * ops[k].i[20] * (double)ktot * ops[k].i[21];
      }
    t_begin(mc ? "fused_substep" : "fused_psi", flops);
    int occ = fused_occ_req;
    if (occ <= 0) { const size_t per_cu = (size_t)160 * 1024 / (fused2_lds ? fused2_lds : 1); occ = per_cu >= 4 ? 4 : (per_cu >= 3 ? 3 : 2); }
    dqmc::launch_fused2_value<real>(st, a, (B + fused2_WT - 1) / fused2_WT, fused2_lds, occ);
    t_end();
    return DQMC_OK;
  }

  // piece pl (0..2) of the three-bf16 split of a float (round to nearest even, residuals exact: common.h bf_split8)
  static uint16_t bf16_piece(float v, int pl) {
    auto rne = [](float f) -> uint16_t {
      uint32_t u; memcpy(&u, &f, 4);
      if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
      return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    auto up = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    uint16_t h = rne(v);
    for (int k = 0; k < pl; ++k) { v = v - up(h); h = rne(v); }
    return h;
  }
  // Fragment-major copy of the Linear weights: [k/4][column block][lane] so that one wave load
  // of 64 consecutive elements is exactly the MFMA B operand (B[k = l>>4][col = l&15]).
  // Per scheduled op two words go to the device: packed-weight offset and "barrier after".
  int pack_fused_weights() {
    std::vector<int32_t> words(2 * (size_t)fused_n_ops, 0);
    std::vector<real> pk;
    op_bf.assign((size_t)fused_n_ops, 0);
    for (int j = 0; j < fused_n_ops && fused_bf && sizeof(real) == 4; ++j) {
      // a layer goes to the bf16 matrix pipe when every piece is whole octets wide and the chunks of 32 k
      // (12 MFMAs of 16 cycles + the operand split) cost less than its k-steps of 4 (2 MFMAs of 32 cycles)
      const dqmc_op& op = ops[f_order[j]];
      if (op.kind != DQMC_OP_LINEAR) continue;
      bool ok = true;
      int chunks = 0, ksteps = 0;
      for (int p = 0; p < op.i[0]; ++p) {
        const int K = op.i[3 + 4 * p];
        if (K % 8 != 0 || K == 0 || (bufs[op.i[1 + 4 * p]].width & 1)) ok = false;      // (even LDS row stride: 8-byte reads of the A octets)
        chunks += (K + 31) / 32; ksteps += (K + 3) / 4;
      }
      op_bf[j] = ok && (fused_bf >= 2 ? 9 * chunks < 2 * ksteps : 4 * chunks <= ksteps);     // (option value 2: the stricter rule; measured 110.8 vs 108.8 us)
    }
    for (int j = 0; j < fused_n_ops; ++j) {
      const dqmc_op& op = ops[f_order[j]];
      words[2 * j + 1] = (j + 1 == fused_n_ops || f_level[j + 1] > f_level[j]) ? 1 : 0;
      if (op.kind != DQMC_OP_LINEAR) continue;
      const int32_t* i = op.i;
      const int ldw = pad4(i[21]), NCB = (ldw + 15) / 16;
      // quad-interleaved: [piece][quad of 4 k-steps][column block][lane][k-step in quad]; every
      // piece is zero padded to whole quads, so one 16-byte load per lane feeds 4 MFMA k-steps
      words[2 * j] = (int32_t)pk.size();
      const real* W = wtmp.data() + i[22];
      int row0 = 0;
      if (op_bf[j]) {
        // bf16 plane layout: [piece][chunk of 32 k][column block][plane][lane][4 words]; word jj of lane l holds
        // k = 32 c + 8 (l >> 4) + 2 jj (low half) and + 1 (high half) of column cb 16 + (l & 15)
        for (int p = 0; p < i[0]; ++p) {
          const int K = i[3 + 4 * p], NC = (K + 31) / 32;
          for (int c = 0; c < NC; ++c)
            for (int cb = 0; cb < NCB; ++cb)
              for (int pl = 0; pl < 3; ++pl)
                for (int l = 0; l < 64; ++l)
                  for (int jj = 0; jj < 4; ++jj) {
                    uint32_t word = 0;
                    for (int h = 0; h < 2; ++h) {
                      const int k = 32 * c + 8 * (l >> 4) + 2 * jj + h, col = cb * 16 + (l & 15);
                      const float wv = (k < K && col < ldw) ? (float)W[(size_t)(row0 + k) * ldw + col] : 0.0f;
                      word |= (uint32_t)bf16_piece(wv, pl) << (16 * h);
                    }
                    real as_real;
                    memcpy(&as_real, &word, 4);       // (float engines only: sizeof(real) == 4)
                    pk.push_back(as_real);
                  }
          row0 += K;
        }
        continue;
      }
      for (int p = 0; p < i[0]; ++p) {
        const int KS = pad4(i[3 + 4 * p]) / 4, NQ = (KS + 3) / 4;
        for (int q = 0; q < NQ; ++q)
          for (int cb = 0; cb < NCB; ++cb)
            for (int l = 0; l < 64; ++l)
              for (int jj = 0; jj < 4; ++jj) {
                const int ks = q * 4 + jj, row = row0 + ks * 4 + (l >> 4), col = cb * 16 + (l & 15);
                pk.push_back((ks < KS && col < ldw) ? W[(size_t)row * ldw + col] : (real)0);
              }
        row0 += KS * 4;
      }
    }
    if (pk.size() > wpk_cap) {
      if (d_wpk) HIP_TRY(hipFree(d_wpk));
      HIP_TRY(hipMalloc((void**)&d_wpk, sizeof(real) * pk.size()));
      wpk_cap = pk.size();
    }
    HIP_TRY(hipMemcpyAsync(d_wpk, pk.data(), sizeof(real) * pk.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_wpk_off, words.data(), sizeof(int32_t) * words.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQMC_OK;
  }

  static bool lanes_supported(int TP) {
    return TP == 1 || TP == 16 || TP == 32 || TP == 48 || TP == 64 || TP == 96 || TP == 128;
  }

  int plan(int B, int TP) {
    buf_off.resize(bufs.size());
    size_t off = 0;
    auto bump = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    for (size_t k = 0; k < bufs.size(); ++k)
      buf_off[k] = bump(sizeof(real) * (size_t)B * bufs[k].rows * lanes_of((int)k, TP) * bufs[k].width);
    {
      int max_ldw = 4;
      for (const auto& o : ops) if (o.kind == DQMC_OP_LINEAR && pad4(o.i[21]) > max_ldw) max_ldw = pad4(o.i[21]);
      off_z = bump(sizeof(real) * (size_t)B * TP * max_ldw);       // per-walker pre-activation rows of the split linear layers
    }
    off_logdet = bump(sizeof(double) * (size_t)B * sys.n_det * TP);
    off_signk = bump(sizeof(int32_t) * (size_t)B * sys.n_det);
    off_cond = bump(sizeof(double) * (size_t)B * sys.n_det);      // conditioning record per determinant (Laplacian mode)
    off_kappa = bump(sizeof(double) * (size_t)B);                 // ... and its psi-weighted sum per walker
    if (ph_n && TP > 1) off_phq = bump(sizeof(double) * (size_t)B * N * dqmc::PH_STRIDE);
    if (off > ws_bytes) {
      if (d_ws) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_ws)); d_ws = nullptr; ws_bytes = 0; }
      hipError_t e = hipMalloc((void**)&d_ws, off);
      if (e != hipSuccess) return fail(DQMC_E_NOMEM, "workspace of " + std::to_string(off) + " bytes: " + hipGetErrorString(e));
      ws_bytes = off;
    }
    last_B = B; last_TP = TP;
    return DQMC_OK;
  }
  real* bptr(int b) { return reinterpret_cast<real*>(d_ws + buf_off[b]); }

  // Workspace bytes per walker for an evaluation with TP lanes (what plan() allocates, without alignment slack).
  size_t ws_bytes_per_walker(int TP) const {
    size_t b = 0;
    for (size_t k = 0; k < bufs.size(); ++k) b += sizeof(real) * (size_t)bufs[k].rows * lanes_of((int)k, TP) * bufs[k].width;
    size_t zrow = 4;
    for (const auto& o : ops) if (o.kind == DQMC_OP_LINEAR && (size_t)pad4(o.i[21]) > zrow) zrow = (size_t)pad4(o.i[21]);
    return b + sizeof(real) * zrow * TP + sizeof(double) * (size_t)sys.n_det * TP + sizeof(int32_t) * (size_t)sys.n_det +
           sizeof(double) * ((size_t)sys.n_det + 1) +           // off_cond, off_kappa
           (ph_n && TP > 1 ? sizeof(double) * (size_t)N * dqmc::PH_STRIDE : 0);
  }

  // Execute the layer program on B walkers, in chunks if the activation workspace of the whole batch would exceed
  // ws_budget (benzene/Psiformer in Laplacian mode needs ~0.2 GB per walker: 2048 walkers per GPU do not fit 288 GB
  // at once).  Chunks are still thousands of MFMA row blocks each.
  int run(const real* r, const real* R, int B, bool laplacian, real* logpsi, int32_t* sign, real* e_loc, real* stats,
          real* grad) {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    const int T = laplacian ? 3 * N + 2 : 1, TP = laplacian ? (T + 15) / 16 * 16 : 1;
    const size_t per = ws_bytes_per_walker(TP);
    long chunk = per ? (long)(ws_budget / per) : B;
    if (chunk < 1) chunk = 1;
    if (laplacian) last_chunks[0] = chunk >= B ? 1 : (int)((B + chunk - 1) / chunk);
    if (chunk >= B) {
      if (laplacian && graph_fits(B))
        return run_graphed(r, R, B, logpsi, sign, e_loc, stats, grad);
      return run_chunk(r, R, B, laplacian, logpsi, sign, e_loc, stats, B, grad, 0);
    }
    for (int b0 = 0; b0 < B; b0 += (int)chunk) {
      const int nb = (B - b0) < chunk ? (B - b0) : (int)chunk;
      const int rc = run_chunk(r + (size_t)b0 * N * 3, R, nb, laplacian, logpsi ? logpsi + b0 : nullptr, sign ? sign + b0 : nullptr,
                               e_loc ? e_loc + b0 : nullptr, stats ? stats + b0 : nullptr, B, grad ? grad + (size_t)b0 * 3 * N : nullptr, b0);
      if (rc) return rc;
    }
    return DQMC_OK;
  }

  // one forward-Laplacian pass through its captured graph (see pass_graph above)
  int run_graphed(const real* r, const real* R, int B, real* logpsi, int32_t* sign, real* e_loc, real* stats, real* grad) {
    if (std::find(graph_warm.begin(), graph_warm.end(), B) == graph_warm.end()) {
      graph_warm.push_back(B);
      return run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
    }
    const void* key[7] = {r, R, logpsi, sign, e_loc, stats, grad};
    PassGraph* hit = nullptr;
    for (auto& g : pgraphs)
      if (g.B == B && g.flag == flag_on && g.ws == d_ws && g.flagp == d_flag && g.epoch == graph_epoch && !memcmp(g.p, key, sizeof(key))) hit = &g;
    if (!hit) {
      // a caller that hands over different buffers on every call would pay a capture (milliseconds) per pass: once captures
      // clearly outnumber replays, the context goes back to eager launches for good
      if (graph_captures >= 8 && graph_hits < graph_captures) {
        graph_broken = true;
        drop_graphs();
        return run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
      }
      ++graph_captures;
      if (!st_g) {
        HIP_TRY(hipStreamCreateWithFlags(&st_g, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev_g0, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_g1, hipEventDisableTiming));
      }
      hipStream_t caller = st;
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(st_g, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        graph_broken = true;
        return run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
      }
      st = st_g;                                   // (run_chunk launches on `st` and forks its side streams from it)
      const int rc = run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
      st = caller;
      const hipError_t ee = hipStreamEndCapture(st_g, &graph);
      hipGraphExec_t exec = nullptr;
      if (rc || ee != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        graph_broken = true;                       // this program / runtime does not capture: eager from now on
        if (rc) return rc;
        return run_chunk(r, R, B, true, logpsi, sign, e_loc, stats, B, grad, 0);
      }
      (void)hipGraphDestroy(graph);
      if (pgraphs.size() >= 12) {                  // (a handful of batch sizes x output buffers; evict the least recently used)
        size_t old = 0;
        for (size_t k = 1; k < pgraphs.size(); ++k) if (pgraphs[k].used < pgraphs[old].used) old = k;
        if (pgraphs[old].exec) (void)hipGraphExecDestroy((hipGraphExec_t)pgraphs[old].exec);
        pgraphs.erase(pgraphs.begin() + (long)old);
      }
      PassGraph g{};
      memcpy(g.p, key, sizeof(key));
      g.B = B; g.flag = flag_on; g.ws = d_ws; g.flagp = d_flag; g.epoch = graph_epoch; g.exec = exec;
      pgraphs.push_back(g);
      hit = &pgraphs.back();
    }
    else ++graph_hits;
    hit->used = ++graph_clock;
    HIP_TRY(hipEventRecord(ev_g0, st));
    HIP_TRY(hipStreamWaitEvent(st_g, ev_g0, 0));
    HIP_TRY(hipGraphLaunch((hipGraphExec_t)hit->exec, st_g));
    HIP_TRY(hipEventRecord(ev_g1, st_g));
    HIP_TRY(hipStreamWaitEvent(st, ev_g1, 0));
    // the host-side layout (buf_off, off_*, last_B, last_TP) follows the replayed pass, so that dqmc_debug_read after it
    // addresses what the graph wrote; the slab is already large enough (the graph was captured on it), nothing is reallocated
    return plan(B, (3 * N + 2 + 15) / 16 * 16);
  }

  int run_chunk(const real* r, const real* R, int B, bool laplacian, real* logpsi, int32_t* sign, real* e_loc, real* stats,
                long stats_ld, real* grad, int b_offset) {
    dqmc::LaneInfo li;
    li.N = N;
    li.T = laplacian ? 3 * N + 2 : 1;
    li.TP = laplacian ? (li.T + 15) / 16 * 16 : 1;
    if (!lanes_supported(li.TP)) return fail(DQMC_E_UNSUPPORTED, "no kernel instance for " + std::to_string(li.TP) + " lanes");
    int rc = plan(B, li.TP);
    if (rc) return rc;
    // pseudo-Hamiltonian: the local-energy pass (not the plain gradient of psi_grad / the Langevin sampler) seeds its
    // derivative lanes with the per-electron Cholesky factors of A(r_i) (ecp/pseudo_hamiltonian.py:115-146)
    const double* phq = nullptr;
    if (laplacian && ph_n && e_loc && !ph_skip) {
      double* q = reinterpret_cast<double*>(d_ws + off_phq);
      dqmc::launch_ph_coeffs<real>(st, r, R, d_ph_nuc, ph_n, d_ph_loc, d_ph_l2, ph_grid, ph_rmax, B, N, q);
      phq = q;
    }
    // edge-stream ops (destination carries pair-compact lanes) go to the companion stream in Laplacian mode, independent
    // node branches to two more (analyse_streams); every op records an event, readers on other streams wait for the
    // events of the buffers they read
    const bool dual = laplacian && dual_stream && !timing_serial() && std::any_of(compact.begin(), compact.end(), [](char c) { return c != 0; });
    const bool multi = dual && multi_stream && (long)B * N * li.TP <= (1L << 20);      // (large batches fill the GPU kernel by kernel)
    struct BufEv { hipEvent_t ev; int sid; };
    std::vector<std::vector<BufEv>> buf_w(dual ? bufs.size() : 0);
    size_t ev_next = 0;
    hipStream_t sl[4] = {st, st, st, st};
    hipEvent_t last_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t zb_reader = nullptr;                    // last reader of the per-walker pre-activation scratch
    auto new_event = [&](hipEvent_t* out) -> int {
      if (ev_next == ms_events.size()) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ms_events.push_back(e); }
      *out = ms_events[ev_next++];
      return DQMC_OK;
    };
    if (dual) {
      if (!st2) HIP_TRY(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
      if (!ev_fork) HIP_TRY(hipEventCreate(&ev_fork));
      sl[1] = st2;
      HIP_TRY(hipEventRecord(ev_fork, st));            // inputs ready, the previous evaluation's readers done
      HIP_TRY(hipStreamWaitEvent(st2, ev_fork, 0));
      if (multi)
        for (int x = 0; x < 2; ++x) {
          if (!st_extra[x]) HIP_TRY(hipStreamCreateWithFlags(&st_extra[x], hipStreamNonBlocking));
          sl[2 + x] = st_extra[x];
          HIP_TRY(hipStreamWaitEvent(st_extra[x], ev_fork, 0));
        }
    }
    auto sid_of = [&](size_t opi) -> int {
      if (!dual) return 0;
      const int sid = op_sid[opi];
      if (sid < 0) return 0;
      return (sid >= 2 && !multi) ? 0 : sid;
    };
    std::vector<int> rd_b, wr_b;
    auto wait_buf = [&](int b, int sid) -> int {        // stream slot `sid` is about to read buffer b
      for (const BufEv& w : buf_w[b]) if (w.sid != sid) HIP_TRY(hipStreamWaitEvent(sl[sid], w.ev, 0));
      return DQMC_OK;
    };
    auto before = [&](const dqmc_op& o, int sid) -> int {
      if (!dual) return DQMC_OK;
      op_io(o, rd_b, wr_b);
      for (int b : rd_b) { const int rcw = wait_buf(b, sid); if (rcw) return rcw; }
      return DQMC_OK;
    };
    auto after = [&](const dqmc_op& o, int sid) -> int {
      if (!dual) return DQMC_OK;
      op_io(o, rd_b, wr_b);
      hipEvent_t e;
      { const int rce = new_event(&e); if (rce) return rce; }
      HIP_TRY(hipEventRecord(e, sl[sid]));
      last_ev[sid] = e;
      for (int b : wr_b) buf_w[b].push_back(BufEv{e, sid});
      return DQMC_OK;
    };
    // second layers of chained MLPs that actually ran inside their parent's launch IN THIS PASS (the chained kernel has
    // instances for some lane counts only, and needs parent and child in the same lane layout: otherwise both layers run
    // as ordinary LINEAR ops)
    std::vector<char> ran_with_parent(ops.size(), 0);
    size_t first_op = 0;
    if (!laplacian && fused_n_ops > 0 && fused2_WT > 0 && (fused_enabled >= 2 || (fused_enabled == 1 && fused_pays(B)))) {
      rc = run_fused2(r, R, B, li);
      if (rc) return rc;
      first_op = (size_t)fused_n_ops;
    }
    for (size_t opi = first_op; opi < ops.size(); ++opi) {
      const dqmc_op& op = ops[opi];
      const int32_t* i = op.i;
      if (ran_with_parent[opi]) continue;               // second layer of a chained MLP: ran with its parent
      const int sid = sid_of(opi);
      const hipStream_t so = sl[sid];
      { const int rcb = before(op, sid); if (rcb) return rcb; }
      switch (op.kind) {
        case DQMC_OP_FEAT_EN:
          t_begin("feat", 0, so);
          dqmc::launch_feat_en<real>(so, r, R, bptr(i[0]), B, sys.n_nuc, sys.n_up, bufs[i[0]].width, li, sys.norm_eps, i[1], i[2], phq);
          t_end();
          break;
        case DQMC_OP_FEAT_EE:
          t_begin("feat", 0, so);
          dqmc::launch_feat_ee<real>(so, r, R, d_it + i[1], bptr(i[0]), B, i[2], li, sys.norm_eps, i[3],
                                     li.TP > 1 && compact[i[0]], phq);
          t_end();
          break;
        case DQMC_OP_LINEAR: {
          dqmc::LinArgs<real> a{};
          a.cfg_bf = linear_bf; a.cfg_bkx = linear_bkx; a.cfg_f64_split = linear_f64_split;
          a.n_pieces = i[0];
          int ktot = 0, w_row = 0, n_bc = 0;
          for (int p = 0; p < i[0]; ++p) {
            const int sb = i[1 + 4 * p];
            a.piece[p].src = bptr(sb);
            a.piece[p].ld = bufs[sb].width;
            a.piece[p].rpw = bufs[sb].rows;
            a.piece[p].r0 = i[2 + 4 * p];
            a.piece[p].K = pad4(i[3 + 4 * p]);
            a.piece[p].bcast = i[4 + 4 * p];
            a.piece[p].w_row = w_row;
            w_row += pad4(i[3 + 4 * p]);
            n_bc += i[4 + 4 * p] ? 1 : 0;
            ktot += i[3 + 4 * p];
          }
          a.W = d_w + i[22];
          a.ldw = pad4(i[21]);
          a.bias = i[23] >= 0 ? d_w + i[23] : nullptr;
          a.dst = bptr(i[17]);
          a.ld_dst = bufs[i[17]].width; a.rpw_dst = bufs[i[17]].rows; a.r0_dst = i[18]; a.col0_dst = i[19];
          a.res = i[25] >= 0 ? bptr(i[25]) : nullptr;
          if (i[25] >= 0) { a.ld_res = bufs[i[25]].width; a.rpw_res = bufs[i[25]].rows; a.r0_res = i[26]; }
          a.res_scale = i[27] ? (real)0.70710678118654752440 : (real)1;
          a.act = i[24]; a.nrows = i[20]; a.B = B; a.T = li.T; a.TP = li.TP;
          if (li.TP > 1 && compact[i[17]]) { a.T = dqmc::PAIR_LANES; a.TP = dqmc::PAIR_LANES; }   // row-wise op on edge rows
          if (mlp_child[opi] >= 0 && dqmc::linear_chain_supported(a.TP, a.ldw, pad4(ops[mlp_child[opi]].i[21])) &&
              compact[ops[mlp_child[opi]].i[17]] == compact[i[17]]) {
            // hidden layer + output layer of a row-wise MLP in one launch (the hidden activations stay in LDS)
            const dqmc_op& ch = ops[mlp_child[opi]];
            const int32_t* c = ch.i;
            { const int rcb = before(ch, sid); if (rcb) return rcb; }
            a.W2 = d_w + c[22]; a.ldw2 = pad4(c[21]); a.bias2 = c[23] >= 0 ? d_w + c[23] : nullptr; a.act2 = c[24];
            a.dst = bptr(c[17]);
            a.ld_dst = bufs[c[17]].width; a.rpw_dst = bufs[c[17]].rows; a.r0_dst = c[18]; a.col0_dst = c[19];
            a.res = c[25] >= 0 ? bptr(c[25]) : nullptr;
            if (c[25] >= 0) { a.ld_res = bufs[c[25]].width; a.rpw_res = bufs[c[25]].rows; a.r0_res = c[26]; }
            a.res_scale = c[27] ? (real)0.70710678118654752440 : (real)1;
            t_begin("linear", 2.0 * (double)B * i[20] * li.T * ((double)ktot * i[21] + (double)c[3] * c[21]), so);
            dqmc::launch_linear_chain<real>(so, a);
            t_end();
            ran_with_parent[mlp_child[opi]] = 1;
            { const int rca = after(op, sid); if (rca) return rca; }
            { const int rca = after(ch, sid); if (rca) return rca; }
            continue;
          }
          if (split_bcast && n_bc > 0 && n_bc < i[0] && i[20] > 1 && !(li.TP > 1 && compact[i[17]])) {
            // Per-walker (broadcast) pieces -- the spin means of the node update, reference gnn/update_features.py:64-106 --
            // contribute the same row to every electron of a walker: their product with W is computed ONCE per walker into
            // a scratch row and enters the main launch as an addend of the pre-activation (for LiH 57 %, for N2 62 % of the
            // K range of the g layers; the dense path multiplied it once per electron).
            dqmc::LinArgs<real> z = a;
            dqmc::LinArgs<real> m = a;
            z.n_pieces = m.n_pieces = 0;
            for (int p = 0; p < i[0]; ++p) {
              if (a.piece[p].bcast) z.piece[z.n_pieces++] = a.piece[p];
              else m.piece[m.n_pieces++] = a.piece[p];
            }
            real* zb = reinterpret_cast<real*>(d_ws + off_z);
            z.bias = nullptr; z.act = 0; z.res = nullptr; z.pre = nullptr;
            z.dst = zb; z.ld_dst = a.ldw; z.rpw_dst = 1; z.r0_dst = 0; z.col0_dst = 0; z.nrows = 1;
            m.pre = zb; m.ld_pre = a.ldw;
            // the per-walker product runs where its inputs (the spin means) were produced, i.e. beside whatever the main
            // stream is still doing for this layer; the scratch row buffer is shared by all layers: its previous reader
            // (the last g layer) must be done before it is overwritten
            int zsid = sid;
            if (dual) {
              for (int p = 0; p < i[0]; ++p)
                if (i[4 + 4 * p] && !buf_w[i[1 + 4 * p]].empty()) zsid = buf_w[i[1 + 4 * p]].back().sid;
              for (int p = 0; p < i[0]; ++p)
                if (i[4 + 4 * p]) { const int rcw = wait_buf(i[1 + 4 * p], zsid); if (rcw) return rcw; }
              if (zb_reader && zsid != sid) HIP_TRY(hipStreamWaitEvent(sl[zsid], zb_reader, 0));
            }
            t_begin("linear", 0, sl[zsid]);
            dqmc::launch_linear<real>(sl[zsid], z);
            t_end();
            if (dual && zsid != sid) {
              hipEvent_t ez;
              { const int rce = new_event(&ez); if (rce) return rce; }
              HIP_TRY(hipEventRecord(ez, sl[zsid]));
              last_ev[zsid] = ez;
              HIP_TRY(hipStreamWaitEvent(so, ez, 0));
            }
            t_begin("linear", 2.0 * (double)B * i[20] * li.T * (double)ktot * i[21], so);
            dqmc::launch_linear<real>(so, m);
            t_end();
            if (dual) {
              { const int rce = new_event(&zb_reader); if (rce) return rce; }
              HIP_TRY(hipEventRecord(zb_reader, so));
            }
            break;
          }
          t_begin("linear", 2.0 * (double)B * i[20] * li.T * (double)ktot * i[21], so);
          dqmc::launch_linear<real>(so, a);
          t_end();
          break;
        }
        case DQMC_OP_SPIN_MEAN:
          t_begin("graph", 0, so);
          dqmc::launch_spin_mean<real>(so, bptr(i[0]), bptr(i[1]), B, i[2], bufs[i[0]].width, li);
          t_end();
          break;
        case DQMC_OP_CONV:
          t_begin("graph", 0, so);
          dqmc::launch_conv<real>(so, bptr(i[0]), bufs[i[0]].rows, bufs[i[0]].width, bptr(i[1]), bufs[i[1]].rows, bufs[i[1]].width, bptr(i[2]),
                                  bufs[i[2]].width, i[3], d_it + i[4], i[5], i[6], B, li, li.TP > 1 && compact[i[0]]);
          t_end();
          break;
        case DQMC_OP_EDGE_SUM:
          t_begin("graph", 0, so);
          dqmc::launch_edge_sum<real>(so, bptr(i[0]), bufs[i[0]].rows, bufs[i[0]].width, bptr(i[2]), bufs[i[2]].width, i[3],
                                      d_it + i[4], i[5], i[6], 1.0 / (double)(i[1] > 0 ? i[1] : 1), B, li,
                                      li.TP > 1 && compact[i[0]]);
          t_end();
          break;
        case DQMC_OP_ATTENTION: {
          // algorithmic flops: S, dP v0 / P v_c, dP_c v_c contractions per lane
          t_begin("attention", 2.0 * B * i[4] * (double)N * (N + i[6]) * i[5] * (li.T == 1 ? 2.0 : 5.0 * li.T));
          int rc2 = DQMC_OK;
          const bool att_mfma = attention_mfma && dqmc::attention_mfma_supported<real>(N, i[5], i[6]) &&
                                (attention_mfma >= 2 || dqmc::attention_mfma_profitable(N));
          // (float64 only: the float32 instance of the split kernel agrees with float64 in the emulator but sent a whole benzene
          // batch to the float64 pass on the MI355X when it was tried -- not understood, not instantiated for the product)
          const int att_split = sizeof(real) == 8 ? (attention_split < 0 ? 1 : attention_split) : 0;
          bool done = false;
          if constexpr (sizeof(real) == 8) {
            if (att_mfma && att_split && li.TP > 1 && dqmc::attention_mfma_split_lds_bytes<real>(N, i[5], i[6]) <= (size_t)160 * 1024 &&
                i[5] >= 16) {
              rc2 = dqmc::launch_attention_mfma_split<real>(st, bptr(i[0]), bptr(i[1]), bptr(i[2]), bptr(i[3]), bufs[i[0]].width, i[4], i[5],
                                                            B, li, i[6], d_w + i[7], d_w + i[8]);
              done = true;
            }
          }
          if (done) {}
          else if (att_mfma)
            rc2 = dqmc::launch_attention_mfma<real>(st, bptr(i[0]), bptr(i[1]), bptr(i[2]), bptr(i[3]), bufs[i[0]].width, i[4], i[5], B, li,
                                                    i[6], d_w + i[7], d_w + i[8], attention_ncb < 0 ? (sizeof(real) == 4 ? 1 : 0) : attention_ncb);
          else
            rc2 = dqmc::launch_attention<real>(st, bptr(i[0]), bptr(i[1]), bptr(i[2]), bptr(i[3]), bufs[i[0]].width,
                                               i[4], i[5], B, li, i[6], d_w + i[7], d_w + i[8]);
          t_end();
          if (rc2) return fail(DQMC_E_HIP, "attention launch failed");
          break;
        }
        case DQMC_OP_CONST:
          t_begin("feat", 0, so);
          dqmc::launch_const_rows<real>(so, d_w + i[1], bptr(i[0]), B, bufs[i[0]].rows, bufs[i[0]].width, li);
          t_end();
          break;
        case DQMC_OP_ROW_SUM:
          t_begin("graph", 0, so);
          dqmc::launch_row_sum<real>(so, bptr(i[0]), bptr(i[1]), B, bufs[i[0]].rows, bufs[i[0]].width, li);
          t_end();
          break;
        case DQMC_OP_ORBITALS:
          t_begin("orbitals", 0, so);
          dqmc::launch_orbitals<real>(so, r, R, bptr(i[0]), bufs[i[0]].width, bptr(i[1]), bufs[i[1]].width, d_w + i[2], d_w + i[3],
                                      d_w + i[4], d_w + i[5], B, sys.n_up, sys.n_nuc, i[6] > 0 ? i[6] : 1, sys.n_det, li,
                                      sys.norm_eps, phq);
          t_end();
          break;
        case DQMC_OP_SLOGDET:
          t_begin("slogdet", 0);
          dqmc::launch_slogdet<real>(st, bptr(i[0]), bufs[i[0]].width, reinterpret_cast<double*>(d_ws + off_logdet),
                                     reinterpret_cast<int32_t*>(d_ws + off_signk), B, sys.n_det, li, slogdet_mfma,
                                     laplacian ? reinterpret_cast<double*>(d_ws + off_cond) : nullptr);
          t_end();
          break;
        case DQMC_OP_FINAL: {
          dqmc::FinalArgs a{};
          a.r = r; a.R = R; a.charges = d_charges;
          a.ecp_loc = d_ecp_loc; a.ecp_nt = ecp_nt_loc;
          a.logdet = reinterpret_cast<double*>(d_ws + off_logdet);
          a.sign_k = reinterpret_cast<int32_t*>(d_ws + off_signk);
          a.jastrow = i[0] >= 0 ? bptr(i[0]) : nullptr;
          a.jas_width = i[0] >= 0 ? bufs[i[0]].width : 0;
          a.conf_coeff = i[1] >= 0 ? d_w + i[1] : nullptr;
          a.alphas = d_w + i[3];
          a.cusp_kind = i[2];
          a.same_scale = op.f[0]; a.anti_scale = op.f[1];
          a.eps = sys.norm_eps;
          a.B = B; a.n_up = sys.n_up; a.n_nuc = sys.n_nuc; a.K = sys.n_det; a.li = li;
          a.logpsi = logpsi; a.sign = sign; a.e_loc = e_loc; a.stats = stats; a.stats_ld = stats_ld; a.grad = grad;
          a.phq = phq;
          if (laplacian) { a.cond = reinterpret_cast<double*>(d_ws + off_cond); a.kappa_out = reinterpret_cast<double*>(d_ws + off_kappa); }
          if (flag_on && laplacian) {
            a.flag_count = d_flag; a.flag_idx = d_flag + 1; a.refine_thresh = refine_thresh; a.thresh_dev = d_thresh; a.b_offset = b_offset;
            a.score_out = d_score;
          }
          t_begin("final", 0);
          dqmc::launch_final<real>(st, a);
          t_end();
          break;
        }
        default:
          return fail(DQMC_E_UNSUPPORTED, "op kind " + std::to_string(op.kind));
      }
      { const int rca = after(op, sid); if (rca) return rca; }
    }
    if (dual)        // join: nothing of this evaluation may still run on another stream when the caller goes on
      for (int x = 1; x < 4; ++x) if (last_ev[x]) HIP_TRY(hipStreamWaitEvent(st, last_ev[x], 0));
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }
  bool timing_serial() const { return false; }
  // The LDS-resident kernel is the faster VALUE path for the small systems it was built for (N <= 4: one launch per
  // Metropolis sub-step) and for small batches of any system (one launch against ~50); for larger systems at large
  // batch the layered MFMA kernels win (N2 / FermiNet, 4096 walkers: 1.38 ms against 1.61 ms per value pass).
  bool fused_pays(int B) const { return N <= 4 || B < 1024; }

  int wf_eval(const void* r, const void* R, int B, void* logpsi, int32_t* sign) override {
    return run((const real*)r, (const real*)R, B, false, (real*)logpsi, sign, nullptr, nullptr, nullptr);
  }
  int local_energy(const void* r, const void* R, int B, void* e_loc, void* stats, void* grad, void* logpsi,
                   int32_t* sign) override {
    last_chunks[0] = last_chunks[1] = 0;
    return lap_refined((const real*)r, (const real*)R, B, (real*)e_loc, (real*)stats, (real*)grad, (real*)logpsi, sign);
  }
  // the pass in this context's own precision: forward-Laplacian evaluation, plus the non-local ECP quadrature when the
  // Hamiltonian has one and a local energy is asked for
  int pass_own(const real* r, const real* R, int B, real* logpsi, int32_t* sign, real* e_loc, real* stats, real* grad) {
    if (ecp_n_nl == 0 || !e_loc || ecp_skip_nl || ecp_defer) return run(r, R, B, true, logpsi, sign, e_loc, stats, grad);
    return local_energy_ecp(r, R, B, e_loc, stats, grad, logpsi, sign);
  }

  // log|psi|, sign and grad log|psi| from the forward-Laplacian pass alone: no potentials beyond k_final's, no
  // non-local ECP quadrature (what value_and_grad(psi) gives the reference's Langevin sampler)
  int psi_grad(const void* r, const void* R, int B, void* logpsi, int32_t* sign, void* grad) override {
    return lap_refined((const real*)r, (const real*)R, B, nullptr, nullptr, (real*)grad, (real*)logpsi, sign);
  }

  // The forward-Laplacian pass; in the float32 build followed by the float64 re-evaluation of the walkers k_final
  // flagged (a few per cent of |psi|^2-distributed walkers of a small system; option "refine" 0 turns it off).
  int ensure_twin() {
    if (twin) return DQMC_OK;
    auto* t = new Engine<double>();
    t->st = st; t->device = device;
    dqmc_system s2 = sys;
    s2.dtype = 1;
    int rc = t->init(&s2, charges_h.data(), bufs.data(), (int)bufs.size(), ops.data(), (int)ops.size(), w64_h.data(), w64_h.size(),
                     h_itable.data(), h_itable.size());
    if (!rc && (!ecp_loc_h.empty() || !ecp_nl_h.empty()))      // the twin carries the whole ECP: its local energies include V_nl
      rc = t->set_ecp(ecp_loc_nt_h, ecp_loc_h.empty() ? nullptr : ecp_loc_h.data(), ecp_nl_L_h, ecp_nl_nt_h, ecp_nl_h.empty() ? nullptr : ecp_nl_h.data());
    if (!rc && !ph_mask_h.empty()) rc = t->set_ph(ph_grid, ph_rmax, ph_loc_h.data(), ph_l2_h.data(), ph_mask_h.data());
    for (size_t k = 0; k < twin_opts.size() && !rc; ++k) rc = t->option(twin_opts[k].first.c_str(), twin_opts[k].second);
    if (rc) { delete t; return rc; }
    t->ws_budget = twin_full_budget ? ws_budget : ws_budget / 2;
    t->timing = timing;
    twin = t;
    return DQMC_OK;
  }
  // float64 results of the n walkers listed in d_flag[1..n] replace the float32 ones.  d_count != nullptr: the list is
  // holds fewer than n entries -- the count is read on the device, n is the (padded) size of this pass (kernels_mcmc.hip:
  // k_refine_gather).
  int refine_listed(const real* r, const real* R, int B, int n, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign,
                    const int32_t* d_count = nullptr, const int32_t* d_list = nullptr, int n_scatter = -1, bool use_score = false) {
    if (!d_list) d_list = d_flag + 1;
    if (n_scatter < 0) n_scatter = n;
    const int n3 = 3 * N, nR3 = 3 * sys.n_nuc;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_r = 0, o_R = o_r + al(sizeof(double) * (size_t)n * n3), o_e = o_R + al(sizeof(double) * nR3),
                 o_s = o_e + al(sizeof(double) * n), o_g = o_s + al(sizeof(double) * 6 * (size_t)n),
                 o_l = o_g + al(sizeof(double) * (size_t)n * n3), o_sg = o_l + al(sizeof(double) * n),
                 tot = o_sg + al(sizeof(int32_t) * n);
    if (tot > ref_bytes) {
      if (d_ref) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_ref)); d_ref = nullptr; ref_bytes = 0; }
      HIP_TRY(hipMalloc((void**)&d_ref, tot));
      ref_bytes = tot;
    }
    double* r64 = (double*)(d_ref + o_r); double* R64 = (double*)(d_ref + o_R); double* e64 = (double*)(d_ref + o_e);
    double* s64 = (double*)(d_ref + o_s); double* g64 = (double*)(d_ref + o_g); double* l64 = (double*)(d_ref + o_l);
    int32_t* sg64 = (int32_t*)(d_ref + o_sg);
    t_begin("refine", 0);
    dqmc::launch_refine_gather(st, (const float*)r, (const float*)R, d_list, d_count, n, n3, nR3, r64, R64);
    t_end();
    twin->ph_skip = (e_loc == nullptr);       // psi_grad / Langevin: the plain gradient, no pseudo-Hamiltonian seeding
    // a Hamiltonian with a non-local ECP: the twin runs the quadrature of its walkers in float64 with the rotation angles
    // of the walkers they stand for (its psi ratios carry the float64 value path's accuracy: float32 ratios alone put
    // ~1e-4 relative on E_loc of a 30-electron Psiformer)
    twin->ecp_skip_nl = (e_loc == nullptr) || ecp_defer;      // (deferred: ecp_mixed adds V_nl to every walker afterwards)
    static_cast<Engine<double>*>(twin)->ecp_seed = ecp_seed;
    static_cast<Engine<double>*>(twin)->ecp_phi = ecp_phi;
    static_cast<Engine<double>*>(twin)->ecp_phi_f32 = true;
    static_cast<Engine<double>*>(twin)->ecp_idx = d_list;
    const int rc = twin->local_energy(r64, R64, n, e64, s64, g64, l64, sg64);
    if (twin->last_chunks[0] > last_chunks[1]) last_chunks[1] = twin->last_chunks[0];
    twin->ph_skip = false;
    twin->ecp_skip_nl = false;
    if (rc) return rc;
    t_begin("refine", 0);
    // (entries past n_scatter -- the calibration sample of a probe call -- are evaluated, read by the host, not written back)
    ref_e64 = e64;
    if (use_score) {
      // probe call: the new threshold needs the float64 energies of the sample on the host first; then only the walkers
      // above it are written back (d_score / refine_thresh at that moment)
      std::vector<double> e_h((size_t)n);
      HIP_TRY(hipMemcpyAsync(e_h.data(), e64, sizeof(double) * e_h.size(), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      probe_sample_e = e_h;
      if (probe_rethreshold) probe_rethreshold();
      n_scatter = n;                       // every evaluated walker above the NEW threshold is written back
    }
    dqmc::launch_refine_scatter(st, d_list, d_count, n, n_scatter, use_score ? d_score : nullptr, refine_thresh, n3, e64, s64, g64, l64, sg64,
                                (float*)e_loc, (float*)stats, (long)B, (float*)grad, (float*)logpsi, sign);
    t_end();
    if (!d_count && !use_score) last_refined += n_scatter;
    return DQMC_OK;
  }
  // error-predictor scores of the last float32 pass that flagged (host copy; walkers of that call, in order)
  int last_score_B = 0;
  int refine_scores(double* out, int n) override {
    if (sizeof(real) != 4 || !d_score || last_score_B < 1) return fail(DQMC_E_UNSUPPORTED, "no float32 pass with the error predictor has run on this context");
    if (n > last_score_B) return fail(DQMC_E_ARG, "more scores requested than the last flagged pass had walkers");
    HIP_TRY(hipMemcpyAsync(out, d_score, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQMC_OK;
  }
  int upload_list(const std::vector<int32_t>& idx) {
    std::vector<int32_t> buf(idx.size() + 1);
    buf[0] = (int32_t)idx.size();
    std::copy(idx.begin(), idx.end(), buf.begin() + 1);
    HIP_TRY(hipMemcpyAsync(d_flag, buf.data(), sizeof(int32_t) * buf.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQMC_OK;
  }
  int lap_refined(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign) {
    const int rc = lap_refined_ecp(r, R, B, e_loc, stats, grad, logpsi, sign);
    refine_info[0] = sizeof(real) == 8 ? 0 : refine; refine_info[1] = refine_thresh; refine_info[2] = probe_c; refine_info[3] = refine_all_calls;
    return rc;
  }
  int lap_refined_ecp(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign) {
    if constexpr (sizeof(real) == 4) {
      if (refine == 1 && ecp_mixed_on && ecp_n_nl > 0 && e_loc && !ecp_skip_nl) {
        // kinetic part first (float32 pass, flagged walkers re-run in float64 WITHOUT the quadrature), then V_nl of every
        // walker with the precision chosen per (nucleus, electron) pair
        ecp_defer = true;
        int rc = lap_refined_core(r, R, B, e_loc, stats, grad, logpsi, sign);
        ecp_defer = false;
        if (rc) return rc;
        return ecp_mixed((const float*)r, (const float*)R, B, (float*)e_loc, (float*)stats);
      }
    }
    return lap_refined_core(r, R, B, e_loc, stats, grad, logpsi, sign);
  }
  // Non-local ECP term of a float32 context with per-pair precision (kernels_ecp.hip: "mixed-precision quadrature"):
  // added to e_loc, stored in stats[3].
  int ecp_mixed(const float* r, const float* R, int B, float* e_loc, float* stats) {
    // No float64 twin for this program (its float64 kernel set does not exist: DQMC_E_UNSUPPORTED -- lap_refined_core has
    // switched the refinement off for the same reason): the quadrature runs entirely in float32, as local_energy_ecp would
    // (every kept pair in the float32 class; the pair cut-off stays).  Any other failure is an error of the call.
    int rc = ensure_twin();
    const bool have_twin = rc == DQMC_OK;
    if (rc && rc != DQMC_E_UNSUPPORTED) return rc;
    const size_t per_walker = (size_t)ecp_n_nl * N * 12, triples_pw = (size_t)ecp_n_nl * N;
    int nbw = (int)(ecp_max_cfg / per_walker);
    nbw = nbw < 1 ? 1 : (nbw > B ? B : nbw);
    const size_t n_cfg = (size_t)nbw * (per_walker + 1), n_tr = (size_t)nbw * triples_pw;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_cls = 0, o_ll = o_cls + al(4 * n_tr), o_lh = o_ll + al(4 * n_tr), o_cnt = o_lh + al(4 * n_tr),
                 o_r32 = o_cnt + 256, o_l32 = o_r32 + al(4 * n_cfg * N * 3), o_s32 = o_l32 + al(4 * n_cfg),
                 o_r64 = o_s32 + al(4 * n_cfg), o_l64 = o_r64 + al(8 * n_cfg * N * 3), o_s64 = o_l64 + al(8 * n_cfg),
                 o_R64 = o_s64 + al(4 * n_cfg), tot = o_R64 + al(8 * 3 * (size_t)sys.n_nuc);
    if (tot > ecpm_bytes) {
      if (d_ecpm) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_ecpm)); d_ecpm = nullptr; ecpm_bytes = 0; }
      hipError_t e = hipMalloc((void**)&d_ecpm, tot);
      if (e != hipSuccess) return fail(DQMC_E_NOMEM, "ECP scratch of " + std::to_string(tot) + " bytes: " + hipGetErrorString(e));
      ecpm_bytes = tot;
    }
    int32_t* cls = (int32_t*)(d_ecpm + o_cls); int32_t* list_l = (int32_t*)(d_ecpm + o_ll); int32_t* list_h = (int32_t*)(d_ecpm + o_lh);
    int32_t* cnt = (int32_t*)(d_ecpm + o_cnt);
    float* rq32 = (float*)(d_ecpm + o_r32); float* lq32 = (float*)(d_ecpm + o_l32); int32_t* sq32 = (int32_t*)(d_ecpm + o_s32);
    double* rq64 = (double*)(d_ecpm + o_r64); double* lq64 = (double*)(d_ecpm + o_l64); int32_t* sq64 = (int32_t*)(d_ecpm + o_s64);
    double* R64 = (double*)(d_ecpm + o_R64);
    if (have_twin) dqmc::launch_refine_gather(st, r, R, nullptr, nullptr, 0, 3 * N, 3 * sys.n_nuc, nullptr, R64);      // (widens R only)
    dqmc::EcpMixArgs a{};
    a.r = r; a.R = R; a.nl_nuc = d_ecp_nuc; a.nl = d_ecp_nl; a.phi = (const float*)ecp_phi; a.seed = ecp_seed;
    a.B = B; a.N = N; a.n_nl = ecp_n_nl; a.L = ecp_L; a.n_t = ecp_nt_nl;
    a.w_heavy = have_twin ? ecp_w_heavy : HUGE_VAL; a.w_skip = ecp_w_skip;
    a.dlog_floor = ecp_dlog_floor;
    ecp_last_counts[0] = ecp_last_counts[1] = ecp_last_counts[2] = 0;
    for (int b0 = 0; b0 < B; b0 += nbw) {
      a.b0 = b0; a.nb = (B - b0) < nbw ? (B - b0) : nbw;
      HIP_TRY(hipMemsetAsync(cnt, 0, 2 * sizeof(int32_t), st));
      a.l32 = nullptr;
      if (have_twin && ecp_dlog_floor > 0) {
        // psi(r) of the chunk's own walkers by both value paths, ahead of the classification (the lists are empty: the
        // configurations are the nb walkers themselves); their disagreement is each walker's float32 error
        t_begin("ecp", 0);
        dqmc::launch_ecp_points_list<float>(st, a, list_l, 0, rq32);
        dqmc::launch_ecp_points_list<double>(st, a, list_h, 0, rq64);
        t_end();
        rc = run((const real*)rq32, (const real*)R, a.nb, false, (real*)lq32, sq32, nullptr, nullptr, nullptr);
        if (rc) return rc;
        rc = twin->wf_eval(rq64, R64, a.nb, lq64, sq64);
        if (rc) return rc;
        a.l32 = lq32; a.l64 = lq64; a.s32 = sq32; a.s64 = sq64;
      }
      t_begin("ecp", 0);
      dqmc::launch_ecp_classify(st, a, cls, list_l, list_h, cnt);
      t_end();
      int32_t n2[2] = {0, 0};
      HIP_TRY(hipMemcpyAsync(n2, cnt, sizeof(n2), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      ecp_last_counts[0] += n2[0]; ecp_last_counts[1] += n2[1];
      ecp_last_counts[2] += (long)a.nb * (long)triples_pw - n2[0] - n2[1];
      t_begin("ecp", 0);
      dqmc::launch_ecp_points_list<float>(st, a, list_l, n2[0], rq32);
      if (have_twin) dqmc::launch_ecp_points_list<double>(st, a, list_h, n2[1], rq64);
      t_end();
      rc = run((const real*)rq32, (const real*)R, a.nb + 12 * n2[0], false, (real*)lq32, sq32, nullptr, nullptr, nullptr);
      if (rc) return rc;
      if (have_twin) {
        rc = twin->wf_eval(rq64, R64, a.nb + 12 * n2[1], lq64, sq64);
        if (rc) return rc;
      } else if (n2[1] != 0) {
        return fail(DQMC_E_HIP, "ECP classification produced float64 pairs without a float64 twin");
      }
      t_begin("ecp", 0);
      dqmc::launch_ecp_reduce_mixed(st, a, cls, lq32, sq32, lq64, sq64, e_loc, stats);
      t_end();
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }
  int lap_refined_core(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign) {
    const int rc = lap_refined_core_(r, R, B, e_loc, stats, grad, logpsi, sign);
    refine_counters[3] += last_refined;
    return rc;
  }
  // more than this many of B walkers above the threshold: the batch goes to float64 whole (hysteresis: see refine_direct_enter)
  bool mostly_flagged(long n_above, int B) {
    const double lim = was_direct ? refine_direct_exit : refine_direct_enter;
    const bool yes = B >= 16 && (double)n_above > lim * (double)B;
    was_direct = yes;
    return yes;
  }
  int lap_refined_core_(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi, int32_t* sign) {
    last_refined = 0;
    ++refine_counters[0];
    if constexpr (sizeof(real) == 8) {
      return pass_own(r, R, B, logpsi, sign, e_loc, stats, grad);
    } else {
      if (!refine) return pass_own(r, R, B, logpsi, sign, e_loc, stats, grad);
      if ((size_t)B + 1 > flag_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (d_flag) { HIP_TRY(hipFree(d_flag)); d_flag = nullptr; }
        HIP_TRY(hipMalloc((void**)&d_flag, sizeof(int32_t) * ((size_t)B + 1)));
        flag_cap = (size_t)B + 1;
      }
      if ((size_t)B > score_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (d_score) { HIP_TRY(hipFree(d_score)); d_score = nullptr; }
        HIP_TRY(hipMalloc((void**)&d_score, sizeof(double) * (size_t)B));
        score_cap = (size_t)B;
      }
      int rc = DQMC_OK;
      // mode 1 on a system where most walkers get flagged (deep attention networks, ill-conditioned Slater matrices of a
      // random-init TransPsiformer): the float32 pass would be wasted, so the following 15 calls go to float64 directly,
      // then the float32 pass is probed again
      const bool direct = refine >= 2 || (refine == 1 && refine_all_calls > 0 && twin);
      if (refine == 1 && refine_all_calls > 0) --refine_all_calls;
      if (direct) {                // the whole forward-Laplacian pass in float64 (float32 stays the sampling dtype)
        ++refine_counters[1];
        last_score_B = 0;          // (no float32 pass, no scores)
        rc = ensure_twin();
        if (rc) return rc;
        std::vector<int32_t> iota((size_t)B);
        for (int k = 0; k < B; ++k) iota[k] = k;
        rc = upload_list(iota);
        if (rc) return rc;
        return refine_listed(r, R, B, B, e_loc, stats, grad, logpsi, sign);
      }
      HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int32_t), st));
      if (!d_thresh) HIP_TRY(hipMalloc((void**)&d_thresh, sizeof(double)));
      if (thresh_uploaded != refine_thresh) {      // (k_final reads the threshold from here: a captured pass must follow a re-calibration)
        HIP_TRY(hipMemcpyAsync(d_thresh, &refine_thresh, sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        thresh_uploaded = refine_thresh;
      }
      flag_on = true;
      rc = pass_own(r, R, B, logpsi, sign, e_loc, stats, grad);
      flag_on = false;
      last_score_B = rc ? 0 : B;
      if (rc) return rc;
      const bool probe = refine == 1 && refine_probe > 0 && e_loc && (calls_since_probe < 0 || calls_since_probe + 1 >= refine_probe);
      if (calls_since_probe >= 0) ++calls_since_probe;
      int32_t n = 0;
      HIP_TRY(hipMemcpyAsync(&n, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (n > B) n = B;
      if (!probe) {
        if (n <= 0) return DQMC_OK;
        rc = ensure_twin();
        if (rc == DQMC_E_UNSUPPORTED && refine == 1) { refine = 0; return DQMC_OK; }   // no float64 kernel set for this program: float32 stands
        if (rc) return rc;
        if (refine == 1 && mostly_flagged(n, B)) {
          // most of the batch is beyond float32: this call and the next 15 evaluate everything in float64
          refine_all_calls = 15;
          ++refine_counters[1];
          std::vector<int32_t> iota((size_t)B);
          for (int k = 0; k < B; ++k) iota[k] = k;
          rc = upload_list(iota);
          if (rc) return rc;
          n = B;
        }
        // the twin's pass replays a captured graph per batch size: round the count up to a multiple of 64 (the surplus rows
        // re-evaluate the first flagged walker and are not written back: k_refine_gather / scatter read the count on the
        // device), so that a handful of sizes serve every step
        const bool padded = n < B && static_cast<Engine<double>*>(twin)->graph_fits((n + 63) / 64 * 64);
        int n_eval = padded ? (n + 63) / 64 * 64 : n;
        if (n_eval > B) n_eval = B;
        const int32_t* d_cnt = padded ? d_flag : nullptr;
        rc = refine_listed(r, R, B, n_eval, e_loc, stats, grad, logpsi, sign, d_cnt);
        last_refined = n;
        HIP_TRY(hipGetLastError());
        return rc;
      }
      // ---- probe call: measure the float32 error per unit of score on a strided sample, re-derive the threshold, and
      // apply it to THIS call as well (a caller that evaluates once gets the calibrated result)
      ++refine_counters[2];
      rc = ensure_twin();
      if (rc == DQMC_E_UNSUPPORTED && refine == 1) { refine = 0; return DQMC_OK; }
      if (rc) return rc;
      std::vector<int32_t> flagged((size_t)n);
      std::vector<double> score((size_t)B);
      std::vector<float> e32((size_t)B);
      if (n) HIP_TRY(hipMemcpyAsync(flagged.data(), d_flag + 1, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(score.data(), d_score, sizeof(double) * (size_t)B, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(e32.data(), e_loc, sizeof(float) * (size_t)B, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      std::vector<char> done((size_t)B, 0);        // 1: evaluated in float64 by the first pass of this call (flagged or sample)
      for (int32_t b : flagged) if (b >= 0 && b < B) done[b] = 1;
      // calibration sample: a strided subset of the WHOLE batch (flagged or not: every walker is an (error, score) pair; a
      // system whose walkers all sit above the current threshold must still be able to move it)
      const int ns = B < refine_sample ? B : refine_sample;
      std::vector<int32_t> sample, list(flagged);
      std::vector<int> sample_pos;                 // position of each sample walker in `list`
      {
        std::vector<int> pos_of((size_t)B, -1);
        for (size_t k = 0; k < flagged.size(); ++k) if (flagged[k] >= 0 && flagged[k] < B) pos_of[flagged[k]] = (int)k;
        for (int j = 0; j < ns; ++j) {
          const int b = (int)((long)j * B / ns);
          if (pos_of[b] < 0) { pos_of[b] = (int)list.size(); list.push_back(b); done[b] = 1; }
          sample.push_back(b);
          sample_pos.push_back(pos_of[b]);
        }
      }
      // ONE float64 pass over flagged + sample; between its evaluation and its write-back the threshold is re-derived from
      // the sample, and only walkers above the NEW threshold are written back: what a walker's result is depends on its
      // score and the threshold alone -- never on having served as a calibration sample or on the threshold before the probe
      rc = upload_list(list);
      if (rc) return rc;
      probe_rethreshold = [&]() {
        std::vector<double> cs;
        for (size_t k = 0; k < sample.size(); ++k) {
          const int32_t b = sample[k];
          const double e64v = probe_sample_e[(size_t)sample_pos[k]];
          const double rel = std::fabs(e64v - (double)e32[b]) / std::fmax(1.0, std::fabs(e64v));
          if (std::isfinite(rel) && std::isfinite(score[b]) && score[b] > 0) cs.push_back(rel / score[b]);
        }
        if (cs.size() >= 2) {
          std::sort(cs.begin(), cs.end());
          const double c = std::fmax(cs[(size_t)(0.9 * (cs.size() - 1) + 0.5)], 1e-12);
          probe_c = probe_c > 0 ? std::sqrt(probe_c * c) : c;         // geometric smoothing over the probes
          refine_thresh = std::fmin(std::fmax(refine_target / probe_c, 1.0), 1e9);
        }
      };
      rc = refine_listed(r, R, B, (int)list.size(), e_loc, stats, grad, logpsi, sign, nullptr, nullptr, 0, true);
      probe_rethreshold = nullptr;
      if (rc) return rc;
      calls_since_probe = 0;
      long n_above = 0;
      for (int b = 0; b < B; ++b) if (!(score[b] <= refine_thresh)) ++n_above;
      std::vector<int32_t> more;
      if (refine == 1 && mostly_flagged(n_above, B)) {
        // most of the batch is beyond float32: the next calls go to float64 directly, and so does the rest of this one
        // (the few walkers below the threshold of such a system are not reliably predicted either)
        refine_all_calls = 15;
        ++refine_counters[1];
        for (int b = 0; b < B; ++b) if (!done[b] || score[b] <= refine_thresh) more.push_back(b);     // not yet written back
        last_refined = B - (int)more.size();
      } else {
        for (int b = 0; b < B; ++b) {
          if (score[b] <= refine_thresh) continue;
          if (done[b]) ++last_refined; else more.push_back(b);
        }
      }
      if (!more.empty()) {
        rc = upload_list(more);
        if (rc) return rc;
        rc = refine_listed(r, R, B, (int)more.size(), e_loc, stats, grad, logpsi, sign);
        if (rc) return rc;
      }
      HIP_TRY(hipGetLastError());
      return DQMC_OK;
    }
  }

  // Effective core potentials: host tables (ecp/gaussian_type_ecp.py:32-93 layout) -> device.
  //   loc[n_nuc][3][2][n_t_loc]  r^-1 / r^0 / r^1 terms: [.,term,0,.] exponents, [.,term,1,.] coefficients
  //   nl [n_nuc][n_l][2][n_t_nl] channels l = 0..n_l-1; nuclei whose block is all zero have no non-local part
  int set_ecp(int n_t_loc, const double* loc, int n_l, int n_t_nl, const double* nl) override {
    if (n_t_loc < 0 || n_l < 0 || n_t_nl < 0) return fail(DQMC_E_ARG, "negative ECP table size");
    ++graph_epoch; drop_graphs();          // (captured passes hold the table pointers)
    if (ph_n && nl && n_l > 0 && n_t_nl > 0) return fail(DQMC_E_ARG, "a pseudo-Hamiltonian and a non-local Gaussian ECP cannot both be set");
    HIP_TRY(hipStreamSynchronize(st));
    if (d_ecp_loc) { HIP_TRY(hipFree(d_ecp_loc)); d_ecp_loc = nullptr; }
    if (d_ecp_nl) { HIP_TRY(hipFree(d_ecp_nl)); d_ecp_nl = nullptr; }
    if (d_ecp_nuc) { HIP_TRY(hipFree(d_ecp_nuc)); d_ecp_nuc = nullptr; }
    ecp_nt_loc = ecp_n_nl = ecp_L = ecp_nt_nl = 0;
    ecp_loc_h.clear(); ecp_loc_nt_h = 0;
    ecp_nl_h.clear(); ecp_nl_L_h = ecp_nl_nt_h = 0;
    if (loc && n_t_loc > 0 && sizeof(real) == 4) { ecp_loc_h.assign(loc, loc + (size_t)sys.n_nuc * 6 * n_t_loc); ecp_loc_nt_h = n_t_loc; }
    if (nl && n_l > 0 && n_t_nl > 0 && sizeof(real) == 4) { ecp_nl_h.assign(nl, nl + (size_t)sys.n_nuc * n_l * 2 * n_t_nl); ecp_nl_L_h = n_l; ecp_nl_nt_h = n_t_nl; }
    if (twin) { const int rc = twin->set_ecp(n_t_loc, loc, n_l, n_t_nl, nl); if (rc) return rc; }
    if (loc && n_t_loc > 0) {
      const size_t n = (size_t)sys.n_nuc * 6 * n_t_loc;
      HIP_TRY(hipMalloc((void**)&d_ecp_loc, sizeof(double) * n));
      HIP_TRY(hipMemcpy(d_ecp_loc, loc, sizeof(double) * n, hipMemcpyHostToDevice));
      ecp_nt_loc = n_t_loc;
    }
    if (nl && n_l > 0 && n_t_nl > 0) {
      const size_t blk = (size_t)n_l * 2 * n_t_nl;
      std::vector<int32_t> nuc;
      std::vector<double> compact;
      for (int a = 0; a < sys.n_nuc; ++a) {          // gaussian_type_ecp.py:121 (nuc_with_nl_pot)
        bool any = false;
        for (size_t k = 0; k < blk; ++k) any = any || nl[a * blk + k] != 0.0;
        if (!any) continue;
        nuc.push_back(a);
        compact.insert(compact.end(), nl + a * blk, nl + (a + 1) * blk);
      }
      if (!nuc.empty()) {
        HIP_TRY(hipMalloc((void**)&d_ecp_nl, sizeof(double) * compact.size()));
        HIP_TRY(hipMalloc((void**)&d_ecp_nuc, sizeof(int32_t) * nuc.size()));
        HIP_TRY(hipMemcpy(d_ecp_nl, compact.data(), sizeof(double) * compact.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_ecp_nuc, nuc.data(), sizeof(int32_t) * nuc.size(), hipMemcpyHostToDevice));
        ecp_n_nl = (int)nuc.size(); ecp_L = n_l; ecp_nt_nl = n_t_nl;
      }
    }
    return DQMC_OK;
  }
  int ecp_rotation(uint64_t seed, const void* phi) override { ecp_seed = seed; ecp_phi = phi; return DQMC_OK; }

  // Pseudo-Hamiltonian tables (ecp/pseudo_hamiltonian.py:73-112): rV_loc and rV_L2 on the regular grid
  // linspace(0, r_max, n_grid), one row per nucleus; rows of nuclei with mask 0 are ignored.  An all-zero mask
  // (or n_grid 0) switches the PH off.
  int set_ph(int n_grid, double r_max, const double* rv_loc, const double* rv_l2, const int32_t* mask) override {
    ++graph_epoch; drop_graphs();
    HIP_TRY(hipStreamSynchronize(st));
    if (d_ph_loc) { HIP_TRY(hipFree(d_ph_loc)); d_ph_loc = nullptr; }
    if (d_ph_l2) { HIP_TRY(hipFree(d_ph_l2)); d_ph_l2 = nullptr; }
    if (d_ph_nuc) { HIP_TRY(hipFree(d_ph_nuc)); d_ph_nuc = nullptr; }
    ph_n = ph_grid = 0; ph_rmax = 0.0;
    ph_loc_h.clear(); ph_l2_h.clear(); ph_mask_h.clear();
    std::vector<int32_t> nuc;
    if (mask && n_grid > 0) for (int a = 0; a < sys.n_nuc; ++a) if (mask[a]) nuc.push_back(a);
    if (nuc.empty()) { if (twin) return twin->set_ph(0, 0.0, nullptr, nullptr, nullptr); return DQMC_OK; }
    if (n_grid < 2 || !(r_max > 0.0) || !rv_loc || !rv_l2) return fail(DQMC_E_ARG, "pseudo-Hamiltonian tables need n_grid >= 2, r_max > 0");
    if (ecp_n_nl) return fail(DQMC_E_ARG, "a pseudo-Hamiltonian and a non-local Gaussian ECP cannot both be set");
    std::vector<double> loc, l2;
    for (int a : nuc) {
      loc.insert(loc.end(), rv_loc + (size_t)a * n_grid, rv_loc + (size_t)(a + 1) * n_grid);
      l2.insert(l2.end(), rv_l2 + (size_t)a * n_grid, rv_l2 + (size_t)(a + 1) * n_grid);
    }
    HIP_TRY(hipMalloc((void**)&d_ph_loc, sizeof(double) * loc.size()));
    HIP_TRY(hipMalloc((void**)&d_ph_l2, sizeof(double) * l2.size()));
    HIP_TRY(hipMalloc((void**)&d_ph_nuc, sizeof(int32_t) * nuc.size()));
    HIP_TRY(hipMemcpy(d_ph_loc, loc.data(), sizeof(double) * loc.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ph_l2, l2.data(), sizeof(double) * l2.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ph_nuc, nuc.data(), sizeof(int32_t) * nuc.size(), hipMemcpyHostToDevice));
    ph_n = (int)nuc.size(); ph_grid = n_grid; ph_rmax = r_max;
    if (sizeof(real) == 4) {
      ph_loc_h.assign(rv_loc, rv_loc + (size_t)sys.n_nuc * n_grid);
      ph_l2_h.assign(rv_l2, rv_l2 + (size_t)sys.n_nuc * n_grid);
      ph_mask_h.assign(mask, mask + sys.n_nuc);
    }
    if (twin) return twin->set_ph(n_grid, r_max, rv_loc, rv_l2, mask);
    return DQMC_OK;
  }

  // E_loc with the non-local ECP term: the Laplacian pass, then 12 N n_nl value-only psi evaluations per
  // walker in batches of <= ecp_max_cfg quadrature walkers (gaussian_type_ecp.py:161-255).
  int local_energy_ecp(const real* r, const real* R, int B, real* e_loc, real* stats, real* grad, real* logpsi,
                       int32_t* sign) {
    const size_t per_walker = (size_t)ecp_n_nl * N * 12;
    int nbw = (int)(ecp_max_cfg / per_walker);
    nbw = nbw < 1 ? 1 : (nbw > B ? B : nbw);
    const size_t n_cfg = (size_t)nbw * per_walker;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_rq = 0, o_lq = o_rq + al(sizeof(real) * n_cfg * N * 3), o_sq = o_lq + al(sizeof(real) * n_cfg),
                 o_l0 = o_sq + al(sizeof(int32_t) * n_cfg), o_s0 = o_l0 + al(sizeof(real) * B),
                 tot = o_s0 + al(sizeof(int32_t) * B);
    if (tot > ecp_bytes) {
      if (d_ecp) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_ecp)); d_ecp = nullptr; ecp_bytes = 0; }
      hipError_t e = hipMalloc((void**)&d_ecp, tot);
      if (e != hipSuccess) return fail(DQMC_E_NOMEM, "ECP scratch of " + std::to_string(tot) + " bytes: " + hipGetErrorString(e));
      ecp_bytes = tot;
    }
    real* rq = (real*)(d_ecp + o_rq); real* lq = (real*)(d_ecp + o_lq); int32_t* sq = (int32_t*)(d_ecp + o_sq);
    real* l0 = logpsi ? logpsi : (real*)(d_ecp + o_l0);
    int32_t* s0 = sign ? sign : (int32_t*)(d_ecp + o_s0);
    int rc = run(r, R, B, true, l0, s0, e_loc, stats, grad);
    if (rc) return rc;
    dqmc::EcpArgs a{};
    a.r = r; a.R = R; a.nl_nuc = d_ecp_nuc; a.nl = d_ecp_nl; a.phi = ecp_phi; a.seed = ecp_seed;
    a.walker_idx = ecp_idx; a.phi_f32 = ecp_phi_f32 ? 1 : 0;
    a.B = B; a.N = N; a.n_nl = ecp_n_nl; a.L = ecp_L; a.n_t = ecp_nt_nl;
    for (int b0 = 0; b0 < B; b0 += nbw) {
      a.b0 = b0; a.nb = (B - b0) < nbw ? (B - b0) : nbw;
      t_begin("ecp", 0);
      dqmc::launch_ecp_points<real>(st, a, rq);
      t_end();
      rc = run(rq, R, (int)((size_t)a.nb * per_walker), false, lq, sq, nullptr, nullptr, nullptr);
      if (rc) return rc;
      t_begin("ecp", 0);
      dqmc::launch_ecp_reduce<real>(st, a, lq, sq, l0, s0, e_loc, stats, (real*)nullptr);
      t_end();
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }

  int mcmc(void* r_, void* logpsi_, int32_t* sign, int32_t* age, void* tau_, const void* R_, int B, int n_sub,
           int max_age, double target, uint64_t seed, const void* noise_, const void* unif_, uint8_t* accept_out,
           double* stats7) override {
    if (B < 1 || n_sub < 0) return fail(DQMC_E_ARG, "bad B / n_sub");
    if ((noise_ == nullptr) != (unif_ == nullptr)) return fail(DQMC_E_ARG, "noise and unif must both be given or both NULL");
    real* r = (real*)r_; real* logpsi = (real*)logpsi_; real* tau = (real*)tau_;
    const real* R = (const real*)R_;
    const size_t n_r = (size_t)B * N * 3;
    // scratch: r_prop, logpsi_prop, sign_prop, noise, unif
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_rp = 0, o_lp = o_rp + al(sizeof(real) * n_r), o_sp = o_lp + al(sizeof(real) * B),
                 o_nz = o_sp + al(sizeof(int32_t) * B), o_un = o_nz + al(sizeof(real) * n_r * (size_t)(n_sub > 0 ? n_sub : 1)),
                 tot = o_un + al(sizeof(real) * B * (size_t)(n_sub > 0 ? n_sub : 1));
    if (tot > mc_bytes) {
      if (d_mc) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_mc)); d_mc = nullptr; }
      HIP_TRY(hipMalloc((void**)&d_mc, tot));
      mc_bytes = tot;
    }
    real* r_prop = (real*)(d_mc + o_rp); real* lp_prop = (real*)(d_mc + o_lp);
    int32_t* s_prop = (int32_t*)(d_mc + o_sp);
    real* nz = (real*)(d_mc + o_nz); real* un = (real*)(d_mc + o_un);
    if (!noise_ && n_sub > 0) {      // all sub-steps' normals and uniforms in ONE launch (Philox is counter based)
      t_begin("mcmc", 0);
      dqmc::launch_rng<real>(st, nz, (long)(n_r * n_sub), un, (long)B * n_sub, seed, (uint64_t)0);
      t_end();
    }
    // whole sub-step in one launch (kernel_fused2.hip: propose in the prologue, determinants / CI sum / accept /
    // tau adaptation in the tail) when the ansatz tail is the plain SLOGDET + FINAL pair and N <= 4
    const bool one_launch = fused_enabled && fused2_WT > 0 && fused_substep && N >= 2 && N <= 4 &&
                            sys.n_nuc <= 8 && fused2_WT <= 16 && (int)ops.size() == fused_n_ops + 2 &&
                            ops[fused_n_ops].kind == DQMC_OP_SLOGDET && ops[fused_n_ops + 1].kind == DQMC_OP_FINAL &&
                            substep_mat_off() >= 0;
    for (int s = 0; s < n_sub; ++s) {
      const real* noise_s; const real* unif_s;
      if (one_launch) {
        if (noise_) {
          noise_s = (const real*)noise_ + (size_t)s * n_r;
          unif_s = (const real*)unif_ + (size_t)s * B;
        } else {
          noise_s = nz + (size_t)s * n_r; unif_s = un + (size_t)s * B;
        }
        dqmc::LaneInfo li; li.N = N; li.T = 1; li.TP = 1;
        int rc = plan(B, 1);
        if (rc) return rc;
        const dqmc_op& fin = ops[fused_n_ops + 1];
        dqmc::FusedMc mc{};
        mc.enabled = 1; mc.noise = noise_s; mc.unif = unif_s; mc.r = r; mc.logpsi = logpsi; mc.sign = sign; mc.age = age;
        mc.tau_in = tau; mc.tau_ring = d_tau_ring; mc.counters = d_nacc; mc.s = s; mc.target = target;
        mc.accept_out = accept_out ? accept_out + (size_t)s * B : nullptr;
        mc.max_age = max_age;
        mc.orb_op = -1;
        for (int j = 0; j < fused_n_ops; ++j) if (ops[f_order[j]].kind == DQMC_OP_ORBITALS) mc.orb_op = j;
        mc.mat_off = substep_mat_off();
        mc.jas_width = fin.i[0] >= 0 ? bufs[fin.i[0]].width : 0;
        mc.cc_off = fin.i[1]; mc.cusp_kind = fin.i[2]; mc.al_off = fin.i[3];
        mc.same_scale = fin.f[0]; mc.anti_scale = fin.f[1];
        rc = run_fused2(nullptr, R, B, li, &mc);
        if (rc) return rc;
        if (s + 1 == n_sub) {
          t_begin("mcmc", 0);
          dqmc::launch_tau_finalize<real>(st, tau, (const real*)d_tau_ring, d_nacc, s, B, target, d_acc);
          t_end();
        }
        continue;
      }
      if (noise_) {
        noise_s = (const real*)noise_ + (size_t)s * n_r;
        unif_s = (const real*)unif_ + (size_t)s * B;
      } else {
        noise_s = nz + (size_t)s * n_r; unif_s = un + (size_t)s * B;
      }
      t_begin("mcmc", 0);
      dqmc::launch_propose<real>(st, r, noise_s, tau, r_prop, (long)n_r);
      t_end();
      int rc = run(r_prop, R, B, false, lp_prop, s_prop, nullptr, nullptr, nullptr);
      if (rc) return rc;
      t_begin("mcmc", 0);
      dqmc::launch_accept<real>(st, r, logpsi, sign, age, r_prop, lp_prop, s_prop, unif_s, max_age, B, N, d_nacc,
                                accept_out ? accept_out + (size_t)s * B : nullptr);
      dqmc::launch_tau_update<real>(st, tau, d_nacc, B, target, d_acc);
      t_end();
    }
    if (stats7) {
      dqmc::launch_sampler_stats<real>(st, r, logpsi, age, tau, d_acc, B, N, sys.norm_eps, d_acc + 1);
      HIP_TRY(hipMemcpyAsync(stats7, d_acc + 1, sizeof(double) * 7, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }

  // ---- Langevin (MALA) and exchange steps (electron_samplers.py:176-330, sampling_utils.py:72-101) ----
  double* d_molz = nullptr;     // device copy of the full nuclear charges the drift cleaning uses
  int upload_molz(const double* z) {
    if (!d_molz) HIP_TRY(hipMalloc((void**)&d_molz, sizeof(double) * sys.n_nuc));
    HIP_TRY(hipMemcpyAsync(d_molz, z, sizeof(double) * sys.n_nuc, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));     // `z` is pageable host memory of the caller
    return DQMC_OK;
  }
  int ensure_mc(size_t tot) {
    if (tot > mc_bytes) {
      if (d_mc) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(d_mc)); d_mc = nullptr; }
      HIP_TRY(hipMalloc((void**)&d_mc, tot));
      mc_bytes = tot;
    }
    return DQMC_OK;
  }
  int langevin_update(const void* r_, const void* R_, const double* molz, int B, const void* tau_, void* logpsi, int32_t* sign,
                      void* force_) override {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    int rc = upload_molz(molz);
    if (rc) return rc;
    const size_t n_r = (size_t)B * N * 3;
    rc = ensure_mc(sizeof(real) * n_r + 256);
    if (rc) return rc;
    real* g = (real*)d_mc;
    rc = lap_refined((const real*)r_, (const real*)R_, B, nullptr, nullptr, g, (real*)logpsi, sign);
    if (rc) return rc;
    t_begin("mcmc", 0);
    dqmc::launch_clean_force<real>(st, g, (const real*)r_, (const real*)R_, d_molz, (const real*)tau_, B, N, sys.n_nuc, (real*)force_);
    t_end();
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }
  int langevin(void* r_, void* logpsi_, int32_t* sign, int32_t* age, void* force_, void* tau_, const void* R_, const double* molz,
               int B, int n_sub, int max_age, double target, uint64_t seed, const void* noise_, const void* unif_,
               uint8_t* accept_out, double* stats7) override {
    if (B < 1 || n_sub < 0) return fail(DQMC_E_ARG, "bad B / n_sub");
    if ((noise_ == nullptr) != (unif_ == nullptr)) return fail(DQMC_E_ARG, "noise and unif must both be given or both NULL");
    int rc = upload_molz(molz);
    if (rc) return rc;
    real* r = (real*)r_; real* logpsi = (real*)logpsi_; real* tau = (real*)tau_; real* force = (real*)force_;
    const real* R = (const real*)R_;
    const size_t n_r = (size_t)B * N * 3, ns = (size_t)(n_sub > 0 ? n_sub : 1);
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_rp = 0, o_g = o_rp + al(sizeof(real) * n_r), o_fp = o_g + al(sizeof(real) * n_r), o_lp = o_fp + al(sizeof(real) * n_r),
                 o_sp = o_lp + al(sizeof(real) * B), o_nz = o_sp + al(sizeof(int32_t) * B), o_un = o_nz + al(sizeof(real) * n_r * ns),
                 tot = o_un + al(sizeof(real) * B * ns);
    rc = ensure_mc(tot);
    if (rc) return rc;
    real* r_prop = (real*)(d_mc + o_rp); real* g_prop = (real*)(d_mc + o_g); real* f_prop = (real*)(d_mc + o_fp);
    real* lp_prop = (real*)(d_mc + o_lp); int32_t* s_prop = (int32_t*)(d_mc + o_sp);
    real* nz = (real*)(d_mc + o_nz); real* un = (real*)(d_mc + o_un);
    if (!noise_ && n_sub > 0) {
      t_begin("mcmc", 0);
      dqmc::launch_rng<real>(st, nz, (long)(n_r * n_sub), un, (long)B * n_sub, seed, (uint64_t)1);
      t_end();
    }
    for (int s = 0; s < n_sub; ++s) {
      const real* noise_s = noise_ ? (const real*)noise_ + (size_t)s * n_r : nz + (size_t)s * n_r;
      const real* unif_s = unif_ ? (const real*)unif_ + (size_t)s * B : un + (size_t)s * B;
      t_begin("mcmc", 0);
      dqmc::launch_langevin_propose<real>(st, r, force, noise_s, tau, r_prop, (long)n_r);
      t_end();
      rc = lap_refined(r_prop, R, B, nullptr, nullptr, g_prop, lp_prop, s_prop);
      if (rc) return rc;
      t_begin("mcmc", 0);
      dqmc::launch_clean_force<real>(st, g_prop, r_prop, R, d_molz, tau, B, N, sys.n_nuc, f_prop);
      dqmc::launch_langevin_accept<real>(st, r, logpsi, sign, age, force, r_prop, lp_prop, s_prop, f_prop, unif_s, tau, max_age, B, N,
                                         d_nacc, accept_out ? accept_out + (size_t)s * B : nullptr);
      dqmc::launch_tau_update<real>(st, tau, d_nacc, B, target, d_acc);
      t_end();
    }
    if (stats7) {
      dqmc::launch_sampler_stats<real>(st, r, logpsi, age, tau, d_acc, B, N, sys.norm_eps, d_acc + 1);
      HIP_TRY(hipMemcpyAsync(stats7, d_acc + 1, sizeof(double) * 7, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }
  int exchange(void* r_, void* logpsi_, int32_t* sign, int32_t* age, const void* tau_, const void* R_, int B, const int32_t* up_idx,
               const int32_t* down_idx, const void* unif_, uint8_t* accept_out, double* stats7) override {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    if (sys.n_up < 1 || sys.n_down < 1) return fail(DQMC_E_ARG, "an exchange step needs electrons of both spins");
    real* r = (real*)r_; real* logpsi = (real*)logpsi_;
    const size_t n_r = (size_t)B * N * 3;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t o_rp = 0, o_lp = o_rp + al(sizeof(real) * n_r), o_sp = o_lp + al(sizeof(real) * B), tot = o_sp + al(sizeof(int32_t) * B);
    int rc = ensure_mc(tot);
    if (rc) return rc;
    real* r_prop = (real*)(d_mc + o_rp); real* lp_prop = (real*)(d_mc + o_lp); int32_t* s_prop = (int32_t*)(d_mc + o_sp);
    t_begin("mcmc", 0);
    dqmc::launch_exchange_propose<real>(st, r, up_idx, down_idx, sys.n_up, B, N, r_prop);
    t_end();
    rc = run(r_prop, (const real*)R_, B, false, lp_prop, s_prop, nullptr, nullptr, nullptr);
    if (rc) return rc;
    t_begin("mcmc", 0);
    // `_accept` without max_age / target_acceptance (electron_samplers.py:312-313): no age override, tau unchanged
    dqmc::launch_accept<real>(st, r, logpsi, sign, age, r_prop, lp_prop, s_prop, (const real*)unif_, -1, B, N, d_nacc, accept_out);
    dqmc::launch_read_accept(st, d_nacc, B, d_acc);
    t_end();
    if (stats7) {
      dqmc::launch_sampler_stats<real>(st, r, logpsi, age, (const real*)tau_, d_acc, B, N, sys.norm_eps, d_acc + 1);
      HIP_TRY(hipMemcpyAsync(stats7, d_acc + 1, sizeof(double) * 7, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    HIP_TRY(hipGetLastError());
    return DQMC_OK;
  }

  int energy_stats(const void* e, const void* w, int B, double* out7) override {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    dqmc::launch_energy_stats<real>(st, (const real*)e, (const real*)w, B, d_acc + 8);
    HIP_TRY(hipMemcpyAsync(out7, d_acc + 8, sizeof(double) * 7, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return DQMC_OK;
  }

  int energy_stats_dev(const void* e, const void* w, int B, double** rec_dev) override {
    if (B < 1) return fail(DQMC_E_ARG, "B must be positive");
    dqmc::launch_energy_stats<real>(st, (const real*)e, (const real*)w, B, d_acc + 8);
    *rec_dev = d_acc + 8;
    return DQMC_OK;
  }

  int debug_read(int buf, double* out, size_t n) override {
    if (last_B == 0) return fail(DQMC_E_ARG, "no evaluation has run yet");
    HIP_TRY(hipStreamSynchronize(st));
    if (buf == -1) {
      const size_t cnt = (size_t)last_B * sys.n_det * last_TP;
      if (n != cnt) return fail(DQMC_E_ARG, "size mismatch");
      HIP_TRY(hipMemcpy(out, d_ws + off_logdet, sizeof(double) * cnt, hipMemcpyDeviceToHost));
      return DQMC_OK;
    }
    if (buf == -4) {   // conditioning record per walker of the last Laplacian-mode evaluation
      if (n != (size_t)last_B || last_TP == 1) return fail(DQMC_E_ARG, "size mismatch or no Laplacian-mode evaluation");
      HIP_TRY(hipMemcpy(out, d_ws + off_kappa, sizeof(double) * n, hipMemcpyDeviceToHost));
      return DQMC_OK;
    }
    if (buf == -3) {   // per-op shader-clock stamps of the fused kernel (workgroup 0)
      if (!d_prof || n > 9 * ops.size() + 80 + 1024 + 2 * 8192) return fail(DQMC_E_ARG, "profile not enabled or size mismatch");
      std::vector<long long> tmp(n);
      HIP_TRY(hipMemcpy(tmp.data(), d_prof, sizeof(long long) * n, hipMemcpyDeviceToHost));
      for (size_t k = 0; k < n; ++k) out[k] = (double)tmp[k];
      return DQMC_OK;
    }
    if (buf == -2) {
      const size_t cnt = (size_t)last_B * sys.n_det;
      if (n != cnt) return fail(DQMC_E_ARG, "size mismatch");
      std::vector<int32_t> tmp(cnt);
      HIP_TRY(hipMemcpy(tmp.data(), d_ws + off_signk, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));
      for (size_t k = 0; k < cnt; ++k) out[k] = tmp[k];
      return DQMC_OK;
    }
    if (buf < 0 || buf >= (int)bufs.size()) return fail(DQMC_E_ARG, "no such buffer");
    const size_t cnt = (size_t)last_B * bufs[buf].rows * last_TP * bufs[buf].width;
    if (n != cnt) return fail(DQMC_E_ARG, "size mismatch: expected " + std::to_string(cnt));
    const int lanes = lanes_of(buf, last_TP);
    const size_t cnt_dev = (size_t)last_B * bufs[buf].rows * lanes * bufs[buf].width;
    std::vector<real> tmp(cnt_dev);
    HIP_TRY(hipMemcpy(tmp.data(), d_ws + buf_off[buf], sizeof(real) * cnt_dev, hipMemcpyDeviceToHost));
    if (lanes == last_TP) {
      for (size_t k = 0; k < cnt; ++k) out[k] = (double)tmp[k];
      return DQMC_OK;
    }
    // pair-compact buffer: expand to the documented full-lane layout [B][rows][TP][width]
    const int T = 3 * N + 2, W = bufs[buf].width, rows = bufs[buf].rows;
    std::fill(out, out + cnt, 0.0);
    for (int b = 0; b < last_B; ++b)
      for (int row = 0; row < rows; ++row) {
        const int rc = pair_rs[buf][2 * row], sd = pair_rs[buf][2 * row + 1];
        for (int ct = 0; ct < lanes; ++ct) {
          int t;
          if (ct == 0) t = 0;
          else if (ct == lanes - 1) t = T - 1;
          else if (ct < 4) t = 1 + 3 * rc + (ct - 1);
          else if (sd < 0 || sd == rc) continue;
          else t = 1 + 3 * sd + (ct - 4);
          const real* src = tmp.data() + (((size_t)b * rows + row) * lanes + ct) * W;
          double* dst = out + (((size_t)b * rows + row) * last_TP + t) * W;
          for (int c = 0; c < W; ++c) dst[c] = (double)src[c];
        }
      }
    return DQMC_OK;
  }
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

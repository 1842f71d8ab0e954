// kernels.h -- host-callable launchers of the gfx950 kernels (explicitly instantiated for
// float and double in the .hip files).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dqmc.h"
#include "common.h"

namespace dqmc {

// ---- kernels_graph.hip ----
template <typename real>
void launch_feat_en(hipStream_t st, const real* r, const real* R, real* x, int B, int n_nuc, int n_up, int width,
                    LaneInfo li, double eps, int log_rescale, int use_spin, const double* phq = nullptr);
template <typename real>
void launch_feat_ee(hipStream_t st, const real* r, const real* R, const int32_t* pairs, real* e, int B, int n_rows, LaneInfo li,
                    double eps, int log_rescale, int compact, const double* phq = nullptr);
template <typename real>
void launch_const_rows(hipStream_t st, const real* tab, real* x, int B, int rows, int width, LaneInfo li);
template <typename real>
void launch_spin_mean(hipStream_t st, const real* x, real* m, int B, int n_up, int width, LaneInfo li);
template <typename real>
void launch_row_sum(hipStream_t st, const real* x, real* s, int B, int rows, int width, LaneInfo li);
template <typename real>
void launch_conv(hipStream_t st, const real* we, int we_rows, int we_width, const real* hx, int hx_rows, int hx_width, real* out,
                 int out_width, int col0, const int32_t* tab, int S, int W, int B, LaneInfo li, int compact);
template <typename real>
void launch_edge_sum(hipStream_t st, const real* e, int e_rows, int e_width, real* out, int out_width, int col0,
                     const int32_t* tab, int S, int W, double scale, int B, LaneInfo li, int compact);

// ---- kernel_linear.hip ----
// One concat piece of the A operand: logical row (b, rr, t) reads
// src[((b*rpw + r0 + (bcast ? 0 : rr))*TP + t)*ld + k], k < K (K % 4 == 0, zero padded).
template <typename real> struct LinPiece {
  const real* src;
  int ld, rpw, r0, K, bcast;
  int w_row;          // first row of this piece's block in W
};
template <typename real> struct LinArgs {
  int n_pieces;
  LinPiece<real> piece[4];
  const real* W;      // [sum K][ldw]
  int ldw;            // pad4(Nout)
  const real* bias;   // [ldw] or nullptr (value lane only)
  real* dst;
  int ld_dst, rpw_dst, r0_dst, col0_dst;
  const real* pre;    // per-walker addend to the pre-activation, real[B][TP][ld_pre], or nullptr: the product of the
  int ld_pre;         //   per-walker (broadcast) pieces, computed once per walker by a separate launch
  const real* res;    // residual input or nullptr: out = (res + y) * res_scale
  int ld_res, rpw_res, r0_res;
  real res_scale;
  int act;            // 0 none, 1 tanh, 2 silu
  int nrows;          // rows per walker in this segment
  int B;
  int T, TP;
  // chained second layer (launch_linear_chain): Y = act2(H W2 + bias2) with H = act(X W + bias) kept on chip;
  // dst / res / col0_dst then describe Y
  const real* W2;     // [ldw][ldw2]
  int ldw2;           // pad4(Nout of the second layer)
  const real* bias2;
  int act2;
  // float64 launches only: the A pieces are FLOAT32 buffers (`src` points at float data, same ld / rows / lanes) that the
  // kernel widens while it stages them -- the float64 tail of a float32 pass reads the head's activations where they lie
  // (plain and 8- / 16- / 32- / 48- / 64-lane tiles; not the split-group and chained instances)
  int src_f32;
  // kernel-selection switches of the calling context (dqmc_set_option "linear_bf" / "linear_bkx" / "linear_f64_split";
  // read on the host by launch_linear only): per launch, so that contexts -- a float32 engine and its float64 twin,
  // contexts of other threads -- do not steer each other and a captured pass keeps what its own context chose
  int cfg_bf, cfg_bkx, cfg_f64_split;
#ifdef DQMC_LIN_PROBE
  // ablation hooks of tools/probe_lin.sh (a build with -DDQMC_LIN_PROBE only, never the product): bit 0 no MFMAs, 1 no epilogue,
  // 2 no A-tile loads, 3 no barriers in the K loop; from bit 4 up: s_sleep units for the second workgroup of every CU at launch
  int cfg_probe;
#endif
};
template <typename real> void launch_linear(hipStream_t st, const LinArgs<real>& a);
template <typename real> void launch_linear_chain(hipStream_t st, const LinArgs<real>& a);
bool linear_chain_supported(int TP, int ldw_hidden, int ldw_out);
constexpr int LINEAR_BF_DEFAULT = 2, LINEAR_BKX_DEFAULT = 3;      // (kernel_linear.hip: what the values select)

// ---- kernel_fused2.hip: LDS-resident value-only psi evaluation, descriptor driven ----
struct FusedBuf {
  int off;        // LDS offset in elements (LDS-resident buffers)
  int stride;     // LDS row stride in elements (width + 2)
  int rows;       // rows per walker
  int width;
  int is_global;  // 1: lives in the HBM workspace (read by later kernels)
  long goff;      // byte offset of the buffer in the workspace
};
// One entry of a wave's work list; built on the host (engine.hip: build_fused2_plan), read through scalar loads.
// quads of k-steps per B-operand group of the fused kernel (kernel_fused2.hip: BSet); lean units hold at most one group
// (float: three slots per column block, which are also the three bf16 planes of a K = 32 chunk of a bf16 unit)
template <typename real> struct FusedGroup { static constexpr int P = sizeof(real) == 4 ? 3 : 2; };
struct FDesc {
  int32_t kind;         // 0 end of list, 1 linear unit, 2 workgroup barrier, 3 structured op (all waves), 4 wave-local LDS fence, 5 lean linear unit (one piece, at most FusedGroup::P quads of k-steps, LDS destination), 6 / 7 linear unit / lean unit on the bf16 matrix pipe (float only; a_ks = octets of k, a_nq = chunks of 32 k, qstride = Vec4 per chunk)
  int32_t op;           // scheduled op index (FusedArgs::ops)
  int32_t ma;           // row blocks of the unit (1..4); column blocks are always 2
  int32_t n_pieces;
  int32_t row0;         // first tile row m = row*WT + wl of the unit
  int32_t rtot;         // rows of the segment in the tile (WT * nrows)
  int32_t col0;         // first output column of the unit
  int32_t ldw;          // padded output width of the layer
  int32_t a_base[4];    // LDS offset of segment row m = 0 of concat piece p
  int32_t a_stride[4];
  int32_t a_ks[4];      // k-steps (pad4(K)/4)
  int32_t a_nq[4];      // quads of 4 k-steps
  int32_t bcast;        // bit p: piece p is one row per walker (row index m & (WT-1))
  int32_t w_off;        // Vec4 index of (quad 0, first column block of the unit, lane 0) in the packed weights
  int32_t w_cb1;        // Vec4 distance to the second column block (0: none, re-read the first)
  int32_t qstride;      // Vec4 per quad of k-steps
  int32_t bias_off;     // into the plain weight buffer, or -1
  int32_t dst_base;     // LDS offset of (segment row 0, col0_dst); HBM destination: buffer index
  int32_t dst_stride;
  int32_t res_base;     // LDS offset of (residual segment row 0, col0_dst), or -1
  int32_t res_stride;
  int32_t flags;        // bits 0-1 activation, bit 2 scale by 1/sqrt(2), bit 3 destination in HBM, bit 4 no partial row/column block
  int32_t g_r0, g_col0; // HBM destination: first row / column
  // first weight group of the wave's NEXT unit (of this unit itself when it is the last): the unit requests it before its
  // own epilogue without touching that unit's descriptor (a dependent scalar load = one more ~250-cycle round trip)
  int32_t nx_w_off, nx_cb1, nx_qstride, nx_nq;
};
static_assert(sizeof(FDesc) == 160, "FDesc is 40 dwords");
// Metropolis sub-step folded into the fused kernel (N <= 4): propose r' = r + tau xi in the prologue, the K
// Slater determinants, the CI sum / cusp / Jastrow and the accept step in the tail -- one launch per sub-step
// instead of seven (electron_samplers.py:102-138).  enabled = 0: plain psi evaluation.
struct FusedMc {
  int enabled;
  const void* noise;      // real[B][N][3]
  const void* unif;       // real[B]
  void* r;                // real[B][N][3] sampler state, updated in place
  void* logpsi;           // real[B]
  int32_t* sign;          // [B]
  int32_t* age;           // [B]
  // Step size without a launch of its own: sub-step s reads the step size and the acceptance count of sub-step
  // s-1 (tau_ring[(s+1)&1], counters[(s+2)%3]), every workgroup derives the same tau_s
  // (electron_samplers.py:121-126), workgroup 0 records it in tau_ring[s&1] and clears counters[(s+1)%3];
  // accepts of this sub-step go to counters[s%3].  k_tau_finalize after the last sub-step writes the caller's tau.
  const void* tau_in;     // real[1]: the caller's step size (used by sub-step 0)
  void* tau_ring;         // real[2]
  int32_t* counters;      // int[3], all zero between calls
  int s;                  // sub-step index within this call
  double target;          // target acceptance (<= 0: no adaptation)
  uint8_t* accept_out;    // [B] or nullptr
  int max_age;
  int orb_op;             // scheduled index of the ORBITALS op (backflow buffer, envelope tables)
  int mat_off;            // LDS offset (reals) of WT*K*N*N free elements for the Slater matrices (dead activations)
  int jas_width;          // > 0: a Jastrow value per walker arrives through the scratch area
  int cc_off, al_off, cusp_kind;
  double same_scale, anti_scale;
};
template <typename real> struct Fused2Args {
  const FDesc* descs;       // device: the four wave lists, concatenated
  const int32_t* wave_begin; // device int[8]: first descriptor of each wave, then the first kind-1 descriptor of each wave (-1: none)
  const ::dqmc_op* ops;     // device: the scheduled ops (structured ops read their fields from here)
  const FusedBuf* fbufs;    // device: LDS offset / stride (tile layout [row][WT]) or HBM offset per buffer
  const real* w;
  const real* wpk;
  const int32_t* itable;
  char* ws;
  const real* r;
  const real* R;
  int B, WT, wt_shift, n_up, n_nuc, K;
  int scratch_off;          // LDS offset (in reals) of the per-tile scratch: positions, Jastrow, log|det|, det signs
  int it_off, n_it;         // LDS byte offset and length of the staged int table
  int ma1;                  // every unit is one row block high: launch the specialised kernel
  int prio_mode;            // 0: hardware default (oldest wave first); 1/2: issue priority rotates among the co-resident workgroups per level / per unit
  int stagger_div;          // workgroups per dispatch wave (the co-resident tiles of a CU are blockIdx / stagger_div apart)
  FusedMc mc;
  long long* prof;          // optional clock stamps of workgroup 0: [wave][256]
  long long* prof_wg;       // optional constant-rate (100 MHz) stamps of EVERY workgroup: [n_blocks][2] start, end
  LaneInfo li;
  double eps;
};
// bytes of that scratch area: r tile [WT][N][3], Jastrow [WT][4], log|det| double[WT][K], sign int[WT][K], CI shift double[WT],
// then the staged int table (n_it ints, 16-byte aligned)
__host__ __device__ inline int fused2_scratch_core(int WT, int N, int K, int real_size) {
  return ((WT * (N * 3 + 4) * real_size + 15) / 16 * 16) + WT * K * 12 + WT * 8;
}
__host__ __device__ inline int fused2_scratch_bytes(int WT, int N, int K, int real_size, int n_it = 0) {
  return (fused2_scratch_core(WT, N, K, real_size) + 15) / 16 * 16 + (4 * n_it + 15) / 16 * 16;
}
template <typename real> void launch_fused2_value(hipStream_t st, const Fused2Args<real>& a, int n_blocks, size_t lds_bytes, int occ);
template <typename real> int fused2_set_lds_limit(size_t lds_bytes);

// ---- kernel_attention.hip ----
// returns 0, or -1 if the (N, head_dim) tile set does not fit the 160 KiB LDS, -2 on a HIP error
template <typename real>
int launch_attention(hipStream_t st, const real* q, const real* k, const real* v, real* out, int width, int H, int hd,
                     int B, LaneInfo li, int n_const, const real* k_const, const real* v_const);
template <typename real> size_t attention_lds_bytes(int N, int hd, int n_const);

// ---- kernel_attention_mfma.hip: contractions on MFMA, float32 and float64 (head_dim % 16 == 0, <= 64 queries / keys) ----
template <typename real> size_t attention_mfma_lds_bytes(int N, int hd, int n_const);
template <typename real> bool attention_mfma_supported(int N, int hd, int n_const);
bool attention_mfma_profitable(int N);
// split variant: eight waves, a pair per query row block (kernel_attention_mfma.hip); returns -1 when the shape is not supported
template <typename real> size_t attention_mfma_split_lds_bytes(int N, int hd, int n_const);
template <typename real>
int launch_attention_mfma_split(hipStream_t st, const real* q, const real* k, const real* v, real* out, int width, int H, int hd, int B,
                                LaneInfo li, int n_const, const real* k_const, const real* v_const);
template <typename real>
int launch_attention_mfma(hipStream_t st, const real* q, const real* k, const real* v, real* out, int width, int H,
                          int hd, int B, LaneInfo li, int n_const, const real* k_const, const real* v_const,
                          int exact_tiles = 1);      // 0: the instance for four key tiles whatever their number (option "attention_ncb")

// ---- kernels_head.hip ----
template <typename real>
void launch_orbitals(hipStream_t st, const real* r, const real* R, const real* bf, int bf_width, real* orb,
                     int orb_width, const real* pi_up, const real* pi_dn, const real* ze_up, const real* ze_dn, int B,
                     int n_up, int n_nuc, int n_env, int K, LaneInfo li, double eps, const double* phq = nullptr);
template <typename real>
void launch_slogdet(hipStream_t st, const real* orb, int orb_width, double* logdet, int32_t* sign_k, int B, int K,
                    LaneInfo li, int use_mfma, double* cond = nullptr);
struct FinalArgs {
  const void* r;          // real[B][N][3]
  const void* R;          // real[n_nuc][3]
  const double* charges;  // device double[n_nuc] (valence charges Z_eff when an ECP is set)
  const double* ecp_loc;  // device double[n_nuc][3][2][ecp_nt] local ECP terms, or nullptr
  int ecp_nt;
  const double* logdet;   // [B][K][TP]
  const int32_t* sign_k;  // [B][K]
  const void* jastrow;    // real[B][1][TP][jas_width] or nullptr
  int jas_width;
  const void* conf_coeff; // real[K] or nullptr
  const void* alphas;     // real[2]
  int cusp_kind;
  double same_scale, anti_scale;
  double eps;
  int B, n_up, n_nuc, K;
  LaneInfo li;
  // outputs (any may be nullptr)
  void* logpsi;   // real[B]
  int32_t* sign;  // [B]
  void* e_loc;    // real[B]
  void* stats;    // real[6][stats_ld], this call's walkers at columns 0..B-1
  long stats_ld;  // leading dimension of stats (the caller's total batch when evaluating a chunk)
  void* grad;     // real[B][3N]
  // float64 refinement (float32 build): the float32 error of E_loc is predicted by
  //   score = (|lap| + |grad|^2) / max(1, |E_loc|)  x  max(1, sum_k |p_k| kappa_k)
  // -- the cancellation in E_kin = -(lap + |grad|^2)/2 (both terms ~ 1/psi^2 near a node) times the conditioning record
  // of the Slater matrices that carry psi (kernels_head.hip, above k_slogdet).  Walkers with score > refine_thresh or a
  // non-finite E_loc are appended to flag_idx (global walker index b_offset + b); nullptr: no flagging.
  // score_out [B_total] (indexed like flag_idx) and kappa_out [B] (this chunk) receive the two records.
  int32_t* flag_count;
  int32_t* flag_idx;
  double refine_thresh;
  const double* thresh_dev;  // when set: the threshold is read from here (a replayed hipGraph of the pass follows re-calibrations)
  int b_offset;
  const double* cond;     // [B][K] conditioning record of the determinant kernels, or nullptr
  double* score_out;
  double* kappa_out;
  // pseudo-Hamiltonian (ecp/pseudo_hamiltonian.py): per-(walker, electron) factors [B][N][PH_STRIDE] the derivative
  // lanes were seeded with (and the electron's share of the local PH potential); nullptr = ordinary kinetic energy
  const double* phq;
};
template <typename real> void launch_final(hipStream_t st, const FinalArgs& a);

// ---- kernels_ecp.hip ----
struct EcpArgs {
  const void* r;           // real[B][N][3], all walkers
  const void* R;           // real[n_nuc][3]
  const int32_t* nl_nuc;   // device [n_nl]: nuclei with a non-local part
  const double* nl;        // device [n_nl][L][2][n_t]: exponents ([.,l,0,.]) and coefficients ([.,l,1,.])
  const void* phi;         // real[B][n_nl][N] rotation angles, or nullptr: Philox(seed)
  // float64 twin evaluating a gathered subset of a float32 context's walkers: walker b of this call is walker
  // walker_idx[b] of the caller's batch (rotation angles -- explicit or Philox -- are keyed by THAT index), and the
  // explicit angles are the caller's float array
  const int32_t* walker_idx;
  int phi_f32;
  uint64_t seed;
  int B, N, n_nl, L, n_t;
  int b0, nb;              // walker chunk of this launch
};
// mixed-precision quadrature of a float32 context (kernels_ecp.hip)
struct EcpMixArgs {
  const float* r;          // float[B][N][3], all walkers of the caller
  const float* R;          // float[n_nuc][3]
  const int32_t* nl_nuc;
  const double* nl;
  const float* phi;        // float[B][n_nl][N] rotation angles, or nullptr: Philox(seed)
  uint64_t seed;
  int B, N, n_nl, L, n_t;
  int b0, nb;              // walker chunk of this launch
  double w_heavy, w_skip;  // class bounds on w = max_l (2l+1) |V_l(r_ia)|
  // Measured float32 error of log|psi(r)| of the chunk's own walkers (both value paths evaluate them before the pairs are
  // classified): l32 / l64 [nb] and their signs, or nullptr.  psi(r) is the denominator of every ratio of a walker, and near
  // a node it is the quantity float32 cannot resolve (while the ratios themselves grow like 1 / psi(r)): the float64 bound
  // of a walker's pairs tightens in proportion, w x max(1, |l32 - l64| / dlog_floor) > w_heavy; a sign mismatch sends
  // every kept pair of the walker to float64.
  const float* l32; const double* l64; const int32_t* s32; const int32_t* s64;
  double dlog_floor;
};
void launch_ecp_classify(hipStream_t st, const EcpMixArgs& a, int32_t* cls, int32_t* list_l, int32_t* list_h, int32_t* counts);
template <typename real_out> void launch_ecp_points_list(hipStream_t st, const EcpMixArgs& a, const int32_t* list, int n_list, real_out* rq);
void launch_ecp_reduce_mixed(hipStream_t st, const EcpMixArgs& a, const int32_t* cls, const float* lq32, const int32_t* sq32,
                             const double* lq64, const int32_t* sq64, float* e_loc, float* stats);
template <typename real> void launch_ecp_points(hipStream_t st, const EcpArgs& a, real* rq);
template <typename real>
void launch_ecp_reduce(hipStream_t st, const EcpArgs& a, const real* logq, const int32_t* signq, const real* log0,
                       const int32_t* sign0, real* e_loc, real* stats, real* v_nl_out);
template <typename real>
void launch_ph_coeffs(hipStream_t st, const real* r, const real* R, const int32_t* ph_nuc, int n_ph, const double* rv_loc,
                      const double* rv_l2, int n_grid, double r_max, int B, int N, double* phq);

// ---- kernels_mcmc.hip ----
template <typename real>
void launch_rng(hipStream_t st, real* noise, long n_noise, real* unif, long n_unif, uint64_t seed, uint64_t stream_id);
template <typename real>
void launch_propose(hipStream_t st, const real* r, const real* noise, const real* tau, real* r_prop, long n);
template <typename real>
void launch_accept(hipStream_t st, real* r, real* logpsi, int32_t* sign, int32_t* age, const real* r_prop,
                   const real* logpsi_prop, const int32_t* sign_prop, const real* unif, int max_age, int B, int N,
                   int32_t* n_accept, uint8_t* accept_out);
template <typename real>
void launch_tau_update(hipStream_t st, real* tau, int32_t* n_accept, int B, double target, double* acc_out);
template <typename real>
void launch_tau_finalize(hipStream_t st, real* tau, const real* tau_ring, int32_t* counters, int s_last, int B, double target,
                         double* acc_out);
template <typename real>
void launch_sampler_stats(hipStream_t st, const real* r, const real* logpsi, const int32_t* age, const real* tau,
                          const double* acc, int B, int N, double eps, double* stats7);
template <typename real>
void launch_energy_stats(hipStream_t st, const real* e_loc, const real* w, int B, double* out7);
template <typename real>
void launch_clean_force(hipStream_t st, const real* grad, const real* r, const real* R, const double* Z, const real* tau, int B,
                        int N, int n_nuc, real* force);
template <typename real>
void launch_langevin_propose(hipStream_t st, const real* r, const real* force, const real* noise, const real* tau, real* r_prop, long n);
template <typename real>
void launch_langevin_accept(hipStream_t st, real* r, real* logpsi, int32_t* sign, int32_t* age, real* force, const real* r_prop,
                            const real* lp_prop, const int32_t* sign_prop, const real* force_prop, const real* unif, const real* tau,
                            int max_age, int B, int N, int32_t* n_accept, uint8_t* accept_out);
template <typename real>
void launch_exchange_propose(hipStream_t st, const real* r, const int32_t* up_idx, const int32_t* down_idx, int n_up, int B, int N,
                             real* r_prop);
void launch_read_accept(hipStream_t st, int32_t* n_accept, int B, double* acc_out);
void launch_widen(hipStream_t st, const float* src, double* dst, long n);
void launch_tail_narrow(hipStream_t st, int n, int n3, const double* e64, const double* st64, const double* g64, const double* lp64,
                        const int32_t* sg64, float* e_loc, float* stats, long stats_ld, float* grad, float* logpsi, int32_t* sign);
void launch_refine_gather(hipStream_t st, const float* r, const float* R, const int32_t* idx, const int32_t* count, int n, int n3,
                          int nR3, double* r64, double* R64);
void launch_refine_scatter(hipStream_t st, const int32_t* idx, const int32_t* count, int n, int n_scatter, const double* score,
                           double thresh, int n3, const double* e64, const double* st64,
                           const double* g64, const double* lp64, const int32_t* sg64, float* e_loc, float* stats,
                           long stats_ld, float* grad, float* logpsi, int32_t* sign);

}  // namespace dqmc

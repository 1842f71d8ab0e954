// kernel_fused.hip -- value-only psi evaluation (the Metropolis hot loop) as ONE kernel.
//
// The MCMC sub-step needs psi at the proposed positions of every walker
// (reference sampling/electron_samplers.py:76-81: vmap(wf)).  Executed op by op this is ~60
// small launches per sub-step and is launch/latency bound; here a workgroup owns a tile of WT
// walkers and runs the whole layer program on it with every activation resident in LDS:
//   pair features -> [w/h/u MLPs, spin means, convolutions, g layer] x L -> Jastrow/backflow
//   heads -> envelope * backflow -> Slater matrix entries (written to HBM for the slogdet kernel).
// Linear layers: A fragments straight from the LDS activation buffers (row stride = width + 2
// floats, i.e. 2 mod 4 => the 16 rows x 2 k of a half-wave hit 32 distinct banks), B fragments
// from a fragment-major packed copy of the weights in L2 (one coalesced 256-byte load per
// 16x4 tile), v_mfma_f32_16x16x4_f32 accumulation, bias/activation/residual fused in the store.
//
// Latency structure (what the SQ counters said mattered): the program itself (ops, buffer
// table) is copied to LDS once so per-op metadata is a ds_read away, ops arrive grouped in
// dependency levels (engine.hip: fused_schedule) and only level boundaries carry a workgroup
// barrier, and the linear "units" of a whole level are dealt round-robin to the waves.
//
// Address spaces are kept explicit for the compiler: LDS is only ever indexed through the
// dynamic-LDS symbol (ds_read/ds_write, never flat_*), HBM/L2 only through kernel-argument
// pointers.
#include "common.h"
#include "kernels.h"

namespace dqmc {

struct BufRef {
  int lds;      // element offset in LDS, or -1: the buffer lives in the HBM workspace
  long goff;    // byte offset in the workspace (HBM buffers)
  int stride;   // row stride in elements
  int rows;     // rows per walker
  int width;
};

// q = m / n, r = m % n for 0 <= m < 2^22, n >= 1 without the ~25-instruction integer division.
struct FastDiv {
  int n;
  float inv;
  __device__ __forceinline__ explicit FastDiv(int n_) : n(n_), inv(1.0f / (float)n_) {}
  __device__ __forceinline__ void divmod(int m, int& q, int& r) const {
    q = (int)((float)m * inv);
    r = m - q * n;
    if (r < 0) { r += n; --q; }
    if (r >= n) { r -= n; ++q; }
  }
};

template <typename real> __device__ __forceinline__ real act_value(int act, real v) {
  if (act == 1) return r_tanh<real>(v);
  if (act == 2) return v / (1 + r_exp<real>(-v));
  return v;
}

// LDS-resident program: [ops][buffer table][op words], then the activation buffers.
template <typename real> struct Meta {
  const dqmc_op* ops;
  const FusedBuf* bufs;
  const int32_t* words;
  __device__ __forceinline__ BufRef buf(int b) const {
    const FusedBuf fb = bufs[b];
    BufRef v;
    v.rows = fb.rows; v.width = fb.width; v.goff = fb.goff;
    v.lds = fb.is_global ? -1 : fb.off;
    v.stride = fb.is_global ? fb.width : fb.stride;
    return v;
  }
};

// One unit of a fused linear layer: MA row blocks x 2 column blocks of
// y = act(concat(pieces) W + b) (+ residual) on the tile.  The k loop is branch-free
// (out-of-range rows / column blocks read a valid dummy location and are dropped at the store)
// and the B fragments of the next 4 k-steps are in flight while the current 4 are multiplied.
template <typename real, int MA>
__device__ __forceinline__ void fused_linear_unit(const FusedArgs<real>& a, const Meta<real>& mt, const dqmc_op& op,
                                                  const real* wpk, int nw, int w0, int rb0, int cg, int slot) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  typedef typename Mfma<real>::acc_t acc_t;
  constexpr int NRW = 2;
  const int32_t* i = op.i;
  const int lane = threadIdx.x & 63;
  const bool stamp = a.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  if (stamp) a.prof[64 + 8 * slot + 0] = clock64();
  const int nrows = i[20];
  const FastDiv fd(nrows);
  const int ldw = (i[21] + 3) / 4 * 4;
  const int Rtot = nw * nrows, NCB = (ldw + 15) / 16;
  int cbi[NRW];
#pragma unroll
  for (int y = 0; y < NRW; ++y) cbi[y] = (cg * NRW + y < NCB) ? cg * NRW + y : NCB - 1;
  // A rows of this lane: (walker, row) of the MA row blocks
  int a_wl[MA], a_rr[MA];
#pragma unroll
  for (int x = 0; x < MA; ++x) {
    int m = (rb0 + x) * 16 + (lane & 15);
    if (m >= Rtot) m = 0;                                 // dummy row, result discarded
    fd.divmod(m, a_wl[x], a_rr[x]);
  }
  acc_t acc[MA][NRW];
#pragma unroll
  for (int x = 0; x < MA; ++x)
#pragma unroll
    for (int y = 0; y < NRW; ++y) acc[x][y] = acc_t{0, 0, 0, 0};
  // bias: issued now so that its L2 round trip overlaps the k loop instead of following it
  real bias_v[NRW];
#pragma unroll
  for (int y = 0; y < NRW; ++y) {
    const int col = cbi[y] * 16 + (lane & 15);
    bias_v[y] = (i[23] >= 0 && col < ldw) ? a.w[i[23] + col] : (real)0;
  }
  if (stamp) a.prof[64 + 8 * slot + 1] = clock64();
  int q0 = 0;
  for (int p = 0; p < i[0]; ++p) {
    const BufRef src = mt.buf(i[1 + 4 * p]);
    const int r0 = i[2 + 4 * p], Kp = (i[3 + 4 * p] + 3) / 4 * 4, bc = i[4 + 4 * p];
    int ao[MA];
#pragma unroll
    for (int x = 0; x < MA; ++x)
      ao[x] = src.lds + (a_wl[x] * src.rows + r0 + (bc ? 0 : a_rr[x])) * src.stride + (lane >> 4);
    const int KS = Kp / 4;
    const int NQ = (KS + 3) / 4;                 // quads of 4 k-steps; the packed weights are zero padded
    // B fragments: one 16-byte load per lane = 4 consecutive k-steps of one 16-column block
    // (fragment-major, quad-interleaved packing, engine.hip: pack_fused_weights).  A ring of D quads
    // (16 k-steps) stays in flight to cover the L2 round trip; A fragments (LDS) run one quad ahead.
    constexpr int D = 4;
    const Vec4<real>* wq[NRW];
#pragma unroll
    for (int y = 0; y < NRW; ++y) wq[y] = reinterpret_cast<const Vec4<real>*>(wpk) + ((long)(q0 * NCB + cbi[y]) * 64 + lane);
    const int qstride = NCB * 64;
    Vec4<real> ring[D][NRW];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (d < NQ) {                                // wave-uniform; short layers load only what they use
#pragma unroll
        for (int y = 0; y < NRW; ++y) ring[d][y] = wq[y][(long)d * qstride];
      }
    }
    if (stamp && p == 0) { a.prof[64 + 8 * slot + 2] = clock64(); a.prof[64 + 8 * slot + 6] = (long long)(ring[0][0].v[0] != 0); a.prof[64 + 8 * slot + 3] = clock64(); }
    real fan[4][MA];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ks = kk < KS ? kk : KS - 1;
#pragma unroll
      for (int x = 0; x < MA; ++x) fan[kk][x] = smem[ao[x] + ks * 4];
    }
    for (int q = 0; q < NQ; q += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (q + d < NQ) {                          // wave-uniform
          Vec4<real> cur[NRW];
#pragma unroll
          for (int y = 0; y < NRW; ++y) cur[y] = ring[d][y];
          if (q + d + D < NQ) {
#pragma unroll
            for (int y = 0; y < NRW; ++y) ring[d][y] = wq[y][(long)(q + d + D) * qstride];
          }
          real fa[4][MA];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int x = 0; x < MA; ++x) fa[kk][x] = fan[kk][x];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {         // A fragments of the next quad (clamped past the end)
            int ks = (q + d + 1) * 4 + kk;
            ks = ks < KS ? ks : KS - 1;
#pragma unroll
            for (int x = 0; x < MA; ++x) fan[kk][x] = smem[ao[x] + ks * 4];
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int x = 0; x < MA; ++x) {
              acc[x][0] = Mfma<real>::run(fa[kk][x], cur[0].v[kk], acc[x][0]);
              acc[x][1] = Mfma<real>::run(fa[kk][x], cur[1].v[kk], acc[x][1]);
            }
        }
      }
    }
    q0 += NQ;
  }
  if (stamp) a.prof[64 + 8 * slot + 4] = clock64();
  // epilogue: bias + activation + residual, store to LDS or HBM
  const BufRef dst = mt.buf(i[17]);
  const bool has_res = i[25] >= 0;
  BufRef res = dst;
  if (has_res) res = mt.buf(i[25]);     // residual inputs are LDS-resident (engine plan)
  const real res_scale = i[27] ? (real)0.70710678118654752440 : (real)1;
  const int act = i[24], col0 = i[19];
  real* dst_g = reinterpret_cast<real*>(a.ws + dst.goff) + (long)w0 * dst.rows * dst.width;
  // phase 1: destination / residual row offsets of the 4*MA rows this lane holds (once, not per column)
  int d_off[MA][4], r_off[MA][4];
  bool r_ok[MA][4];
#pragma unroll
  for (int x = 0; x < MA; ++x)
#pragma unroll
    for (int rgi = 0; rgi < 4; ++rgi) {
      int m = (rb0 + x) * 16 + Mfma<real>::row_of(lane, rgi);
      r_ok[x][rgi] = m < Rtot;
      if (m >= Rtot) m = 0;
      int wl, rr;
      fd.divmod(m, wl, rr);
      d_off[x][rgi] = (wl * dst.rows + i[18] + rr) * dst.stride + col0;
      r_off[x][rgi] = res.lds + (wl * res.rows + i[26] + rr) * res.stride + col0;
    }
  int colv[NRW];
  bool c_ok[NRW];
#pragma unroll
  for (int y = 0; y < NRW; ++y) {
    colv[y] = (cg * NRW + y) * 16 + (lane & 15);
    c_ok[y] = colv[y] < ldw;
    if (!c_ok[y]) colv[y] = 0;
  }
  // phase 2: residual reads issued together (valid dummy addresses for masked elements)
  real rv[MA][4][NRW];
  if (has_res) {
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y) rv[x][rgi][y] = smem[r_off[x][rgi] + colv[y]];
  }
  // phase 3: bias + activation (+ residual) in registers
  real ov[MA][4][NRW];
#pragma unroll
  for (int x = 0; x < MA; ++x)
#pragma unroll
    for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
      for (int y = 0; y < NRW; ++y) ov[x][rgi][y] = acc[x][y][rgi] + bias_v[y];
  if (act == 1) {                                  // wave-uniform: one branch around the whole batch
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y) ov[x][rgi][y] = r_tanh<real>(ov[x][rgi][y]);
  } else if (act == 2) {
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y) ov[x][rgi][y] = act_value<real>(2, ov[x][rgi][y]);
  }
  if (has_res) {
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y) ov[x][rgi][y] = (rv[x][rgi][y] + ov[x][rgi][y]) * res_scale;
  }
  // phase 4: stores; the LDS / HBM choice is wave-uniform
  if (dst.lds >= 0) {
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y)
          if (r_ok[x][rgi] && c_ok[y]) smem[dst.lds + d_off[x][rgi] + colv[y]] = ov[x][rgi][y];
  } else {
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y)
          if (r_ok[x][rgi] && c_ok[y]) dst_g[d_off[x][rgi] + colv[y]] = ov[x][rgi][y];
  }
  if (stamp) a.prof[64 + 8 * slot + 5] = clock64();
}

// Units of one linear layer: (rows_per_unit row blocks) x (2 column blocks); rows_per_unit is
// chosen so that there are at least as many units as waves whenever the layer is big enough.
// `uoff` = units already dealt in this dependency level (keeps the waves evenly loaded).
template <typename real>
__device__ __forceinline__ int fused_linear(const FusedArgs<real>& a, const Meta<real>& mt, const dqmc_op& op, int wpk_off,
                                            int nw, int w0, int uoff, int slot) {
  const int32_t* i = op.i;
  const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const int ldw = (i[21] + 3) / 4 * 4;
  const int Rtot = nw * i[20];
  const int NRB = (Rtot + 15) / 16, NCB = (ldw + 15) / 16;
  const int n_cg = (NCB + 1) / 2;
  int rpu = NRB * n_cg / n_waves;
  rpu = rpu < 1 ? 1 : (rpu > 4 ? 4 : rpu);
  const int n_rg = (NRB + rpu - 1) / rpu;
  const int n_units = n_rg * n_cg;
  const real* wpk = a.wpk + wpk_off;
  int first = (wave - uoff) % n_waves;
  if (first < 0) first += n_waves;
  for (int u = first; u < n_units; u += n_waves) {
    const int rg = u / n_cg, cg = u - rg * n_cg;
    const int rb0 = rg * rpu;
    const int ma = (NRB - rb0) < rpu ? (NRB - rb0) : rpu;
    if (ma == 1) fused_linear_unit<real, 1>(a, mt, op, wpk, nw, w0, rb0, cg, slot);
    else if (ma == 2) fused_linear_unit<real, 2>(a, mt, op, wpk, nw, w0, rb0, cg, slot);
    else if (ma == 3) fused_linear_unit<real, 3>(a, mt, op, wpk, nw, w0, rb0, cg, slot);
    else fused_linear_unit<real, 4>(a, mt, op, wpk, nw, w0, rb0, cg, slot);
  }
  return n_units;
}

template <typename real>
__device__ __forceinline__ void fused_body(const FusedArgs<real>& a) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  const int tid = threadIdx.x, nthr = blockDim.x;
  // ---- program -> LDS ----
  const int o_bufs = fused_meta_off_bufs(a.n_ops), o_words = fused_meta_off_words(a.n_ops, a.n_bufs);
  {
    int32_t* dst = reinterpret_cast<int32_t*>(smem_raw);
    const int n_op_w = a.n_ops * (int)(sizeof(dqmc_op) / 4), n_buf_w = a.n_bufs * (int)(sizeof(FusedBuf) / 4);
    const int32_t* s_ops = reinterpret_cast<const int32_t*>(a.ops);
    const int32_t* s_buf = reinterpret_cast<const int32_t*>(a.fbufs);
    for (int e = tid; e < n_op_w; e += nthr) dst[e] = s_ops[e];
    for (int e = tid; e < n_buf_w; e += nthr) dst[o_bufs / 4 + e] = s_buf[e];
    for (int e = tid; e < 2 * a.n_ops; e += nthr) dst[o_words / 4 + e] = a.op_words[e];
  }
  __syncthreads();
  Meta<real> mt;
  mt.ops = reinterpret_cast<const dqmc_op*>(smem_raw);
  mt.bufs = reinterpret_cast<const FusedBuf*>(smem_raw + o_bufs);
  mt.words = reinterpret_cast<const int32_t*>(smem_raw + o_words);

  const int w0 = blockIdx.x * a.WT;
  const int nw = (a.B - w0) < a.WT ? (a.B - w0) : a.WT;
  const int N = a.li.N, n_up = a.n_up, n_nuc = a.n_nuc, K = a.K;
  const real* r = a.r + (long)w0 * N * 3;
  LaneInfo li = a.li;   // T = TP = 1
  const FastDiv fdN(N);
  int uoff = 0;
  if (a.prof != nullptr && blockIdx.x == 0 && tid == 0) a.prof[0] = clock64();

  for (int k = 0; k < a.n_ops; ++k) {
    const dqmc_op& op = mt.ops[k];
    const int32_t* i = op.i;
    switch (op.kind) {
      case DQMC_OP_FEAT_EN: {
        const BufRef x = mt.buf(i[0]);
        const FastDiv fdn(n_nuc);
        for (int e = tid; e < nw * N * n_nuc; e += nthr) {
          int q, n, wl, el;
          fdn.divmod(e, q, n);
          fdN.divmod(q, wl, el);
          double d[3], f[4];
          for (int c = 0; c < 3; ++c) d[c] = (double)r[(wl * N + el) * 3 + c] - (double)a.R[n * 3 + c];
          pair_feature_lane(d, a.eps, el, -1, 0, li, i[1] != 0, f);
          const int row = x.lds + (wl * N + el) * x.stride;
          for (int c = 0; c < 4; ++c) smem[row + 4 * n + c] = (real)f[c];
          if (n == 0) {
            int c = 4 * n_nuc;
            if (i[2]) smem[row + c++] = (real)(el < n_up ? 1.0 : -1.0);
            for (; c < x.width; ++c) smem[row + c] = (real)0;
          }
        }
        break;
      }
      case DQMC_OP_FEAT_EE: {
        const BufRef eb = mt.buf(i[0]);
        const int32_t* pairs = a.itable + i[1];
        const int n_rows = i[2];
        const FastDiv fdr(n_rows);
        for (int e = tid; e < nw * n_rows; e += nthr) {
          int wl, kr;
          fdr.divmod(e, wl, kr);
          const int rc = pairs[2 * kr], sd = pairs[2 * kr + 1];
          double d[3], f[4];
          for (int c = 0; c < 3; ++c) d[c] = (double)r[(wl * N + rc) * 3 + c] - (double)r[(wl * N + sd) * 3 + c];
          pair_feature_lane(d, a.eps, rc, sd, 0, li, i[3] != 0, f);
          const int row = eb.lds + (wl * n_rows + kr) * eb.stride;
          for (int c = 0; c < 4; ++c) smem[row + c] = (real)f[c];
        }
        break;
      }
      case DQMC_OP_LINEAR:
        uoff += fused_linear<real>(a, mt, op, mt.words[2 * k], nw, w0, uoff, k);
        break;
      case DQMC_OP_SPIN_MEAN: {
        const BufRef x = mt.buf(i[0]), m = mt.buf(i[1]);
        const FastDiv fdw(x.width);
        for (int e = tid; e < nw * 2 * x.width; e += nthr) {
          int q, c;
          fdw.divmod(e, q, c);
          const int which = q & 1, wl = q >> 1;
          const int i0 = which ? n_up : 0, i1 = which ? N : n_up;
          real acc = 0;
          for (int el = i0; el < i1; ++el) acc += smem[x.lds + (wl * N + el) * x.stride + c];
          smem[m.lds + (wl * 2 + which) * m.stride + c] = (i1 > i0) ? acc / (real)(i1 - i0) : (real)0;
        }
        break;
      }
      case DQMC_OP_CONV:
      case DQMC_OP_EDGE_SUM: {
        const bool conv = op.kind == DQMC_OP_CONV;
        const BufRef we = mt.buf(i[0]), out = mt.buf(i[2]);
        BufRef hx = we;
        if (conv) hx = mt.buf(i[1]);
        const real scale = conv ? (real)1 : (real)(1.0 / (double)(i[1] > 0 ? i[1] : 1));
        const int32_t* tab = a.itable + i[4];
        const int S = i[5], W = i[6], col0 = i[3];
        const FastDiv fdw(W);
        for (int e = tid; e < nw * N * W; e += nthr) {
          int q, c, wl, el;
          fdw.divmod(e, q, c);
          fdN.divmod(q, wl, el);
          real acc = 0;
          for (int s = 0; s < S; ++s) {
            const int row = tab[2 * (el * S + s)], snd = tab[2 * (el * S + s) + 1];
            if (row < 0) continue;
            const real ev = smem[we.lds + (wl * we.rows + row) * we.stride + c];
            acc += conv ? ev * smem[hx.lds + (wl * N + snd) * hx.stride + c] : ev;
          }
          smem[out.lds + (wl * N + el) * out.stride + col0 + c] = acc * scale;
        }
        break;
      }
      case DQMC_OP_ROW_SUM: {
        const BufRef x = mt.buf(i[0]), s = mt.buf(i[1]);
        const FastDiv fdw(x.width);
        for (int e = tid; e < nw * x.width; e += nthr) {
          int wl, c;
          fdw.divmod(e, wl, c);
          real acc = 0;
          for (int el = 0; el < x.rows; ++el) acc += smem[x.lds + (wl * x.rows + el) * x.stride + c];
          smem[s.lds + wl * s.stride + c] = acc;
        }
        break;
      }
      case DQMC_OP_ORBITALS: {
        const BufRef bf = mt.buf(i[0]), orb = mt.buf(i[1]);
        real* orb_g = reinterpret_cast<real*>(a.ws + orb.goff) + (long)w0 * orb.rows * orb.width;
        const int KN = K * N;
        const FastDiv fdk(KN);
        for (int e = tid; e < nw * N * KN; e += nthr) {
          int q, kmu, wl, el, kd, mu;
          fdk.divmod(e, q, kmu);
          fdN.divmod(q, wl, el);
          fdN.divmod(kmu, kd, mu);
          const int n_env = i[6] > 0 ? i[6] : 1;          // envelopes per nucleus (kernels_head.hip: k_orbitals)
          const real* pi = a.w + (el < n_up ? i[2] : i[3]) + kmu * n_nuc * n_env;
          const real* ze = a.w + (el < n_up ? i[4] : i[5]) + kmu * n_nuc * n_env;
          real res;
          if (sizeof(real) == 4) {        // float32 build: exponentials on the f32 unit (as k_orbitals, T = 1)
            float acc = 0.f;
            for (int n = 0; n < n_nuc; ++n) {
              float d2 = (float)a.eps;
              for (int c = 0; c < 3; ++c) { const float d = (float)r[(wl * N + el) * 3 + c] - (float)a.R[n * 3 + c]; d2 += d * d; }
              const float rho = sqrtf(d2);
              for (int ev = 0; ev < n_env; ++ev) acc += (float)pi[n * n_env + ev] * expf(-fabsf((float)ze[n * n_env + ev]) * rho);
            }
            res = (real)(acc * (float)smem[bf.lds + (wl * N + el) * bf.stride + kmu]);
          } else {
            double e0 = 0;
            for (int n = 0; n < n_nuc; ++n) {
              double d2 = a.eps;
              for (int c = 0; c < 3; ++c) { const double d = (double)r[(wl * N + el) * 3 + c] - (double)a.R[n * 3 + c]; d2 += d * d; }
              const double rho = sqrt(d2);
              for (int ev = 0; ev < n_env; ++ev) e0 += (double)pi[n * n_env + ev] * exp(-fabs((double)ze[n * n_env + ev]) * rho);
            }
            res = (real)(e0 * (double)smem[bf.lds + (wl * N + el) * bf.stride + kmu]);
          }
          orb_g[(wl * K + kd) * orb.width + el * N + mu] = res;
        }
        break;
      }
      default:
        break;
    }
    if (mt.words[2 * k + 1]) {   // end of a dependency level
      __syncthreads();
      uoff = 0;
    }
    if (a.prof != nullptr && blockIdx.x == 0 && tid == 0) a.prof[k + 1] = clock64();
  }
}

// OCC = workgroups (of 4 waves) the register allocation must leave room for per CU: the kernel is
// latency bound, so co-resident workgroups are what hides the per-op latency chains.
template <typename real, int OCC>
__global__ void __launch_bounds__(256, OCC) k_fused_value(const FusedArgs<real> a) { fused_body<real>(a); }

template <typename real> void launch_fused_value(hipStream_t st, const FusedArgs<real>& a, int n_blocks, size_t lds_bytes, int occ) {
  const dim3 g((unsigned)n_blocks), b(256);
  if (occ >= 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fused_value<real, 4>), g, b, lds_bytes, st, a);
  else if (occ == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fused_value<real, 3>), g, b, lds_bytes, st, a);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fused_value<real, 2>), g, b, lds_bytes, st, a);
}
template <typename real> int fused_set_lds_limit(size_t lds_bytes) {
  int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_value<real, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_value<real, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_value<real, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  return rc;
}

template void launch_fused_value<float>(hipStream_t, const FusedArgs<float>&, int, size_t, int);
template void launch_fused_value<double>(hipStream_t, const FusedArgs<double>&, int, size_t, int);
template int fused_set_lds_limit<float>(size_t);
template int fused_set_lds_limit<double>(size_t);

}  // namespace dqmc

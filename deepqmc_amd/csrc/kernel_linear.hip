// kernel_linear.hip -- the forward-Laplacian linear layer, the dominant kernel of the path.
//
//   Y = act( concat_p(X_p) W + b )  (+ residual),   rows = (walker, row, lane)
//
// Every lane of an activation (value, 3N first derivatives, Laplacian) goes through the
// same matrix, so a layer is ONE GEMM over B*rows*TP stacked rows (reference: what folx's
// forward Laplacian does for hk.Linear, conf/hamil/qc_forward_laplacian.yaml:7-10, restated
// in SURVEY.md appendix C).  The nonlinearity needs all lanes of one (walker,row,column):
//     y = phi(v),  J'_c = phi'(v) J_c,  L' = phi'(v) L + phi''(v) sum_c J_c^2
// Rows are ordered lane-fastest and TP is a multiple of 16, so one (walker,row) group is
// TP/16 consecutive MFMA row blocks held by ONE wave: the chain rule runs on the
// accumulators in registers (sum_c J_c^2 = per-lane partial + two cross-quad shuffles), fused
// with bias, residual and the store.
//
// gfx950 mapping: exact-f32 v_mfma_f32_16x16x4_f32 (or v_mfma_f64_16x16x4_f64 in the float64
// parity build); block = 4 waves stacked in M, wave tile (16*MR) x (16*NR), K staged through
// LDS in chunks of 16 (A transposed on the way in so both fragments are conflict-free
// ds_read_b32), next chunk prefetched into registers while the current one is multiplied.
// The A operand is a virtual concatenation of up to 4 pieces (e.g. [x | mean_up | mean_down |
// conv] for the g layer, reference gnn/electron_gnn.py:239-243), per-walker-mean pieces being
// broadcast by the row mapping instead of materialised.
#include "common.h"
#include "kernels.h"

namespace dqmc {

template <typename real> __device__ __forceinline__ void act_derivs(int act, real v, real& y, real& d1, real& d2) {
  if (act == 1) {
    y = r_tanh<real>(v);
    d1 = 1 - y * y;
    d2 = -2 * y * d1;
  } else if (act == 2) {
    const real s = 1 / (1 + r_exp<real>(-v));
    y = v * s;
    d1 = s * (1 + v * (1 - s));
    d2 = s * (1 - s) * (2 + v * (1 - 2 * s));
  } else if (act == 3) {     // shifted softplus log(1 + e^v) + log(1/2) (reference hkext.py:13-19)
    const real av = v < 0 ? -v : v;
    const real e = r_exp<real>(-av);
    const real s = v >= 0 ? 1 / (1 + e) : e / (1 + e);
    y = (v > 0 ? v : (real)0) + (sizeof(real) == 4 ? (real)log1pf((float)e) : (real)log1p((double)e)) - (real)0.69314718055994530942;
    d1 = s;
    d2 = s * (1 - s);
  } else if (act == 4) {     // multiplicative backflow activation 1 + 2 tanh(v/4) (wf/nn_wave_function.py:17)
    const real t = r_tanh<real>(v * (real)0.25);
    y = 1 + 2 * t;
    d1 = (1 - t * t) * (real)0.5;
    d2 = -t * (1 - t * t) * (real)0.25;
  } else {
    y = v; d1 = 1; d2 = 0;
  }
}

// Row tile / column tile of the calling workgroup.  A layer wider than the column tile launches gy > 1 column tiles per row
// tile, and each of them streams the same A rows; in plain (x, y) order those workgroups are gx dispatches apart and land on
// different XCDs (block b runs on XCD b % 8, each XCD has its own L2), so A is fetched from HBM gy times -- N2 / FermiNet,
// 256-wide layers with 64-column tiles: 11 GB per launch, the launch is HBM-bound at 4 TB/s.  The remap puts the gy column
// tiles of a row tile on ONE XCD, eight dispatches apart, so they run together and share the rows through that XCD's L2.
// Bijective: the first (gx / 8) * 8 row tiles are dealt out as described, the remainder keeps a plain order.
struct TileId { int bx, by; };
__device__ __forceinline__ TileId tile_of_block() {
  const int gx = (int)gridDim.x, gy = (int)gridDim.y;
  TileId t{(int)blockIdx.x, (int)blockIdx.y};
  if (gy == 1) return t;
  const int b = (int)blockIdx.y * gx + (int)blockIdx.x;      // dispatch order
  const int full = gx >> 3, region = full * 8 * gy;
  if (b < region) {
    const int xcd = b & 7, s_ = b >> 3;
    t.by = s_ % gy;
    t.bx = (s_ / gy) * 8 + xcd;
  } else {
    const int r = b - region;
    t.by = r % gy;
    t.bx = full * 8 + r / gy;
  }
  return t;
}

#ifdef DQMC_LIN_PROBE
#define LIN_PROBE(a, bit) (((a).cfg_probe >> (bit)) & 1)
template <typename real> static LinArgs<real> with_probe(const LinArgs<real>& a0) {      // (host: the probe bits of this process)
  LinArgs<real> a = a0;
  static const int pr = getenv("DQMC_LIN_PROBE") ? atoi(getenv("DQMC_LIN_PROBE")) : 0;
  a.cfg_probe = pr;
  return a;
}
#else
#define LIN_PROBE(a, bit) 0
#endif

// A wave-uniform element offset into scalar registers: base (a kernel argument) + offset is then the SGPR base of the global
// loads / stores, which take a 32-bit lane offset on top; values the compiler merely knows to be uniform stay in vector
// registers and every access pays a 64-bit vector add.
__device__ __forceinline__ long uniform_off(long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long)(((unsigned long long)hi << 32) | lo);
}

template <int BN> struct BStride { static constexpr int v = ((BN + 16) % 32 == 16) ? BN + 16 : BN + 32; };

// Where the epilogue puts its results.  HbmSink: the layer's output buffer (+ residual).  LdsSink: the hidden tile of a
// chained two-layer MLP (k_linear<..., CHAIN = true>): row-major [BM][HS], zero where the row / column does not exist,
// so that the second layer can multiply whole chunks.
template <typename real> struct HbmSink {
  // (the fields it needs, by value: a reference to the kernel's argument struct kept the whole struct in scratch)
  real* dst;
  const real* res;
  real res_scale;
  int ld_dst, ld_res, col0_dst, rpw_dst, r0_dst, rpw_res, r0_res, TP, nrows;
  long drow0[2], rrow0[2];
  __device__ __forceinline__ HbmSink(const LinArgs<real>& a)
      : dst(a.dst), res(a.res), res_scale(a.res_scale), ld_dst(a.ld_dst), ld_res(a.ld_res), col0_dst(a.col0_dst), rpw_dst(a.rpw_dst),
        r0_dst(a.r0_dst), rpw_res(a.rpw_res), r0_res(a.r0_res), TP(a.TP), nrows(a.nrows) {}
  // first destination / residual row of group g = (walker, row)
  __device__ __forceinline__ void rows(int g, long& d, long& r) const {
    const int b = g / nrows, rr = g - b * nrows;
    d = ((long)b * rpw_dst + r0_dst + rr) * TP;
    r = res ? ((long)b * rpw_res + r0_res + rr) * TP : 0;
  }
  __device__ __forceinline__ void group(int slot, int g) {
    long d, r;
    rows(g, d, r);
    if (slot) { drow0[1] = d; rrow0[1] = r; } else { drow0[0] = d; rrow0[0] = r; }      // (selects: a per-lane index would put the arrays in scratch)
  }
  // The residual of one element (0 without one).  Callers fetch a BATCH of elements before they store the batch: the
  // destination may alias the residual buffer, so the compiler never moves a residual load above an earlier store itself
  // -- element by element (load, wait for it, add, store) every lane paid one memory latency per element (value rows of
  // a 256 x 128 tile: 64 in a row, 2.2 ms for a launch whose products take 0.6).
  __device__ __forceinline__ real fetch_at(long rrow, int col, bool ok) const {
    if (res == nullptr || !ok) return (real)0;
    return res[rrow * ld_res + col0_dst + col];
  }
  __device__ __forceinline__ void put_at(long drow, int /*tile_row*/, int col, real o, real r, bool ok) {
    if (!ok) return;
    if (res != nullptr) o = (r + o) * res_scale;
    dst[drow * ld_dst + col0_dst + col] = o;
  }
  __device__ __forceinline__ real fetch(int slot, int t, int col, bool ok) const { return fetch_at((slot ? rrow0[1] : rrow0[0]) + t, col, ok); }
  __device__ __forceinline__ void put(int slot, int t, int tile_row, int col, real o, real r, bool ok) { put_at((slot ? drow0[1] : drow0[0]) + t, tile_row, col, o, r, ok); }
  // Uniform-base access (the 16-lane-block groups): `g` is WAVE-UNIFORM, so the 64-bit row arithmetic of a group runs once on
  // the scalar unit and a lane adds a 32-bit element offset (row t of the group, column c of the wave's tile) -- the per-lane
  // form above spent ~30 vector instructions per element on 64-bit multiplies that every lane of the wave computed alike.
  static constexpr bool fills = false;     // (elements that do not exist are skipped, not zero-filled)
  real* udst; const real* ures;
  __device__ __forceinline__ void ugroup(int g, int col_w0) {
    const int b = g / nrows, rr = g - b * nrows;
    udst = dst + uniform_off(((long)b * rpw_dst + r0_dst + rr) * TP * ld_dst + col0_dst + col_w0);
    ures = res ? res + uniform_off(((long)b * rpw_res + r0_res + rr) * TP * ld_res + col0_dst + col_w0) : nullptr;
  }
  // (32-bit BYTE offsets: an element index the compiler would have to scale by sizeof(real) in 64 bits, per access)
  __device__ __forceinline__ real ufetch(int t, int c) const {
    return ures ? *reinterpret_cast<const real*>(reinterpret_cast<const char*>(ures) + (unsigned)((t * ld_res + c) * (int)sizeof(real))) : (real)0;
  }
  __device__ __forceinline__ void uput(int t, int /*tile_row*/, int c, real o, real r) {
    if (ures != nullptr) o = (r + o) * res_scale;
    *reinterpret_cast<real*>(reinterpret_cast<char*>(udst) + (unsigned)((t * ld_dst + c) * (int)sizeof(real))) = o;
  }
  // a PAIR of consecutive groups g, g + 1 (the two 8-lane groups of a row block): the base is group g's, a lane of group
  // g + 1 adds the (uniform, small, non-negative) distance in bytes
  unsigned udd, urd;
  __device__ __forceinline__ void ugroup2(int g, bool second, int col_w0) {
    ugroup(g, col_w0);
    udd = urd = 0;
    if (second) {
      const int b0 = g / nrows, rr0 = g - b0 * nrows, b1 = (g + 1) / nrows, rr1 = g + 1 - b1 * nrows;
      udd = (unsigned)__builtin_amdgcn_readfirstlane((int)((((long)(b1 - b0) * rpw_dst + (rr1 - rr0)) * TP * ld_dst) * (long)sizeof(real)));
      urd = (unsigned)__builtin_amdgcn_readfirstlane((int)((((long)(b1 - b0) * rpw_res + (rr1 - rr0)) * TP * ld_res) * (long)sizeof(real)));
    }
  }
  __device__ __forceinline__ real ufetch2(bool h, int t, int c) const {
    return ures ? *reinterpret_cast<const real*>(reinterpret_cast<const char*>(ures) + ((h ? urd : 0u) + (unsigned)((t * ld_res + c) * (int)sizeof(real)))) : (real)0;
  }
  __device__ __forceinline__ void uput2(bool h, int t, int /*tile_row*/, int c, real o, real r) {
    if (ures != nullptr) o = (r + o) * res_scale;
    *reinterpret_cast<real*>(reinterpret_cast<char*>(udst) + ((h ? udd : 0u) + (unsigned)((t * ld_dst + c) * (int)sizeof(real)))) = o;
  }
};
template <typename real> struct LdsSink {
  real* hs;
  int stride;
  __device__ __forceinline__ void rows(int, long& d, long& r) const { d = 0; r = 0; }
  __device__ __forceinline__ void group(int, int) {}
  __device__ __forceinline__ real fetch_at(long, int, bool) const { return (real)0; }
  __device__ __forceinline__ void put_at(long, int tile_row, int col, real o, real, bool ok) { hs[tile_row * stride + col] = ok ? o : (real)0; }
  __device__ __forceinline__ real fetch(int, int, int, bool) const { return (real)0; }
  __device__ __forceinline__ void put(int, int, int tile_row, int col, real o, real, bool ok) { put_at(0, tile_row, col, o, (real)0, ok); }
  static constexpr bool fills = true;      // rows / columns that do not exist are written as zeros (the second product reads whole chunks)
  int ucol;
  __device__ __forceinline__ void ugroup(int, int col_w0) { ucol = col_w0; }
  __device__ __forceinline__ real ufetch(int, int) const { return (real)0; }
  __device__ __forceinline__ void uput(int, int tile_row, int c, real o, real) { hs[tile_row * stride + ucol + c] = o; }
  __device__ __forceinline__ void ugroup2(int, bool, int col_w0) { ucol = col_w0; }
  __device__ __forceinline__ real ufetch2(bool, int, int) const { return (real)0; }
  __device__ __forceinline__ void uput2(bool, int, int tile_row, int c, real o, real) { hs[tile_row * stride + ucol + c] = o; }
};

// the nonlinearity alone, for a compile-time activation code (value rows: no derivatives, no per-element dispatch)
template <typename real, int ACT> __device__ __forceinline__ real act_value(real v) {
  real y, d1, d2;
  act_derivs<real>(ACT, v, y, d1, d2);
  return y;
}

// Value-only rows (GPW = 0): every accumulator row is a (walker, row) of its own.  Per row block: the four row addresses of
// a lane once, then ALL residual / per-walker pre-activation loads of the block, then the arithmetic and the stores.
template <typename real, int MR, int NR, int ACT, typename Sink>
__device__ __forceinline__ void value_rows_epilogue(typename Mfma<real>::acc_t (&acc)[MR][NR], int nrows, int ld_pre, const real* bias,
                                                    int ldw, const real* pre, int col_w0, int wm, int n_groups, Sink& sink, int bx) {
  constexpr int BM = 64 * MR;
  const int lane = threadIdx.x & 63, cl = lane & 15;
  real bv[NR];
  bool col_ok[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    const int col = col_w0 + n * 16 + cl;
    col_ok[n] = col < ldw;
    bv[n] = (bias != nullptr && col_ok[n]) ? bias[col] : (real)0;
  }
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int r0 = 0; r0 < 4; r0 += 2) {       // two of a lane's four rows at a time: 2 NR loads of each kind in flight, few live registers
      bool m_ok[2];
      int b[2];
      long drow[2], rrow[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int m = bx * BM + wm * (16 * MR) + i * 16 + Mfma<real>::row_of(lane, r0 + q);
        m_ok[q] = m < n_groups;
        b[q] = (m_ok[q] ? m : 0) / nrows;
        sink.rows(m_ok[q] ? m : 0, drow[q], rrow[q]);
      }
      real rres[2][NR], rpre[2][NR];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          const int col = col_w0 + n * 16 + cl;
          rres[q][n] = sink.fetch_at(rrow[q], col, col_ok[n] && m_ok[q]);
          rpre[q][n] = (pre != nullptr && col_ok[n]) ? pre[(long)b[q] * ld_pre + col] : (real)0;
        }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int tile_row = wm * (16 * MR) + i * 16 + Mfma<real>::row_of(lane, r0 + q);
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          const int col = col_w0 + n * 16 + cl;
          real v = acc[i][n][r0 + q];
          if (pre != nullptr && col_ok[n]) v += rpre[q][n];
          if (bias != nullptr && col_ok[n]) v += bv[n];
          sink.put_at(drow[q], tile_row, col, act_value<real, ACT>(v), rres[q][n], col_ok[n] && m_ok[q]);
        }
      }
    }
}

// bias + nonlinearity with the forward-Laplacian chain rule on the accumulators of one wave (header comment), results to
// `sink`.  MR x NR accumulator tiles; wave `wm` of the workgroup's M stack; col_w0 = first column of the wave's tile.
template <typename real, int MR, int NR, int GPW, typename Sink>
__device__ __forceinline__ void lin_epilogue(typename Mfma<real>::acc_t (&acc)[MR][NR], const LinArgs<real>& a, const real* bias, int act,
                                             int ldw, const real* pre, int col_w0, int wm, int n_groups, Sink& sink, int bx,
                                             real* xch = nullptr) {
  constexpr bool HALF = GPW == -1;
  constexpr bool SPLIT = GPW == -2;
  constexpr int GB = GPW > 0 ? MR / GPW : 1;
  const int lane = threadIdx.x & 63, cl = lane & 15;
  if (HALF) {
    // row block i holds groups 2*i (rows 0..7) and 2*i + 1 (rows 8..15), both wave-uniform (see the GPW > 0 branch below:
    // scalar bases, 32-bit lane offsets).  float32 layout: a lane's four rows 4 q + 0..3 lie in ONE group (h = q >> 1): its
    // value lane comes from lane 32 h + column, register 0, sum_c J_c^2 from the lane 16 apart; float64 layout (rows
    // q + 4 reg): registers 0, 1 belong to the first group, 2, 3 to the second, every lane evaluates both.
    constexpr bool F32L = sizeof(real) == 4;
    const int wmu = __builtin_amdgcn_readfirstlane(wm);
    const int Tm1 = a.T - 1;
    const int q = lane >> 4;
    real bvh[NR];
    bool col_ok[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      const int col = col_w0 + n * 16 + cl;
      col_ok[n] = col < ldw;
      bvh[n] = (bias != nullptr && col_ok[n]) ? bias[col] : (real)0;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int g0 = ((bx * 4 + wmu) * MR + i) * 2;
      const bool ok0 = g0 < n_groups, ok1 = g0 + 1 < n_groups;      // wave-uniform
      if (!ok0) {
        if constexpr (Sink::fills) {
          sink.ugroup2(0, false, col_w0);         // (LdsSink: the column base of the zero fill)
#pragma unroll
          for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) sink.uput2(false, 0, wm * (16 * MR) + i * 16 + Mfma<real>::row_of(lane, rg), n * 16 + cl, (real)0, (real)0);
        }
        continue;
      }
      sink.ugroup2(g0, ok1, col_w0);
      real rres[NR][4];                           // (the residuals of the whole row block before its first store)
#pragma unroll
      for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = Mfma<real>::row_of(lane, rg);
          const bool h = row >> 3;
          rres[n][rg] = (col_ok[n] && (!h || ok1)) ? sink.ufetch2(h, row & 7, n * 16 + cl) : (real)0;
        }
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        real y[2], d1[2], d2S[2];
        if constexpr (F32L) {
          real v = __shfl(acc[i][n][0], (lane & 32) + cl, 64) + bvh[n];
          real s_part = 0;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int tt = (4 * q + rg) & 7;
            const real x = acc[i][n][rg];
            if (tt > 0 && tt < Tm1) s_part += x * x;
          }
          const real S = s_part + __shfl_xor(s_part, 16, 64);
          real d2;
          act_derivs<real>(act, v, y[0], d1[0], d2);
          d2S[0] = d2 * S;
          y[1] = y[0]; d1[1] = d1[0]; d2S[1] = d2S[0];
        } else {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const real v = __shfl(acc[i][n][2 * h], cl, 64) + bvh[n];
            real s_part = 0;
#pragma unroll
            for (int rg = 2 * h; rg < 2 * h + 2; ++rg) {
              const int tt = Mfma<real>::row_of(lane, rg) & 7;
              const real x = acc[i][n][rg];
              if (tt > 0 && tt < Tm1) s_part += x * x;
            }
            const real S = quad_sum<real>(s_part);
            real d2;
            act_derivs<real>(act, v, y[h], d1[h], d2);
            d2S[h] = d2 * S;
          }
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = Mfma<real>::row_of(lane, rg), tt = row & 7;
          const bool h = row >> 3;
          const int hs_ = F32L ? 0 : (rg >> 1);
          const real x = acc[i][n][rg];
          // (ONE fused multiply-add on every lane: d1 x rounded on its own and d2 S added afterwards costs the Laplacian lane a rounding per layer)
          real o = r_fma<real>(d1[hs_], x, tt == Tm1 ? d2S[hs_] : (real)0);
          if (tt == 0) o = y[hs_];
          if (tt > Tm1) o = 0;
          const bool ok = col_ok[n] && (!h || ok1);
          if constexpr (Sink::fills) sink.uput2(h, tt, wm * (16 * MR) + i * 16 + row, n * 16 + cl, ok ? o : (real)0, rres[n][rg]);
          else if (ok) sink.uput2(h, tt, wm * (16 * MR) + i * 16 + row, n * 16 + cl, o, rres[n][rg]);
        }
      }
    }
  } else if (SPLIT) {
    // A group of TP = 32 MR lanes spans the TWO waves of a pair (wm & 1 = which half of the lanes): the accumulators of a
    // 96- / 128-lane group in float64 are 96 / 128 registers per column-block pair -- one wave per SIMD when one wave holds
    // them all.  The chain rule needs the value lane (first half, lane 0) and sum_c J_c^2 over ALL derivative lanes: both
    // cross the pair through `xch` (the A tile of the finished K loop).
    const int half = wm & 1, gl = wm >> 1;
    const int g = bx * 2 + gl;
    const bool g_ok = g < n_groups;        // wave-uniform
    const int b = (g_ok ? g : 0) / a.nrows;
    const int t0 = half * (16 * MR);
    sink.group(0, g_ok ? g : 0);
    if (pre != nullptr && g_ok) {
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        const int col = col_w0 + n * 16 + cl;
        if (col < ldw) {
#pragma unroll
          for (int tb = 0; tb < MR; ++tb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const int t = t0 + tb * 16 + Mfma<real>::row_of(lane, rg);
              acc[tb][n][rg] += pre[((long)b * a.TP + t) * a.ld_pre + col];
            }
        }
      }
    }
    real* xv = xch + gl * (3 * NR * 16);
    real* xs = xv + NR * 16;
    __syncthreads();                       // every fragment read of the K loop is done: the A tile becomes the exchange area
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      real s_part = 0;
#pragma unroll
      for (int tb = 0; tb < MR; ++tb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int t = t0 + tb * 16 + Mfma<real>::row_of(lane, rg);
          const real x = acc[tb][n][rg];
          if (t >= 1 && t < a.T - 1) s_part += x * x;
        }
      const real Sh = quad_sum<real>(s_part);
      if (lane < 16) {                     // (lanes 0..15 hold accumulator row 0 in register 0: the value lane of the first half)
        xs[half * (NR * 16) + n * 16 + cl] = Sh;
        if (half == 0) xv[n * 16 + cl] = acc[0][n][0];
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      const int col = col_w0 + n * 16 + cl;
      const bool col_ok = col < ldw;
      real v = xv[n * 16 + cl];
      if (bias != nullptr && col_ok) v += bias[col];
      const real S = xs[n * 16 + cl] + xs[NR * 16 + n * 16 + cl];
      real rres[MR][4];
#pragma unroll
      for (int tb = 0; tb < MR; ++tb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) rres[tb][rg] = sink.fetch(0, t0 + tb * 16 + Mfma<real>::row_of(lane, rg), col, col_ok && g_ok);
      real y, d1, d2;
      act_derivs<real>(act, v, y, d1, d2);
#pragma unroll
      for (int tb = 0; tb < MR; ++tb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = Mfma<real>::row_of(lane, rg);
          const int t = t0 + tb * 16 + row;
          const real x = acc[tb][n][rg];
          real o;
          if (t == 0) o = y;
          else if (t < a.T - 1) o = d1 * x;
          else if (t == a.T - 1) o = d1 * x + d2 * S;
          else o = 0;
          sink.put(0, t, wm * (16 * MR) + tb * 16 + row, col, o, rres[tb][rg], col_ok && g_ok);
        }
    }
  } else if (GPW > 0) {
    constexpr int NG = GPW > 0 ? GPW : 1;
    // A group (walker, row) is GB consecutive row blocks of ONE wave, so everything that names the group -- its index, its
    // walker, the 64-bit bases of its destination / residual / per-walker rows -- is wave-uniform: computed on the scalar unit
    // (the wave index goes through readfirstlane: as a function of threadIdx.x the compiler had to assume it varies per lane
    // and ran all of it, 64-bit multiplies included, on the vector unit for every element: 7 vector instructions per MFMA for
    // the K = 192 node layer of LiH, 30-60 for the narrow edge layers).  A lane adds 32-bit offsets t * ld + column.
    // Everything the stores of a group depend on is requested BEFORE its first store (a load cannot pass an earlier store
    // to a possibly aliasing buffer): the per-walker rows, and the residuals of the whole group where they fit (HOIST: at
    // most 32 registers), else per column block in batches of FB row blocks.
    constexpr bool HOIST = GB * NR * 4 * (int)(sizeof(real) / 4) <= 32;
    const int wmu = __builtin_amdgcn_readfirstlane(wm);
    const int Tm1 = a.T - 1;
    real bv[NR];
    bool col_ok[NR];
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      const int col = col_w0 + n * 16 + cl;
      col_ok[n] = col < ldw;
      bv[n] = (bias != nullptr && col_ok[n]) ? bias[col] : (real)0;
    }
#pragma unroll
    for (int gj = 0; gj < NG; ++gj) {
      const int g = (bx * 4 + wmu) * GPW + gj;
      if (g >= n_groups) {                       // wave-uniform
        if constexpr (Sink::fills) {
          sink.ugroup(0, col_w0);                // (LdsSink: the column base of the zero fill)
#pragma unroll
          for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int tb = 0; tb < GB; ++tb)
#pragma unroll
              for (int rg = 0; rg < 4; ++rg)
                sink.uput(0, wm * (16 * MR) + (gj * GB + tb) * 16 + Mfma<real>::row_of(lane, rg), n * 16 + cl, (real)0, (real)0);
        }
        continue;
      }
      sink.ugroup(g, col_w0);
      if (pre != nullptr) {                      // per-walker part of the pre-activation (all lanes: the layer is linear in them)
        const real* pw = pre + uniform_off((long)(g / a.nrows) * a.TP * a.ld_pre + col_w0);
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          if (col_ok[n]) {
#pragma unroll
            for (int tb = 0; tb < GB; ++tb)
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) {
                const int t = tb * 16 + Mfma<real>::row_of(lane, rg);
                acc[gj * GB + tb][n][rg] += *reinterpret_cast<const real*>(reinterpret_cast<const char*>(pw) + (unsigned)((t * a.ld_pre + n * 16 + cl) * (int)sizeof(real)));
              }
          }
        }
      }
      real rall[HOIST ? NR : 1][HOIST ? GB : 1][4];
      if constexpr (HOIST) {
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
          for (int tb = 0; tb < GB; ++tb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
              rall[n][tb][rg] = col_ok[n] ? sink.ufetch(tb * 16 + Mfma<real>::row_of(lane, rg), n * 16 + cl) : (real)0;
      }
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        real v = __shfl(acc[gj * GB][n][0], cl, 64) + bv[n];
        real s_part = 0;
#pragma unroll
        for (int tb = 0; tb < GB; ++tb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int t = tb * 16 + Mfma<real>::row_of(lane, rg);
            const real x = acc[gj * GB + tb][n][rg];
            if (t >= 1 && t < Tm1) s_part += x * x;
          }
        const real S = quad_sum<real>(s_part);
        real y, d1, d2;
        act_derivs<real>(act, v, y, d1, d2);
        const real d2S = d2 * S;
        // (not hoisted: float64 two row blocks at a time -- the tall tiles have no registers to spare; GB = 4 in float32 too:
        // 16 more registers would cost the 64-lane tiles a workgroup per CU)
        constexpr int FB = HOIST ? GB : (((sizeof(real) == 8 && GB > 2) || GB == 4) ? 2 : GB);
#pragma unroll
        for (int tb0 = 0; tb0 < GB; tb0 += FB) {
          real rres[FB][4];
#pragma unroll
          for (int tb = 0; tb < FB; ++tb)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              if constexpr (HOIST) rres[tb][rg] = rall[n][tb][rg];
              else rres[tb][rg] = (tb0 + tb < GB && col_ok[n]) ? sink.ufetch((tb0 + tb) * 16 + Mfma<real>::row_of(lane, rg), n * 16 + cl) : (real)0;
            }
          if (Sink::fills || col_ok[n]) {
#pragma unroll
            for (int tb = tb0; tb < tb0 + FB && tb < GB; ++tb)
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) {
                const int row = Mfma<real>::row_of(lane, rg);
                const int t = tb * 16 + row;
                const real x = acc[gj * GB + tb][n][rg];
                // (ONE fused multiply-add on every lane, whatever the compiler would contract: d1 x rounded on its own with d2 S added
                // afterwards would cost the Laplacian lane a rounding more per layer)
                real o = r_fma<real>(d1, x, t == Tm1 ? d2S : (real)0);
                if (t == 0) o = y;
                if (t > Tm1) o = 0;
                if (Sink::fills && !col_ok[n]) o = 0;
                sink.uput(t, wm * (16 * MR) + (gj * GB + tb) * 16 + row, n * 16 + cl, o, rres[tb - tb0][rg]);
              }
          }
        }
      }
    }
  } else {
    switch (act) {          // (wave-uniform; one unrolled, branch-free body per activation)
      case 1: value_rows_epilogue<real, MR, NR, 1>(acc, a.nrows, a.ld_pre, bias, ldw, pre, col_w0, wm, n_groups, sink, bx); break;
      case 2: value_rows_epilogue<real, MR, NR, 2>(acc, a.nrows, a.ld_pre, bias, ldw, pre, col_w0, wm, n_groups, sink, bx); break;
      case 3: value_rows_epilogue<real, MR, NR, 3>(acc, a.nrows, a.ld_pre, bias, ldw, pre, col_w0, wm, n_groups, sink, bx); break;
      case 4: value_rows_epilogue<real, MR, NR, 4>(acc, a.nrows, a.ld_pre, bias, ldw, pre, col_w0, wm, n_groups, sink, bx); break;
      default: value_rows_epilogue<real, MR, NR, 0>(acc, a.nrows, a.ld_pre, bias, ldw, pre, col_w0, wm, n_groups, sink, bx); break;
    }
  }
}

// GPW = -1: groups of 8 lanes (the pair-compact edge buffers of common.h), two per MFMA row block.
// MR row blocks x NR column blocks per wave; GPW groups per wave (0: value-only rows); the
// workgroup is 4 waves in M times WN waves in N (WN = 2: 512 threads, BN = 128, so a 128-wide
// layer reads its A rows from HBM once).  A tile in LDS is row-major with stride BK + 2: the
// 16 rows x 2 k of a half-wave fragment read land on 32 distinct banks (18*row mod 32 is a
// permutation of the even banks), and the staging store is two 8-byte writes per thread.
//
// CHAIN (WN = 1, one column tile): a row-wise two-layer MLP  Y = act2(act(X W + b) W2 + b2) (+ residual)  -- the edge
// MLPs w / u and the node MLP h of a message-passing layer (reference gnn/electron_gnn.py:116-160, hkext.py:99-113) --
// in ONE launch: the hidden activations of the tile (chain rule applied, all lanes) go to LDS instead of HBM and are the
// A operand of the second product.  The hidden layer is at most 16 NR wide, the output at most 16 NR2.
//
// BKX: K chunk = 16 BKX.  The small-tile configurations (MR = 1: the narrow layers of the few-electron systems, which
// stream their rows once) are bound by the latency of the chunk loads, not by bandwidth or MFMA rate -- each thread has
// one 16-byte A load and one B load in flight per chunk; BKX = 2 doubles the bytes in flight per workgroup and halves
// the number of load -> barrier -> multiply round trips.
// (second launch bound = waves per SIMD the register allocation must leave room for; the 8-wave value-row tiles -- 256 x 128
// in float32, 128 x 128 in float64 -- are to run two workgroups per CU = four waves per SIMD; the split-group tiles exist to run
// two waves per SIMD -- unconstrained, the 4 x 4 float64 tile takes 200 + 128 registers and one wave remains)
template <typename real, int MR, int NR, int GPW, int WN, bool CHAIN = false, int NR2 = 2, int BKX = 1>
__global__ void __launch_bounds__(256 * WN, GPW == -2 ? 2 : (GPW == 0 && NR == 4 && WN == 2 && MR * sizeof(real) == 16) ? 4 : 1) k_linear(const LinArgs<real> a) {
  constexpr int NT = 256 * WN;
  constexpr int BM = 64 * MR, BN = 16 * NR * WN, BK = 16 * BKX, BK2 = 16;
  constexpr int AS = BK + 2, BS = BStride<BN>::v;
  constexpr bool HALF = GPW == -1;                      // 8-lane groups: two per row block
  constexpr bool SPLIT = GPW == -2;                     // 32 MR-lane groups held by a PAIR of waves (16 MR lanes each)
  constexpr int GB = GPW > 0 ? MR / GPW : 1;            // row blocks per group
  constexpr int APT = MR / WN;                          // A float4 per thread and chunk
  constexpr int NBV = (BK * BN / 4 + NT - 1) / NT;      // B float4 per thread and chunk
  constexpr int HS = CHAIN ? BN + 2 : 1;                // hidden tile stride: 2 * odd -> conflict-free fragment reads
  constexpr int BN2 = 16 * NR2, BS2 = BStride<BN2>::v;
#ifndef DQMC_FRESH_MR_MAX
#define DQMC_FRESH_MR_MAX 3
#endif
  // a fresh accumulator per k chunk (below, at the MFMA loop): float32 Laplacian tiles of up to 3 row blocks per wave (<= 48 lanes: up to 15
  // electrons; N2 / FermiNet 56.3 -> 55.0 ms per step with the 48-lane tiles included, 56.4 without).  Not the value-only rows (Metropolis sub-steps: log|psi| feeds an accept test), not the taller tiles: the second
  // accumulator set costs them a wave of occupancy or spills (<4,4,0,2> 116 -> 128 registers + 144 spilled, <3,4,1,1> 152 -> 180)
  constexpr bool FRESH = sizeof(real) == 4 && GPW != 0 && MR <= DQMC_FRESH_MR_MAX;
  constexpr bool CAN32 = sizeof(real) == 8 && !SPLIT && !CHAIN;      // LinArgs::src_f32 (float32 A pieces under a float64 product)
  static_assert(MR % WN == 0, "MR must be a multiple of WN");
  static_assert(!CHAIN || WN == 1, "chained layers use one column tile");
  static_assert(!SPLIT || (!CHAIN && WN == 1 && 6 * NR * 16 <= BM * AS), "split groups: plain layers, one column tile per workgroup");
  typedef typename Mfma<real>::acc_t acc_t;
  // One LDS block.  CHAIN: the hidden tile Hs ALIASES the staging tiles As / Bs -- it is written by the first layer's epilogue, when
  // the K loop has read them for the last time (one more barrier), so the float64 chained instances need 44 KB instead of 77
  // (three workgroups per CU, which is also what their 152 registers allow, instead of two).
  constexpr int N_AB = BM * AS + BK * BS, N_H = CHAIN ? BM * HS : 0, N_MAIN = N_AB > N_H ? N_AB : N_H;
  __shared__ __attribute__((aligned(32))) real smem_lin[N_MAIN + (CHAIN ? BK2 * BS2 : 0)];
  real* const As = smem_lin;
  real* const Bs = smem_lin + BM * AS;
  real* const Hs = smem_lin;
  real* const Bs2 = smem_lin + N_MAIN;
  static_assert((BM * AS) % 4 == 0 && N_MAIN % 4 == 0, "16-byte (float) / 32-byte (double) aligned LDS tiles");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int n_groups = a.B * a.nrows;            // (walker,row) groups == value-mode rows
  const TileId tile = tile_of_block();
  const int bx = tile.bx;
  const int col_blk0 = tile.by * BN;

  // ---- per-thread A rows ----
  int a_row[APT];         // row inside the tile
  int a_g[APT], a_t[APT]; // group index (b*nrows+rr) and lane t; a_g < 0: out of range
  const int a_kq = tid & 3;
#pragma unroll
  for (int j = 0; j < APT; ++j) {
    const int row = (tid >> 2) + (NT / 4) * j;
    a_row[j] = row;
    int g, t;
    if (HALF) {
      const int w = row / (16 * MR), rb = (row >> 4) % MR;
      g = ((bx * 4 + w) * MR + rb) * 2 + ((row & 15) >> 3);
      t = row & 7;
    } else if (SPLIT) {
      const int w = row / (16 * MR), rb = (row >> 4) % MR;
      g = bx * 2 + (w >> 1);
      t = ((w & 1) * MR + rb) * 16 + (row & 15);
    } else if (GPW > 0) {
      const int w = row / (16 * MR), rb = (row >> 4) % MR;
      g = (bx * 4 + w) * GPW + rb / GB;
      t = (rb % GB) * 16 + (row & 15);
    } else {
      g = bx * BM + row;
      t = 0;
    }
    a_g[j] = g < n_groups ? g : -1;
    a_t[j] = t;
  }

  acc_t acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

#ifdef DQMC_LIN_PROBE
  if ((a.cfg_probe >> 4) && (((blockIdx.x + blockIdx.y * gridDim.x) >> 8) & 1) && (blockIdx.x + blockIdx.y * gridDim.x) < 512)
    for (int z = 0; z < (a.cfg_probe >> 4); ++z) __builtin_amdgcn_s_sleep(127);
#endif
  for (int p = 0; p < a.n_pieces; ++p) {
    const LinPiece<real> pc = a.piece[p];
    const int w_row0 = pc.w_row;   // first W row of the current piece
    const real* a_src[APT];
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      if (a_g[j] >= 0) {
        const int b = a_g[j] / a.nrows, rr = a_g[j] - b * a.nrows;
        const long srow = ((long)b * pc.rpw + pc.r0 + (pc.bcast ? 0 : rr)) * a.TP + a_t[j];
        if (CAN32 && a.src_f32) a_src[j] = reinterpret_cast<const real*>(reinterpret_cast<const float*>(pc.src) + srow * pc.ld);
        else a_src[j] = pc.src + srow * pc.ld;
      } else {
        a_src[j] = nullptr;
      }
    }
    const int n_chunks = (pc.K + BK - 1) / BK;
    Vec4<real> ra[APT][BKX];
    real rb_[NBV][4];          // (scalars: as a conditionally assigned array of over-aligned float64 structs the prefetched weights lived in
                               // scratch -- every chunk's load was waited for at once, stored to scratch and re-read before the LDS store)
    auto load_chunk = [&](int kc) {
#pragma unroll
      for (int j = 0; j < APT; ++j)
#pragma unroll
        for (int x = 0; x < BKX; ++x) {
          const int k0 = kc * BK + 16 * x + 4 * a_kq;
          if (CAN32 && a.src_f32) {
            if (a_src[j] != nullptr && k0 < pc.K) {
              const Vec4<float> t = *reinterpret_cast<const Vec4<float>*>(reinterpret_cast<const float*>(a_src[j]) + k0);
              ra[j][x] = Vec4<real>{{(real)t.v[0], (real)t.v[1], (real)t.v[2], (real)t.v[3]}};
            } else {
              ra[j][x] = Vec4<real>{{0, 0, 0, 0}};
            }
          } else if (a_src[j] != nullptr && k0 < pc.K && !LIN_PROBE(a, 2)) ra[j][x] = *reinterpret_cast<const Vec4<real>*>(a_src[j] + k0);
          else ra[j][x] = Vec4<real>{{0, 0, 0, 0}};
        }
#pragma unroll
      for (int j = 0; j < NBV; ++j) {
        const int f = tid + NT * j;
        const int k = f / (BN / 4), n4 = f % (BN / 4);
        const int kk = kc * BK + k, col = col_blk0 + 4 * n4;
        Vec4<real> t{{0, 0, 0, 0}};
        if (f < BK * BN / 4 && kk < pc.K && col < a.ldw) t = *reinterpret_cast<const Vec4<real>*>(a.W + (long)(w_row0 + kk) * a.ldw + col);
#pragma unroll
        for (int x = 0; x < 4; ++x) rb_[j][x] = t.v[x];
      }
    };
    load_chunk(0);
    for (int kc = 0; kc < n_chunks; ++kc) {
      if (!LIN_PROBE(a, 3)) __syncthreads();  // previous chunk's fragments have been read
#pragma unroll
      for (int j = 0; j < APT; ++j)
#pragma unroll
        for (int x = 0; x < BKX; ++x) {
          Vec2<real>* dst = reinterpret_cast<Vec2<real>*>(&As[a_row[j] * AS + 16 * x + 4 * a_kq]);
          dst[0] = Vec2<real>{{ra[j][x].v[0], ra[j][x].v[1]}};
          dst[1] = Vec2<real>{{ra[j][x].v[2], ra[j][x].v[3]}};
        }
#pragma unroll
      for (int j = 0; j < NBV; ++j) {
        const int f = tid + NT * j;
        if (f < BK * BN / 4) {
          const int k = f / (BN / 4), n4 = f % (BN / 4);
          *reinterpret_cast<Vec4<real>*>(&Bs[k * BS + 4 * n4]) = Vec4<real>{{rb_[j][0], rb_[j][1], rb_[j][2], rb_[j][3]}};
        }
      }
      if (!LIN_PROBE(a, 3)) __syncthreads();
      if (kc + 1 < n_chunks) load_chunk(kc + 1);  // prefetch while the MFMAs run
      // float32: every chunk of 16 / 32 k is summed into a FRESH accumulator and its result added to the running sum.  The MFMA rounds
      // its accumulator once per k-step of 4, and in one long chain each of those K / 4 roundings is as large as the sum has grown;
      // chunked, the roundings inside a chunk are relative to the chunk's partial sum and only K / 16 additions happen at full size.
      // Measured on the MI355X (LiH / PauliNet, 4096 walkers, tools/gpu_r05_h.sh): median float32 error of E_loc 1.39e-7 -> 1.01e-7, error
      // scale m of the refinement 1.24e-8 -> 8.7e-9, walkers re-evaluated in float64 20.7 % -> 12.6 % (tests/f32_model.py with an MFMA-chain
      // accumulator had predicted x 0.6-0.7).  float64 passes keep the single chain.
      acc_t part[FRESH ? MR : 1][FRESH ? NR : 1];
      if (FRESH) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) part[FRESH ? i : 0][FRESH ? j : 0] = acc_t{0, 0, 0, 0};
      }
      if (!LIN_PROBE(a, 0))
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        const int kcol = kk * 4 + (lane >> 4);
        real fa[MR], fb[NR];
#pragma unroll
        for (int i = 0; i < MR; ++i) fa[i] = As[(wm * (16 * MR) + i * 16 + (lane & 15)) * AS + kcol];
#pragma unroll
        for (int j = 0; j < NR; ++j) fb[j] = Bs[kcol * BS + wn * (16 * NR) + j * 16 + (lane & 15)];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            if (FRESH) part[FRESH ? i : 0][FRESH ? j : 0] = Mfma<real>::run(fa[i], fb[j], part[FRESH ? i : 0][FRESH ? j : 0]);
            else acc[i][j] = Mfma<real>::run(fa[i], fb[j], acc[i][j]);
          }
      }
      if (FRESH) {
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j)
#pragma unroll
            for (int x = 0; x < 4; ++x) acc[i][j][x] += part[FRESH ? i : 0][FRESH ? j : 0][x];
      }
    }
  }

  if (!CHAIN) {
#ifdef DQMC_LIN_PROBE
    if (LIN_PROBE(a, 1)) {      // (the accumulators stay live through a store that never happens)
      real t = 0;
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (t == (real)12345.678) a.dst[0] = t;
      return;
    }
#endif
    HbmSink<real> sink(a);
    lin_epilogue<real, MR, NR, GPW>(acc, a, a.bias, a.act, a.ldw, a.pre, col_blk0 + wn * (16 * NR), wm, n_groups, sink, bx, As);
    return;
  }
  // ---- chained second layer: hidden tile -> LDS, then Y = act2(H W2 + b2) from there ----
  auto second_layer = [&](acc_t (&h)[MR][NR], const real* bias1, const real* W2, const real* bias2, const LinArgs<real>& out) {
    // W2 (<= 16 NR rows x 16 NR2 columns: one 4-vector per thread and chunk) is requested NOW, ahead of the hidden layer's
    // epilogue, and waits in registers: fetched chunk by chunk between the barriers of the loop below, every chunk cost one
    // L2 round trip that nothing overlapped
    static_assert(BK2 * BN2 / 4 <= NT, "one W2 vector per thread and chunk");
    const int K2 = a.ldw;                                   // hidden width (a multiple of 4, zero padded in Hs up to BN)
    const int n_chunks2 = (K2 + BK2 - 1) / BK2;             // <= NR: the hidden tile is BN = 16 NR columns wide
    real w2r[NR][4];                                        // (scalars, not Vec4 structs: the over-aligned float64 struct array stayed in scratch)
#pragma unroll
    for (int kc = 0; kc < NR; ++kc) {
      const int k = tid / (BN2 / 4), n4 = tid % (BN2 / 4);
      const int kk = kc * BK2 + k, col = 4 * n4;
      Vec4<real> t{{0, 0, 0, 0}};
      if (tid < BK2 * BN2 / 4 && kc < n_chunks2 && kk < K2 && col < a.ldw2) t = *reinterpret_cast<const Vec4<real>*>(W2 + (long)kk * a.ldw2 + col);
#pragma unroll
      for (int x = 0; x < 4; ++x) w2r[kc][x] = t.v[x];
    }
    {
      __syncthreads();                                      // every wave has read the last chunk of As / Bs: Hs may overwrite them
      LdsSink<real> hsink{Hs, HS};
      lin_epilogue<real, MR, NR, GPW>(h, a, bias1, a.act, a.ldw, (const real*)nullptr, 0, wm, n_groups, hsink, bx);
    }
    acc_t acc2[MR][NR2];
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR2; ++j) acc2[i][j] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kc = 0; kc < NR; ++kc) {
      if (kc < n_chunks2) {                                 // (block-uniform; a guard, not a break: the unrolled kc indexes w2r in registers)
      __syncthreads();                                      // hidden tile complete (kc = 0) / previous chunk of W2 consumed
      if (tid < BK2 * BN2 / 4) {
        const int k = tid / (BN2 / 4), n4 = tid % (BN2 / 4);
        *reinterpret_cast<Vec4<real>*>(&Bs2[k * BS2 + 4 * n4]) = Vec4<real>{{w2r[kc][0], w2r[kc][1], w2r[kc][2], w2r[kc][3]}};
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int kcol = kk * 4 + (lane >> 4);
        real fa[MR], fb[NR2];                                // (kc < NR: every k-step lies inside the BN = 16 NR columns of the hidden tile)
#pragma unroll
        for (int i = 0; i < MR; ++i) fa[i] = Hs[(wm * (16 * MR) + i * 16 + (lane & 15)) * HS + kc * BK2 + kcol];
#pragma unroll
        for (int j = 0; j < NR2; ++j) fb[j] = Bs2[kcol * BS2 + j * 16 + (lane & 15)];
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR2; ++j) acc2[i][j] = Mfma<real>::run(fa[i], fb[j], acc2[i][j]);
      }
      }
    }
    HbmSink<real> sink(out);
    lin_epilogue<real, MR, NR2, GPW>(acc2, out, bias2, a.act2, a.ldw2, (const real*)nullptr, 0, wm, n_groups, sink, bx);
  };
  second_layer(acc, a.bias, a.W2, a.bias2, a);
}

// ---- float32 layers on the bf16 matrix pipe (common.h: "float32 products on the bf16 matrix pipe") ----
// Same tiling, row mapping and epilogue as k_linear<float, ...>; the K loop runs in chunks of 32:
//   * the A tile stays float32 in LDS (row stride 34 words: the eight consecutive values a lane needs are four
//     conflict-free 8-byte reads) and is split into its three bf16 pieces in registers, once per row block and chunk
//     (44 VALU instructions against NR x NP MFMAs);
//   * the weights are split while they are staged: a thread takes two k rows x four columns, and writes one 16-byte
//     row segment per plane into Bs[plane][k pair][column] (row stride BN + 4 words: the four k groups of a fragment
//     read fall on distinct banks);
//   * NP = 9: all products of the pieces, the result is a float32 product sum with FEWER roundings than the
//     v_mfma_f32_16x16x4_f32 chain (144 instead of 256 matrix-pipe cycles per 16 x 16 x 32 block) -- the Laplacian
//     pass, whose derivative lanes cancel; NP = 6 (96 cycles) drops the three terms below 2^-24: value-only rows.
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// (second launch bound = waves per SIMD the register allocation must leave room for: the 48-row x 64-column wave tiles of
// the 48-lane groups need 172 registers unconstrained -- two waves per SIMD -- and fit three with 168)
template <int MR, int NR, int GPW, int WN, int NP>
__global__ void __launch_bounds__(256 * WN, (MR == 3 && WN == 1) ? 3 : 1) k_linear_bf(const LinArgs<float> a) {
  typedef float real;
  constexpr int NT = 256 * WN;
  constexpr int BM = 64 * MR, BN = 16 * NR * WN, BK = 32;
  constexpr int AS = BK + 2, BSTR = BN + 4;
  constexpr bool HALF = GPW == -1;
  constexpr int GB = GPW > 0 ? MR / GPW : 1;
  constexpr int APT = MR / WN;                                  // A rows per thread and chunk (two 16-byte loads each)
  constexpr int NBI = ((BK / 2) * (BN / 4) + NT - 1) / NT;       // B items (2 k rows x 4 columns) per thread and chunk
  constexpr bool HOLD_A = MR <= NR;                              // which operand's pieces stay in registers across the other's loop
#ifndef DQMC_BF_FRESH
#define DQMC_BF_FRESH 1
#endif
  // Laplacian tiles: the nine product passes of a 32-wide k chunk go to a FRESH accumulator per block, added to the running sum once
  // (k_linear: FRESH, same reason: nine roundings per chunk at the size of the whole sum become one).  N2 / FermiNet, 4096 walkers, same
  // call (tools/gpu_r05_h.sh): median float32 error of E_loc 2.03e-7 -> 1.17e-7, error scale m 5.6e-9 -> 3.7e-9, walkers re-evaluated in
  // float64 16.2 % -> 8.4 %, 56.4 -> 52.7 ms per step -- although <3,4,1,1,9> now spills 16 registers at its 168-register bound (2 before).
  constexpr bool BF_FRESH = DQMC_BF_FRESH && NP == 9;
  static_assert(MR % WN == 0, "MR must be a multiple of WN");
  typedef Mfma<float>::acc_t acc_t;
  HIP_DYNAMIC_SHARED(char, smem_raw)
  float* As = reinterpret_cast<float*>(smem_raw);                                  // [BM][AS]
  uint32_t* Bs = reinterpret_cast<uint32_t*>(smem_raw + (size_t)BM * AS * 4);     // [3][16][BSTR]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int n_groups = a.B * a.nrows;
  const TileId tile = tile_of_block();
  const int bx = tile.bx;
  const int col_blk0 = tile.by * BN;
  const int kg = lane >> 4, l15 = lane & 15;

  int a_row[APT], a_g[APT], a_t[APT];
  const int a_kq = tid & 3;
#pragma unroll
  for (int j = 0; j < APT; ++j) {
    const int row = (tid >> 2) + (NT / 4) * j;
    a_row[j] = row;
    int g, t;
    if (HALF) {
      const int w = row / (16 * MR), rb = (row >> 4) % MR;
      g = ((bx * 4 + w) * MR + rb) * 2 + ((row & 15) >> 3);
      t = row & 7;
    } else if (GPW > 0) {
      const int w = row / (16 * MR), rb = (row >> 4) % MR;
      g = (bx * 4 + w) * GPW + rb / GB;
      t = (rb % GB) * 16 + (row & 15);
    } else {
      g = bx * BM + row;
      t = 0;
    }
    a_g[j] = g < n_groups ? g : -1;
    a_t[j] = t;
  }

  acc_t acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  for (int p = 0; p < a.n_pieces; ++p) {
    const LinPiece<real> pc = a.piece[p];
    const int w_row0 = pc.w_row;
    const real* a_src[APT];
#pragma unroll
    for (int j = 0; j < APT; ++j) {
      if (a_g[j] >= 0) {
        const int b = a_g[j] / a.nrows, rr = a_g[j] - b * a.nrows;
        const long srow = ((long)b * pc.rpw + pc.r0 + (pc.bcast ? 0 : rr)) * a.TP + a_t[j];
        a_src[j] = pc.src + srow * pc.ld;
      } else {
        a_src[j] = nullptr;
      }
    }
    const int n_chunks = (pc.K + BK - 1) / BK;
    Vec4<real> ra[APT][2], rb_[NBI][2];
    auto load_chunk = [&](int kc) {
#pragma unroll
      for (int j = 0; j < APT; ++j)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const int k0 = kc * BK + 16 * x + 4 * a_kq;
          if (a_src[j] != nullptr && k0 < pc.K) ra[j][x] = *reinterpret_cast<const Vec4<real>*>(a_src[j] + k0);
          else ra[j][x] = Vec4<real>{{0, 0, 0, 0}};
        }
#pragma unroll
      for (int j = 0; j < NBI; ++j) {
        const int f = tid + NT * j;
        const int kp = f / (BN / 4), n4 = f % (BN / 4);
        const int kk = kc * BK + 2 * kp, col = col_blk0 + 4 * n4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (f < (BK / 2) * (BN / 4) && kk + h < pc.K && col < a.ldw)
            rb_[j][h] = *reinterpret_cast<const Vec4<real>*>(a.W + (long)(w_row0 + kk + h) * a.ldw + col);
          else
            rb_[j][h] = Vec4<real>{{0, 0, 0, 0}};
        }
      }
    };
    load_chunk(0);
    for (int kc = 0; kc < n_chunks; ++kc) {
      __syncthreads();  // previous chunk's fragments have been read
#pragma unroll
      for (int j = 0; j < APT; ++j)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          Vec2<real>* dst = reinterpret_cast<Vec2<real>*>(&As[a_row[j] * AS + 16 * x + 4 * a_kq]);
          dst[0] = Vec2<real>{{ra[j][x].v[0], ra[j][x].v[1]}};
          dst[1] = Vec2<real>{{ra[j][x].v[2], ra[j][x].v[3]}};
        }
#pragma unroll
      for (int j = 0; j < NBI; ++j) {
        const int f = tid + NT * j;
        if (f < (BK / 2) * (BN / 4)) {
          const int kp = f / (BN / 4), n4 = f % (BN / 4);
          BfFrag* d0 = reinterpret_cast<BfFrag*>(&Bs[(0 * 16 + kp) * BSTR + 4 * n4]);
          BfFrag* d1 = reinterpret_cast<BfFrag*>(&Bs[(1 * 16 + kp) * BSTR + 4 * n4]);
          BfFrag* d2 = reinterpret_cast<BfFrag*>(&Bs[(2 * 16 + kp) * BSTR + 4 * n4]);
          uint32_t w0[4], w1[4], w2[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {                        // column c of the item: k (low half) and k + 1 (high half)
            const float x0 = rb_[j][0].v[c], x1 = rb_[j][1].v[c];
            const uint32_t h = bf_pack2(x0, x1);
            const float r0 = x0 - bf_lo_as_float(h), r1 = x1 - bf_hi_as_float(h);
            const uint32_t m = bf_pack2(r0, r1);
            w0[c] = h; w1[c] = m; w2[c] = bf_pack2(r0 - bf_lo_as_float(m), r1 - bf_hi_as_float(m));
          }
          *d0 = BfFrag{{w0[0], w0[1], w0[2], w0[3]}};
          *d1 = BfFrag{{w1[0], w1[1], w1[2], w1[3]}};
          *d2 = BfFrag{{w2[0], w2[1], w2[2], w2[3]}};
        }
      }
      __syncthreads();
      if (kc + 1 < n_chunks) load_chunk(kc + 1);  // prefetch while the MFMAs run
      auto read_a = [&](int i, BfFrag& p0, BfFrag& p1, BfFrag& p2) {
        float av[8];
        const Vec2<real>* src = reinterpret_cast<const Vec2<real>*>(&As[(wm * (16 * MR) + i * 16 + l15) * AS + 8 * kg]);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const Vec2<real> t = src[q]; av[2 * q] = t.v[0]; av[2 * q + 1] = t.v[1]; }
        bf_split8(av, p0, p1, p2);
      };
      auto read_b = [&](int j, BfFrag& p0, BfFrag& p1, BfFrag& p2) {
        const int col = wn * (16 * NR) + j * 16 + l15;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          p0.w[q] = Bs[(0 * 16 + 4 * kg + q) * BSTR + col];
          p1.w[q] = Bs[(1 * 16 + 4 * kg + q) * BSTR + col];
          p2.w[q] = Bs[(2 * 16 + 4 * kg + q) * BSTR + col];
        }
      };
      if (HOLD_A) {
        BfFrag fa[MR][3];
#pragma unroll
        for (int i = 0; i < MR; ++i) read_a(i, fa[i][0], fa[i][1], fa[i][2]);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          BfFrag b0, b1, b2;
          read_b(j, b0, b1, b2);
          // the products of this chunk are summed into a fresh accumulator and added to the running sum once (k_linear: FRESH)
          acc_t part[MR];
#pragma unroll
          for (int i = 0; i < MR; ++i) part[i] = BF_FRESH ? acc_t{0, 0, 0, 0} : acc[i][j];
          if (NP == 9) {
#pragma unroll
            for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][2], b2, part[i]);
#pragma unroll
            for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][2], b1, part[i]);
#pragma unroll
            for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][1], b2, part[i]);
          }
#pragma unroll
          for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][2], b0, part[i]);
#pragma unroll
          for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][1], b1, part[i]);
#pragma unroll
          for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][0], b2, part[i]);
#pragma unroll
          for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][1], b0, part[i]);
#pragma unroll
          for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][0], b1, part[i]);
#pragma unroll
          for (int i = 0; i < MR; ++i) part[i] = mfma_bf16(fa[i][0], b0, part[i]);
#pragma unroll
          for (int i = 0; i < MR; ++i) {
            if (BF_FRESH) {
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[i][j][x] += part[i][x];
            } else {
              acc[i][j] = part[i];
            }
          }
          sched_fence();       // (keeps the scheduler from hoisting every column block's fragment reads: registers)
        }
      } else {
        BfFrag fb[NR][3];
#pragma unroll
        for (int j = 0; j < NR; ++j) read_b(j, fb[j][0], fb[j][1], fb[j][2]);
#pragma unroll
        for (int i = 0; i < MR; ++i) {
          BfFrag a0, a1, a2;
          read_a(i, a0, a1, a2);
          acc_t part[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) part[j] = BF_FRESH ? acc_t{0, 0, 0, 0} : acc[i][j];
          if (NP == 9) {
#pragma unroll
            for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a2, fb[j][2], part[j]);
#pragma unroll
            for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a2, fb[j][1], part[j]);
#pragma unroll
            for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a1, fb[j][2], part[j]);
          }
#pragma unroll
          for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a2, fb[j][0], part[j]);
#pragma unroll
          for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a1, fb[j][1], part[j]);
#pragma unroll
          for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a0, fb[j][2], part[j]);
#pragma unroll
          for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a1, fb[j][0], part[j]);
#pragma unroll
          for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a0, fb[j][1], part[j]);
#pragma unroll
          for (int j = 0; j < NR; ++j) part[j] = mfma_bf16(a0, fb[j][0], part[j]);
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            if (BF_FRESH) {
#pragma unroll
              for (int x = 0; x < 4; ++x) acc[i][j][x] += part[j][x];
            } else {
              acc[i][j] = part[j];
            }
          }
          sched_fence();
        }
      }
    }
  }
  HbmSink<real> sink(a);
  lin_epilogue<real, MR, NR, GPW>(acc, a, a.bias, a.act, a.ldw, a.pre, col_blk0 + wn * (16 * NR), wm, n_groups, sink, bx);
}

// "linear_bf" (dqmc_set_option, per context: LinArgs::cfg_bf): 0 = float32 MFMAs everywhere, 1 = every float32 layer of sufficient depth on the bf16
// pipe, 2 (default) = only the Laplacian tiles of the 48-lane groups (11-15 electrons), the one place where it measured
// FASTER.  Same-call A/Bs on the MI355X, ms per step, value 1 against 0: LiH / PauliNet 5.46 vs 5.46, N2 / FermiNet
// 52.9 -> 54.4, benzene / Psiformer (256 walkers) 264 -> 284, although the matrix pipe does 44 % less work.  What the
// counters say for the 128 x 128 tiles of LiH (K = 192): 7 VALU instructions per MFMA -- the forward-Laplacian epilogue
// plus the operand splits outweigh the MFMAs (12 k VALU cycles against 6.9 k matrix-pipe cycles per wave), LDS bank
// conflicts on a third of the LDS cycles, and 116 registers against 78 (2 instead of 3 workgroups per CU).  The N2 tiles
// (K = 256 .. 768, 133 TF/s = 85 % of the float32 MFMA peak with f32 MFMAs) needed 172 registers -- two waves per SIMD,
// 3.12 against 2.71 ms per launch; held to 168 (three waves) they come out ahead: N2 53.1 -> 52.1 ms per step with value
// 2, helped by the accuracy of the nine-product sum, which rounds less often than the f32 MFMA chain (refined walkers
// 5.4 % -> 3.9 % on N2; 4.3 % -> 4.0 % on LiH with value 1).  The value-row tiles stay slower with the split (63 -> 68 us).
// a layer goes to the bf16 pipe when its chunks of 32 k (NP MFMAs of 16 cycles per block) cost less than its k-steps of
// 4 (one MFMA of 32 cycles): pieces of a few k would multiply mostly padding
static bool bf_pays(const LinArgs<float>& a, int np) {
  long chunks = 0, ksteps = 0;
  for (int p = 0; p < a.n_pieces; ++p) { chunks += (a.piece[p].K + 31) / 32; ksteps += (a.piece[p].K + 3) / 4; }
  return a.cfg_bf != 0 && chunks * np * 16 + chunks * 12 < ksteps * 32;
}
template <int MR, int NR, int GPW, int WN> static void launch_bf(hipStream_t st, const LinArgs<float>& a, unsigned gx, unsigned gy) {
  constexpr int BM = 64 * MR, BN = 16 * NR * WN;
  constexpr size_t lds = (size_t)BM * 34 * 4 + (size_t)3 * 16 * (BN + 4) * 4;
  constexpr int NP = GPW == 0 ? 6 : 9;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear_bf<MR, NR, GPW, WN, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear_bf<MR, NR, GPW, WN, NP>), dim3(gx, gy), dim3(256 * WN), lds, st, a);
}
template <typename real, int MR, int NR, int GPW, int WN> struct BfLaunch {
  static bool run(hipStream_t, const LinArgs<real>&, unsigned, unsigned) { return false; }
};
template <int MR, int NR, int GPW, int WN> struct BfLaunch<float, MR, NR, GPW, WN> {
  static bool run(hipStream_t st, const LinArgs<float>& a, unsigned gx, unsigned gy) {
    if (!bf_pays(a, GPW == 0 ? 6 : 9)) return false;
    // 2 (default): the Laplacian tiles of the 48-lane groups (11-15 electrons: measured faster there) and of the 16-lane groups (up
    // to 4 electrons: same speed since the epilogue went lean, but the nine-product sum rounds less often than the f32 MFMA chain --
    // LiH / PauliNet, 4096 walkers: 872 -> 739 walkers re-evaluated in float64 per E_loc call, 2.41 -> 2.30 ms per call)
    if (a.cfg_bf >= 2 && !(GPW > 0 && ((MR == 3 && WN == 1) || MR == GPW))) return false;
    launch_bf<MR, NR, GPW, WN>(st, a, gx, gy);
    return true;
  }
};

// A/B hook (dqmc_set_option "linear_bkx", per context: LinArgs::cfg_bkx): 1 = 16-wide chunks everywhere, 2 = 32-wide chunks for the float32 small tiles,
// 3 (default) = for the float64 small tiles (the refinement twin's batches of a few hundred walkers), 4 = both (float32 too: LiH / PauliNet
// E_loc pass 2.41 -> 2.38 ms, but the fresh accumulator of a 32-wide chunk rounds twice as often as that of a 16-wide one -- N2 / FermiNet:
// error scale m 4.1e-9 -> 4.7e-9, 14.4 % -> 17.4 % of the walkers re-evaluated in float64, 49.1 -> 50.6 ms per step)
// float64, 96- / 128-lane groups (28 / 42 electrons): MR = 6 / 8 row blocks per wave; with two column blocks the accumulators
// alone are 96 / 128 registers and ONE wave per SIMD remains (hence the split groups below)
template <typename real> static bool wide_chunks(const LinArgs<real>& a) {
  int kmax = 0;
  for (int p = 0; p < a.n_pieces; ++p) kmax = a.piece[p].K > kmax ? a.piece[p].K : kmax;
  const bool on = sizeof(real) == 4 ? (a.cfg_bkx == 2 || a.cfg_bkx == 4) : (a.cfg_bkx == 3 || a.cfg_bkx == 4);
  return on && kmax >= 32;
}
template <typename real, int MR, int NR, int GPW, int WN> static void launch_cfg(hipStream_t st, const LinArgs<real>& a) {
  constexpr int BM = 64 * MR, BN = 16 * NR * WN;
  const long n_groups = (long)a.B * a.nrows;
  const unsigned gx = GPW == -2 ? (unsigned)((n_groups + 1) / 2)
                      : GPW < 0 ? (unsigned)((n_groups + 8 * MR - 1) / (8 * MR))
                      : GPW > 0 ? (unsigned)((n_groups + 4 * GPW - 1) / (4 * GPW)) : (unsigned)((n_groups + BM - 1) / BM);
  const unsigned gy = (unsigned)((a.ldw + BN - 1) / BN);
  if (BfLaunch<real, MR, NR, GPW, WN>::run(st, a, gx, gy)) return;
  if (MR == 1 && WN == 1 && GPW != 0 && wide_chunks(a))
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear<real, MR, NR, GPW, WN, false, 2, (MR == 1 && WN == 1 && GPW != 0) ? 2 : 1>), dim3(gx, gy), dim3(256 * WN), 0, st, a);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear<real, MR, NR, GPW, WN>), dim3(gx, gy), dim3(256 * WN), 0, st, a);
}
template <typename real, int MR, int NR, int GPW> static void launch_chain_cfg(hipStream_t st, const LinArgs<real>& a) {
  constexpr int BM = 64 * MR;
  const long n_groups = (long)a.B * a.nrows;
  const unsigned gx = GPW < 0 ? (unsigned)((n_groups + 8 * MR - 1) / (8 * MR))
                      : GPW > 0 ? (unsigned)((n_groups + 4 * GPW - 1) / (4 * GPW)) : (unsigned)((n_groups + BM - 1) / BM);
  if (MR == 1 && GPW != 0 && wide_chunks(a))
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear<real, MR, NR, GPW, 1, true, 2, (MR == 1 && GPW != 0) ? 2 : 1>), dim3(gx, 1), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_linear<real, MR, NR, GPW, 1, true, 2>), dim3(gx, 1), dim3(256), 0, st, a);
}
template <typename real, int MR, int GPW> static void launch_chain_nr(hipStream_t st, const LinArgs<real>& a) {
  if (a.ldw > 32) launch_chain_cfg<real, MR, 4, GPW>(st, a);
  else if (a.ldw > 16) launch_chain_cfg<real, MR, 2, GPW>(st, a);
  else launch_chain_cfg<real, MR, 1, GPW>(st, a);
}
// Shapes the chained kernel is instantiated for: hidden width <= 64, output width <= 32, 8- / 16- / 32-lane groups or
// value-only rows.
bool linear_chain_supported(int TP, int ldw_hidden, int ldw_out) {
  return (TP == 1 || TP == 8 || TP == 16 || TP == 32) && ldw_hidden <= 64 && ldw_out <= 32;
}
template <typename real> void launch_linear_chain(hipStream_t st, const LinArgs<real>& a_in) {
#ifdef DQMC_LIN_PROBE
  const LinArgs<real> a = with_probe(a_in);
#else
  const LinArgs<real>& a = a_in;
#endif
  switch (a.TP) {
    case 1: launch_chain_nr<real, 1, 0>(st, a); break;
    case 8: launch_chain_nr<real, 1, -1>(st, a); break;
    case 16: launch_chain_nr<real, 1, 1>(st, a); break;
    case 32: launch_chain_nr<real, 2, 1>(st, a); break;
    default: break;
  }
}

template <typename real, int MR, int GPW> static void launch_nr(hipStream_t st, const LinArgs<real>& a) {
  constexpr int NR_MAX = sizeof(real) == 8 ? 2 : 4;   // accumulator registers: MR*NR*4 per lane
  constexpr bool WIDE = sizeof(real) == 4 && MR % 2 == 0;   // 8 waves, BN = 128: the A rows of a wide layer are read half as often
  if (a.ldw > 64 && WIDE) launch_cfg<real, MR, (NR_MAX >= 4 ? 4 : 2), GPW, (WIDE ? 2 : 1)>(st, a);
  else if (a.ldw > 32 && NR_MAX >= 4) launch_cfg<real, MR, (NR_MAX >= 4 ? 4 : 2), GPW, 1>(st, a);
  else if (a.ldw > 16) launch_cfg<real, MR, 2, GPW, 1>(st, a);
  else launch_cfg<real, MR, 1, GPW, 1>(st, a);
}

// float64, 96- / 128-lane groups (28 / 42 electrons: what the refinement twin of the attention ansatzes runs): with one wave
// per group the accumulators of MR = 6 / 8 row blocks x 2 column blocks alone are 96 / 128 registers, the kernel needs
// 282 / 362 and ONE wave per SIMD remains (a quarter of the time of a benzene step at ~55 % of the f64 MFMA peak).
// Split groups (GPW = -2): a PAIR of waves holds a group, 3 / 4 row blocks each, FOUR column blocks (the A tile of a chunk
// feeds 64 columns instead of 32: half the L2 -> LDS traffic per flop); value lane and sum_c J_c^2 cross the pair through
// LDS in the epilogue.  Option "linear_f64_split" (default 1) / LinArgs::cfg_f64_split.
template <typename real, int MRH> static bool launch_split(hipStream_t st, const LinArgs<real>& a) {
  if constexpr (sizeof(real) == 8) {
    if (!a.cfg_f64_split || a.src_f32) return false;
    if (a.ldw > 32) launch_cfg<real, MRH, 4, -2, 1>(st, a);
    else if (a.ldw > 16) launch_cfg<real, MRH, 2, -2, 1>(st, a);
    else launch_cfg<real, MRH, 1, -2, 1>(st, a);
    return true;
  }
  return false;
}

template <typename real> void launch_linear(hipStream_t st, const LinArgs<real>& a_in) {
#ifdef DQMC_LIN_PROBE
  const LinArgs<real> a = with_probe(a_in);
#else
  const LinArgs<real>& a = a_in;
#endif
  switch (a.TP) {
    case 1: {
      // value-only rows (Metropolis sub-steps of the larger ansatzes): small batches would leave CUs idle with
      // 256-row tiles, so the tile height follows the row count (aim: >= 8 workgroups per CU)
      const long rows = (long)a.B * a.nrows;
      const long col_blocks = (a.ldw + 127) / 128;
      // (measured on N2 / FermiNet, 57 k rows x 256 columns: 64-row tiles 35.5 ms of linear time per step, 128-row 36.9,
      // 256-row 49.1 -- the tall tiles pay only when there are thousands of them)
      if constexpr (sizeof(real) == 8) {
        // float64 value rows by the hundred thousand (the heavy pairs of an ECP quadrature): 128 x 128 tiles on 8 waves --
        // launch_nr's 32-column float64 tiles stream the A rows of a 256-wide layer eight times (38 TF/s measured)
        if (a.ldw > 64 && (rows + 127) / 128 * col_blocks >= 1024) { launch_cfg<real, 2, 4, 0, 2>(st, a); break; }
      }
      if ((rows + 255) / 256 * col_blocks >= 2048) launch_nr<real, 4, 0>(st, a);
      else if ((rows + 127) / 128 * col_blocks >= 2048) launch_nr<real, 2, 0>(st, a);
      else launch_nr<real, 1, 0>(st, a);
      break;
    }
    // Laplacian-mode batches of a few hundred walkers (the float64 refinement pass, small evaluation batches) would
    // occupy a fraction of the 256 CUs with the 16-groups-per-workgroup tiles: shrink the tile until >= 512 workgroups
    // Laplacian-mode layers of the small systems (8-lane pair-compact edge rows; 16 lanes = up to 4 electrons).  These
    // layers are narrow (K, N <= 448 / 128) and stream their rows from HBM once: what they need is MANY workgroups in
    // flight, not tall tiles.  Measured at 4096 LiH walkers: 64-row tiles for the narrow layers and 128-row x 128-column
    // tiles (8 waves) for the wide ones take the whole E_loc pass from 2.06 to 1.80 ms against the 256-row tiles that suit
    // the large GEMMs of the bigger systems; small batches (the float64 refinement pass) get the small tiles anyway.
    case 8: launch_nr<real, 1, -1>(st, a); break;
    case 16: {
      const long wg2 = ((long)a.B * a.nrows + 7) / 8 * ((a.ldw + 127) / 128);      // workgroups of the 128 x 128 tiling
      if (a.ldw > 64 && wg2 >= 256) launch_nr<real, 2, 2>(st, a);
      else launch_nr<real, 1, 1>(st, a);
      break;
    }
    case 32: launch_nr<real, 2, 1>(st, a); break;      // (H2O / PauliNet, 4096 walkers: E_loc-only +3 % against <4, 2>)
    case 48: launch_nr<real, 3, 1>(st, a); break;
    case 64: launch_nr<real, 4, 1>(st, a); break;
    case 96: if (!launch_split<real, 3>(st, a)) launch_nr<real, 6, 1>(st, a); break;
    case 128: if (!launch_split<real, 4>(st, a)) launch_nr<real, 8, 1>(st, a); break;
    default: break;  // rejected by the engine before launch (lanes_supported)
  }
}

template void launch_linear<float>(hipStream_t, const LinArgs<float>&);
template void launch_linear<double>(hipStream_t, const LinArgs<double>&);
template void launch_linear_chain<float>(hipStream_t, const LinArgs<float>&);
template void launch_linear_chain<double>(hipStream_t, const LinArgs<double>&);

}  // namespace dqmc

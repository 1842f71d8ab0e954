// kernel_attention_mfma.hip -- forward-Laplacian multi-head attention with the contractions on MFMA.
//
// Same mathematics as kernel_attention.hip (hk.MultiHeadAttention at reference gnn/update_features.py:
// 273-278 / :436-444; propagation rules of SURVEY.md appendix C), restructured for the matrix cores:
// one workgroup per (walker, head), wave w owns the 16 queries of row block w and ALL keys, so every row
// reduction of the softmax algebra (over keys) stays inside the wave -- 16-lane shuffles, no LDS round trip.
// Per derivative lane t the wave issues
//     S-phase:  dS  = (q_t k0^T + q0 k_t^T)/sqrt(hd),   QK += q_t k_t^T          (K = hd)
//     O-phase:  out = dP_t v0 + P v_t,                  OL += dP_t v_t            (K = keys)
// as 16x16x4 MFMAs (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64: same fragment layout) on fragments read
// from LDS tiles (row stride hd + 2: conflict-free A/B reads), keeps P, the running sums
// A1 = sum_c dP_c*(dS_c - m_c), QK, OL in accumulator-layout registers for the whole lane loop, and turns dP
// (accumulator layout: column per lane) into an A operand (row per lane) through a 16 x keys scratch tile per wave.
//
// Round 4: float64 too -- the refinement twin IS the hot path of the attention ansatzes (BASELINE configs[3..4] send
// 28-100 % of their walkers through it) and its attention was scalar FMA code.  What had kept float64 out was the
// LDS: eight tiles of 48 x 66 doubles are 203 KB.  The value-lane operands that are A fragments -- q0 and P -- now live
// in REGISTERS (16 + 16 values per lane: a workgroup of this kernel is alone on its CU, one wave per SIMD with 512
// registers each), their LDS tiles are gone, and the dP scratch exists for the active waves only: benzene (42
// electrons, head_dim 64) 146 KB in float64, 73 KB in float32 (two workgroups per CU instead of one).
// head_dim a multiple of 16 up to 64, at most 64 queries and 64 keys.
#include "common.h"
#include "kernels.h"

namespace dqmc {

namespace {
constexpr int MAXC = 4;    // 16-wide tiles along keys and along head_dim
constexpr int MAXK = 16;   // k-steps of 4 along head_dim / keys

template <typename real> __device__ __forceinline__ real row16_sum(real v) {   // sum over the 16 lanes sharing lane >> 4
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
template <typename real> __device__ __forceinline__ real rmax(real a, real b) { return a > b ? a : b; }
template <typename real> __device__ __forceinline__ real row16_max(real v) {
  v = rmax(v, (real)__shfl_xor(v, 1, 64)); v = rmax(v, (real)__shfl_xor(v, 2, 64));
  v = rmax(v, (real)__shfl_xor(v, 4, 64)); v = rmax(v, (real)__shfl_xor(v, 8, 64));
  return v;
}
}  // namespace

// (second launch bound: float32 tile sets are <= 80 KB, two workgroups share a CU and each wave gets 256 registers;
// a float64 workgroup has the CU to itself)
// NCB = key tiles (compile time: the accumulator-layout state P, A1, QK, dS is NCB x 4 values per lane, and the float64
// instance lives at the 512-register limit -- what does not fit travels through AGPR copies, 12 VALU instructions per MFMA by
// the SQ counters of round 4 with NCB fixed at 4)
template <typename real, int NCB>
__global__ void __launch_bounds__(256, sizeof(real) == 4 ? 2 : 1)
k_attention_mfma(const real* __restrict__ q, const real* __restrict__ k, const real* __restrict__ v, real* __restrict__ out, int width,
                 int H, int hd, LaneInfo li, int n_const, const real* __restrict__ k_const, const real* __restrict__ v_const) {
  typedef typename Mfma<real>::acc_t acc_t;
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* sm = reinterpret_cast<real*>(smem_raw);
  const int N = li.N, T = li.T, TP = li.TP;
  const int M = n_const + N;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int S = hd + 2;                               // row stride of the [rows][hd] tiles
  const int n_cb = NCB < (M + 15) / 16 ? NCB : (M + 15) / 16, n_db = hd / 16;     // key tiles (<= NCB; the launcher picks NCB = their number), head_dim tiles
  const int n_rb = (N + 15) / 16;                     // query row blocks = active waves
  const int M16 = n_cb * 16, N16 = n_rb * 16;
  const int SA = M16 + 2;                             // row stride of the per-wave [16][keys] A-operand tile
  const int nkd = hd / 4, nkm = M16 / 4;              // k-steps of the S-phase / of the O-phase
  // (value-only launches have no current-lane k / v tiles: the launcher allocates 30 instead of 47 KB for 30 electrons x 64,
  // five instead of three workgroups share a CU -- the launch is bound by the latency of its tile loads)
  real* k0 = sm;                real* v0 = k0 + M16 * S;
  real* qc = v0 + M16 * S;      real* kc = qc + N16 * S;   real* vc = kc + (T > 1 ? M16 * S : 0);
  real* DA = vc + (T > 1 ? M16 * S : 0);              // [n_rb waves][16][SA]  dP_t (and once P) as A operand
  const real sc = (real)(1.0 / sqrt((double)hd));
  const long row0 = (long)b * N * TP;
  const int col0 = h * hd;
  const bool active = wave < n_rb;
  const int i_base = wave * 16;                       // first query of this wave

  // Tile loads: thread (r_in, v4) moves one 4-vector per pass of rpp rows; the (<= MAXP) loads of a lane's q, k and
  // v tiles are all issued before any is used, and the next lane's are in flight while the current one is
  // multiplied (registers, not LDS, are the second buffer).
  constexpr int MAXP = NCB;                            // passes of rpp >= 16 rows cover the 16 NCB key rows (and the query rows)
  const int vpr = hd / 4;                              // 4-vectors per row
  const int rpp = 256 / vpr;                           // rows per pass
  const int r_in = tid / vpr, v4 = tid - r_in * vpr;
  const bool ld_thread = r_in < rpp;
  const real* qg = q + row0 * width + col0 + 4 * v4;
  const real* kg = k + row0 * width + col0 + 4 * v4;
  const real* vg = v + row0 * width + col0 + 4 * v4;
  Vec4<real> Rq[MAXP], Rk[MAXP], Rv[MAXP];
  auto issue = [&](int t) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int row = p * rpp + r_in;
      Rq[p] = Vec4<real>{{0, 0, 0, 0}};
      Rk[p] = Vec4<real>{{0, 0, 0, 0}};
      Rv[p] = Vec4<real>{{0, 0, 0, 0}};
      if (!ld_thread) continue;
      if (row < N) Rq[p] = *reinterpret_cast<const Vec4<real>*>(qg + ((long)row * TP + t) * width);
      if (row < n_const) {
        if (t == 0) {
          Rk[p] = *reinterpret_cast<const Vec4<real>*>(k_const + (long)row * (H * hd) + col0 + 4 * v4);
          Rv[p] = *reinterpret_cast<const Vec4<real>*>(v_const + (long)row * (H * hd) + col0 + 4 * v4);
        }
      } else if (row < M) {
        const long off = ((long)(row - n_const) * TP + t) * width;
        Rk[p] = *reinterpret_cast<const Vec4<real>*>(kg + off);
        Rv[p] = *reinterpret_cast<const Vec4<real>*>(vg + off);
      }
    }
  };
  auto put = [&](real* dq, real* dk, real* dv) {   // registers -> LDS tiles (rows beyond N / M are zero)
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int row = p * rpp + r_in;
      if (!ld_thread) continue;
      if (row < N16) {
        Vec2<real>* d2 = reinterpret_cast<Vec2<real>*>(dq + row * S + 4 * v4);
        d2[0] = Vec2<real>{{Rq[p].v[0], Rq[p].v[1]}}; d2[1] = Vec2<real>{{Rq[p].v[2], Rq[p].v[3]}};
      }
      if (row < M16) {
        Vec2<real>* k2 = reinterpret_cast<Vec2<real>*>(dk + row * S + 4 * v4);
        k2[0] = Vec2<real>{{Rk[p].v[0], Rk[p].v[1]}}; k2[1] = Vec2<real>{{Rk[p].v[2], Rk[p].v[3]}};
        Vec2<real>* v2 = reinterpret_cast<Vec2<real>*>(dv + row * S + 4 * v4);
        v2[0] = Vec2<real>{{Rv[p].v[0], Rv[p].v[1]}}; v2[1] = Vec2<real>{{Rv[p].v[2], Rv[p].v[3]}};
      }
    }
  };
  // rows of this lane in accumulator layout
  int irow[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) irow[rg] = i_base + Mfma<real>::row_of(lane, rg);

  issue(0);
  put(qc, k0, v0);                                    // (q0 passes through the current-lane tile on its way to registers)
  if (T > 1) issue(1);
  __syncthreads();

  acc_t P[NCB], A1[NCB], QK[NCB], OL[MAXC];
  real A2[4] = {0, 0, 0, 0};
  real qa[MAXK], pa[MAXK];                            // q0 and P as A fragments: row l15, k = 4 kk + l4
#pragma unroll
  for (int c = 0; c < NCB; ++c) { P[c] = acc_t{0, 0, 0, 0}; A1[c] = acc_t{0, 0, 0, 0}; QK[c] = acc_t{0, 0, 0, 0}; }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) OL[c] = acc_t{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < MAXK; ++kk) { qa[kk] = 0; pa[kk] = 0; }
  real* myDA = DA + (active ? wave : 0) * 16 * SA;

  // ---- value lane: P = softmax(q0 k0^T / sqrt(hd)) over the M keys, out_0 = P v0 ----
  if (active) {
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk)
      if (kk < nkd) qa[kk] = qc[(i_base + l15) * S + kk * 4 + l4];
    acc_t s0[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) s0[c] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      if (kk >= nkd) break;
#pragma unroll
      for (int c = 0; c < NCB; ++c)
        if (c < n_cb) s0[c] = Mfma<real>::run(qa[kk], k0[(c * 16 + l15) * S + kk * 4 + l4], s0[c]);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      real mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < NCB; ++c)
        if (c < n_cb && c * 16 + l15 < M) mx = rmax(mx, (real)(s0[c][rg] * sc));
      mx = row16_max<real>(mx);
      real sum = 0;
#pragma unroll
      for (int c = 0; c < NCB; ++c) {
        real e = 0;
        if (c < n_cb && c * 16 + l15 < M) e = r_exp<real>(s0[c][rg] * sc - mx);
        P[c][rg] = e;
        sum += e;
      }
      sum = row16_sum<real>(sum);
      const real inv = 1 / sum;
#pragma unroll
      for (int c = 0; c < NCB; ++c) P[c][rg] *= inv;
    }
    // P as an A operand (row = query, k = key): through the wave's scratch tile, then held in registers
#pragma unroll
    for (int c = 0; c < NCB; ++c)
      if (c < n_cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) myDA[Mfma<real>::row_of(lane, rg) * SA + c * 16 + l15] = P[c][rg];
    wave_lds_fence();
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk)
      if (kk < nkm) pa[kk] = myDA[l15 * SA + kk * 4 + l4];
    acc_t o[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) o[c] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      if (kk >= nkm) break;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_db) o[c] = Mfma<real>::run(pa[kk], v0[(kk * 4 + l4) * S + c * 16 + l15], o[c]);
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < n_db)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          if (irow[rg] < N) out[(row0 + (long)irow[rg] * TP) * width + col0 + c * 16 + l15] = o[c][rg];
  }
  if (T == 1) return;

  // ---- derivative lanes, then the Laplacian lane (t = T-1) ----
  for (int t = 1; t < T; ++t) {
    const bool lap = t == T - 1;
    __syncthreads();                                  // previous lane's tiles (t = 1: q0 in qc) are no longer read
    put(qc, kc, vc);
    __syncthreads();
    if (t + 1 < T) issue(t + 1);                      // in flight while this lane is multiplied
    if (!active) continue;
    // S-phase
    acc_t ds[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) ds[c] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      if (kk >= nkd) break;
      const int ko = kk * 4 + l4;
      const real a0 = qa[kk], at = qc[(i_base + l15) * S + ko];
#pragma unroll
      for (int c = 0; c < NCB; ++c)
        if (c < n_cb) {
          const real b0 = k0[(c * 16 + l15) * S + ko], bt = kc[(c * 16 + l15) * S + ko];
          ds[c] = Mfma<real>::run(at, b0, ds[c]);
          ds[c] = Mfma<real>::run(a0, bt, ds[c]);
          if (!lap) QK[c] = Mfma<real>::run(at, bt, QK[c]);
        }
    }
    // softmax algebra on the wave's 16 x M block (accumulator layout), row sums by 16-lane shuffles
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      real m = 0;
#pragma unroll
      for (int c = 0; c < NCB; ++c)
        if (c < n_cb) {
          ds[c][rg] = lap ? (ds[c][rg] + 2 * QK[c][rg]) * sc : ds[c][rg] * sc;     // dS_c  or  L_S
          m += P[c][rg] * ds[c][rg];
        }
      m = row16_sum<real>(m);                          // rowsum(P*dS_c)  or  rowsum(P*L_S)
      real s2 = 0;
#pragma unroll
      for (int c = 0; c < NCB; ++c)
        if (c < n_cb) {
          real dp;
          if (!lap) {
            const real cc = ds[c][rg] - m;
            dp = P[c][rg] * cc;
            A1[c][rg] += dp * cc;
            s2 += dp * ds[c][rg];
          } else {
            dp = A1[c][rg] + P[c][rg] * (ds[c][rg] - m - A2[rg]);                     // L_P
          }
          myDA[Mfma<real>::row_of(lane, rg) * SA + c * 16 + l15] = dp;
        }
      if (!lap) A2[rg] += row16_sum<real>(s2);
    }
    wave_lds_fence();
    // O-phase (the wave reads back only what it wrote itself: LDS operations of a wave complete in order)
    acc_t o[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) o[c] = lap ? OL[c] : acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      if (kk >= nkm) break;
      const int jo = kk * 4 + l4;
      const real adp = myDA[l15 * SA + jo], ap = pa[kk];
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_db) {
          const real b0 = v0[jo * S + c * 16 + l15], bt = vc[jo * S + c * 16 + l15];
          o[c] = Mfma<real>::run(adp, b0, o[c]);
          o[c] = Mfma<real>::run(ap, bt, o[c]);
          if (!lap) OL[c] = Mfma<real>::run(adp, bt + bt, OL[c]);                     // 2 sum_c dP_c v_c
        }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < n_db)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          if (irow[rg] < N) out[(row0 + (long)irow[rg] * TP + t) * width + col0 + c * 16 + l15] = o[c][rg];
  }
  __syncthreads();
  for (int t = T; t < TP; ++t)
    for (int e = tid; e < N * hd; e += nthr) {
      const int i = e / hd, d = e - i * hd;
      out[(row0 + (long)i * TP + t) * width + col0 + d] = 0;
    }
}

// ---- split variant (round 4): a PAIR of waves per query row block ---------------------------------------------------------
// k_attention_mfma gives a row block of 16 queries to one wave: 28 electrons (C4H4) keep two of the four waves busy, 42 three,
// and the float64 instance holds so much accumulator-layout state per wave that it lives on AGPR copies (12 VALU instructions
// per MFMA by the SQ counters).  Here the workgroup has EIGHT waves, wave = 2 rb + half: in the S-phase a wave owns the key
// tiles c = half, half + 2 of its row block (state P, A1, QK, dS for two tiles instead of four), in the O-phase the head_dim
// tiles d = half, half + 2 (OL, out for two tiles) with the contraction over ALL keys.  What crosses the pair goes through LDS:
// the row maximum and row sum of the softmax once, the row sums m_t = sum_keys P dS_t once per lane (and the running A2 sums
// with the Laplacian lane), dP_t into the row block's shared scratch tile.  Four workgroup barriers per lane instead of two.
template <typename real>
__global__ void __launch_bounds__(512, 2)
k_attention_mfma_split(const real* __restrict__ q, const real* __restrict__ k, const real* __restrict__ v, real* __restrict__ out, int width,
                       int H, int hd, LaneInfo li, int n_const, const real* __restrict__ k_const, const real* __restrict__ v_const) {
  typedef typename Mfma<real>::acc_t acc_t;
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* sm = reinterpret_cast<real*>(smem_raw);
  const int N = li.N, T = li.T, TP = li.TP;
  const int M = n_const + N;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int rb = wave >> 1, half = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int S = hd + 2;
  const int n_cb = (M + 15) / 16, n_db = hd / 16, n_rb = (N + 15) / 16;
  const int M16 = n_cb * 16, N16 = n_rb * 16;
  const int SA = M16 + 2;
  const int nkd = hd / 4, nkm = M16 / 4;
  real* k0 = sm;                real* v0 = k0 + M16 * S;
  real* qc = v0 + M16 * S;      real* kc = qc + N16 * S;   real* vc = kc + M16 * S;
  real* DA = vc + M16 * S;                            // [n_rb][16][SA]  P once, then dP_t, as A operand (shared by the pair)
  real* XC = DA + n_rb * 16 * SA;                     // [4 slots][n_rb][2 halves][16 rows][2]  partial row sums crossing the pair
  const real sc = (real)(1.0 / sqrt((double)hd));
  const long row0 = (long)b * N * TP;
  const int col0 = h * hd;
  const bool active = rb < n_rb;
  const int i_base = rb * 16;

  constexpr int MAXP = 2;                              // 512 threads: >= 32 rows per pass
  const int vpr = hd / 4;
  const int rpp = 512 / vpr;
  const int r_in = tid / vpr, v4 = tid - r_in * vpr;
  const real* qg = q + row0 * width + col0 + 4 * v4;
  const real* kg = k + row0 * width + col0 + 4 * v4;
  const real* vg = v + row0 * width + col0 + 4 * v4;
  Vec4<real> Rq[MAXP], Rk[MAXP], Rv[MAXP];
  auto issue = [&](int t) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int row = p * rpp + r_in;
      Rq[p] = Vec4<real>{{0, 0, 0, 0}};
      Rk[p] = Vec4<real>{{0, 0, 0, 0}};
      Rv[p] = Vec4<real>{{0, 0, 0, 0}};
      if (row < N) Rq[p] = *reinterpret_cast<const Vec4<real>*>(qg + ((long)row * TP + t) * width);
      if (row < n_const) {
        if (t == 0) {
          Rk[p] = *reinterpret_cast<const Vec4<real>*>(k_const + (long)row * (H * hd) + col0 + 4 * v4);
          Rv[p] = *reinterpret_cast<const Vec4<real>*>(v_const + (long)row * (H * hd) + col0 + 4 * v4);
        }
      } else if (row < M) {
        const long off = ((long)(row - n_const) * TP + t) * width;
        Rk[p] = *reinterpret_cast<const Vec4<real>*>(kg + off);
        Rv[p] = *reinterpret_cast<const Vec4<real>*>(vg + off);
      }
    }
  };
  auto put = [&](real* dq, real* dk, real* dv) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int row = p * rpp + r_in;
      if (row < N16) {
        Vec2<real>* d2 = reinterpret_cast<Vec2<real>*>(dq + row * S + 4 * v4);
        d2[0] = Vec2<real>{{Rq[p].v[0], Rq[p].v[1]}}; d2[1] = Vec2<real>{{Rq[p].v[2], Rq[p].v[3]}};
      }
      if (row < M16) {
        Vec2<real>* k2 = reinterpret_cast<Vec2<real>*>(dk + row * S + 4 * v4);
        k2[0] = Vec2<real>{{Rk[p].v[0], Rk[p].v[1]}}; k2[1] = Vec2<real>{{Rk[p].v[2], Rk[p].v[3]}};
        Vec2<real>* v2 = reinterpret_cast<Vec2<real>*>(dv + row * S + 4 * v4);
        v2[0] = Vec2<real>{{Rv[p].v[0], Rv[p].v[1]}}; v2[1] = Vec2<real>{{Rv[p].v[2], Rv[p].v[3]}};
      }
    }
  };
  int irow[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) irow[rg] = i_base + Mfma<real>::row_of(lane, rg);
  // partial row quantities crossing the pair: slot s, value index x (0 / 1)
  auto xc_put = [&](int slot, int x, const real (&val)[4]) {
    if (active && l15 == 0) {
      real* dst = XC + (((slot * n_rb + rb) * 2 + half) * 16) * 2;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) dst[Mfma<real>::row_of(lane, rg) * 2 + x] = val[rg];
    }
  };
  auto xc_other = [&](int slot, int x, int rg) -> real {
    return XC[(((slot * n_rb + rb) * 2 + (half ^ 1)) * 16 + Mfma<real>::row_of(lane, rg)) * 2 + x];
  };

  issue(0);
  put(qc, k0, v0);
  if (T > 1) issue(1);
  __syncthreads();

  acc_t P[2], A1[2], QK[2], OL[2];
  real A2[4] = {0, 0, 0, 0};                          // this half's share of the running sums
  real qa[MAXK], pa[MAXK];
#pragma unroll
  for (int j = 0; j < 2; ++j) { P[j] = acc_t{0, 0, 0, 0}; A1[j] = acc_t{0, 0, 0, 0}; QK[j] = acc_t{0, 0, 0, 0}; OL[j] = acc_t{0, 0, 0, 0}; }
#pragma unroll
  for (int kk = 0; kk < MAXK; ++kk) { qa[kk] = 0; pa[kk] = 0; }
  real* myDA = DA + (active ? rb : 0) * 16 * SA;

  // ---- value lane ----
  acc_t s0[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) s0[j] = acc_t{0, 0, 0, 0};
  real part[4];
  if (active) {
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk)
      if (kk < nkd) qa[kk] = qc[(i_base + l15) * S + kk * 4 + l4];
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      if (kk >= nkd) break;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = half + 2 * j;
        if (c < n_cb) s0[j] = Mfma<real>::run(qa[kk], k0[(c * 16 + l15) * S + kk * 4 + l4], s0[j]);
      }
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      real mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = half + 2 * j;
        if (c < n_cb && c * 16 + l15 < M) mx = rmax(mx, (real)(s0[j][rg] * sc));
      }
      part[rg] = row16_max<real>(mx);
    }
  }
  xc_put(0, 0, part);
  __syncthreads();
  if (active) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const real mx = rmax(part[rg], xc_other(0, 0, rg));
      real sum = 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = half + 2 * j;
        real e = 0;
        if (c < n_cb && c * 16 + l15 < M) e = r_exp<real>(s0[j][rg] * sc - mx);
        P[j][rg] = e;
        sum += e;
      }
      part[rg] = row16_sum<real>(sum);
    }
  }
  xc_put(1, 0, part);
  __syncthreads();
  if (active) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const real inv = 1 / (part[rg] + xc_other(1, 0, rg));
#pragma unroll
      for (int j = 0; j < 2; ++j) P[j][rg] *= inv;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = half + 2 * j;
      if (c < n_cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) myDA[Mfma<real>::row_of(lane, rg) * SA + c * 16 + l15] = P[j][rg];
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk)
      if (kk < nkm) pa[kk] = myDA[l15 * SA + kk * 4 + l4];
    acc_t o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) o[j] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < MAXK; ++kk) {
      if (kk >= nkm) break;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int d = half + 2 * j;
        if (d < n_db) o[j] = Mfma<real>::run(pa[kk], v0[(kk * 4 + l4) * S + d * 16 + l15], o[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int d = half + 2 * j;
      if (d < n_db)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          if (irow[rg] < N) out[(row0 + (long)irow[rg] * TP) * width + col0 + d * 16 + l15] = o[j][rg];
    }
  }
  if (T == 1) return;

  // ---- derivative lanes, then the Laplacian lane ----
  for (int t = 1; t < T; ++t) {
    const bool lap = t == T - 1;
    __syncthreads();                                  // previous lane's tiles and dP scratch are no longer read
    put(qc, kc, vc);
    __syncthreads();
    if (t + 1 < T) issue(t + 1);
    acc_t ds[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) ds[j] = acc_t{0, 0, 0, 0};
    real mp[4] = {0, 0, 0, 0};
    if (active) {
#pragma unroll
      for (int kk = 0; kk < MAXK; ++kk) {
        if (kk >= nkd) break;
        const int ko = kk * 4 + l4;
        const real a0 = qa[kk], at = qc[(i_base + l15) * S + ko];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = half + 2 * j;
          if (c < n_cb) {
            const real b0 = k0[(c * 16 + l15) * S + ko], bt = kc[(c * 16 + l15) * S + ko];
            ds[j] = Mfma<real>::run(at, b0, ds[j]);
            ds[j] = Mfma<real>::run(a0, bt, ds[j]);
            if (!lap) QK[j] = Mfma<real>::run(at, bt, QK[j]);
          }
        }
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        real m = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = half + 2 * j;
          if (c < n_cb) {
            ds[j][rg] = lap ? (ds[j][rg] + 2 * QK[j][rg]) * sc : ds[j][rg] * sc;     // dS_c  or  L_S
            m += P[j][rg] * ds[j][rg];
          }
        }
        mp[rg] = row16_sum<real>(m);
      }
    }
    const int slot = 2 + (t & 1);
    xc_put(slot, 0, mp);
    if (lap) xc_put(slot, 1, A2);
    __syncthreads();
    if (active) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const real m = mp[rg] + xc_other(slot, 0, rg);
        const real a2 = lap ? A2[rg] + xc_other(slot, 1, rg) : (real)0;
        real s2 = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = half + 2 * j;
          if (c < n_cb) {
            real dp;
            if (!lap) {
              const real cc = ds[j][rg] - m;
              dp = P[j][rg] * cc;
              A1[j][rg] += dp * cc;
              s2 += dp * ds[j][rg];
            } else {
              dp = A1[j][rg] + P[j][rg] * (ds[j][rg] - m - a2);                       // L_P
            }
            myDA[Mfma<real>::row_of(lane, rg) * SA + c * 16 + l15] = dp;
          }
        }
        if (!lap) A2[rg] += row16_sum<real>(s2);
      }
    }
    __syncthreads();                                  // dP_t of both halves is in the row block's scratch tile
    if (active) {
      acc_t o[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) o[j] = lap ? OL[j] : acc_t{0, 0, 0, 0};
#pragma unroll
      for (int kk = 0; kk < MAXK; ++kk) {
        if (kk >= nkm) break;
        const int jo = kk * 4 + l4;
        const real adp = myDA[l15 * SA + jo], ap = pa[kk];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int d = half + 2 * j;
          if (d < n_db) {
            const real b0 = v0[jo * S + d * 16 + l15], bt = vc[jo * S + d * 16 + l15];
            o[j] = Mfma<real>::run(adp, b0, o[j]);
            o[j] = Mfma<real>::run(ap, bt, o[j]);
            if (!lap) OL[j] = Mfma<real>::run(adp, bt + bt, OL[j]);                   // 2 sum_c dP_c v_c
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int d = half + 2 * j;
        if (d < n_db)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            if (irow[rg] < N) out[(row0 + (long)irow[rg] * TP + t) * width + col0 + d * 16 + l15] = o[j][rg];
      }
    }
  }
  __syncthreads();
  for (int t = T; t < TP; ++t)
    for (int e = tid; e < N * hd; e += nthr) {
      const int i = e / hd, d = e - i * hd;
      out[(row0 + (long)i * TP + t) * width + col0 + d] = 0;
    }
}

template <typename real> size_t attention_mfma_split_lds_bytes(int N, int hd, int n_const) {
  const size_t M16 = ((size_t)N + n_const + 15) / 16 * 16, N16 = ((size_t)N + 15) / 16 * 16;
  return sizeof(real) * ((N16 + 4 * M16) * (hd + 2) + (N16 / 16) * 16 * (M16 + 2) + 4 * (N16 / 16) * 2 * 16 * 2);
}
template <typename real>
int launch_attention_mfma_split(hipStream_t st, const real* q, const real* k, const real* v, real* out, int width, int H, int hd, int B,
                                LaneInfo li, int n_const, const real* k_const, const real* v_const) {
  const size_t lds = attention_mfma_split_lds_bytes<real>(li.N, hd, n_const);
  if (lds > 160 * 1024 || hd % 16 != 0 || hd > 64 || hd < 16 || li.N > 64 || li.N + n_const > 64) return -1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_mfma_split<real>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return -2;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention_mfma_split<real>), dim3((unsigned)(B * H)), dim3(512), lds, st, q, k, v, out, width, H, hd,
                     li, n_const, k_const, v_const);
  return 0;
}
template int launch_attention_mfma_split<double>(hipStream_t, const double*, const double*, const double*, double*, int, int, int, int, LaneInfo,
                                                 int, const double*, const double*);
template size_t attention_mfma_split_lds_bytes<double>(int, int, int);

template <typename real> size_t attention_mfma_lds_bytes(int N, int hd, int n_const) {
  const size_t M16 = ((size_t)N + n_const + 15) / 16 * 16, N16 = ((size_t)N + 15) / 16 * 16;
  return sizeof(real) * ((N16 + 4 * M16) * (hd + 2) + (N16 / 16) * 16 * (M16 + 2));
}

template <typename real> bool attention_mfma_supported(int N, int hd, int n_const) {
  return hd % 16 == 0 && hd <= 64 && N <= 64 && N + n_const <= 64 && attention_mfma_lds_bytes<real>(N, hd, n_const) <= 160 * 1024;
}
// Measured on MI355X (float32, 4 heads x 64, attention time per VMC step, scalar kernel -> this one): 42 electrons
// 135 -> 31 ms, 28 electrons 66 -> 49 ms, 14 electrons 29 -> 67 ms, 4 electrons 10 -> 75 ms: with a single query
// row block three of the four waves idle through the per-lane tile traffic, so the scalar kernel keeps the
// small systems.
bool attention_mfma_profitable(int N) { return N > 16; }

template <typename real, int NCB>
static int launch_attention_mfma_ncb(hipStream_t st, const real* q, const real* k, const real* v, real* out, int width, int H, int hd, int B,
                                     LaneInfo li, int n_const, const real* k_const, const real* v_const, size_t lds) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_mfma<real, NCB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return -2;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention_mfma<real, NCB>), dim3((unsigned)(B * H)), dim3(256), lds, st, q, k, v, out, width, H, hd,
                     li, n_const, k_const, v_const);
  return 0;
}
template <typename real>
int launch_attention_mfma(hipStream_t st, const real* q, const real* k, const real* v, real* out, int width, int H,
                          int hd, int B, LaneInfo li, int n_const, const real* k_const, const real* v_const, int exact_tiles) {
  size_t lds = attention_mfma_lds_bytes<real>(li.N, hd, n_const);
  if (li.T == 1) {                                   // value-only: k0, v0, q and the per-wave P tiles (the kernel lays DA behind q)
    const size_t M16 = ((size_t)li.N + n_const + 15) / 16 * 16, N16 = ((size_t)li.N + 15) / 16 * 16;
    lds = sizeof(real) * ((N16 + 2 * M16) * (hd + 2) + (N16 / 16) * 16 * (M16 + 2));
  }
  const int n_cb = exact_tiles ? (li.N + n_const + 15) / 16 : 4;
  if (n_cb <= 2) return launch_attention_mfma_ncb<real, 2>(st, q, k, v, out, width, H, hd, B, li, n_const, k_const, v_const, lds);
  if (n_cb == 3) return launch_attention_mfma_ncb<real, 3>(st, q, k, v, out, width, H, hd, B, li, n_const, k_const, v_const, lds);
  return launch_attention_mfma_ncb<real, 4>(st, q, k, v, out, width, H, hd, B, li, n_const, k_const, v_const, lds);
}

template size_t attention_mfma_lds_bytes<float>(int, int, int);
template size_t attention_mfma_lds_bytes<double>(int, int, int);
template bool attention_mfma_supported<float>(int, int, int);
template bool attention_mfma_supported<double>(int, int, int);
template int launch_attention_mfma<float>(hipStream_t, const float*, const float*, const float*, float*, int, int, int, int, LaneInfo, int,
                                          const float*, const float*, int);
template int launch_attention_mfma<double>(hipStream_t, const double*, const double*, const double*, double*, int, int, int, int, LaneInfo,
                                           int, const double*, const double*, int);

}  // namespace dqmc

// kernel_attention_mfma.hip -- forward-Laplacian multi-head attention with the contractions on MFMA.
//
// Same mathematics as kernel_attention.hip (hk.MultiHeadAttention at reference gnn/update_features.py:
// 273-278 / :436-444; propagation rules of SURVEY.md appendix C), restructured for the matrix cores:
// one workgroup per (walker, head), wave w owns the 16 queries of row block w and ALL keys, so every row
// reduction of the softmax algebra (over keys) stays inside the wave -- 16-lane shuffles, no LDS round trip.
// Per derivative lane t the wave issues
//     S-phase:  dS  = (q_t k0^T + q0 k_t^T)/sqrt(hd),   QK += q_t k_t^T          (K = hd)
//     O-phase:  out = dP_t v0 + P v_t,                  OL += dP_t v_t            (K = keys)
// as v_mfma_f32_16x16x4_f32 on fragments read from LDS tiles (row stride hd + 2: conflict-free A/B reads),
// keeps P, the running sums A1 = sum_c dP_c*(dS_c - m_c), QK, OL in accumulator-layout registers for the
// whole lane loop, and turns dP (accumulator layout: column per lane) into an A operand (row per lane) through
// a 16 x keys scratch tile per wave.  float32 only (the float64 parity build keeps the scalar kernel: its
// tiles would not fit the LDS), head_dim a multiple of 16 up to 64, at most 64 queries and 64 keys.
#include "common.h"
#include "kernels.h"

namespace dqmc {

namespace {
constexpr int MAXC = 4;   // 16-wide tiles along keys and along head_dim
typedef Mfma<float>::acc_t acc_t;

__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes sharing lane >> 4
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64)); v = fmaxf(v, __shfl_xor(v, 8, 64));
  return v;
}
}  // namespace

__global__ void __launch_bounds__(256) k_attention_mfma(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, float* __restrict__ out,
                                                        int width, int H, int hd, LaneInfo li, int n_const,
                                                        const float* __restrict__ k_const,
                                                        const float* __restrict__ v_const) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  float* sm = reinterpret_cast<float*>(smem_raw);
  const int N = li.N, T = li.T, TP = li.TP;
  const int M = n_const + N;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int S = hd + 2;                               // row stride of the [rows][hd] tiles
  const int n_cb = (M + 15) / 16, n_db = hd / 16;     // key tiles, head_dim tiles
  const int n_rb = (N + 15) / 16;                     // query row blocks = active waves
  const int M16 = n_cb * 16, N16 = n_rb * 16;
  const int SA = M16 + 2;                             // row stride of the per-wave [16][keys] A-operand tiles
  float* q0 = sm;               float* k0 = q0 + N16 * S;   float* v0 = k0 + M16 * S;
  float* qc = v0 + M16 * S;     float* kc = qc + N16 * S;   float* vc = kc + M16 * S;
  float* PA = vc + M16 * S;                           // [4 waves][16][SA]  P as A operand
  float* DA = PA + 4 * 16 * SA;                       // [4 waves][16][SA]  dP_t as A operand
  const float sc = (float)(1.0 / sqrt((double)hd));
  const long row0 = (long)b * N * TP;
  const int col0 = h * hd;
  const bool active = wave < n_rb;
  const int i_base = wave * 16;                       // first query of this wave

  // Tile loads: thread (r_in, v4) moves one float4 per pass of rpp rows; the (<= MAXP) loads of a lane's q, k and
  // v tiles are all issued before any is used, and the next lane's are in flight while the current one is
  // multiplied (registers, not LDS, are the second buffer).
  constexpr int MAXP = 4;
  const int vpr = hd / 4;                              // float4 per row
  const int rpp = 256 / vpr;                           // rows per pass
  const int r_in = tid / vpr, v4 = tid - r_in * vpr;
  const bool ld_thread = r_in < rpp;
  const float* qg = q + row0 * width + col0 + 4 * v4;
  const float* kg = k + row0 * width + col0 + 4 * v4;
  const float* vg = v + row0 * width + col0 + 4 * v4;
  Vec4<float> Rq[MAXP], Rk[MAXP], Rv[MAXP];
  auto issue = [&](int t) {
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int row = p * rpp + r_in;
      Rq[p] = Vec4<float>{{0.f, 0.f, 0.f, 0.f}};
      Rk[p] = Vec4<float>{{0.f, 0.f, 0.f, 0.f}};
      Rv[p] = Vec4<float>{{0.f, 0.f, 0.f, 0.f}};
      if (!ld_thread) continue;
      if (row < N) Rq[p] = *reinterpret_cast<const Vec4<float>*>(qg + ((long)row * TP + t) * width);
      if (row < n_const) {
        if (t == 0) {
          Rk[p] = *reinterpret_cast<const Vec4<float>*>(k_const + (long)row * (H * hd) + col0 + 4 * v4);
          Rv[p] = *reinterpret_cast<const Vec4<float>*>(v_const + (long)row * (H * hd) + col0 + 4 * v4);
        }
      } else if (row < M) {
        const long off = ((long)(row - n_const) * TP + t) * width;
        Rk[p] = *reinterpret_cast<const Vec4<float>*>(kg + off);
        Rv[p] = *reinterpret_cast<const Vec4<float>*>(vg + off);
      }
    }
  };
  auto put = [&](float* dq, float* dk, float* dv) {   // registers -> LDS tiles (rows beyond N / M are zero)
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int row = p * rpp + r_in;
      if (!ld_thread) continue;
      if (row < N16) {
        Vec2<float>* d2 = reinterpret_cast<Vec2<float>*>(dq + row * S + 4 * v4);
        d2[0] = Vec2<float>{{Rq[p].v[0], Rq[p].v[1]}}; d2[1] = Vec2<float>{{Rq[p].v[2], Rq[p].v[3]}};
      }
      if (row < M16) {
        Vec2<float>* k2 = reinterpret_cast<Vec2<float>*>(dk + row * S + 4 * v4);
        k2[0] = Vec2<float>{{Rk[p].v[0], Rk[p].v[1]}}; k2[1] = Vec2<float>{{Rk[p].v[2], Rk[p].v[3]}};
        Vec2<float>* v2 = reinterpret_cast<Vec2<float>*>(dv + row * S + 4 * v4);
        v2[0] = Vec2<float>{{Rv[p].v[0], Rv[p].v[1]}}; v2[1] = Vec2<float>{{Rv[p].v[2], Rv[p].v[3]}};
      }
    }
  };
  // rows of this lane in accumulator layout and their validity
  int irow[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) irow[rg] = i_base + Mfma<float>::row_of(lane, rg);

  issue(0);
  put(q0, k0, v0);
  if (T > 1) issue(1);
  __syncthreads();

  acc_t P[MAXC], A1[MAXC], QK[MAXC], OL[MAXC];
  float A2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < MAXC; ++c) { P[c] = acc_t{0, 0, 0, 0}; A1[c] = acc_t{0, 0, 0, 0}; QK[c] = acc_t{0, 0, 0, 0}; OL[c] = acc_t{0, 0, 0, 0}; }
  float* myPA = PA + wave * 16 * SA;
  float* myDA = DA + wave * 16 * SA;

  // ---- value lane: P = softmax(q0 k0^T / sqrt(hd)) over the M keys, out_0 = P v0 ----
  if (active) {
    acc_t s0[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) s0[c] = acc_t{0, 0, 0, 0};
    for (int kk = 0; kk < hd / 4; ++kk) {
      const float a0 = q0[(i_base + l15) * S + kk * 4 + l4];
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_cb) s0[c] = Mfma<float>::run(a0, k0[(c * 16 + l15) * S + kk * 4 + l4], s0[c]);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float mx = -INFINITY;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_cb && c * 16 + l15 < M) mx = fmaxf(mx, s0[c][rg] * sc);
      mx = row16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        float e = 0.f;
        if (c < n_cb && c * 16 + l15 < M) e = expf(s0[c][rg] * sc - mx);
        P[c][rg] = e;
        sum += e;
      }
      sum = row16_sum(sum);
      const float inv = 1.f / sum;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) P[c][rg] *= inv;
    }
    // P as an A operand (row = query, k = key): through the wave's scratch tile
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < n_cb)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) myPA[Mfma<float>::row_of(lane, rg) * SA + c * 16 + l15] = P[c][rg];
  }
  __syncthreads();
  if (active) {
    acc_t o[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) o[c] = acc_t{0, 0, 0, 0};
    for (int kk = 0; kk < M16 / 4; ++kk) {
      const float ap = myPA[l15 * SA + kk * 4 + l4];
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_db) o[c] = Mfma<float>::run(ap, v0[(kk * 4 + l4) * S + c * 16 + l15], o[c]);
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
    __syncthreads();                                  // previous lane's tiles are no longer read
    put(qc, kc, vc);
    __syncthreads();
    if (t + 1 < T) issue(t + 1);                      // in flight while this lane is multiplied
    if (!active) continue;
    // S-phase
    acc_t ds[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) ds[c] = acc_t{0, 0, 0, 0};
    for (int kk = 0; kk < hd / 4; ++kk) {
      const int ko = kk * 4 + l4;
      const float a0 = q0[(i_base + l15) * S + ko], at = qc[(i_base + l15) * S + ko];
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_cb) {
          const float b0 = k0[(c * 16 + l15) * S + ko], bt = kc[(c * 16 + l15) * S + ko];
          ds[c] = Mfma<float>::run(at, b0, ds[c]);
          ds[c] = Mfma<float>::run(a0, bt, ds[c]);
          if (!lap) QK[c] = Mfma<float>::run(at, bt, QK[c]);
        }
    }
    // softmax algebra on the wave's 16 x M block (accumulator layout), row sums by 16-lane shuffles
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float m = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_cb) {
          ds[c][rg] = lap ? (ds[c][rg] + 2.f * QK[c][rg]) * sc : ds[c][rg] * sc;     // dS_c  or  L_S
          m += P[c][rg] * ds[c][rg];
        }
      m = row16_sum(m);                                // rowsum(P*dS_c)  or  rowsum(P*L_S)
      float s2 = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_cb) {
          float dp;
          if (!lap) {
            const float cc = ds[c][rg] - m;
            dp = P[c][rg] * cc;
            A1[c][rg] += dp * cc;
            s2 += dp * ds[c][rg];
          } else {
            dp = A1[c][rg] + P[c][rg] * (ds[c][rg] - m - A2[rg]);                     // L_P
          }
          myDA[Mfma<float>::row_of(lane, rg) * SA + c * 16 + l15] = dp;
        }
      if (!lap) A2[rg] += row16_sum(s2);
    }
    wave_lds_fence();
    // O-phase (the wave reads back only what it wrote itself: LDS operations of a wave complete in order)
    acc_t o[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) o[c] = lap ? OL[c] : acc_t{0, 0, 0, 0};
    for (int kk = 0; kk < M16 / 4; ++kk) {
      const int jo = kk * 4 + l4;
      const float adp = myDA[l15 * SA + jo], ap = myPA[l15 * SA + jo];
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < n_db) {
          const float b0 = v0[jo * S + c * 16 + l15], bt = vc[jo * S + c * 16 + l15];
          o[c] = Mfma<float>::run(adp, b0, o[c]);
          o[c] = Mfma<float>::run(ap, bt, o[c]);
          if (!lap) OL[c] = Mfma<float>::run(adp, bt + bt, OL[c]);                     // 2 sum_c dP_c v_c
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
      out[(row0 + (long)i * TP + t) * width + col0 + d] = 0.f;
    }
}

size_t attention_mfma_lds_bytes(int N, int hd, int n_const) {
  const size_t M16 = ((size_t)N + n_const + 15) / 16 * 16, N16 = ((size_t)N + 15) / 16 * 16;
  return sizeof(float) * ((2 * N16 + 4 * M16) * (hd + 2) + 2 * 4 * 16 * (M16 + 2));
}

bool attention_mfma_supported(int N, int hd, int n_const) {
  return hd % 16 == 0 && hd <= 64 && N <= 64 && N + n_const <= 64 && attention_mfma_lds_bytes(N, hd, n_const) <= 160 * 1024;
}
// Measured on MI355X (4 heads x 64, attention time per VMC step, scalar kernel -> this one): 42 electrons
// 135 -> 31 ms, 28 electrons 66 -> 49 ms, 14 electrons 29 -> 67 ms, 4 electrons 10 -> 75 ms: with a single query
// row block three of the four waves idle through the per-lane tile traffic, so the scalar kernel keeps the
// small systems.
bool attention_mfma_profitable(int N) { return N > 16; }

int launch_attention_mfma(hipStream_t st, const float* q, const float* k, const float* v, float* out, int width, int H,
                          int hd, int B, LaneInfo li, int n_const, const float* k_const, const float* v_const) {
  const size_t lds = attention_mfma_lds_bytes(li.N, hd, n_const);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return -2;
  hipLaunchKernelGGL(k_attention_mfma, dim3((unsigned)(B * H)), dim3(256), lds, st, q, k, v, out, width, H, hd, li,
                     n_const, k_const, v_const);
  return 0;
}

}  // namespace dqmc

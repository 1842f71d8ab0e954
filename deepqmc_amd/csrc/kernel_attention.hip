// kernel_attention.hip -- multi-head self attention over the electrons of one walker in
// forward-Laplacian form (Psiformer layers: hk.MultiHeadAttention called at reference
// gnn/update_features.py:273-278; propagation rules of SURVEY.md appendix C, the dense analogue
// of the reference's folxext.py:71-172).
//
// One workgroup per (walker, head).  The value-lane tiles q0, k0, v0 and P = softmax(q0 k0^T/sqrt(hd))
// stay in LDS; the 3N derivative lanes are streamed through one LDS tile set, each contributing
//   dS_c = (q_c k0^T + q0 k_c^T)/sqrt(hd),  m_c = rowsum(P * dS_c),  dP_c = P * (dS_c - m_c),
//   out_c = dP_c v0 + P v_c
// and accumulating what the Laplacian lane needs:  sum_c dP_c*(dS_c - m_c),  sum_c rowsum(dP_c*dS_c),
// sum_c q_c k_c^T,  sum_c dP_c v_c.  Activation layout as everywhere: X[(b*N + i)*TP + t][H*hd].
//
// Nuclear tokens (TransPsiformer, update_features.py:385-451 with elec_to_nuc = false): the electron
// queries additionally attend to n_const key / value rows that do not depend on the electrons (folded on
// the host, deepqmc_amd/nuclear_stream.py).  They come first in the key order -- as in the reference's
// concatenate([nuclei, electrons]) -- and their derivative lanes are zero.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dqmc {

template <typename real>
__global__ void __launch_bounds__(256) k_attention(const real* __restrict__ q, const real* __restrict__ k,
                                                   const real* __restrict__ v, real* __restrict__ out, int width,
                                                   int H, int hd, LaneInfo li, int n_const,
                                                   const real* __restrict__ k_const, const real* __restrict__ v_const,
                                                   int NQ) {
  // blockIdx.y selects a block of NQ queries (the rows of the attention matrix are independent given all keys): the
  // query-indexed tiles shrink accordingly, which is what lets 42 electrons x 64 features fit the LDS in float64
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* sm = reinterpret_cast<real*>(smem_raw);
  const int NE = li.N, T = li.T, TP = li.TP;           // NE electrons = keys; N = queries of this workgroup
  const int i_lo = blockIdx.y * NQ;
  const int N = (NE - i_lo) < NQ ? (NE - i_lo) : NQ;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int S = hd + 1;              // padded row stride of the [rows][hd] tiles
  const int M = n_const + NE;        // keys: constant rows first, then the electrons
  const int NN = N * M, NH = N * S, MH = M * S;
  real* q0 = sm;            real* k0 = q0 + NH;       real* v0 = k0 + MH;
  real* qc = v0 + MH;       real* kc = qc + NH;       real* vc = kc + MH;
  real* OL = vc + MH;                                 // [N][S]   2 sum_c dP_c v_c
  real* P = OL + NH;        real* dS = P + NN;        real* dP = dS + NN;
  real* A1 = dP + NN;       real* QK = A1 + NN;       // [N][M] accumulators
  real* mrow = QK + NN;     real* A2 = mrow + N;      // [N]
  const real sc = (real)(1.0 / sqrt((double)hd));
  const long row0 = (long)b * NE * TP;                // row of (b, i=0, t=0)
  const long qrow0 = row0 + (long)i_lo * TP;          // row of this workgroup's first query
  const int col0 = h * hd;

  auto load_tile = [&](const real* src, int t, real* dst) {
    for (int e = tid; e < N * hd; e += nthr) {
      const int i = e / hd, d = e - i * hd;
      dst[i * S + d] = src[(qrow0 + (long)i * TP + t) * width + col0 + d];
    }
  };
  auto load_kv = [&](const real* src, const real* cst, int t, real* dst) {   // [M][hd]: constants, then electrons
    for (int e = tid; e < M * hd; e += nthr) {
      const int j = e / hd, d = e - j * hd;
      real x;
      if (j < n_const) x = t == 0 ? cst[(long)j * (H * hd) + col0 + d] : (real)0;
      else x = src[(row0 + (long)(j - n_const) * TP + t) * width + col0 + d];
      dst[j * S + d] = x;
    }
  };
  load_tile(q, 0, q0); load_kv(k, k_const, 0, k0); load_kv(v, v_const, 0, v0);
  for (int e = tid; e < NN; e += nthr) { A1[e] = 0; QK[e] = 0; }
  for (int e = tid; e < NH; e += nthr) OL[e] = 0;
  if (tid < N) A2[tid] = 0;
  __syncthreads();
  // S0 and the softmax over keys
  for (int e = tid; e < NN; e += nthr) {
    const int i = e / M, j = e - i * M;
    real s = 0;
    for (int d = 0; d < hd; ++d) s += q0[i * S + d] * k0[j * S + d];
    dS[e] = s * sc;
  }
  __syncthreads();
  if (tid < N) {
    real mx = dS[tid * M];
    for (int j = 1; j < M; ++j) mx = dS[tid * M + j] > mx ? dS[tid * M + j] : mx;
    real sum = 0;
    for (int j = 0; j < M; ++j) { const real ev = r_exp<real>(dS[tid * M + j] - mx); P[tid * M + j] = ev; sum += ev; }
    const real inv = 1 / sum;
    for (int j = 0; j < M; ++j) P[tid * M + j] *= inv;
  }
  __syncthreads();
  // value lane: out_0 = P v0
  for (int e = tid; e < N * hd; e += nthr) {
    const int i = e / hd, d = e - i * hd;
    real o = 0;
    for (int j = 0; j < M; ++j) o += P[i * M + j] * v0[j * S + d];
    out[(qrow0 + (long)i * TP) * width + col0 + d] = o;
  }
  if (T == 1) return;

  for (int t = 1; t < T; ++t) {       // derivative lanes, then the Laplacian lane (t = T-1)
    const bool lap = t == T - 1;
    __syncthreads();
    load_tile(q, t, qc); load_kv(k, k_const, t, kc); load_kv(v, v_const, t, vc);
    __syncthreads();
    // register tiles of TI x TJ logits per thread: the q / k rows of a tile are read once per d and feed TI * TJ * 3
    // multiply-adds (one logit per thread meant 4 LDS reads per 3 multiply-adds: the loop ran at the LDS read rate)
    {
      constexpr int TI = 2, TJ = 4;
      const int nti = (N + TI - 1) / TI, ntj = (M + TJ - 1) / TJ;
      for (int e = tid; e < nti * ntj; e += nthr) {
        const int ti = e / ntj, tj = e - ti * ntj;
        int io[TI], jo[TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a) { const int i = ti * TI + a; io[a] = (i < N ? i : N - 1) * S; }
#pragma unroll
        for (int c = 0; c < TJ; ++c) { const int j = tj * TJ + c; jo[c] = (j < M ? j : M - 1) * S; }
        real s[TI][TJ], qk[TI][TJ];
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
          for (int c = 0; c < TJ; ++c) { s[a][c] = 0; qk[a][c] = 0; }
        for (int d = 0; d < hd; ++d) {
          real qa[TI], q0a[TI], kb[TJ], k0b[TJ];
#pragma unroll
          for (int a = 0; a < TI; ++a) { qa[a] = qc[io[a] + d]; q0a[a] = q0[io[a] + d]; }
#pragma unroll
          for (int c = 0; c < TJ; ++c) { kb[c] = kc[jo[c] + d]; k0b[c] = k0[jo[c] + d]; }
#pragma unroll
          for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int c = 0; c < TJ; ++c) {
              s[a][c] += qa[a] * k0b[c] + q0a[a] * kb[c];
              qk[a][c] += qa[a] * kb[c];
            }
        }
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
          for (int c = 0; c < TJ; ++c) {
            const int i = ti * TI + a, j = tj * TJ + c;
            if (i < N && j < M) {
              const int o = i * M + j;
              if (!lap) { dS[o] = s[a][c] * sc; QK[o] += qk[a][c]; }
              else dS[o] = (s[a][c] + 2 * QK[o]) * sc;             // L_S
            }
          }
      }
    }
    __syncthreads();
    if (tid < N) {
      real m = 0;
      for (int j = 0; j < M; ++j) m += P[tid * M + j] * dS[tid * M + j];
      mrow[tid] = m;                                   // rowsum(P*dS_c)  or  rowsum(P*L_S)
    }
    __syncthreads();
    if (!lap) {
      for (int e = tid; e < NN; e += nthr) {
        const int i = e / M;
        const real c = dS[e] - mrow[i];
        const real p = P[e] * c;
        dP[e] = p;
        A1[e] += p * c;
      }
      __syncthreads();
      if (tid < N) {
        real s2 = 0;
        for (int j = 0; j < M; ++j) s2 += dP[tid * M + j] * dS[tid * M + j];
        A2[tid] += s2;
      }
      {     // TI queries x TD features per thread
        constexpr int TI = 2, TD = 4;
        const int nti = (N + TI - 1) / TI, ntd = hd / TD;
        for (int e = tid; e < nti * ntd; e += nthr) {
          const int ti = e / ntd, d0 = (e - ti * ntd) * TD;
          int io[TI];
#pragma unroll
          for (int a = 0; a < TI; ++a) { const int i = ti * TI + a; io[a] = (i < N ? i : N - 1) * M; }
          real o[TI][TD], ol[TI][TD];
#pragma unroll
          for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int c = 0; c < TD; ++c) { o[a][c] = 0; ol[a][c] = 0; }
          for (int j = 0; j < M; ++j) {
            real dp[TI], pp[TI], v0d[TD], vcd[TD];
#pragma unroll
            for (int a = 0; a < TI; ++a) { dp[a] = dP[io[a] + j]; pp[a] = P[io[a] + j]; }
#pragma unroll
            for (int c = 0; c < TD; ++c) { v0d[c] = v0[j * S + d0 + c]; vcd[c] = vc[j * S + d0 + c]; }
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
              for (int c = 0; c < TD; ++c) {
                o[a][c] += dp[a] * v0d[c] + pp[a] * vcd[c];
                ol[a][c] += dp[a] * vcd[c];
              }
          }
#pragma unroll
          for (int a = 0; a < TI; ++a) {
            const int i = ti * TI + a;
            if (i >= N) continue;
#pragma unroll
            for (int c = 0; c < TD; ++c) {
              OL[i * S + d0 + c] += 2 * ol[a][c];
              out[(qrow0 + (long)i * TP + t) * width + col0 + d0 + c] = o[a][c];
            }
          }
        }
      }
    } else {
      // L_P = A1 + P*(L_S - rowsum(P*L_S) - A2);  out_L = L_P v0 + P v_L + 2 sum_c dP_c v_c
      for (int e = tid; e < NN; e += nthr) {
        const int i = e / M;
        dP[e] = A1[e] + P[e] * (dS[e] - mrow[i] - A2[i]);
      }
      __syncthreads();
      for (int e = tid; e < N * hd; e += nthr) {
        const int i = e / hd, d = e - i * hd;
        real o = OL[i * S + d];
        for (int j = 0; j < M; ++j) o += dP[i * M + j] * v0[j * S + d] + P[i * M + j] * vc[j * S + d];
        out[(qrow0 + (long)i * TP + t) * width + col0 + d] = o;
      }
    }
  }
  for (int t = T; t < TP; ++t)
    for (int e = tid; e < N * hd; e += nthr) {
      const int i = e / hd, d = e - i * hd;
      out[(qrow0 + (long)i * TP + t) * width + col0 + d] = 0;
    }
}

// LDS bytes with the queries split into blocks of NQ (NQ = N: one workgroup per (walker, head))
template <typename real> static size_t attention_lds_split(int N, int hd, int n_const, int NQ) {
  const size_t M = (size_t)N + n_const;
  return sizeof(real) * (((size_t)3 * NQ + 4 * M) * (hd + 1) + (size_t)5 * NQ * M + 2 * NQ);
}
// smallest number of query blocks whose tile set fits the 160 KiB LDS (0: none does)
template <typename real> static int attention_query_blocks(int N, int hd, int n_const) {
  for (int qs = 1; qs <= N; ++qs) {
    const int NQ = (N + qs - 1) / qs;
    if (attention_lds_split<real>(N, hd, n_const, NQ) <= (size_t)160 * 1024) return qs;
  }
  return 0;
}
template <typename real> size_t attention_lds_bytes(int N, int hd, int n_const) {
  const int qs = attention_query_blocks<real>(N, hd, n_const);
  if (qs == 0) return (size_t)1 << 30;
  return attention_lds_split<real>(N, hd, n_const, (N + qs - 1) / qs);
}

template <typename real>
int launch_attention(hipStream_t st, const real* q, const real* k, const real* v, real* out, int width, int H, int hd,
                     int B, LaneInfo li, int n_const, const real* k_const, const real* v_const) {
  int qs = attention_query_blocks<real>(li.N, hd, n_const);
  if (qs == 0) return -1;
  if (const char* f = getenv("DQMC_ATTN_QSPLIT")) {      // test hook: force at least this many query blocks
    const int want = atoi(f);
    if (want > qs && want <= li.N) qs = want;
  }
  const int NQ = (li.N + qs - 1) / qs;
  const size_t lds = attention_lds_split<real>(li.N, hd, n_const, NQ);
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention<real>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return -2;
  // few electrons: one wave per (walker, head) instead of four mostly idle ones (measured: N = 4 10.4 -> 7.0 ms
  // per step; N = 14 is faster with four waves)
  const unsigned nthr = li.N <= 8 ? 64u : 256u;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention<real>), dim3((unsigned)(B * H), (unsigned)qs), dim3(nthr), lds, st, q, k, v, out,
                     width, H, hd, li, n_const, k_const, v_const, NQ);
  return 0;
}

template size_t attention_lds_bytes<float>(int, int, int);
template size_t attention_lds_bytes<double>(int, int, int);
template int launch_attention<float>(hipStream_t, const float*, const float*, const float*, float*, int, int, int, int,
                                     LaneInfo, int, const float*, const float*);
template int launch_attention<double>(hipStream_t, const double*, const double*, const double*, double*, int, int,
                                      int, int, LaneInfo, int, const double*, const double*);

}  // namespace dqmc

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
#include "common.h"
#include "kernels.h"

namespace dqmc {

template <typename real>
__global__ void __launch_bounds__(256) k_attention(const real* __restrict__ q, const real* __restrict__ k,
                                                   const real* __restrict__ v, real* __restrict__ out, int width,
                                                   int H, int hd, LaneInfo li) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* sm = reinterpret_cast<real*>(smem_raw);
  const int N = li.N, T = li.T, TP = li.TP;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int S = hd + 1;              // padded row stride of the [N][hd] tiles
  const int NN = N * N, NH = N * S;
  real* q0 = sm;            real* k0 = q0 + NH;       real* v0 = k0 + NH;
  real* qc = v0 + NH;       real* kc = qc + NH;       real* vc = kc + NH;
  real* OL = vc + NH;                                 // [N][S]   2 sum_c dP_c v_c
  real* P = OL + NH;        real* dS = P + NN;        real* dP = dS + NN;
  real* A1 = dP + NN;       real* QK = A1 + NN;       // [N][N] accumulators
  real* mrow = QK + NN;     real* A2 = mrow + N;      // [N]
  const real sc = (real)(1.0 / sqrt((double)hd));
  const long row0 = (long)b * N * TP;                 // row of (b, i=0, t=0)
  const int col0 = h * hd;

  auto load_tile = [&](const real* src, int t, real* dst) {
    for (int e = tid; e < N * hd; e += nthr) {
      const int i = e / hd, d = e - i * hd;
      dst[i * S + d] = src[(row0 + (long)i * TP + t) * width + col0 + d];
    }
  };
  load_tile(q, 0, q0); load_tile(k, 0, k0); load_tile(v, 0, v0);
  for (int e = tid; e < NN; e += nthr) { A1[e] = 0; QK[e] = 0; }
  for (int e = tid; e < NH; e += nthr) OL[e] = 0;
  if (tid < N) A2[tid] = 0;
  __syncthreads();
  // S0 and the softmax over keys
  for (int e = tid; e < NN; e += nthr) {
    const int i = e / N, j = e - i * N;
    real s = 0;
    for (int d = 0; d < hd; ++d) s += q0[i * S + d] * k0[j * S + d];
    dS[e] = s * sc;
  }
  __syncthreads();
  if (tid < N) {
    real mx = dS[tid * N];
    for (int j = 1; j < N; ++j) mx = dS[tid * N + j] > mx ? dS[tid * N + j] : mx;
    real sum = 0;
    for (int j = 0; j < N; ++j) { const real ev = r_exp<real>(dS[tid * N + j] - mx); P[tid * N + j] = ev; sum += ev; }
    const real inv = 1 / sum;
    for (int j = 0; j < N; ++j) P[tid * N + j] *= inv;
  }
  __syncthreads();
  // value lane: out_0 = P v0
  for (int e = tid; e < N * hd; e += nthr) {
    const int i = e / hd, d = e - i * hd;
    real o = 0;
    for (int j = 0; j < N; ++j) o += P[i * N + j] * v0[j * S + d];
    out[(row0 + (long)i * TP) * width + col0 + d] = o;
  }
  if (T == 1) return;

  for (int t = 1; t < T; ++t) {       // derivative lanes, then the Laplacian lane (t = T-1)
    const bool lap = t == T - 1;
    __syncthreads();
    load_tile(q, t, qc); load_tile(k, t, kc); load_tile(v, t, vc);
    __syncthreads();
    for (int e = tid; e < NN; e += nthr) {
      const int i = e / N, j = e - i * N;
      real s = 0, qk = 0;
      for (int d = 0; d < hd; ++d) {
        s += qc[i * S + d] * k0[j * S + d] + q0[i * S + d] * kc[j * S + d];
        qk += qc[i * S + d] * kc[j * S + d];
      }
      if (!lap) { dS[e] = s * sc; QK[e] += qk; }
      else dS[e] = (s + 2 * QK[e]) * sc;             // L_S
    }
    __syncthreads();
    if (tid < N) {
      real m = 0;
      for (int j = 0; j < N; ++j) m += P[tid * N + j] * dS[tid * N + j];
      mrow[tid] = m;                                   // rowsum(P*dS_c)  or  rowsum(P*L_S)
    }
    __syncthreads();
    if (!lap) {
      for (int e = tid; e < NN; e += nthr) {
        const int i = e / N;
        const real c = dS[e] - mrow[i];
        const real p = P[e] * c;
        dP[e] = p;
        A1[e] += p * c;
      }
      __syncthreads();
      if (tid < N) {
        real s2 = 0;
        for (int j = 0; j < N; ++j) s2 += dP[tid * N + j] * dS[tid * N + j];
        A2[tid] += s2;
      }
      for (int e = tid; e < N * hd; e += nthr) {
        const int i = e / hd, d = e - i * hd;
        real o = 0, ol = 0;
        for (int j = 0; j < N; ++j) {
          const real dp = dP[i * N + j];
          o += dp * v0[j * S + d] + P[i * N + j] * vc[j * S + d];
          ol += dp * vc[j * S + d];
        }
        OL[i * S + d] += 2 * ol;
        out[(row0 + (long)i * TP + t) * width + col0 + d] = o;
      }
    } else {
      // L_P = A1 + P*(L_S - rowsum(P*L_S) - A2);  out_L = L_P v0 + P v_L + 2 sum_c dP_c v_c
      for (int e = tid; e < NN; e += nthr) {
        const int i = e / N;
        dP[e] = A1[e] + P[e] * (dS[e] - mrow[i] - A2[i]);
      }
      __syncthreads();
      for (int e = tid; e < N * hd; e += nthr) {
        const int i = e / hd, d = e - i * hd;
        real o = OL[i * S + d];
        for (int j = 0; j < N; ++j) o += dP[i * N + j] * v0[j * S + d] + P[i * N + j] * vc[j * S + d];
        out[(row0 + (long)i * TP + t) * width + col0 + d] = o;
      }
    }
  }
  for (int t = T; t < TP; ++t)
    for (int e = tid; e < N * hd; e += nthr) {
      const int i = e / hd, d = e - i * hd;
      out[(row0 + (long)i * TP + t) * width + col0 + d] = 0;
    }
}

template <typename real> size_t attention_lds_bytes(int N, int hd) {
  return sizeof(real) * ((size_t)7 * N * (hd + 1) + (size_t)5 * N * N + 2 * N);
}

template <typename real>
int launch_attention(hipStream_t st, const real* q, const real* k, const real* v, real* out, int width, int H, int hd,
                     int B, LaneInfo li) {
  const size_t lds = attention_lds_bytes<real>(li.N, hd);
  if (lds > 160 * 1024) return -1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention<real>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return -2;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_attention<real>), dim3((unsigned)(B * H)), dim3(256), lds, st, q, k, v, out,
                     width, H, hd, li);
  return 0;
}

template size_t attention_lds_bytes<float>(int, int);
template size_t attention_lds_bytes<double>(int, int);
template int launch_attention<float>(hipStream_t, const float*, const float*, const float*, float*, int, int, int, int,
                                     LaneInfo);
template int launch_attention<double>(hipStream_t, const double*, const double*, const double*, double*, int, int,
                                      int, int, LaneInfo);

}  // namespace dqmc

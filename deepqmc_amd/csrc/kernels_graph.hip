// kernels_graph.hip -- input features and the small structured ops of the electron GNN,
// all in forward-Laplacian (lane-stacked) form.  One thread per output element
// (walker, row, lane, column) with the column fastest => coalesced row-major traffic.
#include "common.h"
#include "kernels.h"

namespace dqmc {

// Electron-nucleus input features, reference gnn/electron_gnn.py:596-625:
// x[b][i][t][4a + f] = lane t of [|d|, d_x, d_y, d_z], d = r_i - R_a (+ spin column).
template <typename real>
__global__ void __launch_bounds__(256) k_feat_en(const real* __restrict__ r, const real* __restrict__ R, real* __restrict__ x,
                                                 int B, int n_nuc, int n_up, int width, LaneInfo li, double eps,
                                                 int log_rescale, int use_spin, const double* __restrict__ phq) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * li.N * li.TP * n_nuc;
  if (idx >= total) return;
  const int a = (int)(idx % n_nuc);
  long q = idx / n_nuc;
  const int t = (int)(q % li.TP); q /= li.TP;
  const int i = (int)(q % li.N);
  const int b = (int)(q / li.N);
  double d[3];
  for (int k = 0; k < 3; ++k) d[k] = (double)r[((long)b * li.N + i) * 3 + k] - (double)R[a * 3 + k];
  double f[4];
  pair_feature_lane(d, eps, i, -1, t, li, log_rescale != 0, f, phq ? phq + ((long)b * li.N + i) * PH_STRIDE : nullptr);
  real* row = x + (((long)b * li.N + i) * li.TP + t) * width;
  for (int k = 0; k < 4; ++k) row[4 * a + k] = (real)f[k];
  if (a == 0) {  // spin column and zero padding of the row tail
    int c = 4 * n_nuc;
    if (use_spin) row[c++] = (t == 0) ? (real)(i < n_up ? 1.0 : -1.0) : (real)0;
    for (; c < width; ++c) row[c] = (real)0;
  }
}

// Electron-electron edge features, reference gnn/graph.py:23-31 (d = r_recv - r_send).
// A negative sender s names nucleus -1 - s: d = r_recv - R (the 'ne' edges of the convolution, gnn/graph.py:100-121).
template <typename real>
__global__ void __launch_bounds__(256) k_feat_ee(const real* __restrict__ r, const real* __restrict__ R,
                                                 const int32_t* __restrict__ pairs,
                                                 real* __restrict__ e, int B, int n_rows, LaneInfo li, double eps,
                                                 int log_rescale, int compact, const double* __restrict__ phq) {
  // compact: the destination carries the 8 pair lanes of common.h (Laplacian mode only)
  const int TPd = compact ? PAIR_LANES : li.TP;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * n_rows * TPd;
  if (idx >= total) return;
  const int td = (int)(idx % TPd);
  long q = idx / TPd;
  const int k = (int)(q % n_rows);
  const int b = (int)(q / n_rows);
  const int rc = pairs[2 * k], sd = pairs[2 * k + 1];
  const int t = compact ? pair_lane_full(td, li.T, rc, sd) : td;
  Vec4<real> o;
  if (t < 0) {
    for (int c = 0; c < 4; ++c) o.v[c] = (real)0;
  } else {
    double d[3];
    for (int c = 0; c < 3; ++c)
      d[c] = (double)r[((long)b * li.N + rc) * 3 + c] - (sd >= 0 ? (double)r[((long)b * li.N + sd) * 3 + c] : (double)R[(-1 - sd) * 3 + c]);
    double f[4];
    pair_feature_lane(d, eps, rc, sd, t, li, log_rescale != 0, f, phq ? phq + ((long)b * li.N + rc) * PH_STRIDE : nullptr,
                      (phq && sd >= 0) ? phq + ((long)b * li.N + sd) * PH_STRIDE : nullptr);
    for (int c = 0; c < 4; ++c) o.v[c] = (real)f[c];
  }
  *reinterpret_cast<Vec4<real>*>(e + idx * 4) = o;
}

// Mean over spin-up / spin-down electrons, reference gnn/update_features.py:86-102.
template <typename real>
__global__ void __launch_bounds__(256) k_spin_mean(const real* __restrict__ x, real* __restrict__ m, int B, int n_up,
                                                   int width, LaneInfo li) {
  // four consecutive features per thread (widths are multiples of 4): 16-byte loads and stores
  const int w4 = width >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * 2 * li.TP * w4;
  if (idx >= total) return;
  const int c = (int)(idx % w4) << 2;
  long q = idx / w4;
  const int t = (int)(q % li.TP); q /= li.TP;
  const int which = (int)(q % 2);
  const int b = (int)(q / 2);
  const int i0 = which ? n_up : 0, i1 = which ? li.N : n_up;
  Vec4<real> acc{{0, 0, 0, 0}};
  for (int i = i0; i < i1; ++i) {
    const Vec4<real> v = *reinterpret_cast<const Vec4<real>*>(x + (((long)b * li.N + i) * li.TP + t) * width + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc.v[j] += v.v[j];
  }
  const real inv = (i1 > i0) ? (real)1 / (real)(i1 - i0) : (real)0;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc.v[j] = (i1 > i0) ? acc.v[j] / (real)(i1 - i0) : (real)0;
  (void)inv;
  *reinterpret_cast<Vec4<real>*>(m + (((long)b * 2 + which) * li.TP + t) * width + c) = acc;
}

// Sum over all rows of a walker (Jastrow sum_first, wf/omni.py:35-40).
template <typename real>
__global__ void __launch_bounds__(256) k_row_sum(const real* __restrict__ x, real* __restrict__ s, int B, int rows,
                                                 int width, LaneInfo li) {
  const int w4 = width >> 2;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * li.TP * w4;
  if (idx >= total) return;
  const int c = (int)(idx % w4) << 2;
  long q = idx / w4;
  const int t = (int)(q % li.TP);
  const int b = (int)(q / li.TP);
  Vec4<real> acc{{0, 0, 0, 0}};
  for (int i = 0; i < rows; ++i) {
    const Vec4<real> v = *reinterpret_cast<const Vec4<real>*>(x + (((long)b * rows + i) * li.TP + t) * width + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc.v[j] += v.v[j];
  }
  *reinterpret_cast<Vec4<real>*>(s + ((long)b * li.TP + t) * width + c) = acc;
}

// Convolution feature, reference gnn/graph.py:226-335: out[i] = sum_s we[row(i,s)] * hx[send(i,s)]
// with the product rule across lanes (value, d/dr_c, Laplacian).  tab: int [N][S][2].
template <typename real, int VW>
__global__ void __launch_bounds__(256) k_conv(const real* __restrict__ we, int we_rows, int we_width,
                                              const real* __restrict__ hx, int hx_rows, int hx_width, real* __restrict__ out,
                                              int out_width, int col0, const int32_t* __restrict__ tab, int S, int W,
                                              int B, LaneInfo li, int compact) {
  // `compact`: the edge operand carries the 8 pair lanes of common.h; lane t of edge (i, snd) is then lane
  // pair_lane(t) of its row, or zero.
  // One thread per (walker, receiver, VW consecutive columns) walks the lanes once: out_t = sum_s (a_t h_0 + a_0 h_t),
  // and the Laplacian lane adds 2 sum_s sum_c a_c h_c from a running dot product -- every input element is read once
  // and no thread carries the whole Laplacian sum alone.  VW = 4: 16-byte loads and stores (the 4-byte version issued
  // four times the memory instructions for the same bytes and ran at 2 TB/s).
  typedef VecN<real, VW> vec;
  const int Wv = W / VW;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * li.N * Wv;
  if (idx >= total) return;
  const int c = (int)(idx % Wv) * VW;
  const long q = idx / Wv;
  const int i = (int)(q % li.N);
  const int b = (int)(q / li.N);
  const int T = li.T;
  const int TPe = (compact && T > 1) ? PAIR_LANES : li.TP;
  real* o = out + (((long)b * li.N + i) * li.TP) * out_width + col0 + c;
  constexpr int SMAX = 4;       // up to SMAX senders: their rows and value lanes stay in registers (static indices only)
  if (S <= SMAX) {
    const real* ap[SMAX];
    const real* hp[SMAX];
    int sndv[SMAX];
    bool ok[SMAX];
    vec a0[SMAX], h0[SMAX];
    vec acc0 = vec::zero(), dot = vec::zero();
#pragma unroll
    for (int u = 0; u < SMAX; ++u) {
      const int row = u < S ? tab[2 * (i * S + u)] : -1, snd = u < S ? tab[2 * (i * S + u) + 1] : 0;
      ok[u] = row >= 0;
      sndv[u] = snd;
      const int hrow = snd >= 0 ? snd : -1 - snd;        // a negative sender is nucleus -1 - snd (row of a nuclear node buffer)
      ap[u] = we + (((long)b * we_rows + (ok[u] ? row : 0)) * TPe) * we_width + c;
      hp[u] = hx + (((long)b * hx_rows + (ok[u] ? hrow : 0)) * li.TP) * hx_width + c;
      a0[u] = ok[u] ? vec::load(ap[u]) : vec::zero();
      h0[u] = ok[u] ? vec::load(hp[u]) : vec::zero();
      acc0.fma(a0[u], h0[u]);
    }
    acc0.store(o);
    for (int t = 1; t < T; ++t) {
      vec acc = vec::zero();
#pragma unroll
      for (int u = 0; u < SMAX; ++u) {
        if (!ok[u]) continue;
        const int ta = (compact && T > 1) ? pair_lane(t, T, i, sndv[u]) : t;
        const vec at = ta >= 0 ? vec::load(ap[u] + (long)ta * we_width) : vec::zero(), ht = vec::load(hp[u] + (long)t * hx_width);
        acc.fma(at, h0[u]);
        acc.fma(a0[u], ht);
        if (t < T - 1) dot.fma(at, ht);
      }
      if (t == T - 1) acc.axpy((real)2, dot);
      acc.store(o + (long)t * out_width);
    }
  } else {
    vec acc0 = vec::zero(), dot = vec::zero();
    for (int s = 0; s < S; ++s) {
      const int row = tab[2 * (i * S + s)], snd = tab[2 * (i * S + s) + 1];
      if (row < 0) continue;
      const int hrow = snd >= 0 ? snd : -1 - snd;
      acc0.fma(vec::load(we + (((long)b * we_rows + row) * TPe) * we_width + c), vec::load(hx + (((long)b * hx_rows + hrow) * li.TP) * hx_width + c));
    }
    acc0.store(o);
    for (int t = 1; t < T; ++t) {
      vec acc = vec::zero();
      for (int s = 0; s < S; ++s) {
        const int row = tab[2 * (i * S + s)], snd = tab[2 * (i * S + s) + 1];
        if (row < 0) continue;
        const real* a = we + (((long)b * we_rows + row) * TPe) * we_width + c;
        const real* h = hx + (((long)b * hx_rows + (snd >= 0 ? snd : -1 - snd)) * li.TP) * hx_width + c;
        const int ta = (compact && T > 1) ? pair_lane(t, T, i, snd) : t;
        const vec at = ta >= 0 ? vec::load(a + (long)ta * we_width) : vec::zero(), ht = vec::load(h + (long)t * hx_width);
        acc.fma(at, vec::load(h));
        acc.fma(vec::load(a), ht);
        if (t < T - 1) dot.fma(at, ht);
      }
      if (t == T - 1) acc.axpy((real)2, dot);
      acc.store(o + (long)t * out_width);
    }
  }
  for (int t = T; t < li.TP; ++t) vec::zero().store(o + (long)t * out_width);
}

// Edge sum/mean feature, reference gnn/update_features.py:109-159 (linear in the lanes).
template <typename real>
__global__ void __launch_bounds__(256) k_edge_sum(const real* __restrict__ e, int e_rows, int e_width,
                                                  real* __restrict__ out, int out_width, int col0,
                                                  const int32_t* __restrict__ tab, int S, int W, real scale, int B,
                                                  LaneInfo li, int compact) {
  // four consecutive features per thread when the width allows (16-byte loads / stores, 4x fewer index computations)
  const bool v4 = (W & 3) == 0 && (col0 & 3) == 0;
  const int Wv = v4 ? W >> 2 : W;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * li.N * li.TP * Wv;
  if (idx >= total) return;
  const int c = (int)(idx % Wv) * (v4 ? 4 : 1);
  long q = idx / Wv;
  const int t = (int)(q % li.TP); q /= li.TP;
  const int i = (int)(q % li.N);
  const int b = (int)(q / li.N);
  const bool cp = compact && li.T > 1;
  const int TPe = cp ? PAIR_LANES : li.TP;
  real* o = out + (((long)b * li.N + i) * li.TP + t) * out_width + col0 + c;
  if (v4) {
    Vec4<real> acc{{0, 0, 0, 0}};
    if (t < li.T)
      for (int s = 0; s < S; ++s) {
        const int row = tab[2 * (i * S + s)], snd = tab[2 * (i * S + s) + 1];
        if (row < 0) continue;
        const int te = cp ? pair_lane(t, li.T, i, snd) : t;
        if (te >= 0) {
          const Vec4<real> v = *reinterpret_cast<const Vec4<real>*>(e + (((long)b * e_rows + row) * TPe + te) * e_width + c);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc.v[j] += v.v[j];
        }
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc.v[j] *= scale;
    *reinterpret_cast<Vec4<real>*>(o) = acc;
    return;
  }
  real acc = 0;
  if (t < li.T)
    for (int s = 0; s < S; ++s) {
      const int row = tab[2 * (i * S + s)], snd = tab[2 * (i * S + s) + 1];
      if (row < 0) continue;
      const int te = cp ? pair_lane(t, li.T, i, snd) : t;
      if (te >= 0) acc += e[(((long)b * e_rows + row) * TPe + te) * e_width + c];
    }
  o[0] = acc * scale;
}

// ---------------------------------------------------------------------------------------------
// launchers

static inline unsigned nblk(long total) { return (unsigned)((total + 255) / 256); }

template <typename real>
void launch_feat_en(hipStream_t st, const real* r, const real* R, real* x, int B, int n_nuc, int n_up, int width,
                    LaneInfo li, double eps, int log_rescale, int use_spin, const double* phq) {
  const long total = (long)B * li.N * li.TP * n_nuc;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_feat_en<real>), dim3(nblk(total)), dim3(256), 0, st, r, R, x, B, n_nuc, n_up,
                     width, li, eps, log_rescale, use_spin, phq);
}
template <typename real>
void launch_feat_ee(hipStream_t st, const real* r, const real* R, const int32_t* pairs, real* e, int B, int n_rows, LaneInfo li,
                    double eps, int log_rescale, int compact, const double* phq) {
  const long total = (long)B * n_rows * (compact ? PAIR_LANES : li.TP);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_feat_ee<real>), dim3(nblk(total)), dim3(256), 0, st, r, R, pairs, e, B, n_rows,
                     li, eps, log_rescale, compact, phq);
}
// Constant rows (learned embeddings that do not depend on the electron positions: hk.Embed of the electron /
// nuclear embeddings, gnn/electron_gnn.py:497-503,596-625 with positional_embeddings = false): value lane from the
// weight table, all derivative lanes zero.
template <typename real>
__global__ void __launch_bounds__(256) k_const_rows(const real* __restrict__ tab, real* __restrict__ x, int B, int rows, int width,
                                                    LaneInfo li) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * rows * li.TP * width;
  if (idx >= total) return;
  const int c = (int)(idx % width);
  long q = idx / width;
  const int t = (int)(q % li.TP); q /= li.TP;
  const int row = (int)(q % rows);
  x[idx] = t == 0 ? tab[(long)row * width + c] : (real)0;
}
template <typename real>
void launch_const_rows(hipStream_t st, const real* tab, real* x, int B, int rows, int width, LaneInfo li) {
  const long total = (long)B * rows * li.TP * width;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_const_rows<real>), dim3(nblk(total)), dim3(256), 0, st, tab, x, B, rows, width, li);
}
template <typename real>
void launch_spin_mean(hipStream_t st, const real* x, real* m, int B, int n_up, int width, LaneInfo li) {
  const long total = (long)B * 2 * li.TP * (width / 4);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_spin_mean<real>), dim3(nblk(total)), dim3(256), 0, st, x, m, B, n_up, width, li);
}
template <typename real>
void launch_row_sum(hipStream_t st, const real* x, real* s, int B, int rows, int width, LaneInfo li) {
  const long total = (long)B * li.TP * (width / 4);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_row_sum<real>), dim3(nblk(total)), dim3(256), 0, st, x, s, B, rows, width, li);
}
template <typename real>
void launch_conv(hipStream_t st, const real* we, int we_rows, int we_width, const real* hx, int hx_rows, int hx_width, real* out,
                 int out_width, int col0, const int32_t* tab, int S, int W, int B, LaneInfo li, int compact) {
  if ((W & 3) == 0 && (col0 & 3) == 0 && (we_width & 3) == 0 && (hx_width & 3) == 0 && (out_width & 3) == 0) {
    const long total = (long)B * li.N * (W / 4);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv<real, 4>), dim3(nblk(total)), dim3(256), 0, st, we, we_rows, we_width, hx, hx_rows,
                       hx_width, out, out_width, col0, tab, S, W, B, li, compact);
  } else {
    const long total = (long)B * li.N * W;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_conv<real, 1>), dim3(nblk(total)), dim3(256), 0, st, we, we_rows, we_width, hx, hx_rows,
                       hx_width, out, out_width, col0, tab, S, W, B, li, compact);
  }
}
template <typename real>
void launch_edge_sum(hipStream_t st, const real* e, int e_rows, int e_width, real* out, int out_width, int col0,
                     const int32_t* tab, int S, int W, double scale, int B, LaneInfo li, int compact) {
  const long total = (long)B * li.N * li.TP * (((W & 3) == 0 && (col0 & 3) == 0) ? W / 4 : W);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_edge_sum<real>), dim3(nblk(total)), dim3(256), 0, st, e, e_rows, e_width, out,
                     out_width, col0, tab, S, W, (real)scale, B, li, compact);
}

#define DQMC_INST(real)                                                                                              \
  template void launch_feat_en<real>(hipStream_t, const real*, const real*, real*, int, int, int, int, LaneInfo,     \
                                     double, int, int, const double*);                                               \
  template void launch_feat_ee<real>(hipStream_t, const real*, const real*, const int32_t*, real*, int, int, LaneInfo, double, \
                                     int, int, const double*);                                                       \
  template void launch_const_rows<real>(hipStream_t, const real*, real*, int, int, int, LaneInfo);                   \
  template void launch_spin_mean<real>(hipStream_t, const real*, real*, int, int, int, LaneInfo);                    \
  template void launch_row_sum<real>(hipStream_t, const real*, real*, int, int, int, LaneInfo);                      \
  template void launch_conv<real>(hipStream_t, const real*, int, int, const real*, int, int, real*, int, int,        \
                                  const int32_t*, int, int, int, LaneInfo, int);                                     \
  template void launch_edge_sum<real>(hipStream_t, const real*, int, int, real*, int, int, const int32_t*, int, int, \
                                      double, int, LaneInfo, int);
DQMC_INST(float)
DQMC_INST(double)
#undef DQMC_INST

}  // namespace dqmc

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
#include "common.h"
#include "kernels.h"
#include "../../include/dqmc.h"

namespace dqmc {

template <typename real> struct BufView {
  real* lds;     // nullptr for a global buffer
  real* glb;     // global base (walker 0 of the tile), layout [walker][rows][width]
  int stride;    // row stride in elements
  int rows;      // rows per walker
  int width;
};

template <typename real>
__device__ __forceinline__ BufView<real> view(const FusedArgs<real>& a, real* smem, int b, int w0) {
  const FusedBuf fb = a.fbufs[b];
  BufView<real> v;
  v.rows = fb.rows; v.width = fb.width;
  if (fb.is_global) {
    v.lds = nullptr;
    v.glb = reinterpret_cast<real*>(a.ws + fb.goff) + (long)w0 * fb.rows * fb.width;
    v.stride = fb.width;
  } else {
    v.lds = smem + fb.off;
    v.glb = nullptr;
    v.stride = fb.stride;
  }
  return v;
}

template <typename real> __device__ __forceinline__ real act_value(int act, real v) {
  if (act == 1) return r_tanh<real>(v);
  if (act == 2) return v / (1 + r_exp<real>(-v));
  return v;
}

// y = act(concat(pieces) W + b) (+ residual) on the tile, MFMA 16x16x4.
template <typename real>
__device__ void fused_linear(const FusedArgs<real>& a, real* smem, const dqmc_op& op, long wpk_off, int nw, int w0) {
  typedef typename Mfma<real>::acc_t acc_t;
  constexpr int MRW = 4, NRW = 2;
  const int32_t* i = op.i;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const int nrows = i[20], nout = i[21];
  const int ldw = (nout + 3) / 4 * 4;
  const int Rtot = nw * nrows;
  const int NRB = (Rtot + 15) / 16, NCB = (ldw + 15) / 16;
  const int n_rg = (NRB + MRW - 1) / MRW, n_cg = (NCB + NRW - 1) / NRW;
  const BufView<real> dst = view<real>(a, smem, i[17], w0);
  const real* bias = i[23] >= 0 ? a.w + i[23] : nullptr;
  const bool has_res = i[25] >= 0;
  BufView<real> res = dst;
  if (has_res) res = view<real>(a, smem, i[25], w0);
  const real res_scale = i[27] ? (real)0.70710678118654752440 : (real)1;
  const real* wpk = a.wpk + wpk_off;

  for (int u = wave; u < n_rg * n_cg; u += n_waves) {
    const int rg = u / n_cg, cg = u - rg * n_cg;
    acc_t acc[MRW][NRW];
#pragma unroll
    for (int x = 0; x < MRW; ++x)
#pragma unroll
      for (int y = 0; y < NRW; ++y) acc[x][y] = acc_t{0, 0, 0, 0};
    int ks0 = 0;
    for (int p = 0; p < i[0]; ++p) {
      const BufView<real> src = view<real>(a, smem, i[1 + 4 * p], w0);
      const int r0 = i[2 + 4 * p], Kp = (i[3 + 4 * p] + 3) / 4 * 4, bc = i[4 + 4 * p];
      const real* ap[MRW];
#pragma unroll
      for (int x = 0; x < MRW; ++x) {
        const int m = (rg * MRW + x) * 16 + (lane & 15);
        if (m < Rtot) {
          const int wl = m / nrows, rr = m - wl * nrows;
          ap[x] = src.lds + (long)(wl * src.rows + r0 + (bc ? 0 : rr)) * src.stride + (lane >> 4);
        } else {
          ap[x] = nullptr;
        }
      }
      const int KS = Kp / 4;
#pragma unroll 4
      for (int ks = 0; ks < KS; ++ks) {
        real fb[NRW], fa[MRW];
#pragma unroll
        for (int y = 0; y < NRW; ++y) {
          const int cb = cg * NRW + y;
          fb[y] = cb < NCB ? wpk[((long)(ks0 + ks) * NCB + cb) * 64 + lane] : (real)0;
        }
#pragma unroll
        for (int x = 0; x < MRW; ++x) fa[x] = ap[x] != nullptr ? ap[x][ks * 4] : (real)0;
#pragma unroll
        for (int x = 0; x < MRW; ++x)
#pragma unroll
          for (int y = 0; y < NRW; ++y) acc[x][y] = Mfma<real>::run(fa[x], fb[y], acc[x][y]);
      }
      ks0 += KS;
    }
    // epilogue: bias + activation + residual, store to LDS or HBM
#pragma unroll
    for (int x = 0; x < MRW; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi) {
        const int m = (rg * MRW + x) * 16 + Mfma<real>::row_of(lane, rgi);
        if (m >= Rtot) continue;
        const int wl = m / nrows, rr = m - wl * nrows;
#pragma unroll
        for (int y = 0; y < NRW; ++y) {
          const int col = (cg * NRW + y) * 16 + (lane & 15);
          if (col >= ldw) continue;
          real v = acc[x][y][rgi];
          if (bias != nullptr) v += bias[col];
          v = act_value<real>(i[24], v);
          if (has_res) {
            const long ro = (long)(wl * res.rows + i[26] + rr) * res.stride + i[19] + col;
            v = ((res.lds ? res.lds[ro] : res.glb[ro]) + v) * res_scale;
          }
          const long o = (long)(wl * dst.rows + i[18] + rr) * dst.stride + i[19] + col;
          if (dst.lds) dst.lds[o] = v; else dst.glb[o] = v;
        }
      }
  }
}

template <typename real>
__global__ void __launch_bounds__(256) k_fused_value(const FusedArgs<real> a) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  const int w0 = blockIdx.x * a.WT;
  const int nw = (a.B - w0) < a.WT ? (a.B - w0) : a.WT;
  const int N = a.li.N, n_up = a.n_up, n_nuc = a.n_nuc, K = a.K;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const real* r = a.r + (long)w0 * N * 3;
  LaneInfo li = a.li;   // T = TP = 1

  for (int k = 0; k < a.n_ops; ++k) {
    const dqmc_op& op = a.ops[k];
    const int32_t* i = op.i;
    switch (op.kind) {
      case DQMC_OP_FEAT_EN: {
        const BufView<real> x = view<real>(a, smem, i[0], w0);
        for (int e = tid; e < nw * N * n_nuc; e += nthr) {
          const int n = e % n_nuc, q = e / n_nuc, el = q % N, wl = q / N;
          double d[3], f[4];
          for (int c = 0; c < 3; ++c) d[c] = (double)r[(wl * N + el) * 3 + c] - (double)a.R[n * 3 + c];
          pair_feature_lane(d, a.eps, el, -1, 0, li, i[1] != 0, f);
          real* row = x.lds + (long)(wl * N + el) * x.stride;
          for (int c = 0; c < 4; ++c) row[4 * n + c] = (real)f[c];
          if (n == 0) {
            int c = 4 * n_nuc;
            if (i[2]) row[c++] = (real)(el < n_up ? 1.0 : -1.0);
            for (; c < x.width; ++c) row[c] = (real)0;
          }
        }
        break;
      }
      case DQMC_OP_FEAT_EE: {
        const BufView<real> eb = view<real>(a, smem, i[0], w0);
        const int32_t* pairs = a.itable + i[1];
        const int n_rows = i[2];
        for (int e = tid; e < nw * n_rows; e += nthr) {
          const int kr = e % n_rows, wl = e / n_rows;
          const int rc = pairs[2 * kr], sd = pairs[2 * kr + 1];
          double d[3], f[4];
          for (int c = 0; c < 3; ++c) d[c] = (double)r[(wl * N + rc) * 3 + c] - (double)r[(wl * N + sd) * 3 + c];
          pair_feature_lane(d, a.eps, rc, sd, 0, li, i[3] != 0, f);
          real* row = eb.lds + (long)(wl * n_rows + kr) * eb.stride;
          for (int c = 0; c < 4; ++c) row[c] = (real)f[c];
        }
        break;
      }
      case DQMC_OP_LINEAR:
        fused_linear<real>(a, smem, op, a.wpk_off[k], nw, w0);
        break;
      case DQMC_OP_SPIN_MEAN: {
        const BufView<real> x = view<real>(a, smem, i[0], w0), m = view<real>(a, smem, i[1], w0);
        for (int e = tid; e < nw * 2 * x.width; e += nthr) {
          const int c = e % x.width, q = e / x.width, which = q & 1, wl = q >> 1;
          const int i0 = which ? n_up : 0, i1 = which ? N : n_up;
          real acc = 0;
          for (int el = i0; el < i1; ++el) acc += x.lds[(long)(wl * N + el) * x.stride + c];
          m.lds[(long)(wl * 2 + which) * m.stride + c] = (i1 > i0) ? acc / (real)(i1 - i0) : (real)0;
        }
        break;
      }
      case DQMC_OP_CONV:
      case DQMC_OP_EDGE_SUM: {
        const bool conv = op.kind == DQMC_OP_CONV;
        const BufView<real> we = view<real>(a, smem, i[0], w0), out = view<real>(a, smem, i[2], w0);
        BufView<real> hx = we;
        if (conv) hx = view<real>(a, smem, i[1], w0);
        const real scale = conv ? (real)1 : (real)(1.0 / (double)(i[1] > 0 ? i[1] : 1));
        const int32_t* tab = a.itable + i[4];
        const int S = i[5], W = i[6], col0 = i[3];
        for (int e = tid; e < nw * N * W; e += nthr) {
          const int c = e % W, q = e / W, el = q % N, wl = q / N;
          real acc = 0;
          for (int s = 0; s < S; ++s) {
            const int row = tab[2 * (el * S + s)], snd = tab[2 * (el * S + s) + 1];
            if (row < 0) continue;
            const real ev = we.lds[(long)(wl * we.rows + row) * we.stride + c];
            acc += conv ? ev * hx.lds[(long)(wl * N + snd) * hx.stride + c] : ev;
          }
          out.lds[(long)(wl * N + el) * out.stride + col0 + c] = acc * scale;
        }
        break;
      }
      case DQMC_OP_ROW_SUM: {
        const BufView<real> x = view<real>(a, smem, i[0], w0), s = view<real>(a, smem, i[1], w0);
        for (int e = tid; e < nw * x.width; e += nthr) {
          const int c = e % x.width, wl = e / x.width;
          real acc = 0;
          for (int el = 0; el < x.rows; ++el) acc += x.lds[(long)(wl * x.rows + el) * x.stride + c];
          s.lds[(long)wl * s.stride + c] = acc;
        }
        break;
      }
      case DQMC_OP_ORBITALS: {
        const BufView<real> bf = view<real>(a, smem, i[0], w0), orb = view<real>(a, smem, i[1], w0);
        const int KN = K * N;
        for (int e = tid; e < nw * N * KN; e += nthr) {
          const int kmu = e % KN, q = e / KN, el = q % N, wl = q / N;
          const int kd = kmu / N, mu = kmu - kd * N;
          const real* pi = a.w + (el < n_up ? i[2] : i[3]) + (long)kmu * n_nuc;
          const real* ze = a.w + (el < n_up ? i[4] : i[5]) + (long)kmu * n_nuc;
          double e0 = 0;
          for (int n = 0; n < n_nuc; ++n) {
            double d2 = a.eps;
            for (int c = 0; c < 3; ++c) { const double d = (double)r[(wl * N + el) * 3 + c] - (double)a.R[n * 3 + c]; d2 += d * d; }
            e0 += (double)pi[n] * exp(-fabs((double)ze[n]) * sqrt(d2));
          }
          const double b0 = (double)bf.lds[(long)(wl * N + el) * bf.stride + kmu];
          orb.glb[(long)(wl * K + kd) * orb.width + el * N + mu] = (real)(e0 * b0);
        }
        break;
      }
      default:
        break;
    }
    __syncthreads();
  }
}

template <typename real> void launch_fused_value(hipStream_t st, const FusedArgs<real>& a, int n_blocks, size_t lds_bytes) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fused_value<real>), dim3((unsigned)n_blocks), dim3(256), lds_bytes, st, a);
}
template <typename real> int fused_set_lds_limit(size_t lds_bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_value<real>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

template void launch_fused_value<float>(hipStream_t, const FusedArgs<float>&, int, size_t);
template void launch_fused_value<double>(hipStream_t, const FusedArgs<double>&, int, size_t);
template int fused_set_lds_limit<float>(size_t);
template int fused_set_lds_limit<double>(size_t);

}  // namespace dqmc

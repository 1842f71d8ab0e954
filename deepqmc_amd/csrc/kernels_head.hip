// kernels_head.hip -- the Slater-determinant head of the wave function and the local-energy
// assembly: envelope * backflow -> K Slater matrices (with lanes) -> slogdet and its
// derivative traces (LDS-resident inverse, wave shuffles for the trace reductions) ->
// CI sum + cusps + Jastrow -> kinetic energy + Coulomb terms.
#include "common.h"
#include "kernels.h"

namespace dqmc {

// A[b][k][t][i*N + mu] = lane t of envelope(i; k,mu) * backflow(i; k,mu).
// Envelope: sum_a pi[k*N+mu][a] * exp(-|zeta[k*N+mu][a]| rho_ia)   (reference wf/env.py:57-75,
// isotropic, per-orbital exponents; a = nucleus * n_env + e runs over n_env envelopes per nucleus:
// n_env = 1 is one shell per nucleus, n_env = 3 with pi = 1 the SimplifiedNucleusDependentEnvelopes of
// env.py:110-226 whose exponents the host reads out of the nuclear stream); it depends on r_i only, so its
// derivative lanes are the three of electron i and the Laplacian.  Backflow is dense.
template <typename real>
__global__ void __launch_bounds__(256) k_orbitals(const real* __restrict__ r, const real* __restrict__ R,
                                                  const real* __restrict__ bf, int bf_width, real* __restrict__ orb,
                                                  int orb_width, const real* __restrict__ pi_up,
                                                  const real* __restrict__ pi_dn, const real* __restrict__ ze_up,
                                                  const real* __restrict__ ze_dn, int B, int n_up, int n_nuc, int n_env,
                                                  int K, LaneInfo li, double eps, const double* __restrict__ phq) {
  // One thread per (walker, electron, orbital k*N+mu): the envelope value / gradient / Laplacian are
  // computed once (n_nuc*n_env exponentials) and then applied to all TP lanes of the backflow row;
  // neighbouring threads walk neighbouring orbitals, so the backflow reads and the Slater-matrix
  // writes of every lane are contiguous runs of N elements.
  const int N = li.N, KN = K * N;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * N * KN;
  const bool need_d = li.T > 1;
  const int L = n_nuc * n_env;
  if (!need_d && sizeof(real) == 4) {
    // Value-only evaluation of the float32 build (Metropolis sub-steps, ECP quadrature walkers: hundreds of thousands of
    // rows per launch).  The launch was bound by its arithmetic -- every thread took n_nuc square roots and L libm
    // exponentials (~ 540 instructions for benzene, 6.8 ms per 690 k rows) for 8 bytes of traffic.  Now: the block's
    // electron-nucleus distances once, into LDS (a block of 256 orbitals spans 1 + 256 / KN electrons); exponentials
    // on the hardware exp2 unit with the rounding error of x log2(e) carried along (1-2 ulp, as accurate as expf).
    constexpr int RHO_CAP = 2048;
    __shared__ float rho_s[RHO_CAP];
    const long idx0 = (long)blockIdx.x * blockDim.x;
    const long idx1 = idx0 + blockDim.x < total ? idx0 + blockDim.x : total;      // (idx0 < total: the grid is sized to it)
    const long q0 = idx0 / KN, q1 = (idx1 - 1) / KN;
    const int nq = (int)(q1 - q0 + 1);
    const bool staged = (long)nq * n_nuc <= RHO_CAP;      // (uniform over the block)
    if (staged) {
      for (int e = threadIdx.x; e < nq * n_nuc; e += blockDim.x) {
        const int qq = e / n_nuc, nuc = e - qq * n_nuc;
        const real* rq = r + (q0 + qq) * 3;
        float d2 = (float)eps;
        for (int c = 0; c < 3; ++c) { const float d = (float)rq[c] - (float)R[nuc * 3 + c]; d2 += d * d; }
        rho_s[e] = sqrtf(d2);
      }
      __syncthreads();
    }
    if (idx >= total) return;
    // (64-bit divisions are ~100 instructions each: one pair per block on uniform values, 32-bit arithmetic per thread)
    const int lo = (int)(idx0 - q0 * KN) + (int)threadIdx.x;
    const int dq = lo / KN, kmu = lo - dq * KN;
    const long q = q0 + dq;
    const long b0 = q0 / N;
    const int ii = (int)(q0 - b0 * N) + dq;
    const int b = (int)b0 + ii / N, i = ii % N;
    const int k = kmu / N, mu = kmu - k * N;
    const real* pi = (i < n_up ? pi_up : pi_dn) + (long)kmu * L;
    const real* ze = (i < n_up ? ze_up : ze_dn) + (long)kmu * L;
    const real* rp = r + q * 3;
    const float* rho_q = rho_s + (q - q0) * n_nuc;
    float acc = 0.f, rho = 0.f;
    int nuc = 0, ev = 0;
    auto entry = [&](float pa, float za) {
      if (ev == 0) {
        if (staged) {
          rho = rho_q[nuc];
        } else {
          float d2 = (float)eps;
          for (int c = 0; c < 3; ++c) { const float d = (float)rp[c] - (float)R[nuc * 3 + c]; d2 += d * d; }
          rho = sqrtf(d2);
        }
      }
      const float x = -fabsf(za) * rho;
      const float t = x * 1.44269504f;
      const float e = __builtin_fmaf(x, 1.44269504f, -t) + x * 1.92596303e-8f;      // x log2(e) = t + e
      const float p2 = __builtin_amdgcn_exp2f(t);
      acc += pa * __builtin_fmaf(p2, e * 0.693147181f, p2);
      if (++ev == n_env) { ev = 0; ++nuc; }
    };
    // The table rows of neighbouring threads (orbitals) are L elements apart, so scalar reads touch 64 cache lines per
    // wave load; rows with L % 4 == 0 are fetched as float4 (4x fewer requests).
    if ((L & 3) == 0) {
      for (int a4 = 0; a4 < L; a4 += 4) {
        const Vec4<real> p4 = *reinterpret_cast<const Vec4<real>*>(pi + a4), z4 = *reinterpret_cast<const Vec4<real>*>(ze + a4);
#pragma unroll
        for (int j = 0; j < 4; ++j) entry((float)p4.v[j], (float)z4.v[j]);
      }
    } else {
      for (int a = 0; a < L; ++a) entry((float)pi[a], (float)ze[a]);
    }
    orb[(((long)b * K + k) * li.TP) * orb_width + i * N + mu] =
        (real)(acc * (float)bf[(((long)b * N + i) * li.TP) * bf_width + kmu]);
    return;
  }
  if (idx >= total) return;
  const int kmu = (int)(idx % KN);
  const long q = idx / KN;
  const int i = (int)(q % N);
  const int b = (int)(q / N);
  const int k = kmu / N, mu = kmu - k * N;
  const real* pi = (i < n_up ? pi_up : pi_dn) + (long)kmu * n_nuc * n_env;
  const real* ze = (i < n_up ? ze_up : ze_dn) + (long)kmu * n_nuc * n_env;
  double e0 = 0, eL = 0, eJ[3] = {0, 0, 0};
  // (table rows as float4 where L % 4 == 0: see the value-only branch)
  const bool vec = (L & 3) == 0;
  const real* rp = r + ((long)b * N + i) * 3;
  {
    const double* Q = phq ? phq + ((long)b * N + i) * PH_STRIDE : nullptr;
    double rho = 0, u[3] = {0, 0, 0}, g2 = 0, lr = 0;
    int nuc = 0, ev = 0;
    auto entry = [&](double pa, double za) {
      if (ev == 0) {
        double d[3];
        for (int c = 0; c < 3; ++c) d[c] = (double)rp[c] - (double)R[nuc * 3 + c];
        const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        rho = sqrt(eps + d2);
        const double ir = 1.0 / rho;                         // one division per nucleus
        for (int c = 0; c < 3; ++c) u[c] = d[c] * ir;
        g2 = d2 * ir * ir;                                   // |grad rho|^2
        lr = 3.0 * ir - d2 * ir * ir * ir;                   // Laplacian of rho
        if (Q) {   // pseudo-Hamiltonian lanes: derivatives along the columns of Q, tr(A Hess) with A = Q Q^T
          double qd[3], trA = 0.0;
          for (int c = 0; c < 3; ++c) {
            qd[c] = (Q[0 + c] * d[0] + Q[3 + c] * d[1] + Q[6 + c] * d[2]) * ir;
            trA += Q[c] * Q[c] + Q[3 + c] * Q[3 + c] + Q[6 + c] * Q[6 + c];
          }
          for (int c = 0; c < 3; ++c) u[c] = qd[c];
          g2 = qd[0] * qd[0] + qd[1] * qd[1] + qd[2] * qd[2];
          lr = trA * ir - g2 * ir;
        }
      }
      const double z = fabs(za);
      const double w = pa * exp(-z * rho);
      e0 += w;
      if (need_d) {
        for (int c = 0; c < 3; ++c) eJ[c] += -z * w * u[c];
        eL += w * (z * z * g2 - z * lr);
      }
      if (++ev == n_env) { ev = 0; ++nuc; }
    };
    if (vec) {
      for (int a4 = 0; a4 < L; a4 += 4) {
        const Vec4<real> p4 = *reinterpret_cast<const Vec4<real>*>(pi + a4), z4 = *reinterpret_cast<const Vec4<real>*>(ze + a4);
#pragma unroll
        for (int j = 0; j < 4; ++j) entry((double)p4.v[j], (double)z4.v[j]);
      }
    } else {
      for (int a = 0; a < L; ++a) entry((double)pi[a], (double)ze[a]);
    }
  }
  const real* brow = bf + (((long)b * N + i) * li.TP) * bf_width + kmu;
  real* orow = orb + (((long)b * K + k) * li.TP) * orb_width + i * N + mu;
  const double b0 = (double)brow[0];
  orow[0] = (real)(e0 * b0);
  if (!need_d) return;
  double lap_cross = 0;
  for (int t = 1; t < li.T - 1; ++t) {
    const double bt = (double)brow[(long)t * bf_width];
    double o = e0 * bt;
    const int c = t - 1;
    if (c / 3 == i) {
      const int c3 = c - 3 * i;
      const double ej = c3 == 0 ? eJ[0] : c3 == 1 ? eJ[1] : eJ[2];      // (selects: a runtime index would put eJ in scratch)
      o += ej * b0; lap_cross += 2.0 * ej * bt;
    }
    orow[(long)t * orb_width] = (real)o;
  }
  const int tl = li.T - 1;
  orow[(long)tl * orb_width] = (real)(e0 * (double)brow[(long)tl * bf_width] + eL * b0 + lap_cross);
  for (int t = li.T; t < li.TP; ++t) orow[(long)t * orb_width] = (real)0;
}

// Conditioning record of the Laplacian-mode determinant kernels (float32 build: what the refinement keys on besides
// the node cancellation): kappa = (1/N) sum_ij |A_ij| |(A^-1)_ji|, the sum of the MAGNITUDES of the terms of
// tr(A^-1 A) = N.  It bounds the response of log|det A| -- and, with dA_c in place of A, of the derivative traces --
// to a relative perturbation eps of every matrix entry (|d log det| <= eps N kappa), is 1 for a diagonal matrix and,
// unlike |A| |A^-1|, does not change when rows or columns are rescaled (envelopes decaying at different rates).
// float32 orbitals carry eps ~ 6e-8 per entry: kappa ~ 1e4..1e6 (random-init TransPsiformer) means float32 cannot
// deliver 1e-5 on E_loc whatever the kernels do.

// slogdet of one N x N matrix per wave plus its forward-Laplacian lanes:
//   J_c = tr(A^-1 dA_c),   L = tr(A^-1 A_L) - sum_c tr((A^-1 dA_c)^2)        
// Gauss-Jordan with partial pivoting in LDS (double), sign = (-1)^swaps * prod sign(pivot)
// -- the restatement of LAPACK getrf's sign/log|det| used by jnp.linalg.slogdet
// (reference wf/nn_wave_function.py:36-39).  One 64-lane workgroup per (walker, determinant).
template <typename real, int NMAX>
__global__ void __launch_bounds__(64) k_slogdet(const real* __restrict__ orb, int orb_width,
                                                double* __restrict__ logdet, int32_t* __restrict__ sign_k, int K,
                                                LaneInfo li, double* __restrict__ cond) {
  __shared__ double A[NMAX * NMAX];
  __shared__ double Inv[NMAX * NMAX];
  __shared__ double M[NMAX * NMAX];
  __shared__ double colp[NMAX];
  __shared__ int piv_s;
  const int N = li.N, NN = N * N;
  const int lane = threadIdx.x;
  const long bk = blockIdx.x;  // b*K + k
  const real* base = orb + bk * li.TP * orb_width;
  for (int e = lane; e < NN; e += 64) {
    A[e] = (double)base[e];
    Inv[e] = (e / N == e % N) ? 1.0 : 0.0;
  }
  __syncthreads();
  double logabs = 0.0;
  int sgn = 1;
  for (int p = 0; p < N; ++p) {
    if (lane == 0) {
      int best = p;
      double bv = fabs(A[p * N + p]);
      for (int i = p + 1; i < N; ++i) {
        const double v = fabs(A[i * N + p]);
        if (v > bv) { bv = v; best = i; }   // first maximum, as LAPACK idamax
      }
      piv_s = best;
    }
    __syncthreads();
    const int q = piv_s;
    if (q != p) {
      for (int j = lane; j < N; j += 64) {
        double t0 = A[p * N + j]; A[p * N + j] = A[q * N + j]; A[q * N + j] = t0;
        t0 = Inv[p * N + j]; Inv[p * N + j] = Inv[q * N + j]; Inv[q * N + j] = t0;
      }
      sgn = -sgn;
    }
    __syncthreads();
    const double piv = A[p * N + p];
    logabs += log(fabs(piv));
    if (piv < 0) sgn = -sgn;
    if (piv == 0) sgn = 0;
    __syncthreads();
    const double ip = 1.0 / piv;
    for (int j = lane; j < N; j += 64) { A[p * N + j] *= ip; Inv[p * N + j] *= ip; }
    for (int i = lane; i < N; i += 64) colp[i] = A[i * N + p];
    __syncthreads();
    for (int e = lane; e < NN; e += 64) {
      const int i = e / N, j = e - i * N;
      if (i != p) {
        const double f = colp[i];
        A[e] -= f * A[p * N + j];
        Inv[e] -= f * Inv[p * N + j];
      }
    }
    __syncthreads();
  }
  if (lane == 0) {
    logdet[bk * li.TP] = logabs;
    sign_k[bk] = sgn;
  }
  if (li.T == 1) return;
  if (cond) {      // componentwise condition number of the matrix (comment above k_slogdet)
    double c = 0.0;
    for (int e = lane; e < NN; e += 64) {
      const int i = e / N, j = e - i * N;
      c += fabs((double)base[e]) * fabs(Inv[j * N + i]);
    }
    c = wave_sum<double>(c);
    if (lane == 0) cond[bk] = c / N;
  }
  double tr2_sum = 0.0;
  for (int t = 1; t < li.T; ++t) {
    const real* At = base + (long)t * orb_width;
    for (int e = lane; e < NN; e += 64) A[e] = (double)At[e];
    __syncthreads();
    double tr = 0.0;
    for (int e = lane; e < NN; e += 64) {
      const int i = e / N, l = e - i * N;
      double m = 0.0;
      for (int j = 0; j < N; ++j) m += Inv[i * N + j] * A[j * N + l];
      M[e] = m;
      if (i == l) tr += m;
    }
    __syncthreads();
    tr = wave_sum<double>(tr);
    if (t < li.T - 1) {
      double t2 = 0.0;
      for (int e = lane; e < NN; e += 64) {
        const int i = e / N, l = e - i * N;
        t2 += M[e] * M[l * N + i];
      }
      tr2_sum += wave_sum<double>(t2);
      if (lane == 0) logdet[bk * li.TP + t] = tr;
    } else if (lane == 0) {
      logdet[bk * li.TP + t] = tr - tr2_sum;
    }
    __syncthreads();
  }
  for (int t = li.T + lane; t < li.TP; t += 64) logdet[bk * li.TP + t] = 0.0;
}

// Laplacian-mode variant for 5..16 electrons: ONE wave per matrix like k_slogdet, but the per-lane products
// M_c = A^-1 dA_c (3N + 1 of them per matrix: what the kernel spends its time on) are single 16 x 16 x 16 tiles on the
// float64 matrix cores: A^-1 (zero padded) sits in registers as the A operand for the whole lane loop, dA_c is read
// straight from HBM in B-operand order (row j of dA_c = 16 consecutive columns per quarter wave: coalesced), and
// four v_mfma_f64_16x16x4_f64 replace ~N^3 scalar multiply-adds fed from LDS.  tr(M_c) is one wave reduction per lane;
// sum_c tr(M_c^2) -- M_c transposed through a 16 x 17 LDS tile -- is accumulated per thread and reduced once at the end.
// N2 / FermiNet (14 electrons, 16 determinants, 4096 walkers): 3.2 ms -> see DESIGN.md section 4.
template <typename real>
__global__ void __launch_bounds__(64) k_slogdet_w16(const real* __restrict__ orb, int orb_width, double* __restrict__ logdet,
                                                    int32_t* __restrict__ sign_k, LaneInfo li, double* __restrict__ cond) {
  constexpr int NS = 17;
  __shared__ double A[16 * NS];
  __shared__ double Inv[16 * NS];
  __shared__ double colp[16];
  __shared__ int piv_s;
  const int N = li.N, NN = N * N, T = li.T;
  const int lane = threadIdx.x, l15 = lane & 15, l4 = lane >> 4;
  const long bk = blockIdx.x;  // b*K + k
  const real* base = orb + bk * li.TP * orb_width;
  for (int e = lane; e < 16 * 16; e += 64) {
    const int i = e >> 4, j = e & 15;
    const bool in = i < N && j < N;
    A[i * NS + j] = in ? (double)base[i * N + j] : 0.0;
    Inv[i * NS + j] = (in && i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  double logabs = 0.0;
  int sgn = 1;
  for (int p = 0; p < N; ++p) {
    if (lane == 0) {
      int best = p;
      double bv = fabs(A[p * NS + p]);
      for (int i = p + 1; i < N; ++i) {
        const double v = fabs(A[i * NS + p]);
        if (v > bv) { bv = v; best = i; }   // first maximum, as LAPACK idamax
      }
      piv_s = best;
    }
    __syncthreads();
    const int q = piv_s;
    if (q != p) {
      for (int j = lane; j < N; j += 64) {
        double t0 = A[p * NS + j]; A[p * NS + j] = A[q * NS + j]; A[q * NS + j] = t0;
        t0 = Inv[p * NS + j]; Inv[p * NS + j] = Inv[q * NS + j]; Inv[q * NS + j] = t0;
      }
      sgn = -sgn;
    }
    __syncthreads();
    const double piv = A[p * NS + p];
    logabs += log(fabs(piv));
    if (piv < 0) sgn = -sgn;
    if (piv == 0) sgn = 0;
    __syncthreads();
    const double ip = 1.0 / piv;
    for (int j = lane; j < N; j += 64) { A[p * NS + j] *= ip; Inv[p * NS + j] *= ip; }
    for (int i = lane; i < N; i += 64) colp[i] = A[i * NS + p];
    __syncthreads();
    for (int e = lane; e < NN; e += 64) {
      const int i = e / N, j = e - i * N;
      if (i != p) {
        const double f = colp[i];
        A[i * NS + j] -= f * A[p * NS + j];
        Inv[i * NS + j] -= f * Inv[p * NS + j];
      }
    }
    __syncthreads();
  }
  if (lane == 0) {
    logdet[bk * li.TP] = logabs;
    sign_k[bk] = sgn;
  }
  if (T == 1) return;
  if (cond) {      // kappa = sum |A_ij| |(A^-1)_ji| / N (comment above k_slogdet)
    double c = 0.0;
    for (int e = lane; e < NN; e += 64) {
      const int i = e / N, j = e - i * N;
      c += fabs((double)base[e]) * fabs(Inv[j * NS + i]);
    }
    c = wave_sum<double>(c);
    if (lane == 0) cond[bk] = c / N;
  }
  typedef Mfma<double>::acc_t acc_t;
  double fa[4];                                            // A^-1 as MFMA A fragments: row l15, k = kk*4 + l4
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fa[kk] = Inv[l15 * NS + kk * 4 + l4];
  double* Ms = A;                                          // the elimination is over: A's tile holds M_c (transposition)
  // B fragments of dA_c: element (k = kk*4 + l4, n = l15) = dA_c[j = k][l = n]
  int boff[4];
  bool bok[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int j = kk * 4 + l4;
    bok[kk] = j < N && l15 < N;
    boff[kk] = bok[kk] ? j * N + l15 : 0;
  }
  real fbn[4];
  auto issue = [&](int t) {
    const real* At = base + (long)t * orb_width;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fbn[kk] = At[boff[kk]];
  };
  issue(1);
  double t2_acc = 0.0;
  for (int t = 1; t < T; ++t) {
    double fb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fb[kk] = bok[kk] ? (double)fbn[kk] : 0.0;
    if (t + 1 < T) issue(t + 1);                           // next lane in flight during the products and reductions
    acc_t acc = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = Mfma<double>::run(fa[kk], fb[kk], acc);
    double tr = 0.0;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) if (Mfma<double>::row_of(lane, rg) == l15) tr += acc[rg];
    const bool need2 = t < T - 1;
    if (need2) {
      wave_lds_fence();                                    // previous lane's reads of Ms are done
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) Ms[Mfma<double>::row_of(lane, rg) * NS + l15] = acc[rg];
      wave_lds_fence();
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) t2_acc += acc[rg] * Ms[l15 * NS + Mfma<double>::row_of(lane, rg)];
    }
    tr = wave_sum<double>(tr);
    if (need2) {
      if (lane == 0) logdet[bk * li.TP + t] = tr;
    } else {
      const double tr2_sum = wave_sum<double>(t2_acc);
      if (lane == 0) logdet[bk * li.TP + t] = tr - tr2_sum;
    }
  }
  for (int t = T + lane; t < li.TP; t += 64) logdet[bk * li.TP + t] = 0.0;
}

// One 16 x 16 tile of A^-1 dA_c: the A fragments (a row block of A^-1, the same for every lane) live in
// registers, the NK B fragments are requested from LDS before the first MFMA, so the chain of dependent
// v_mfma_f64_16x16x4_f64 is not interleaved with LDS round trips.
template <int NK>
__device__ __forceinline__ Mfma<double>::acc_t slogdet_tile_mm(const double (&fa)[12], const double* ib, int NS) {
  double fb[NK];
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) fb[kk] = ib[(kk * 4) * NS];
  Mfma<double>::acc_t acc = Mfma<double>::acc_t{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < NK; ++kk) acc = Mfma<double>::run(fa[kk], fb[kk], acc);
  return acc;
}

// Laplacian-mode variant for larger determinants (N > 8): the per-lane products M_c = A^-1 dA_c -- N^3 each,
// 3N + 1 of them per matrix, what dominates this kernel from ~14 electrons on -- run on the float64 matrix cores
// (v_mfma_f64_16x16x4_f64) with A^-1 as the A operand and dA_c as the B operand, both in LDS; four waves share
// the (N/16)^2 output tiles; traces tr(M) and tr(M^2) from the LDS copy of M.  The inverse itself is the same
// pivoted Gauss-Jordan as k_slogdet (256 threads instead of 64).  One workgroup per (walker, determinant).
template <typename real>
__global__ void __launch_bounds__(256, 2) k_slogdet_mfma(const real* __restrict__ orb, int orb_width,
                                                      double* __restrict__ logdet, int32_t* __restrict__ sign_k,
                                                      LaneInfo li, double* __restrict__ cond) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  const int N = li.N, NN = N * N, T = li.T;
  const int nt = (N + 15) / 16, N16 = nt * 16, NS = N16 + 1;
  double* A = reinterpret_cast<double*>(smem_raw);     // [N16][NS] value matrix, then dA_c (B operand)
  double* Inv = A + N16 * NS;                           // [N16][NS] (A operand), zero outside N x N
  double* Mx = Inv + N16 * NS;                          // [N16][NS] M_c
  double* colp = Mx + N16 * NS;                         // [N16]
  __shared__ int piv_s;
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const long bk = blockIdx.x;  // b*K + k
  const real* base = orb + bk * li.TP * orb_width;
  for (int e = tid; e < N16 * N16; e += nthr) {
    const int i = e / N16, j = e - i * N16;
    const bool in = i < N && j < N;
    A[i * NS + j] = in ? (double)base[i * N + j] : 0.0;
    Inv[i * NS + j] = (in && i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  double logabs = 0.0;
  int sgn = 1;
  // elements of the elimination update owned by this thread (the same for every pivot: no integer division in the loop)
  constexpr int MAXU = 9;                                   // N * N <= 2304 = 9 x 256
  int e_row[MAXU], e_off[MAXU];
#pragma unroll
  for (int u = 0; u < MAXU; ++u) {
    const int e = tid + u * 256;
    const int i = e / N, j = e - i * N;
    e_row[u] = e < NN ? i : -1;
    e_off[u] = i * NS + j;
  }
  for (int p = 0; p < N; ++p) {
    if (wave == 0) {
      // pivot search by one wave (N <= 48 rows): largest |A[i][p]|, i >= p; ties to the smaller row, as LAPACK idamax
      double bv = (lane >= p && lane < N) ? fabs(A[lane * NS + p]) : -1.0;
      int best = lane;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) {
        const double ov = __shfl_xor(bv, m, 64);
        const int ob = __shfl_xor(best, m, 64);
        if (ov > bv || (ov == bv && ob < best)) { bv = ov; best = ob; }
      }
      if (lane == 0) piv_s = best;
    }
    __syncthreads();
    const int q = piv_s;
    if (q != p) {
      for (int j = tid; j < N; j += nthr) {
        double t0 = A[p * NS + j]; A[p * NS + j] = A[q * NS + j]; A[q * NS + j] = t0;
        t0 = Inv[p * NS + j]; Inv[p * NS + j] = Inv[q * NS + j]; Inv[q * NS + j] = t0;
      }
      sgn = -sgn;
    }
    __syncthreads();
    const double piv = A[p * NS + p];
    if (tid == 0) logabs += log(fabs(piv));                 // (only thread 0 writes log|det|)
    if (piv < 0) sgn = -sgn;
    if (piv == 0) sgn = 0;
    __syncthreads();
    const double ip = 1.0 / piv;
    for (int j = tid; j < N; j += nthr) { A[p * NS + j] *= ip; Inv[p * NS + j] *= ip; }
    for (int i = tid; i < N; i += nthr) colp[i] = A[i * NS + p];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
      const int i = e_row[u];
      if (i >= 0 && i != p) {
        const double f = colp[i];
        const int j = e_off[u] - i * NS;
        A[e_off[u]] -= f * A[p * NS + j];
        Inv[e_off[u]] -= f * Inv[p * NS + j];
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    logdet[bk * li.TP] = logabs;
    sign_k[bk] = sgn;
  }
  if (cond) {      // kappa = sum |A_ij| |(A^-1)_ji| / N; colp is free after the elimination
    double c = 0.0;
    for (int e = tid; e < NN; e += nthr) {
      const int i = e / N, j = e - i * N;
      c += fabs((double)base[e]) * fabs(Inv[j * NS + i]);
    }
    c = wave_sum<double>(c);
    if (lane == 0) colp[wave] = c;
    __syncthreads();
    if (tid == 0) cond[bk] = (colp[0] + colp[1] + colp[2] + colp[3]) / N;
    __syncthreads();
  }
  // ---- derivative lanes (round 4: software-pipelined, LDS traffic cut) ----
  // SQ counters of the previous structure on benzene (42 electrons; profiles/r04_pmc_sq_counters_benzene.json): the LDS
  // unit of a CU was busy for the whole launch -- every lane staged dA_c through LDS (9 writes per thread, 36 fragment
  // reads per MFMA wave), then all 256 threads re-read M for the two traces and reduced them with 24 ds_bpermute and two
  // workgroup barriers -- for 36 MFMAs of work: 15 % of a benzene step.  Now: waves 0 .. nt-1 fetch their B fragments of
  // dA_c straight from global memory in operand order (two lanes ahead, registers), multiply, and leave M_c in one of two
  // LDS buffers; wave 3 reduces the traces of the PREVIOUS lane from the other buffer meanwhile.  One barrier per lane.
  typedef Mfma<double>::acc_t acc_t;
  double fa[12];                                            // this wave's row block of A^-1 as MFMA A fragments
#pragma unroll
  for (int kk = 0; kk < 12; ++kk) fa[kk] = (wave < nt && kk * 4 < N16) ? Inv[(wave * 16 + l15) * NS + kk * 4 + l4] : 0.0;
  __syncthreads();                                          // (Inv and A are free from here on)
  double* Mbuf[2] = {Mx, A};
  const int nk = N16 / 4;
  real fbA[3][12], fbB[3][12];                              // B fragments of two lanes in flight: dA_c[4 kk + l4][16 cb + l15]
  auto load_frags = [&](real (&fb)[3][12], int t) {
    const real* At = base + (long)t * orb_width;
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
      for (int kk = 0; kk < 12; ++kk) {
        const int k = kk * 4 + l4, col = cb * 16 + l15;
        fb[cb][kk] = (t < T && cb < nt && k < N && col < N) ? At[k * N + col] : (real)0;
      }
  };
  double tr2_sum = 0.0;
  // iteration `it`: the MFMA waves produce M of lane it (it < T) into Mbuf[it & 1]; the trace wave reduces lane it - 1
  auto step = [&](real (&fb)[3][12], int it) {
    if (wave < nt) {
      if (it < T) {
        double* Mo = Mbuf[it & 1];
#pragma unroll
        for (int cb = 0; cb < 3; ++cb)
          if (cb < nt) {
            acc_t acc = acc_t{0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 12; ++kk)
              if (kk < nk) acc = Mfma<double>::run(fa[kk], (double)fb[cb][kk], acc);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) Mo[(wave * 16 + Mfma<double>::row_of(lane, rg)) * NS + cb * 16 + l15] = acc[rg];
          }
        load_frags(fb, it + 2);                             // this register set is free again: two lanes ahead
      }
    } else if (wave == 3 && it >= 2 && it <= T) {          // (T odd: the last pair of steps reaches it = T + 1, which has no lane to reduce)
      const int t = it - 1;                                 // 1 <= t < T
      const double* Mi = Mbuf[t & 1];
      const bool need2 = t < T - 1;
      double tr = (lane < N) ? Mi[lane * NS + lane] : 0.0, t2 = 0.0;
      if (need2 && lane < N) {
        // (four independent partial sums: the LDS reads of four terms are in flight together instead of one dependent
        // read -> multiply -> add chain per term)
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        const double* col = Mi + lane;
        const double* row = Mi + lane * NS;
        int i = 0;
        for (; i + 4 <= N; i += 4) {
          const double c0 = col[(i + 0) * NS], c1 = col[(i + 1) * NS], c2 = col[(i + 2) * NS], c3 = col[(i + 3) * NS];
          const double r0 = row[i + 0], r1 = row[i + 1], r2 = row[i + 2], r3 = row[i + 3];
          p0 += c0 * r0; p1 += c1 * r1; p2 += c2 * r2; p3 += c3 * r3;
        }
        for (; i < N; ++i) p0 += col[i * NS] * row[i];
        t2 = (p0 + p1) + (p2 + p3);
      }
      tr = wave_sum<double>(tr);
      if (need2) {
        tr2_sum += wave_sum<double>(t2);
        if (lane == 0) logdet[bk * li.TP + t] = tr;
      } else if (lane == 0) {
        logdet[bk * li.TP + t] = tr - tr2_sum;
      }
    }
    __syncthreads();
  };
  if (wave < nt) { load_frags(fbA, 1); load_frags(fbB, 2); }
  for (int it = 1; it <= T; it += 2) {
    step(fbA, it);
    step(fbB, it + 1);
  }
  for (int t = T + tid; t < li.TP; t += nthr) logdet[bk * li.TP + t] = 0.0;
}

// Value-only variant (Metropolis sub-steps, ECP quadrature walkers; T = 1): sign and log|det| need only
// the LU factorisation, not the inverse.  One wave per matrix, the matrix in LDS (double, odd row stride);
// per pivot: wave-wide arg-max over the column (first maximum, as LAPACK idamax), row swap, and the rank-1
// update of the trailing block dealt over an 8 x 8 lane grid.  Two barriers per pivot instead of five, no
// serial pivot search, a third of the flops of the Gauss-Jordan kernel above.
template <typename real>
__global__ void __launch_bounds__(64) k_slogdet_lu(const real* __restrict__ orb, int orb_width,
                                                   double* __restrict__ logdet, int32_t* __restrict__ sign_k,
                                                   LaneInfo li) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  double* A = reinterpret_cast<double*>(smem_raw);
  const int N = li.N, NS = N | 1;
  const int lane = threadIdx.x;
  const long bk = blockIdx.x;  // b*K + k
  const real* base = orb + bk * li.TP * orb_width;
  for (int e = lane; e < N * N; e += 64) {
    const int i = e / N, j = e - i * N;
    A[i * NS + j] = (double)base[e];
  }
  __syncthreads();
  double logabs = 0.0;
  int sgn = 1;
  const int li8 = lane >> 3, lj8 = lane & 7;
  for (int p = 0; p < N; ++p) {
    double bv = -1.0;
    int q = p;
    for (int i = p + lane; i < N; i += 64) {
      const double x = fabs(A[i * NS + p]);
      if (x > bv) { bv = x; q = i; }
    }
    for (int m = 1; m < 64; m <<= 1) {
      const double ov = __shfl_xor(bv, m, 64);
      const int oq = __shfl_xor(q, m, 64);
      if (ov > bv || (ov == bv && oq < q)) { bv = ov; q = oq; }
    }
    if (q != p) {
      for (int j = lane; j < N; j += 64) {
        const double t0 = A[p * NS + j]; A[p * NS + j] = A[q * NS + j]; A[q * NS + j] = t0;
      }
      sgn = -sgn;
    }
    __syncthreads();
    const double piv = A[p * NS + p];
    logabs += log(fabs(piv));
    if (piv < 0) sgn = -sgn;
    if (piv == 0) { sgn = 0; break; }      // singular: log|det| = -inf, sign 0 (wave-uniform exit)
    const double ip = 1.0 / piv;
    for (int i = p + 1 + li8; i < N; i += 8) {
      const double f = A[i * NS + p] * ip;
      for (int j = p + 1 + lj8; j < N; j += 8) A[i * NS + j] -= f * A[p * NS + j];
    }
    __syncthreads();
  }
  if (lane == 0) {
    logdet[bk * li.TP] = logabs;
    sign_k[bk] = sgn;
  }
}

// Value-only variant for 5..16 electrons: SIXTEEN lanes per matrix (four matrices per wave), lane i keeps row i in
// registers; no LDS, no barriers.  Per pivot: arg-max of |A[.][p]| over the unused rows by four xor-shuffles inside the
// 16-lane group (larger value wins, ties go to the row that LAPACK's physically swapped order would meet first: the
// smaller current POSITION, tracked per lane), the pivot row is broadcast column by column, every other unused row
// eliminates in its own registers.  Rows are never moved: the position bookkeeping reproduces getrf's swap count, so
// sign = (-1)^swaps * prod sign(pivot) is the same integer as from k_slogdet_lu (whose 64-lane, LDS-resident version
// spent ~7 k cycles per pivot on barriers and a 64-wide reduction for a 14-row column).
template <typename real>
__global__ void __launch_bounds__(256) k_slogdet_lu16(const real* __restrict__ orb, int orb_width, double* __restrict__ logdet,
                                                      int32_t* __restrict__ sign_k, long n_mat, LaneInfo li) {
  const int N = li.N;
  const int lane = threadIdx.x & 63, row = lane & 15, g0 = lane & 48;      // g0: first lane of this matrix's group
  const long bk_raw = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool live = bk_raw < n_mat;
  const long bk = live ? bk_raw : n_mat - 1;               // idle groups shadow the last matrix (all lanes stay in the shuffles)
  const real* base = orb + bk * li.TP * orb_width;
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = (row < N && j < N) ? (double)base[row * N + j] : 0.0;
  int pos = row;                                           // current position of this row in getrf's swapped order
  bool used = row >= N;
  double logabs = 0.0;
  int sgn = 1;
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    if (p < N) {                                           // (uniform over the launch)
      double bv = used ? -1.0 : fabs(a[p]);
      int bpos = used ? 1 << 20 : pos, bl = row;
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        const double ov = __shfl_xor(bv, m, 64);
        const int op = __shfl_xor(bpos, m, 64), ol = __shfl_xor(bl, m, 64);
        if (ov > bv || (ov == bv && op < bpos)) { bv = ov; bpos = op; bl = ol; }
      }
      const int ql = g0 + bl;                              // lane holding the pivot row
      const double piv = __shfl(a[p], ql, 64);
      if (bpos != p) sgn = -sgn;                           // getrf swaps rows p and bpos
      if (pos == p && row != bl) pos = bpos;               // the row that sat at position p moves to where the pivot row was
      if (row == bl) { pos = p; used = true; }
      logabs += log(fabs(piv));
      if (piv < 0) sgn = -sgn;
      if (piv == 0) sgn = 0;
      const double f = (used || piv == 0) ? 0.0 : a[p] / piv;
#pragma unroll
      for (int j = p + 1; j < 16; ++j) {
        if (j < N) {
          const double pj = __shfl(a[j], ql, 64);
          a[j] -= f * pj;
        }
      }
    }
  }
  if (live && row == 0) {
    logdet[bk * li.TP] = logabs;
    sign_k[bk] = sgn;
  }
}

// Value-only variant for 17..44 electrons: the register scheme of k_slogdet_lu16 with LPM = 32 (two matrices per wave,
// N <= 32) or 64 lanes per matrix, lane i keeps row i (NC >= N columns) in registers.  The LDS-resident k_slogdet_lu
// above spends two workgroup barriers and a 64-wide reduction per pivot on a matrix that fits the register file: 4.9 ms
// for the 368 k 30 x 30 determinants of one quadrature batch of benzene + ECP.  Same pivoting rule (first maximum in
// getrf's swapped order), same sign convention.
template <typename real, int NC, int LPM>
__global__ void __launch_bounds__(256) k_slogdet_reg(const real* __restrict__ orb, int orb_width, double* __restrict__ logdet,
                                                     int32_t* __restrict__ sign_k, long n_mat, LaneInfo li) {
  const int N = li.N;
  const int lane = threadIdx.x & 63, row = lane & (LPM - 1), g0 = lane & ~(LPM - 1);      // g0: first lane of this matrix's group
  const long bk_raw = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LPM;
  const bool live = bk_raw < n_mat;
  const long bk = live ? bk_raw : n_mat - 1;               // idle groups shadow the last matrix (all lanes stay in the shuffles)
  const real* base = orb + bk * li.TP * orb_width;
  double a[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) a[j] = (row < N && j < N) ? (double)base[row * N + j] : 0.0;
  int pos = row;                                           // current position of this row in getrf's swapped order
  bool used = row >= N;
  double logabs = 0.0;
  int sgn = 1;
#pragma unroll
  for (int p = 0; p < NC; ++p) {
    if (p < N) {                                           // (uniform over the launch)
      double bv = used ? -1.0 : fabs(a[p]);
      int bpos = used ? 1 << 20 : pos, bl = row;
#pragma unroll
      for (int m = 1; m < LPM; m <<= 1) {
        const double ov = __shfl_xor(bv, m, 64);
        const int op = __shfl_xor(bpos, m, 64), ol = __shfl_xor(bl, m, 64);
        if (ov > bv || (ov == bv && op < bpos)) { bv = ov; bpos = op; bl = ol; }
      }
      const int ql = g0 + bl;                              // lane holding the pivot row
      const double piv = __shfl(a[p], ql, 64);
      if (bpos != p) sgn = -sgn;                           // getrf swaps rows p and bpos
      if (pos == p && row != bl) pos = bpos;               // the row that sat at position p moves to where the pivot row was
      if (row == bl) { pos = p; used = true; }
      logabs += log(fabs(piv));
      if (piv < 0) sgn = -sgn;
      if (piv == 0) sgn = 0;
      const double f = (used || piv == 0) ? 0.0 : a[p] / piv;
#pragma unroll
      for (int j = p + 1; j < NC; ++j) {
        if (j < N) {
          const double pj = __shfl(a[j], ql, 64);
          a[j] -= f * pj;
        }
      }
    }
  }
  if (live && row == 0) {
    logdet[bk * li.TP] = logabs;
    sign_k[bk] = sgn;
  }
}

// Small-matrix variant (N <= 4: H2, LiH, Be ...): one thread per (walker, determinant), the
// matrix, its inverse and one derivative lane at a time in registers; row swaps are predicated
// so every index is a compile-time constant.  Same pivoting rule and sign convention as above.
template <typename real, int N>
__global__ void __launch_bounds__(256) k_slogdet_small(const real* __restrict__ orb, int orb_width,
                                                       double* __restrict__ logdet, int32_t* __restrict__ sign_k,
                                                       long n_mat, LaneInfo li, double* __restrict__ cond) {
  const long bk = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (bk >= n_mat) return;
  const real* base = orb + bk * li.TP * orb_width;
  double A[N][N], Inv[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) { A[i][j] = (double)base[i * N + j]; Inv[i][j] = (i == j) ? 1.0 : 0.0; }
  double logabs = 0.0;
  int sgn = 1;
#pragma unroll
  for (int p = 0; p < N; ++p) {
    int best = p;
    double bv = fabs(A[p][p]);
#pragma unroll
    for (int i = p + 1; i < N; ++i) {
      const double v = fabs(A[i][p]);
      if (v > bv) { bv = v; best = i; }
    }
#pragma unroll
    for (int i = p + 1; i < N; ++i) {
      const bool sw = (best == i);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const double a0 = A[p][j], a1 = A[i][j], b0 = Inv[p][j], b1 = Inv[i][j];
        A[p][j] = sw ? a1 : a0; A[i][j] = sw ? a0 : a1;
        Inv[p][j] = sw ? b1 : b0; Inv[i][j] = sw ? b0 : b1;
      }
    }
    if (best != p) sgn = -sgn;
    const double piv = A[p][p];
    logabs += log(fabs(piv));
    if (piv < 0) sgn = -sgn;
    if (piv == 0) sgn = 0;
    const double ip = 1.0 / piv;
#pragma unroll
    for (int j = 0; j < N; ++j) { A[p][j] *= ip; Inv[p][j] *= ip; }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i == p) continue;
      const double f = A[i][p];
#pragma unroll
      for (int j = 0; j < N; ++j) { A[i][j] -= f * A[p][j]; Inv[i][j] -= f * Inv[p][j]; }
    }
  }
  logdet[bk * li.TP] = logabs;
  sign_k[bk] = sgn;
  if (li.T == 1) return;
  if (cond) {
    double c = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) c += fabs((double)base[i * N + j]) * fabs(Inv[j][i]);
    cond[bk] = c / N;
  }
  double tr2_sum = 0.0;
  for (int t = 1; t < li.T; ++t) {
    const real* At = base + (long)t * orb_width;
    double D[N][N], M[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) D[i][j] = (double)At[i * N + j];
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int l = 0; l < N; ++l) {
        double m = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) m += Inv[i][j] * D[j][l];
        M[i][l] = m;
        if (i == l) tr += m;
      }
    if (t < li.T - 1) {
      double t2 = 0.0;
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int l = 0; l < N; ++l) t2 += M[i][l] * M[l][i];
      tr2_sum += t2;
      logdet[bk * li.TP + t] = tr;
    } else {
      logdet[bk * li.TP + t] = tr - tr2_sum;
    }
  }
  for (int t = li.T; t < li.TP; ++t) logdet[bk * li.TP + t] = 0.0;
}

// CI sum, cusps, Jastrow, and (Laplacian mode) the local energy.  SIXTEEN lanes per walker (the one-thread-per-walker
// version was a 100 us serial chain even for 32 walkers): determinants, derivative lanes and particle pairs are dealt
// to the lanes of a group and combined with xor-shuffles inside the group; double arithmetic (the cancellation
// Delta + |grad|^2 is the sensitive spot).
__device__ __forceinline__ double grp16_sum(double v) {
  for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double grp16_max(double v) {
  for (int m = 1; m < 16; m <<= 1) v = fmax(v, __shfl_xor(v, m, 64));
  return v;
}
template <typename real>
__global__ void __launch_bounds__(256) k_final(const FinalArgs a) {
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = threadIdx.x & 15;
  const int b_raw = gtid >> 4;
  const bool live = b_raw < a.B;
  const int b = live ? b_raw : a.B - 1;            // idle groups shadow the last walker (all lanes stay in the shuffles)
  const int N = a.li.N, T = a.li.T, TP = a.li.TP, K = a.K;
  const real* r = reinterpret_cast<const real*>(a.r) + (long)b * N * 3;
  const real* R = reinterpret_cast<const real*>(a.R);
  const double* x = a.logdet + (long)b * K * TP;
  const int32_t* sk = a.sign_k + (long)b * K;
  const real* cc = reinterpret_cast<const real*>(a.conf_coeff);
  // exp-normalised CI sum, reference wf/nn_wave_function.py:152-160
  double shift = -INFINITY;
  for (int k = l; k < K; k += 16) shift = fmax(shift, x[(long)k * TP]);
  shift = grp16_max(shift);
  if (isinf(shift)) shift = 0.0;
  double psi = 0.0;
  for (int k = l; k < K; k += 16) psi += (cc ? (double)cc[k] : 1.0) * sk[k] * exp(x[(long)k * TP] - shift);
  psi = grp16_sum(psi);
  double logpsi = log(fabs(psi)) + shift;
  const int sign = (psi > 0) - (psi < 0);
  // pair terms: cusp value (wf/cusp.py:5-26,68-78), its Laplacian, and the electron-electron repulsion
  const real* al = reinterpret_cast<const real*>(a.alphas);
  const int n_pairs = N * (N - 1) / 2;
  double cusp = 0.0, cusp_lap = 0.0, v_el = 0.0;
  {
    int i = 0, j = 1, p0 = 0;                        // pair p <-> (i, j), i < j, row-major
    for (int p = l; p < n_pairs; p += 16) {
      while (p - p0 >= N - 1 - i) { p0 += N - 1 - i; ++i; }
      j = i + 1 + (p - p0);
      double d2 = 0.0;
      for (int c = 0; c < 3; ++c) { const double d = (double)r[i * 3 + c] - (double)r[j * 3 + c]; d2 += d * d; }
      const double rho = sqrt(a.eps + d2);
      v_el += 1.0 / rho;                                    // reference physics.py:119-121 (safe norm)
      if (a.cusp_kind) {
        const bool same = (i < a.n_up) == (j < a.n_up);
        const double sc = same ? a.same_scale : a.anti_scale, alp = (double)al[same ? 0 : 1];
        double g1, g2;
        if (a.cusp_kind == 1) {
          const double u = 1 + alp * rho;
          cusp += -sc / (alp * u);
          g1 = sc / (u * u); g2 = -2 * sc * alp / (u * u * u);
        } else {
          const double u = alp + rho;
          cusp += -sc * alp * alp / u;
          g1 = sc * alp * alp / (u * u); g2 = -2 * sc * alp * alp / (u * u * u);
        }
        // both electrons: 2 * (g'' |grad rho|^2 + g' Lap rho)
        if (a.phq) {    // pseudo-Hamiltonian: sum over the two ends of g'' |Q^T grad rho|^2 + g' tr(A Hess rho)
          double dAd = 0.0, trA = 0.0;
          const double dv[3] = {(double)r[i * 3] - (double)r[j * 3], (double)r[i * 3 + 1] - (double)r[j * 3 + 1],
                                (double)r[i * 3 + 2] - (double)r[j * 3 + 2]};
          for (int end = 0; end < 2; ++end) {
            const double* Q = a.phq + ((long)b * N + (end ? j : i)) * PH_STRIDE;
            for (int c = 0; c < 3; ++c) {
              const double qd = Q[0 + c] * dv[0] + Q[3 + c] * dv[1] + Q[6 + c] * dv[2];
              dAd += qd * qd;
              trA += Q[c] * Q[c] + Q[3 + c] * Q[3 + c] + Q[6 + c] * Q[6 + c];
            }
          }
          cusp_lap += g2 * dAd / (rho * rho) + g1 * (trA / rho - dAd / (rho * rho * rho));
        } else
        cusp_lap += 2.0 * (g2 * d2 / (rho * rho) + g1 * (3.0 / rho - d2 / (rho * rho * rho)));
      }
    }
  }
  cusp = grp16_sum(cusp);
  const real* jas = reinterpret_cast<const real*>(a.jastrow);
  const real* jrow = jas ? jas + (long)b * TP * a.jas_width : nullptr;
  logpsi += cusp + (jrow ? (double)jrow[0] : 0.0);
  if (live && l == 0) {
    if (a.logpsi) reinterpret_cast<real*>(a.logpsi)[b] = (real)logpsi;
    if (a.sign) a.sign[b] = sign;
  }
  if (T == 1) return;

  // ---- gradient and Laplacian of log|psi| ----
  // per determinant (lanes over k): p_k and p_k (L_k + sum_t J_kt^2)
  double lap = 0.0;
  for (int k = l; k < K; k += 16) {
    const double pk = (cc ? (double)cc[k] : 1.0) * sk[k] * exp(x[(long)k * TP] - shift) / psi;
    double s2 = 0.0;
    for (int t = 1; t < T - 1; ++t) { const double j = x[(long)k * TP + t]; s2 += j * j; }
    lap += pk * (x[(long)k * TP + T - 1] + s2);
  }
  lap = grp16_sum(lap);
  // conditioning of the determinants that carry psi: sum_k |p_k| kappa_k (kernels above)
  double kappa = 0.0;
  if (a.cond) {
    for (int k = l; k < K; k += 16)
      kappa += fabs((cc ? (double)cc[k] : 1.0) * exp(x[(long)k * TP] - shift) / psi) * (sk[k] ? a.cond[(long)b * K + k] : 0.0);
    kappa = grp16_sum(kappa);
  }
  // per derivative lane (lanes over t): g_t = sum_k p_k J_kt (+ Jastrow + cusp gradients)
  real* grad = reinterpret_cast<real*>(a.grad);
  double sumJ2 = 0.0, qf2 = 0.0, first_order = 0.0;
  for (int t = 1 + l; t < T - 1; t += 16) {
    const int c = t - 1, e = c / 3, xyz = c - 3 * e;
    double g = 0.0;
    for (int k = 0; k < K; ++k) {
      const double pk = (cc ? (double)cc[k] : 1.0) * sk[k] * exp(x[(long)k * TP] - shift) / psi;
      g += pk * x[(long)k * TP + t];
    }
    sumJ2 += g * g;   // CI part only: subtracted in the Laplacian of the log-sum
    if (jrow) g += (double)jrow[(long)t * a.jas_width];
    if (a.cusp_kind) {  // d/dr_e of the pair cusps involving electron e
      for (int j = 0; j < N; ++j) {
        if (j == e) continue;
        double dv[3], d2 = a.eps;
        for (int c2 = 0; c2 < 3; ++c2) { dv[c2] = (double)r[e * 3 + c2] - (double)r[j * 3 + c2]; d2 += dv[c2] * dv[c2]; }
        const double rho = sqrt(d2);
        const bool same = (e < a.n_up) == (j < a.n_up);
        const double sc = same ? a.same_scale : a.anti_scale, alp = (double)al[same ? 0 : 1];
        const double g1 = a.cusp_kind == 1 ? sc / ((1 + alp * rho) * (1 + alp * rho))
                                           : sc * alp * alp / ((alp + rho) * (alp + rho));
        if (a.phq) {
          const double* Q = a.phq + ((long)b * N + e) * PH_STRIDE;
          g += g1 * (Q[0 + xyz] * dv[0] + Q[3 + xyz] * dv[1] + Q[6 + xyz] * dv[2]) / rho;
        } else
        g += g1 * dv[xyz] / rho;
      }
    }
    if (a.phq) first_order += a.phq[((long)b * N + e) * PH_STRIDE + 9 + xyz] * g;   // b . grad_r = (Q^-1 b) . grad_v
    qf2 += g * g;
    if (live && grad) grad[(long)b * (3 * N) + c] = (real)g;
  }
  sumJ2 = grp16_sum(sumJ2);
  qf2 = grp16_sum(qf2);
  cusp_lap = grp16_sum(cusp_lap);
  v_el = grp16_sum(v_el);
  lap += cusp_lap - sumJ2;
  if (jrow) lap += (double)jrow[(long)(T - 1) * a.jas_width];
  double v_loc = 0.0;
  for (int q = l; q < N * a.n_nuc; q += 16) {
    const int i = q / a.n_nuc, n = q - i * a.n_nuc;
    double d2 = 0.0;
    for (int c = 0; c < 3; ++c) { const double d = (double)r[i * 3 + c] - (double)R[n * 3 + c]; d2 += d * d; }
    const double rn = sqrt(d2);
    v_loc -= a.charges[n] / rn;                           // reference physics.py:131-133 (plain norm)
    if (a.ecp_loc) {                                      // ecp/gaussian_type_ecp.py:127-159: r^-1, r^0, r^1 Gaussians
      const double* p = a.ecp_loc + (long)n * 6 * a.ecp_nt;
      for (int t = 0; t < a.ecp_nt; ++t) {
        v_loc += p[1 * a.ecp_nt + t] / rn * exp(-p[0 * a.ecp_nt + t] * d2);
        v_loc += p[3 * a.ecp_nt + t] * exp(-p[2 * a.ecp_nt + t] * d2);
        v_loc += p[5 * a.ecp_nt + t] * rn * exp(-p[4 * a.ecp_nt + t] * d2);
      }
    }
  }
  v_loc = grp16_sum(v_loc);
  // nuclear repulsion of the geometry of THIS call (reference physics.py:112-116: recomputed from phys_conf.R)
  double e_nuc = 0.0;
  for (int n = 0; n < a.n_nuc; ++n)
    for (int m = n + 1; m < a.n_nuc; ++m) {
      double d2 = 0.0;
      for (int c = 0; c < 3; ++c) { const double d = (double)R[n * 3 + c] - (double)R[m * 3 + c]; d2 += d * d; }
      e_nuc += a.charges[n] * a.charges[m] / sqrt(d2);
    }
  double e_kin = -0.5 * (lap + qf2);                        // reference physics.py:108
  if (a.phq) {
    // ecp/pseudo_hamiltonian.py:236-278: the lanes are derivatives in the transformed coordinates (A = Q Q^T carries
    // the 1/2 of the kinetic energy), E_kin = sum_i b_i . grad_i - (tr-Laplacian + |grad_v|^2); :173-190 local term
    first_order = grp16_sum(first_order);
    e_kin = first_order - (lap + qf2);
    double vph = 0.0;
    for (int i = l; i < N; i += 16) vph += a.phq[((long)b * N + i) * PH_STRIDE + 12];
    v_loc += grp16_sum(vph);
  }
  const double e_loc = e_kin + v_loc + v_el + e_nuc;        // reference hamil.py:172 (V_nl is added by k_ecp_reduce)
  if (!live || l != 0) return;
  if (a.e_loc) reinterpret_cast<real*>(a.e_loc)[b] = (real)e_loc;
  if (a.flag_idx) {      // float32 build: hand ill-conditioned walkers to the float64 refinement pass
    // float32 error predictor (engine.hip: lap_refined): node cancellation x conditioning of the determinants
    const double score = (fabs(lap) + qf2) / fmax(1.0, fabs(e_loc)) * fmax(1.0, kappa);
    if (a.score_out) a.score_out[a.b_offset + b] = score;
    if (!(score <= (a.thresh_dev ? *a.thresh_dev : a.refine_thresh))) a.flag_idx[atomicAdd(a.flag_count, 1)] = a.b_offset + b;     // NaN / inf land here too
  }
  if (a.kappa_out) a.kappa_out[b] = kappa;
  if (a.stats) {
    real* s = reinterpret_cast<real*>(a.stats);
    s[0L * a.stats_ld + b] = (real)v_el;
    s[1L * a.stats_ld + b] = (real)e_kin;
    s[2L * a.stats_ld + b] = (real)v_loc;
    s[3L * a.stats_ld + b] = (real)0;
    s[4L * a.stats_ld + b] = (real)lap;
    s[5L * a.stats_ld + b] = (real)qf2;
  }
}

template <typename real>
void launch_orbitals(hipStream_t st, const real* r, const real* R, const real* bf, int bf_width, real* orb,
                     int orb_width, const real* pi_up, const real* pi_dn, const real* ze_up, const real* ze_dn, int B,
                     int n_up, int n_nuc, int n_env, int K, LaneInfo li, double eps, const double* phq) {
  const long total = (long)B * li.N * K * li.N;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_orbitals<real>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r, R,
                     bf, bf_width, orb, orb_width, pi_up, pi_dn, ze_up, ze_dn, B, n_up, n_nuc, n_env, K, li, eps, phq);
}

// use_mfma (engine option "slogdet_mfma"): 1 = f64-MFMA derivative traces (N > 16: k_slogdet_mfma, 4 waves per matrix;
// 5 <= N <= 16: k_slogdet_w16, one wave per matrix), 2 = k_slogdet_mfma from N > 8 on, 3 = as 1 without k_slogdet_w16, 0 = never
template <typename real>
void launch_slogdet(hipStream_t st, const real* orb, int orb_width, double* logdet, int32_t* sign_k, int B, int K,
                    LaneInfo li, int use_mfma, double* cond) {
  const int slogdet_use_mfma = use_mfma;
  const unsigned grid = (unsigned)((long)B * K);
  const long n_mat = (long)B * K;
  const unsigned gsm = (unsigned)((n_mat + 255) / 256);
  if (li.N == 2)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_small<real, 2>), dim3(gsm), dim3(256), 0, st, orb, orb_width, logdet,
                       sign_k, n_mat, li, cond);
  else if (li.N == 3)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_small<real, 3>), dim3(gsm), dim3(256), 0, st, orb, orb_width, logdet,
                       sign_k, n_mat, li, cond);
  else if (li.N == 4)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_small<real, 4>), dim3(gsm), dim3(256), 0, st, orb, orb_width, logdet,
                       sign_k, n_mat, li, cond);
  else if (li.T == 1 && li.N <= 16 && slogdet_use_mfma != 3)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_lu16<real>), dim3((unsigned)((n_mat * 16 + 255) / 256)), dim3(256), 0, st, orb, orb_width,
                       logdet, sign_k, n_mat, li);
  else if (li.T == 1 && li.N <= 32 && slogdet_use_mfma != 3)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_reg<real, 32, 32>), dim3((unsigned)((n_mat * 32 + 255) / 256)), dim3(256), 0, st, orb, orb_width,
                       logdet, sign_k, n_mat, li);
  else if (li.T == 1 && li.N <= 44 && slogdet_use_mfma != 3)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_reg<real, 44, 64>), dim3((unsigned)((n_mat * 64 + 255) / 256)), dim3(256), 0, st, orb, orb_width,
                       logdet, sign_k, n_mat, li);
  else if (li.T == 1)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_lu<real>), dim3(grid), dim3(64), sizeof(double) * li.N * (li.N | 1), st,
                       orb, orb_width, logdet, sign_k, li);
  else if (li.N <= 48 && ((slogdet_use_mfma == 1 && li.N > 16) || (slogdet_use_mfma == 2 && li.N > 8) || (slogdet_use_mfma == 3 && li.N > 16)))     // >= 2 x 2 tiles (14 x 14: the wave-per-matrix kernel is faster); N16 <= 48: 57 KB of LDS
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_mfma<real>), dim3(grid), dim3(256),
                       sizeof(double) * ((size_t)3 * (((li.N + 15) / 16) * 16) * ((((li.N + 15) / 16) * 16) + 1) + (((li.N + 15) / 16) * 16) + 16),
                       st, orb, orb_width, logdet, sign_k, li, cond);
  else if (li.N <= 16 && slogdet_use_mfma >= 1 && slogdet_use_mfma != 3)     // one wave per matrix, derivative traces on the f64 matrix cores
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet_w16<real>), dim3(grid), dim3(64), 0, st, orb, orb_width, logdet, sign_k, li, cond);
  else if (li.N <= 8)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet<real, 8>), dim3(grid), dim3(64), 0, st, orb, orb_width, logdet,
                       sign_k, K, li, cond);
  else if (li.N <= 16)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet<real, 16>), dim3(grid), dim3(64), 0, st, orb, orb_width, logdet,
                       sign_k, K, li, cond);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_slogdet<real, 44>), dim3(grid), dim3(64), 0, st, orb, orb_width, logdet,
                       sign_k, K, li, cond);
}

template <typename real> void launch_final(hipStream_t st, const FinalArgs& a) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_final<real>), dim3((unsigned)((a.B + 15) / 16)), dim3(256), 0, st, a);
}

#define DQMC_INST(real)                                                                                              \
  template void launch_orbitals<real>(hipStream_t, const real*, const real*, const real*, int, real*, int,           \
                                      const real*, const real*, const real*, const real*, int, int, int, int, int,   \
                                      LaneInfo, double, const double*);                                              \
  template void launch_slogdet<real>(hipStream_t, const real*, int, double*, int32_t*, int, int, LaneInfo, int,  \
                                     double*);                                                                   \
  template void launch_final<real>(hipStream_t, const FinalArgs&);
DQMC_INST(float)
DQMC_INST(double)
#undef DQMC_INST

}  // namespace dqmc

// kernels_ecp.hip -- non-local part of Gaussian-type effective core potentials.
//
// Reference: ecp/gaussian_type_ecp.py:161-255 (`nonloc_potential`) + ecp/ecp_utils.py:23-95.
//   V_nl = sum_{a in ecp} sum_i sum_l (2l+1)/12 * V_l(|r_i - R_a|) * sum_{q=1..12} P_l(cos theta_q) psi(r_i -> x_q)/psi(r)
// with x_q the 12 vertices of an icosahedron of radius |r_i - R_a| around R_a, its pole rotated onto the
// electron-nucleus axis and the whole turned about that axis by a random angle in [0, pi/5).
// The 12 N n_ecp value-only psi evaluations per walker are what costs (229 GFLOP/walker for benzene,
// SURVEY.md section 8); they run through the ordinary value path on a batch of "quadrature walkers"
// that k_ecp_points writes, and k_ecp_reduce folds the psi ratios back into E_loc.
// The local part (gaussian_type_ecp.py:127-159) lives in k_final next to the Coulomb terms.
#include "common.h"
#include "kernels.h"

namespace dqmc {

// cos(theta_q) of the unit icosahedron of ecp_utils.py:23-32: poles, then alternating
// atan(2) / pi - atan(2) rings.
__device__ __forceinline__ double ico_cos(int q) {
  if (q == 0) return 1.0;
  if (q == 1) return -1.0;
  return (q & 1) ? -0.44721359549995793 : 0.44721359549995793;   // -+ 1/sqrt(5)
}
__device__ __forceinline__ double ico_phi(int q) {
  if (q < 2) return 0.0;
  const int j = (q - 2) >> 1;
  return 0.62831853071795865 * ((q & 1) ? (2 * j - 1) : (2 * j));  // pi/5 * ...
}

// Quadrature walkers of the chunk [b0, b0 + nb): config ((bl*n_nl + j)*N + i)*12 + q is walker b0+bl
// with electron i moved to vertex q around nucleus nl_nuc[j].  One thread per coordinate.
template <typename real>
__global__ void __launch_bounds__(256) k_ecp_points(const EcpArgs a, real* __restrict__ rq) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int N = a.N;
  const long per_cfg = (long)N * 3;
  const long n_cfg = (long)a.nb * a.n_nl * N * 12;
  if (idx >= n_cfg * per_cfg) return;
  const long cfg = idx / per_cfg;
  const int ec = (int)(idx - cfg * per_cfg), e = ec / 3, c = ec - 3 * e;
  const int q = (int)(cfg % 12);
  const long t1 = cfg / 12;
  const int i = (int)(t1 % N);
  const long t2 = t1 / N;
  const int j = (int)(t2 % a.n_nl), bl = (int)(t2 / a.n_nl);
  const int b = a.b0 + bl;
  const real* r = reinterpret_cast<const real*>(a.r) + (long)b * per_cfg;
  if (e != i) { rq[idx] = r[ec]; return; }
  const real* R = reinterpret_cast<const real*>(a.R) + 3 * a.nl_nuc[j];
  const double dx = (double)r[3 * i] - (double)R[0], dy = (double)r[3 * i + 1] - (double)R[1],
               dz = (double)r[3 * i + 2] - (double)R[2];
  const double radius = sqrt(dx * dx + dy * dy + dz * dz);
  double ct = dz / radius;                                   // ecp_utils.py:53 (clip to [-1, 1])
  ct = ct < -1.0 ? -1.0 : (ct > 1.0 ? 1.0 : ct);
  const double st = sqrt(1.0 - ct * ct);                     // sin(arccos(.)) >= 0
  const double phi = atan2(dy, dx);
  double phr;
  const long bg = a.walker_idx ? (long)a.walker_idx[b] : (long)b;          // index that keys the rotation angle
  if (a.phi != nullptr) {
    const long pidx = (bg * a.n_nl + j) * N + i;
    phr = a.phi_f32 ? (double)reinterpret_cast<const float*>(a.phi)[pidx] : (double)reinterpret_cast<const real*>(a.phi)[pidx];
  } else {
    uint32_t o[4];
    philox4x32(a.seed, 0x45435000ull + (uint64_t)j, (uint64_t)bg * (uint64_t)N + (uint64_t)i, o);
    phr = u01(o[0], o[1]) * 0.62831853071795865;             // U[0, pi/5), ecp_utils.py:55
  }
  // u = rot_z(phr) * unit vertex; v = rot_y(theta) u; w = rot_z(phi) v
  const double cq = ico_cos(q), sq = sqrt(1.0 - cq * cq), pq = ico_phi(q) + phr;
  const double ux = sq * cos(pq), uy = sq * sin(pq), uz = cq;
  const double vx = ct * ux + st * uz, vy = uy, vz = -st * ux + ct * uz;
  const double cp = cos(phi), sp = sin(phi);
  const double w[3] = {cp * vx - sp * vy, sp * vx + cp * vy, vz};
  rq[idx] = (real)(radius * w[c] + (double)R[c]);
}

// One thread per walker of the chunk: V_nl from the psi ratios; added to e_loc and stored in stats[3].
template <typename real>
__global__ void __launch_bounds__(64) k_ecp_reduce(const EcpArgs a, const real* __restrict__ logq,
                                                   const int32_t* __restrict__ signq, const real* __restrict__ log0,
                                                   const int32_t* __restrict__ sign0, real* __restrict__ e_loc,
                                                   real* __restrict__ stats, real* __restrict__ v_nl_out) {
  const int bl = blockIdx.x * blockDim.x + threadIdx.x;
  if (bl >= a.nb) return;
  const int b = a.b0 + bl, N = a.N;
  const real* r = reinterpret_cast<const real*>(a.r) + (long)b * N * 3;
  const double l0 = (double)log0[b];
  const int s0 = sign0[b];
  double total = 0.0;
  for (int j = 0; j < a.n_nl; ++j) {
    const real* R = reinterpret_cast<const real*>(a.R) + 3 * a.nl_nuc[j];
    const double* nl = a.nl + (long)j * a.L * 2 * a.n_t;
    for (int i = 0; i < N; ++i) {
      double d2 = 0.0;
      for (int c = 0; c < 3; ++c) { const double d = (double)r[3 * i + c] - (double)R[c]; d2 += d * d; }
      const long base = (((long)bl * a.n_nl + j) * N + i) * 12;
      double ratio[12];
      for (int q = 0; q < 12; ++q)
        ratio[q] = exp((double)logq[base + q] - l0) * (double)(signq[base + q] * s0);   // ecp_utils.py:92-95
      for (int l = 0; l < a.L; ++l) {
        double vl = 0.0;                                                               // V_l(r), :207-211
        for (int t = 0; t < a.n_t; ++t) vl += nl[(l * 2 + 1) * a.n_t + t] * exp(-d2 * nl[(l * 2) * a.n_t + t]);
        if (vl == 0.0) continue;
        double integral = 0.0;
        for (int q = 0; q < 12; ++q) {
          const double x = ico_cos(q);
          double pm = 1.0, p = x;                     // Legendre recurrence
          double pl = l == 0 ? 1.0 : x;
          for (int m = 2; m <= l; ++m) { pl = ((2 * m - 1) * x * p - (m - 1) * pm) / m; pm = p; p = pl; }
          integral += ratio[q] * pl;
        }
        total += vl * (2 * l + 1) / 12.0 * integral;
      }
    }
  }
  if (e_loc) e_loc[b] = (real)((double)e_loc[b] + total);
  if (stats) stats[3L * a.B + b] = (real)total;
  if (v_nl_out) v_nl_out[b] = (real)total;
}

template <typename real> void launch_ecp_points(hipStream_t st, const EcpArgs& a, real* rq) {
  const long n = (long)a.nb * a.n_nl * a.N * 12 * a.N * 3;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ecp_points<real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, rq);
}
template <typename real>
void launch_ecp_reduce(hipStream_t st, const EcpArgs& a, const real* logq, const int32_t* signq, const real* log0,
                       const int32_t* sign0, real* e_loc, real* stats, real* v_nl_out) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ecp_reduce<real>), dim3((unsigned)((a.nb + 63) / 64)), dim3(64), 0, st, a, logq,
                     signq, log0, sign0, e_loc, stats, v_nl_out);
}

// ---- pseudo-Hamiltonian coefficients ----------------------------------------------------------------------------
// Reference: ecp/pseudo_hamiltonian.py:173-234.  For electron i of walker b, with d_I = r_i - R_I, rho_I = |d_I|
// (plain norm, geom/general.py:19-21) and the tabulated radial functions rV_loc / rV_L2 of the PH nuclei (linear
// interpolation on the regular grid [0, r_max], zero outside -- RegularGridInterpolator(fill_value=0), :95-101):
//   A_i = 1/2 I + sum_I [ rV_L2(rho) rho I - rV_L2(rho)/rho d d^T ],    b_i = sum_I 2 rV_L2(rho)/rho d,
//   V_ph(b) = sum_i sum_I rV_loc(rho)/rho.
// Written per (walker, electron): the lower Cholesky factor Q of A (jax.scipy.linalg.cholesky(A, lower=True), :257)
// Q^-1 b (so that b . grad_r = (Q^-1 b) . grad_v, :268-271) and the electron's share of V_ph (k_final sums them).
__device__ __forceinline__ double ph_interp(const double* tab, int n_grid, double inv_h, double rho) {
  const double x = rho * inv_h;
  if (!(x >= 0.0) || x > (double)(n_grid - 1)) return 0.0;
  int k = (int)x;
  if (k > n_grid - 2) k = n_grid - 2;
  const double f = x - k;
  return tab[k] + f * (tab[k + 1] - tab[k]);
}
template <typename real>
__global__ void __launch_bounds__(256) k_ph_coeffs(const real* __restrict__ r, const real* __restrict__ R,
                                                   const int32_t* __restrict__ ph_nuc, int n_ph,
                                                   const double* __restrict__ rv_loc, const double* __restrict__ rv_l2,
                                                   int n_grid, double r_max, int B, int N, double* __restrict__ phq) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * N) return;
  const double inv_h = (double)(n_grid - 1) / r_max;
  const real* ri = r + (long)idx * 3;
  double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 0.5}}, bv[3] = {0, 0, 0}, vloc = 0.0;
  for (int j = 0; j < n_ph; ++j) {
    const real* Rn = R + ph_nuc[j] * 3;
    double d[3], d2 = 0.0;
    for (int c = 0; c < 3; ++c) { d[c] = (double)ri[c] - (double)Rn[c]; d2 += d[c] * d[c]; }
    const double rho = sqrt(d2);
    const double rvl2 = ph_interp(rv_l2 + (long)j * n_grid, n_grid, inv_h, rho);
    const double v = rvl2 / rho;
    vloc += ph_interp(rv_loc + (long)j * n_grid, n_grid, inv_h, rho) / rho;
    for (int a = 0; a < 3; ++a) {
      bv[a] += 2.0 * v * d[a];
      A[a][a] += rvl2 * rho;
      for (int c = 0; c < 3; ++c) A[a][c] -= v * d[a] * d[c];
    }
  }
  // lower Cholesky factor (a non-positive pivot gives NaN, as LAPACK potrf does through jax)
  double Q[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  Q[0][0] = sqrt(A[0][0]);
  Q[1][0] = A[1][0] / Q[0][0];
  Q[2][0] = A[2][0] / Q[0][0];
  Q[1][1] = sqrt(A[1][1] - Q[1][0] * Q[1][0]);
  Q[2][1] = (A[2][1] - Q[2][0] * Q[1][0]) / Q[1][1];
  Q[2][2] = sqrt(A[2][2] - Q[2][0] * Q[2][0] - Q[2][1] * Q[2][1]);
  double y[3];
  y[0] = bv[0] / Q[0][0];
  y[1] = (bv[1] - Q[1][0] * y[0]) / Q[1][1];
  y[2] = (bv[2] - Q[2][0] * y[0] - Q[2][1] * y[1]) / Q[2][2];
  double* o = phq + (long)idx * PH_STRIDE;
  for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) o[3 * a + c] = Q[a][c];
  for (int a = 0; a < 3; ++a) o[9 + a] = y[a];
  o[12] = vloc;
}
template <typename real>
void launch_ph_coeffs(hipStream_t st, const real* r, const real* R, const int32_t* ph_nuc, int n_ph, const double* rv_loc,
                      const double* rv_l2, int n_grid, double r_max, int B, int N, double* phq) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ph_coeffs<real>), dim3((unsigned)(((long)B * N + 255) / 256)), dim3(256), 0, st, r, R,
                     ph_nuc, n_ph, rv_loc, rv_l2, n_grid, r_max, B, N, phq);
}

#define DQMC_INST(real)                                                                                            \
  template void launch_ph_coeffs<real>(hipStream_t, const real*, const real*, const int32_t*, int, const double*,  \
                                       const double*, int, double, int, int, double*);                             \
  template void launch_ecp_points<real>(hipStream_t, const EcpArgs&, real*);                                       \
  template void launch_ecp_reduce<real>(hipStream_t, const EcpArgs&, const real*, const int32_t*, const real*,     \
                                        const int32_t*, real*, real*, real*);
DQMC_INST(float)
DQMC_INST(double)
#undef DQMC_INST

}  // namespace dqmc

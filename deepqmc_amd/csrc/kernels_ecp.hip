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

// ---- mixed-precision quadrature of a float32 context (engine.hip: ecp_mixed) ------------------------------------------------
// The float32 value path puts ~1e-4 on log|psi| of a 30-electron Psiformer, i.e. on every psi ratio; V_nl is a sum of
// 12 N n_ecp of them weighted by (2l+1)/12 V_l(|r_i - R_a|), and V_l is a Gaussian of the electron-nucleus distance: a
// few electron-nucleus pairs carry the sum.  Every (walker, ECP nucleus a, electron i) triple is classified by
//     w = max_l (2l+1) |V_l(|r_i - R_a|)|      (the bound of its contribution per unit of its quadrature mean)
//   w >  w_heavy : its 12 psi ratios are evaluated by the float64 twin (numerator AND denominator in float64),
//   w <= w_skip  : dropped -- its contribution is below w_skip |mean ratio|, ~1e-10 Ha,
//   otherwise    : float32, with psi(r) of the walker evaluated through the SAME float32 value path as the 12 moved
//                  configurations (the errors of numerator and denominator are those of one arithmetic).
// The class depends on the walker's geometry alone.  Classified triples are compacted with an atomic counter: list ORDER
// varies from call to call, results do not (every configuration is evaluated independently and stored at its own slot).
constexpr int32_t ECP_SKIP = INT32_MIN;
__device__ __forceinline__ void ecp_vertex(const double ri[3], const double Rn[3], double phr, int q, double out[3]) {
  const double dx = ri[0] - Rn[0], dy = ri[1] - Rn[1], dz = ri[2] - Rn[2];
  const double radius = sqrt(dx * dx + dy * dy + dz * dz);
  double ct = dz / radius;                                   // ecp_utils.py:53 (clip to [-1, 1])
  ct = ct < -1.0 ? -1.0 : (ct > 1.0 ? 1.0 : ct);
  const double st = sqrt(1.0 - ct * ct);
  const double phi = atan2(dy, dx);
  const double cq = ico_cos(q), sq = sqrt(1.0 - cq * cq), pq = ico_phi(q) + phr;
  const double ux = sq * cos(pq), uy = sq * sin(pq), uz = cq;
  const double vx = ct * ux + st * uz, vy = uy, vz = -st * ux + ct * uz;
  const double cp = cos(phi), sp = sin(phi);
  out[0] = radius * (cp * vx - sp * vy) + Rn[0];
  out[1] = radius * (sp * vx + cp * vy) + Rn[1];
  out[2] = radius * vz + Rn[2];
}
__device__ __forceinline__ double ecp_angle(const EcpMixArgs& a, int b, int j, int i) {
  if (a.phi != nullptr) return (double)a.phi[((long)b * a.n_nl + j) * a.N + i];
  uint32_t o[4];
  philox4x32(a.seed, 0x45435000ull + (uint64_t)j, (uint64_t)b * (uint64_t)a.N + (uint64_t)i, o);
  return u01(o[0], o[1]) * 0.62831853071795865;             // U[0, pi/5), ecp_utils.py:55
}
// one thread per triple t = (bl * n_nl + j) * N + i of the chunk
__global__ void __launch_bounds__(256) k_ecp_classify(const EcpMixArgs a, int32_t* __restrict__ cls, int32_t* __restrict__ list_l,
                                                      int32_t* __restrict__ list_h, int32_t* __restrict__ counts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.nb * a.n_nl * a.N) return;
  const int i = t % a.N, j = (t / a.N) % a.n_nl, bl = t / (a.N * a.n_nl);
  const float* r = a.r + ((long)(a.b0 + bl) * a.N + i) * 3;
  const float* R = a.R + 3 * a.nl_nuc[j];
  double d2 = 0.0;
  for (int c = 0; c < 3; ++c) { const double d = (double)r[c] - (double)R[c]; d2 += d * d; }
  const double* nl = a.nl + (long)j * a.L * 2 * a.n_t;
  double w = 0.0;
  for (int l = 0; l < a.L; ++l) {
    double vl = 0.0;
    for (int k = 0; k < a.n_t; ++k) vl += nl[(l * 2 + 1) * a.n_t + k] * exp(-d2 * nl[(l * 2) * a.n_t + k]);
    w = fmax(w, (2 * l + 1) * fabs(vl));
  }
  if (!(w > a.w_skip)) { cls[t] = ECP_SKIP; return; }
  double amp = 1.0;            // how much worse than ordinary float32 this walker's psi(r) is (EcpMixArgs::l32)
  if (a.l32 != nullptr) {
    const double dl = fabs((double)a.l32[bl] - a.l64[bl]);
    amp = (a.s32[bl] != a.s64[bl] || !(dl == dl)) ? HUGE_VAL : fmax(1.0, dl / a.dlog_floor);
  }
  if (w * amp > a.w_heavy) { const int pos = atomicAdd(&counts[1], 1); list_h[pos] = t; cls[t] = -(pos + 1); }
  else { const int pos = atomicAdd(&counts[0], 1); list_l[pos] = t; cls[t] = pos; }
}
// configurations of one class: [0, nb) the walkers of the chunk themselves, then 12 per listed triple.  One thread per coordinate.
template <typename real_out>
__global__ void __launch_bounds__(256) k_ecp_points_list(const EcpMixArgs a, const int32_t* __restrict__ list, int n_list,
                                                         real_out* __restrict__ rq) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int N = a.N;
  const long per_cfg = (long)N * 3;
  const long n_cfg = (long)a.nb + 12L * n_list;
  if (idx >= n_cfg * per_cfg) return;
  const long cfg = idx / per_cfg;
  const int ec = (int)(idx - cfg * per_cfg), e = ec / 3, c = ec - 3 * e;
  if (cfg < a.nb) { rq[idx] = (real_out)a.r[(long)(a.b0 + cfg) * per_cfg + ec]; return; }
  const long k = cfg - a.nb;
  const int t = list[k / 12], q = (int)(k % 12);
  const int i = t % N, j = (t / N) % a.n_nl, bl = t / (N * a.n_nl);
  const int b = a.b0 + bl;
  const float* r = a.r + (long)b * per_cfg;
  if (e != i) { rq[idx] = (real_out)r[ec]; return; }
  const float* R = a.R + 3 * a.nl_nuc[j];
  const double ri[3] = {(double)r[3 * i], (double)r[3 * i + 1], (double)r[3 * i + 2]};
  const double Rn[3] = {(double)R[0], (double)R[1], (double)R[2]};
  double x[3];
  ecp_vertex(ri, Rn, ecp_angle(a, b, j, i), q, x);
  rq[idx] = (real_out)x[c];
}
// one thread per walker of the chunk: V_nl from the ratios of both classes; added to e_loc, stored in stats[3]
__global__ void __launch_bounds__(64) k_ecp_reduce_mixed(const EcpMixArgs a, const int32_t* __restrict__ cls,
                                                         const float* __restrict__ lq32, const int32_t* __restrict__ sq32,
                                                         const double* __restrict__ lq64, const int32_t* __restrict__ sq64,
                                                         float* __restrict__ e_loc, float* __restrict__ stats) {
  const int bl = blockIdx.x * blockDim.x + threadIdx.x;
  if (bl >= a.nb) return;
  const int b = a.b0 + bl, N = a.N;
  const float* r = a.r + (long)b * N * 3;
  const double l0_32 = (double)lq32[bl], l0_64 = lq64[bl];
  const int s0_32 = sq32[bl], s0_64 = sq64[bl];
  double total = 0.0;
  for (int j = 0; j < a.n_nl; ++j) {
    const float* R = a.R + 3 * a.nl_nuc[j];
    const double* nl = a.nl + (long)j * a.L * 2 * a.n_t;
    for (int i = 0; i < N; ++i) {
      const int32_t c = cls[((long)bl * a.n_nl + j) * N + i];
      if (c == ECP_SKIP) continue;
      double d2 = 0.0;
      for (int x = 0; x < 3; ++x) { const double d = (double)r[3 * i + x] - (double)R[x]; d2 += d * d; }
      double ratio[12];
      if (c >= 0) {
        const long base = (long)a.nb + 12L * c;
        for (int q = 0; q < 12; ++q) ratio[q] = exp((double)lq32[base + q] - l0_32) * (double)(sq32[base + q] * s0_32);
      } else {
        const long base = (long)a.nb + 12L * (-(long)c - 1);
        for (int q = 0; q < 12; ++q) ratio[q] = exp(lq64[base + q] - l0_64) * (double)(sq64[base + q] * s0_64);
      }
      for (int l = 0; l < a.L; ++l) {
        double vl = 0.0;
        for (int t = 0; t < a.n_t; ++t) vl += nl[(l * 2 + 1) * a.n_t + t] * exp(-d2 * nl[(l * 2) * a.n_t + t]);
        if (vl == 0.0) continue;
        double integral = 0.0;
        for (int q = 0; q < 12; ++q) {
          const double x = ico_cos(q);
          double pm = 1.0, p = x;
          double pl = l == 0 ? 1.0 : x;
          for (int m = 2; m <= l; ++m) { pl = ((2 * m - 1) * x * p - (m - 1) * pm) / m; pm = p; p = pl; }
          integral += ratio[q] * pl;
        }
        total += vl * (2 * l + 1) / 12.0 * integral;
      }
    }
  }
  if (e_loc) e_loc[b] = (float)((double)e_loc[b] + total);
  if (stats) stats[3L * a.B + b] = (float)total;
}
void launch_ecp_classify(hipStream_t st, const EcpMixArgs& a, int32_t* cls, int32_t* list_l, int32_t* list_h, int32_t* counts) {
  const long n = (long)a.nb * a.n_nl * a.N;
  hipLaunchKernelGGL(k_ecp_classify, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, cls, list_l, list_h, counts);
}
template <typename real_out> void launch_ecp_points_list(hipStream_t st, const EcpMixArgs& a, const int32_t* list, int n_list, real_out* rq) {
  const long n = ((long)a.nb + 12L * n_list) * a.N * 3;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ecp_points_list<real_out>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, list, n_list, rq);
}
template void launch_ecp_points_list<float>(hipStream_t, const EcpMixArgs&, const int32_t*, int, float*);
template void launch_ecp_points_list<double>(hipStream_t, const EcpMixArgs&, const int32_t*, int, double*);
void launch_ecp_reduce_mixed(hipStream_t st, const EcpMixArgs& a, const int32_t* cls, const float* lq32, const int32_t* sq32,
                             const double* lq64, const int32_t* sq64, float* e_loc, float* stats) {
  hipLaunchKernelGGL(k_ecp_reduce_mixed, dim3((unsigned)((a.nb + 63) / 64)), dim3(64), 0, st, a, cls, lq32, sq32, lq64, sq64, e_loc, stats);
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

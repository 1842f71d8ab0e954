// kernels_mcmc.hip -- Metropolis propose / accept / step-size adaptation, sampler statistics and
// the per-rank energy record.  Reference: sampling/electron_samplers.py:102-163.
#include "common.h"
#include "kernels.h"

namespace dqmc {

// noise ~ N(0,1) (Box-Muller, two per counter), unif ~ U[0,1).
template <typename real>
__global__ void __launch_bounds__(256) k_rng(real* __restrict__ noise, long n_noise, real* __restrict__ unif,
                                             long n_unif, uint64_t seed, uint64_t stream_id) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n_pairs = (n_noise + 1) / 2;
  uint32_t o[4];
  if (idx < n_pairs) {
    philox4x32(seed, 2 * stream_id, (uint64_t)idx, o);
    const double u1 = 1.0 - u01(o[0], o[1]);   // (0,1]
    const double u2 = u01(o[2], o[3]);
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586 * u2;
    noise[2 * idx] = (real)(rad * cos(ang));
    if (2 * idx + 1 < n_noise) noise[2 * idx + 1] = (real)(rad * sin(ang));
  }
  if (idx < n_unif) {
    philox4x32(seed, 2 * stream_id + 1, (uint64_t)idx, o);
    if (sizeof(real) == 4) unif[idx] = (real)((float)(o[0] >> 8) * (1.0f / 16777216.0f));
    else unif[idx] = (real)u01(o[0], o[1]);
  }
}

// r' = r + tau * xi   (electron_samplers.py:102-104)
template <typename real>
__global__ void __launch_bounds__(256) k_propose(const real* __restrict__ r, const real* __restrict__ noise,
                                                 const real* __restrict__ tau, real* __restrict__ r_prop, long n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) r_prop[idx] = r[idx] + tau[0] * noise[idx];
}

// accept = 2 (log|psi'| - log|psi|) > log u  [| age >= max_age];  select r/psi; age bookkeeping
// (electron_samplers.py:106-138).  A NaN proposal compares false and is rejected.
template <typename real>
__global__ void __launch_bounds__(256) k_accept(real* __restrict__ r, real* __restrict__ logpsi,
                                                int32_t* __restrict__ sign, int32_t* __restrict__ age,
                                                const real* __restrict__ r_prop, const real* __restrict__ lp_prop,
                                                const int32_t* __restrict__ sign_prop, const real* __restrict__ unif,
                                                int max_age, int B, int N, int32_t* __restrict__ n_accept,
                                                uint8_t* __restrict__ accept_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const real log_prob = 2 * (lp_prop[b] - logpsi[b]);
  const real lu = sizeof(real) == 4 ? (real)logf((float)unif[b]) : (real)log((double)unif[b]);
  bool acc = log_prob > lu;
  if (max_age >= 0) acc = acc || (age[b] >= max_age);
  if (acc) {
    for (int k = 0; k < 3 * N; ++k) r[(long)b * 3 * N + k] = r_prop[(long)b * 3 * N + k];
    logpsi[b] = lp_prop[b];
    sign[b] = sign_prop[b];
    age[b] = 0;
    atomicAdd(n_accept, 1);
  } else {
    age[b] = age[b] + 1;
  }
  if (accept_out) accept_out[b] = acc ? 1 : 0;
}

// tau <- tau * max(acceptance, 0.05) / target   (electron_samplers.py:121-126); resets the counter.
template <typename real>
__global__ void k_tau_update(real* __restrict__ tau, int32_t* __restrict__ n_accept, int B, double target,
                             double* __restrict__ acc_out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const real acceptance = (real)n_accept[0] / (real)B;
    if (target > 0) {
      const real m = acceptance > (real)0.05 ? acceptance : (real)0.05;
      tau[0] = tau[0] / ((real)target / m);
    }
    acc_out[0] = (double)n_accept[0] / (double)B;
    n_accept[0] = 0;
  }
}

// After the last one-launch sub-step (index s_last) of a call: the step size for the next call from the last
// acceptance count, the acceptance for the sampler statistics, counters back to zero (FusedMc in kernels.h).
template <typename real>
__global__ void k_tau_finalize(real* __restrict__ tau, const real* __restrict__ tau_ring, int32_t* __restrict__ counters,
                               int s_last, int B, double target, double* __restrict__ acc_out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int n = counters[s_last % 3];
    real t = tau_ring[s_last & 1];
    const real acceptance = (real)n / (real)B;
    if (target > 0) {
      const real m = acceptance > (real)0.05 ? acceptance : (real)0.05;
      t = t / ((real)target / m);
    }
    tau[0] = t;
    acc_out[0] = (double)n / (double)B;
    counters[0] = 0; counters[1] = 0; counters[2] = 0;
  }
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum<double>(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double t = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}
__device__ __forceinline__ double block_max(double v, double* sh) {
  for (int m = 1; m < 64; m <<= 1) v = fmax(v, __shfl_xor(v, m, 64));
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double t = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, sh[i]);
  return t;
}

// compute_stats (electron_samplers.py:154-163): acceptance, tau, age mean/max, log|psi| mean/std
// (population), mean pairwise e-e distance.  One 1024-thread block (16 waves: with 256 threads a thread walked 16 walkers x N (N - 1) / 2
// float64 square roots, 29 us on the critical path of every VMC step).
template <typename real>
__global__ void __launch_bounds__(1024) k_sampler_stats(const real* __restrict__ r, const real* __restrict__ logpsi,
                                                       const int32_t* __restrict__ age, const real* __restrict__ tau,
                                                       const double* __restrict__ acc, int B, int N, double eps,
                                                       double* __restrict__ out7) {
  __shared__ double sh[16];
  double s_age = 0, m_age = 0, s_lp = 0, s_d = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    s_age += age[b];
    m_age = fmax(m_age, (double)age[b]);
    s_lp += (double)logpsi[b];
    const real* rb = r + (long)b * 3 * N;
    for (int i = 0; i < N; ++i)
      for (int j = i + 1; j < N; ++j) {
        double d2 = eps;
        for (int c = 0; c < 3; ++c) { const double d = (double)rb[3 * i + c] - (double)rb[3 * j + c]; d2 += d * d; }
        s_d += sqrt(d2);
      }
  }
  s_age = block_sum(s_age, sh);
  m_age = block_max(m_age, sh);
  s_lp = block_sum(s_lp, sh);
  s_d = block_sum(s_d, sh);
  const double mean_lp = s_lp / B;
  double v = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) { const double d = (double)logpsi[b] - mean_lp; v += d * d; }
  v = block_sum(v, sh);
  if (threadIdx.x == 0) {
    out7[0] = acc[0];
    out7[1] = (double)tau[0];
    out7[2] = s_age / B;
    out7[3] = m_age;
    out7[4] = mean_lp;
    out7[5] = sqrt(v / B);
    out7[6] = s_d / ((double)B * (N * (N - 1) / 2));
  }
}

// Per-rank record {n, sum_w, sum_wE, sum_E, M2, min, max} of the energy reduction
// (observable.py:474-479, parallel.py:175-225), merged across ranks on the host.
template <typename real>
__global__ void __launch_bounds__(1024) k_energy_stats(const real* __restrict__ e, const real* __restrict__ w, int B,
                                                       double* __restrict__ out7) {
  __shared__ double sh[16];
  double sw = 0, swe = 0, se = 0, mn = INFINITY, mx = -INFINITY;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const double x = (double)e[b], ww = w ? (double)w[b] : 1.0;
    sw += ww; swe += ww * x; se += x;
    mn = fmin(mn, x); mx = fmax(mx, x);
  }
  sw = block_sum(sw, sh); swe = block_sum(swe, sh); se = block_sum(se, sh);
  mx = block_max(mx, sh);
  mn = -block_max(-mn, sh);
  const double mean = se / B;
  double m2 = 0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) { const double d = (double)e[b] - mean; m2 += d * d; }
  m2 = block_sum(m2, sh);
  if (threadIdx.x == 0) {
    out7[0] = B; out7[1] = sw; out7[2] = swe; out7[3] = se; out7[4] = m2; out7[5] = mn; out7[6] = mx;
  }
}

template <typename real>
void launch_rng(hipStream_t st, real* noise, long n_noise, real* unif, long n_unif, uint64_t seed, uint64_t stream_id) {
  const long n = ((n_noise + 1) / 2 > n_unif) ? (n_noise + 1) / 2 : n_unif;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rng<real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, noise, n_noise,
                     unif, n_unif, seed, stream_id);
}
template <typename real>
void launch_propose(hipStream_t st, const real* r, const real* noise, const real* tau, real* r_prop, long n) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_propose<real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r, noise,
                     tau, r_prop, n);
}
template <typename real>
void launch_accept(hipStream_t st, real* r, real* logpsi, int32_t* sign, int32_t* age, const real* r_prop,
                   const real* logpsi_prop, const int32_t* sign_prop, const real* unif, int max_age, int B, int N,
                   int32_t* n_accept, uint8_t* accept_out) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_accept<real>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, r, logpsi,
                     sign, age, r_prop, logpsi_prop, sign_prop, unif, max_age, B, N, n_accept, accept_out);
}
template <typename real>
void launch_tau_update(hipStream_t st, real* tau, int32_t* n_accept, int B, double target, double* acc_out) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tau_update<real>), dim3(1), dim3(64), 0, st, tau, n_accept, B, target, acc_out);
}
template <typename real>
void launch_tau_finalize(hipStream_t st, real* tau, const real* tau_ring, int32_t* counters, int s_last, int B, double target,
                         double* acc_out) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tau_finalize<real>), dim3(1), dim3(64), 0, st, tau, tau_ring, counters, s_last, B, target,
                     acc_out);
}
template <typename real>
void launch_sampler_stats(hipStream_t st, const real* r, const real* logpsi, const int32_t* age, const real* tau,
                          const double* acc, int B, int N, double eps, double* stats7) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sampler_stats<real>), dim3(1), dim3(1024), 0, st, r, logpsi, age, tau, acc, B, N,
                     eps, stats7);
}
template <typename real>
void launch_energy_stats(hipStream_t st, const real* e_loc, const real* w, int B, double* out7) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_energy_stats<real>), dim3(1), dim3(1024), 0, st, e_loc, w, B, out7);
}

// ---- Metropolis-adjusted Langevin sampler and opposite-spin exchange steps --------------------------------
// Drift cleaning (reference sampling/sampling_utils.py:72-101): one thread per (walker, electron).  z = r_i - R of the
// nearest nucleus; crossover parameter a = (1 + f^.z^)/2 + Z^2 z^2 / (10 (4 + Z^2 z^2)) with the FULL nuclear charge
// Z of that nucleus (mol.charges, not the valence charge); force scaled by 2 / (sqrt(1 + 2 a |f|^2 tau) + 1) and
// then capped so that tau |f| <= |z| (a step cannot overshoot the nearest nucleus).  eps = finfo(real).eps.
template <typename real>
__global__ void __launch_bounds__(256) k_clean_force(const real* __restrict__ grad, const real* __restrict__ r,
                                                     const real* __restrict__ R, const double* __restrict__ Z,
                                                     const real* __restrict__ tau_p, int B, int N, int n_nuc,
                                                     real* __restrict__ force) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * N) return;
  const double eps = sizeof(real) == 4 ? 1.1920928955078125e-07 : 2.220446049250313e-16;
  const double tau = (double)tau_p[0];
  const real* ri = r + idx * 3;
  double f[3] = {(double)grad[idx * 3], (double)grad[idx * 3 + 1], (double)grad[idx * 3 + 2]};
  double z[3] = {0, 0, 0}, z2 = INFINITY, zc = 0;
  for (int n = 0; n < n_nuc; ++n) {                    // argmin keeps the first of equal distances (jnp.argmin)
    double d[3], d2 = 0;
    for (int c = 0; c < 3; ++c) { d[c] = (double)ri[c] - (double)R[n * 3 + c]; d2 += d[c] * d[c]; }
    if (d2 < z2) { z2 = d2; z[0] = d[0]; z[1] = d[1]; z[2] = d[2]; zc = Z[n]; }
  }
  const double zn = sqrt(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  const double f2 = f[0] * f[0] + f[1] * f[1] + f[2] * f[2];
  const double fn = fmax(sqrt(f2), eps);
  const double fz = (f[0] * z[0] + f[1] * z[1] + f[2] * z[2]) / (fn * zn);
  const double Z2z2 = zc * zc * z2;
  const double a = (1.0 + fz) / 2.0 + Z2z2 / (10.0 * (4.0 + Z2z2));
  const double factor = 2.0 / (sqrt(1.0 + 2.0 * a * f2 * tau) + 1.0);
  for (int c = 0; c < 3; ++c) f[c] *= factor;
  const double fn2 = fmax(sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]), eps);
  const double cap = fmin(1.0, sqrt(z2) / (tau * fn2));
  for (int c = 0; c < 3; ++c) force[idx * 3 + c] = (real)(f[c] * cap);
}
// r' = r + tau F + sqrt(tau) xi   (electron_samplers.py:214-221)
template <typename real>
__global__ void __launch_bounds__(256) k_langevin_propose(const real* __restrict__ r, const real* __restrict__ force,
                                                          const real* __restrict__ noise, const real* __restrict__ tau,
                                                          real* __restrict__ r_prop, long n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const real t = tau[0];
  const real st = sizeof(real) == 4 ? (real)sqrtf((float)t) : (real)sqrt((double)t);
  r_prop[idx] = r[idx] + t * force[idx] + st * noise[idx];
}
// log of the Green's-function ratio + 2 (log|psi'| - log|psi|) > log u  [| age >= max_age]; select the state
// including the drift (electron_samplers.py:223-232 + :106-138).  One thread per walker.
template <typename real>
__global__ void __launch_bounds__(256) k_langevin_accept(real* __restrict__ r, real* __restrict__ logpsi,
                                                         int32_t* __restrict__ sign, int32_t* __restrict__ age,
                                                         real* __restrict__ force, const real* __restrict__ r_prop,
                                                         const real* __restrict__ lp_prop, const int32_t* __restrict__ sign_prop,
                                                         const real* __restrict__ force_prop, const real* __restrict__ unif,
                                                         const real* __restrict__ tau_p, int max_age, int B, int N,
                                                         int32_t* __restrict__ n_accept, uint8_t* __restrict__ accept_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const real tau = tau_p[0];
  const long o = (long)b * 3 * N;
  real log_G = 0;
  for (int k = 0; k < 3 * N; ++k) {
    const real f0 = force[o + k], f1 = force_prop[o + k];
    log_G += (f0 + f1) * ((r[o + k] - r_prop[o + k]) + tau / 2 * (f0 - f1));
  }
  const real log_prob = log_G + 2 * (lp_prop[b] - logpsi[b]);
  const real lu = sizeof(real) == 4 ? (real)logf((float)unif[b]) : (real)log((double)unif[b]);
  bool acc = log_prob > lu;
  if (max_age >= 0) acc = acc || (age[b] >= max_age);
  if (acc) {
    for (int k = 0; k < 3 * N; ++k) { r[o + k] = r_prop[o + k]; force[o + k] = force_prop[o + k]; }
    logpsi[b] = lp_prop[b];
    sign[b] = sign_prop[b];
    age[b] = 0;
    atomicAdd(n_accept, 1);
  } else {
    age[b] = age[b] + 1;
  }
  if (accept_out) accept_out[b] = acc ? 1 : 0;
}
// r' = r with the positions of spin-up electron up_idx[b] and spin-down electron n_up + down_idx[b] swapped
// (electron_samplers.py:277-288)
template <typename real>
__global__ void __launch_bounds__(256) k_exchange_propose(const real* __restrict__ r, const int32_t* __restrict__ up_idx,
                                                          const int32_t* __restrict__ down_idx, int n_up, int B, int N,
                                                          real* __restrict__ r_prop) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * N) return;
  const int b = (int)(idx / N), i = (int)(idx - (long)b * N);
  // indices are clamped into their spin blocks: a caller's out-of-range choice can never read outside r
  const int n_dn = N - n_up, ui = up_idx[b], di = down_idx[b];
  const int u = ui < 0 ? 0 : (ui >= n_up ? n_up - 1 : ui), d = n_up + (di < 0 ? 0 : (di >= n_dn ? n_dn - 1 : di));
  const int src = i == u ? d : (i == d ? u : i);
  for (int c = 0; c < 3; ++c) r_prop[idx * 3 + c] = r[((long)b * N + src) * 3 + c];
}
// acceptance of a step whose count sits in n_accept -> acc_out, counter back to zero (no step-size change)
__global__ void k_read_accept(int32_t* __restrict__ n_accept, int B, double* __restrict__ acc_out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { acc_out[0] = (double)n_accept[0] / (double)B; n_accept[0] = 0; }
}
template <typename real>
void launch_clean_force(hipStream_t st, const real* grad, const real* r, const real* R, const double* Z, const real* tau, int B,
                        int N, int n_nuc, real* force) {
  const long n = (long)B * N;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_clean_force<real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, grad, r, R, Z, tau,
                     B, N, n_nuc, force);
}
template <typename real>
void launch_langevin_propose(hipStream_t st, const real* r, const real* force, const real* noise, const real* tau, real* r_prop, long n) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_langevin_propose<real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r, force, noise,
                     tau, r_prop, n);
}
template <typename real>
void launch_langevin_accept(hipStream_t st, real* r, real* logpsi, int32_t* sign, int32_t* age, real* force, const real* r_prop,
                            const real* lp_prop, const int32_t* sign_prop, const real* force_prop, const real* unif, const real* tau,
                            int max_age, int B, int N, int32_t* n_accept, uint8_t* accept_out) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_langevin_accept<real>), dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, r, logpsi, sign,
                     age, force, r_prop, lp_prop, sign_prop, force_prop, unif, tau, max_age, B, N, n_accept, accept_out);
}
template <typename real>
void launch_exchange_propose(hipStream_t st, const real* r, const int32_t* up_idx, const int32_t* down_idx, int n_up, int B, int N,
                             real* r_prop) {
  const long n = (long)B * N;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_exchange_propose<real>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, r, up_idx,
                     down_idx, n_up, B, N, r_prop);
}
void launch_read_accept(hipStream_t st, int32_t* n_accept, int B, double* acc_out) {
  hipLaunchKernelGGL(k_read_accept, dim3(1), dim3(64), 0, st, n_accept, B, acc_out);
}

// ---- float64 refinement of flagged walkers (engine.hip: Engine<float>::lap_refined) ----
// The list idx[0..n) is either complete on the host side (count == nullptr) or still being produced on the device: then
// `count` points at the number of flagged walkers and n is the CAPACITY the float64 pass was enqueued for -- slots past
// min(*count, n) are filled with a valid walker (the first flagged one, or walker 0) and not scattered back, so the
// host never has to wait for the count before it enqueues the pass.
// gather: positions of the flagged walkers (and the geometry) widened to double
__global__ void __launch_bounds__(256) k_refine_gather(const float* __restrict__ r, const float* __restrict__ R,
                                                       const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int n,
                                                       int n3, int nR3, double* __restrict__ r64, double* __restrict__ R64) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nR3) R64[e] = (double)R[e];
  if (e >= (long)n * n3) return;
  const int j = (int)(e / n3), c = (int)(e - (long)j * n3);
  long b;
  if (count) {
    const int live = *count < n ? *count : n;
    b = live > 0 ? idx[j < live ? j : 0] : 0;
  } else {
    b = idx[j];
  }
  r64[e] = (double)r[b * n3 + c];
}
// scatter: the float64 results replace the float32 ones of those walkers
__global__ void __launch_bounds__(256) k_refine_scatter(const int32_t* __restrict__ idx, const int32_t* __restrict__ count, int n,
                                                        int n_scatter, const double* __restrict__ score, double thresh, int n3,
                                                        const double* __restrict__ e64, const double* __restrict__ st64,
                                                        const double* __restrict__ g64, const double* __restrict__ lp64,
                                                        const int32_t* __restrict__ sg64, float* __restrict__ e_loc,
                                                        float* __restrict__ stats, long stats_ld, float* __restrict__ grad,
                                                        float* __restrict__ logpsi, int32_t* __restrict__ sign) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_scatter || (count && j >= *count)) return;
  const long b = idx[j];
  if (score && score[b] <= thresh) return;      // probe call: evaluated under the old threshold, not flagged under the new one
  if (e_loc) e_loc[b] = (float)e64[j];
  if (stats) for (int k = 0; k < 6; ++k) stats[k * stats_ld + b] = (float)st64[(long)k * n + j];
  if (grad) for (int c = 0; c < n3; ++c) grad[b * n3 + c] = (float)g64[(long)j * n3 + c];
  if (logpsi) logpsi[b] = (float)lp64[j];
  if (sign) sign[b] = sg64[j];
}
// ---- float64 tail of a float32 forward-Laplacian pass (engine_refine.inl: run_tail) ----
// widen: a float32 activation buffer (or the walker positions) into the float64 twin's workspace, four elements per thread
__global__ void __launch_bounds__(256) k_widen(const float* __restrict__ src, double* __restrict__ dst, long n) {
  const long e = 4 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (e + 3 < n && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {      // (workspace buffers are 256-byte aligned; a caller's r / R view may not be)
    const Vec4<float> v = *reinterpret_cast<const Vec4<float>*>(src + e);
    *reinterpret_cast<Vec2<double>*>(dst + e) = Vec2<double>{{(double)v.v[0], (double)v.v[1]}};
    *reinterpret_cast<Vec2<double>*>(dst + e + 2) = Vec2<double>{{(double)v.v[2], (double)v.v[3]}};
  } else {
    for (long k = e; k < n && k < e + 4; ++k) dst[k] = (double)src[k];
  }
}
void launch_widen(hipStream_t st, const float* src, double* dst, long n) {
  if (n < 1) return;
  hipLaunchKernelGGL(k_widen, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, st, src, dst, n);
}
// narrow: the float64 results of the tail -> the caller's float32 arrays (walker j of the chunk; stats columns with the
// caller's leading dimension)
__global__ void __launch_bounds__(256) k_tail_narrow(int n, int n3, const double* __restrict__ e64, const double* __restrict__ st64,
                                                     const double* __restrict__ g64, const double* __restrict__ lp64,
                                                     const int32_t* __restrict__ sg64, float* __restrict__ e_loc,
                                                     float* __restrict__ stats, long stats_ld, float* __restrict__ grad,
                                                     float* __restrict__ logpsi, int32_t* __restrict__ sign) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (grad && e < (long)n * n3) grad[e] = (float)g64[e];
  if (e >= n) return;
  if (e_loc) e_loc[e] = (float)e64[e];
  if (stats) for (int k = 0; k < 6; ++k) stats[k * stats_ld + e] = (float)st64[(long)k * n + e];
  if (logpsi) logpsi[e] = (float)lp64[e];
  if (sign) sign[e] = sg64[e];
}
void launch_tail_narrow(hipStream_t st, int n, int n3, const double* e64, const double* st64, const double* g64, const double* lp64,
                        const int32_t* sg64, float* e_loc, float* stats, long stats_ld, float* grad, float* logpsi, int32_t* sign) {
  const long tot = grad ? (long)n * n3 : n;
  hipLaunchKernelGGL(k_tail_narrow, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, n, n3, e64, st64, g64, lp64, sg64, e_loc, stats,
                     stats_ld, grad, logpsi, sign);
}
void launch_refine_gather(hipStream_t st, const float* r, const float* R, const int32_t* idx, const int32_t* count, int n, int n3,
                          int nR3, double* r64, double* R64) {
  long tot = (long)n * n3;
  if (tot < nR3) tot = nR3;
  hipLaunchKernelGGL(k_refine_gather, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, r, R, idx, count, n, n3, nR3, r64, R64);
}
void launch_refine_scatter(hipStream_t st, const int32_t* idx, const int32_t* count, int n, int n_scatter, const double* score,
                           double thresh, int n3, const double* e64, const double* st64,
                           const double* g64, const double* lp64, const int32_t* sg64, float* e_loc, float* stats,
                           long stats_ld, float* grad, float* logpsi, int32_t* sign) {
  if (n_scatter < 1) return;
  hipLaunchKernelGGL(k_refine_scatter, dim3((unsigned)((n_scatter + 255) / 256)), dim3(256), 0, st, idx, count, n, n_scatter, score, thresh, n3, e64, st64, g64, lp64,
                     sg64, e_loc, stats, stats_ld, grad, logpsi, sign);
}

#define DQMC_INST(real)                                                                                             \
  template void launch_rng<real>(hipStream_t, real*, long, real*, long, uint64_t, uint64_t);                        \
  template void launch_propose<real>(hipStream_t, const real*, const real*, const real*, real*, long);              \
  template void launch_accept<real>(hipStream_t, real*, real*, int32_t*, int32_t*, const real*, const real*,        \
                                    const int32_t*, const real*, int, int, int, int32_t*, uint8_t*);                \
  template void launch_tau_update<real>(hipStream_t, real*, int32_t*, int, double, double*);                        \
  template void launch_tau_finalize<real>(hipStream_t, real*, const real*, int32_t*, int, int, double, double*);    \
  template void launch_sampler_stats<real>(hipStream_t, const real*, const real*, const int32_t*, const real*,      \
                                           const double*, int, int, double, double*);                               \
  template void launch_energy_stats<real>(hipStream_t, const real*, const real*, int, double*);                     \
  template void launch_clean_force<real>(hipStream_t, const real*, const real*, const real*, const double*, const real*, int, int, \
                                         int, real*);                                                               \
  template void launch_langevin_propose<real>(hipStream_t, const real*, const real*, const real*, const real*, real*, long); \
  template void launch_langevin_accept<real>(hipStream_t, real*, real*, int32_t*, int32_t*, real*, const real*, const real*, \
                                             const int32_t*, const real*, const real*, const real*, int, int, int, int32_t*, \
                                             uint8_t*);                                                             \
  template void launch_exchange_propose<real>(hipStream_t, const real*, const int32_t*, const int32_t*, int, int, int, real*);
DQMC_INST(float)
DQMC_INST(double)
#undef DQMC_INST

}  // namespace dqmc

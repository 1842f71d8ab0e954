// common.h -- shared device helpers for the gfx950 local-energy kernels.
//
// Activation layout (include/dqmc.h): real X[B][rows][TP][width]; element
// (b,row,t,col) at ((b*rows + row)*TP + t)*width + col.  Lane t = 0 is the value,
// t = 1..3N the derivative w.r.t. electron coordinate c = t-1, t = 3N+1 the Laplacian.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dqmc {

struct LaneInfo {
  int T;    // used lanes (1 or 3N+2)
  int TP;   // allocated lanes (1 or round_up(T,16))
  int N;    // electrons
};

// MFMA 16x16x4 wrappers.  A operand: lane l holds A[row = l&15][k = l>>4]; B operand:
// B[k = l>>4][col = l&15]; C/D: col = l&15, four rows per lane -- f32: (l>>4)*4 + reg,
// f64: (l>>4) + 4*reg (MI355X guide, "Fragment layout").
typedef float f32x4 __attribute__((vector_size(16)));
typedef double f64x4 __attribute__((vector_size(32)));

template <typename real> struct Mfma;
template <> struct Mfma<float> {
  typedef f32x4 acc_t;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row_of(int lane, int reg) { return ((lane >> 4) << 2) + reg; }
};
template <> struct Mfma<double> {
  typedef f64x4 acc_t;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row_of(int lane, int reg) { return (lane >> 4) + (reg << 2); }
};

// ---- float32 products on the bf16 matrix pipe ----
// gfx950 has no fast f32 MFMA (v_mfma_f32_16x16x4_f32 runs at the VALU rate, 1/16 of the bf16 rate).  A float is
// EXACTLY the sum of three bf16 numbers (8 + 8 + 8 significand bits, round-to-nearest residuals), so
//   a b = sum_{i,j} a_i b_j,   a = a_0 + a_1 + a_2,  b = b_0 + b_1 + b_2,
// and the six products with i + j <= 2 carry everything above 2^-24 |a b| (the dropped ones are <= 2^-24, 2^-24,
// 2^-32 |a b| with random signs: the same order as the rounding of an f32 accumulation).  Six
// v_mfma_f32_16x16x32_bf16 (K = 32 each, 16 cycles) replace eight v_mfma_f32_16x16x4_f32 (K = 4 each, 32 cycles):
// 96 instead of 256 matrix-pipe cycles per 16 x 16 x 32 block, accumulated in f32 as before.
// Operand layout of v_mfma_f32_16x16x32_bf16: lane l holds row (A) / column (B) l & 15 and the eight k values
// 8 (l >> 4) .. 8 (l >> 4) + 7, two per register, even k in the low half; C/D as the 16x16x4 f32 form.
struct BfFrag { uint32_t w[4]; };
#if defined(__HIPCC__)
typedef __bf16 dqmc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 dqmc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float dqmc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf_pack2(float lo, float hi) {       // v_cvt_pk_bf16_f32 (round to nearest even)
  const dqmc_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dqmc_bf16x2));
}
__device__ __forceinline__ f32x4 mfma_bf16(const BfFrag& a, const BfFrag& b, f32x4 c) {
  typedef float cvec __attribute__((ext_vector_type(4)));
  const cvec r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dqmc_bf16x8, a), __builtin_bit_cast(dqmc_bf16x8, b),
                                                        __builtin_bit_cast(cvec, c), 0, 0, 0);
  return __builtin_bit_cast(f32x4, r);
}
#else
// (SIMT emulator build: bf_pack2 and the MFMA come from tests/simt/hip/hip_runtime.h)
inline f32x4 mfma_bf16(const BfFrag& a, const BfFrag& b, f32x4 c) { return simt_mfma_f32_16x16x32_bf16(a.w, b.w, c); }
#endif
__device__ __forceinline__ float bf_lo_as_float(uint32_t w) { uint32_t u = w << 16; float f; __builtin_memcpy(&f, &u, 4); return f; }
__device__ __forceinline__ float bf_hi_as_float(uint32_t w) { uint32_t u = w & 0xffff0000u; float f; __builtin_memcpy(&f, &u, 4); return f; }
// eight consecutive floats -> the three bf16 pieces of each (44 VALU instructions)
__device__ __forceinline__ void bf_split8(const float (&a)[8], BfFrag& p0, BfFrag& p1, BfFrag& p2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a0 = a[2 * j], a1 = a[2 * j + 1];
    const uint32_t h = bf_pack2(a0, a1);
    const float r0 = a0 - bf_lo_as_float(h), r1 = a1 - bf_hi_as_float(h);
    const uint32_t m = bf_pack2(r0, r1);
    const float s0 = r0 - bf_lo_as_float(m), s1 = r1 - bf_hi_as_float(m);
    p0.w[j] = h; p1.w[j] = m; p2.w[j] = bf_pack2(s0, s1);
  }
}

template <typename real> struct Vec4;   // 4 consecutive reals, naturally aligned
template <> struct __attribute__((aligned(16))) Vec4<float> { float v[4]; };
template <> struct __attribute__((aligned(16))) Vec4<double> { double v[4]; };      // (two 16-byte accesses; 32-byte alignment kept conditionally assigned prefetch registers in scratch)

template <typename real> struct Vec2;   // 2 consecutive reals, naturally aligned
template <> struct __attribute__((aligned(8))) Vec2<float> { float v[2]; };
template <> struct __attribute__((aligned(16))) Vec2<double> { double v[2]; };

// VW consecutive reals as a value (VW = 1 or 4): element-parallel kernels are written once for both widths
template <typename real, int VW> struct VecN {
  real v[VW];
  static __device__ __forceinline__ VecN zero() { VecN r; for (int j = 0; j < VW; ++j) r.v[j] = 0; return r; }
  static __device__ __forceinline__ VecN load(const real* p) {
    VecN r;
    if (VW == 4) { const Vec4<real> t = *reinterpret_cast<const Vec4<real>*>(p); for (int j = 0; j < VW; ++j) r.v[j] = t.v[j]; }
    else for (int j = 0; j < VW; ++j) r.v[j] = p[j];
    return r;
  }
  __device__ __forceinline__ void store(real* p) const {
    if (VW == 4) { Vec4<real> t; for (int j = 0; j < 4; ++j) t.v[j] = v[j < VW ? j : 0]; *reinterpret_cast<Vec4<real>*>(p) = t; }
    else for (int j = 0; j < VW; ++j) p[j] = v[j];
  }
  __device__ __forceinline__ void fma(const VecN& a, const VecN& b) { for (int j = 0; j < VW; ++j) v[j] += a.v[j] * b.v[j]; }
  __device__ __forceinline__ void axpy(real s, const VecN& a) { for (int j = 0; j < VW; ++j) v[j] += s * a.v[j]; }
};

// a * b + c with one rounding
template <typename real> __device__ __forceinline__ real r_fma(real a, real b, real c);
template <> __device__ __forceinline__ float r_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
template <> __device__ __forceinline__ double r_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename real> __device__ __forceinline__ real r_tanh(real x);
// f32 tanh in ~15 VALU instructions (libm tanhf is ~100 and dominated the small layers of the
// fused kernel): odd polynomial for |x| < 0.25 (truncation < 3e-9), (1 - e)/(1 + e) with
// e = exp(-2|x|) on the hardware exp2 / rcp otherwise.  Max error ~2 ulp of the result.
template <> __device__ __forceinline__ float r_tanh<float>(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float poly = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
  const float e = __expf(-2.0f * ax);
  const float big = copysignf(__fdividef(1.0f - e, 1.0f + e), x);
  return ax < 0.25f ? poly : big;
}
template <> __device__ __forceinline__ double r_tanh<double>(double x) { return tanh(x); }
template <typename real> __device__ __forceinline__ real r_exp(real x);
template <> __device__ __forceinline__ float r_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double r_exp<double>(double x) { return exp(x); }

// Ordering point between LDS writes of a wave and reads of the same locations by other lanes of that wave.
// The hardware executes a wave's LDS operations in order, so only the compiler (and the CPU emulation harness,
// where lanes are fibers) needs a barrier here.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_wave_barrier();
}

// Sum over the four 16-lane quads of a wave (same lane&15).
template <typename real> __device__ __forceinline__ real quad_sum(real v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
template <typename real> __device__ __forceinline__ real wave_sum(real v) {
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based: (seed, stream, index) -> 4 x u32 ----
__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32(uint64_t seed, uint64_t stream, uint64_t idx, uint32_t out[4]) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {  // [0,1) with 53 bits
  return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

// Pair-compact lanes.  An edge row (receiver i, sender j) of the two-particle stream depends on the coordinates
// of electrons i and j only (reference gnn/graph.py:23-31: d = r_recv - r_send; the edge MLPs act row-wise), so
// of its 3N derivative lanes at most 6 are non-zero.  Edge buffers therefore carry 8 lanes instead of
// round_up(3N+2, 16):   0 value | 1..3 d/dr_recv | 4..6 d/dr_send | 7 Laplacian
// (a nuclear sender or a self edge leaves lanes 4..6 zero).  pair_lane() maps a full lane t to the compact
// lane of edge (recv, send), or -1 if that lane is identically zero.  This is the block sparsity that folx
// exploits for the reference (last paragraph).
constexpr int PAIR_LANES = 8;
// Pseudo-Hamiltonian coefficients per (walker, electron), double[PH_STRIDE]: the row-major 3x3 lower Cholesky
// factor Q of A (9), Q^-1 b (3), and the electron's local PH potential (1); written by k_ph_coeffs (kernels_ecp.hip).
constexpr int PH_STRIDE = 13;
__device__ __forceinline__ int pair_lane(int t, int T, int recv, int send) {
  if (t == 0) return 0;
  if (t == T - 1) return PAIR_LANES - 1;
  const int e = (t - 1) / 3, x = (t - 1) - 3 * e;
  if (e == recv) return 1 + x;
  if (e == send) return 4 + x;
  return -1;
}
// inverse: full lane of compact lane ct (or -1: a lane that is zero for this edge)
__device__ __forceinline__ int pair_lane_full(int ct, int T, int recv, int send) {
  if (ct == 0) return 0;
  if (ct == PAIR_LANES - 1) return T - 1;
  if (ct < 4) return 1 + 3 * recv + (ct - 1);
  if (send < 0 || send == recv) return -1;
  return 1 + 3 * send + (ct - 4);
}

// Lane-t value of the pair features [rho, d_x, d_y, d_z] (optionally * log1p(rho)/rho) of the
// difference d = r_recv - r_send (send < 0: a nucleus, no dependence), rho = sqrt(eps + d.d).
// Reference: gnn/edge_features.py:21-123 + utils.py:79-85; derivative lanes per SURVEY.md
// appendix C.  All arithmetic in double (inputs are 3 coordinates; cost is negligible).
//
// Pseudo-Hamiltonian seeding (reference ecp/pseudo_hamiltonian.py:115-146): with Q_recv / Q_send given (row-major
// 3x3 lower Cholesky factors of the per-electron matrices A = Q Q^T), derivative lane (e, x) is the derivative
// along column x of Q_e (the coordinate change r_e = Q_e v_e with Q held fixed) and the last lane is
// sum_e tr(A_e Hess_e f) instead of the Laplacian.  NULL factors = the identity = the plain Laplacian.
__device__ __forceinline__ void pair_feature_lane(const double d[3], double eps, int recv, int send, int t,
                                                  const LaneInfo li, bool log_rescale, double out[4],
                                                  const double* Qr = nullptr, const double* Qs = nullptr) {
  const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const double rho = sqrt(eps + d2);
  const int n_ends = (recv == send) ? 0 : (send >= 0 ? 2 : 1);
  // A_eff = sum over the electron ends of Q Q^T:  trA = tr(A_eff),  Ad = A_eff d,  dAd = d.A_eff.d
  double trA = 3.0 * n_ends, dAd = n_ends * d2, Ad[3] = {n_ends * d[0], n_ends * d[1], n_ends * d[2]};
  if (Qr && n_ends) {
    trA = 0.0; dAd = 0.0; Ad[0] = Ad[1] = Ad[2] = 0.0;
    for (int end = 0; end < n_ends; ++end) {
      const double* Q = end ? Qs : Qr;
      for (int c = 0; c < 3; ++c) {                    // column c of Q
        const double qd = Q[0 + c] * d[0] + Q[3 + c] * d[1] + Q[6 + c] * d[2];
        dAd += qd * qd;
        for (int a = 0; a < 3; ++a) { Ad[a] += Q[3 * a + c] * qd; }
        trA += Q[c] * Q[c] + Q[3 + c] * Q[3 + c] + Q[6 + c] * Q[6 + c];
      }
    }
  }
  // lane of rho and of the three components of d
  double rho_t = 0.0, dc_t[3] = {0.0, 0.0, 0.0};
  double sgn = 0.0;
  int x = -1;
  if (t == 0) {
    rho_t = rho;
    dc_t[0] = d[0]; dc_t[1] = d[1]; dc_t[2] = d[2];
  } else if (t < li.T - 1) {
    const int c = t - 1, e = c / 3;
    x = c - 3 * e;
    sgn = (e == recv ? 1.0 : 0.0) - (e == send ? 1.0 : 0.0);
    if (Qr && sgn != 0.0) {
      const double* Q = (e == recv) ? Qr : Qs;
      for (int a = 0; a < 3; ++a) dc_t[a] = sgn * Q[3 * a + x];
      rho_t = (dc_t[0] * d[0] + dc_t[1] * d[1] + dc_t[2] * d[2]) / rho;
    } else {
      rho_t = sgn * d[x] / rho;
      dc_t[x] = sgn;
    }
  } else if (t == li.T - 1) {
    rho_t = trA / rho - dAd / (rho * rho * rho);
  }
  if (!log_rescale) {
    out[0] = rho_t; out[1] = dc_t[0]; out[2] = dc_t[1]; out[3] = dc_t[2];
    return;
  }
  const double l1 = log1p(rho), ir = 1.0 / (1.0 + rho);
  const double s = l1 / rho;
  const double s1 = (ir - s) / rho;
  const double s2 = (-ir * ir - 2.0 * s1) / rho;
  const double sumJ2 = dAd / (rho * rho);                     // sum_c rho_c^2
  if (t == 0) {
    out[0] = l1;
    for (int a = 0; a < 3; ++a) out[1 + a] = d[a] * s;
  } else if (t < li.T - 1) {
    out[0] = rho_t * ir;
    for (int a = 0; a < 3; ++a) out[1 + a] = dc_t[a] * s + d[a] * s1 * rho_t;
  } else if (t == li.T - 1) {
    out[0] = rho_t * ir - sumJ2 * ir * ir;
    const double sL = s1 * rho_t + s2 * sumJ2;
    for (int a = 0; a < 3; ++a) out[1 + a] = d[a] * sL + 2.0 * s1 * Ad[a] / rho;
  } else {
    out[0] = out[1] = out[2] = out[3] = 0.0;
  }
}

}  // namespace dqmc

// spec_device.h -- device helpers shared by the PLAN-SPECIALISED Metropolis sub-step kernels (csrc/gen/*.hip, written by
// deepqmc_amd/codegen/substep.py from a layer program; reference sampling/electron_samplers.py:76-138).
//
// Layout of such a kernel ("row-in-lanes"): ONE WAVE owns a tile of 16 / N walkers and keeps every activation in registers.
// An MFMA computes Y^T = W^T X^T: the A operand is a 16 x 32 tile of W^T (a 1 KB fragment of the weight tape, read from the LDS
// ring with one ds_read_b128 per lane), the B operand is X^T -- lane l holds the activation row c = l & 15 = (walker, electron)
// and the eight k-slots 8 (l >> 4) .. + 7 --, and the result leaves lane l with output features 16 b + 4 (l >> 4) + 0..3 of the
// SAME row c: exactly what the next layer needs as ITS B operand (k-slot order within a 32-chunk is free as long as the packed
// weights use the same order).  So the layers chain through registers without an LDS round trip, the row-wise graph ops of the
// GNN (sender gather, spin means, sums over electrons) are permutations inside the 4-lane quad of a walker (DPP), and the LDS
// holds nothing but the weight ring shared by the four waves (= four tiles) of a workgroup.
#pragma once

#include "common.h"
#include "kernels.h"

namespace dqmc {

// LU with partial pivoting of one N x N Slater matrix (LDS, `real` entries as the ORBITALS op would have stored
// them) in double registers: sign and log|det| -- the arithmetic of k_slogdet_small, value lane only.
template <typename real, int N>
__device__ __forceinline__ void fused2_det(const real* m, double& logabs, int& sgn) {
  double A[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) A[i][j] = (double)m[i * N + j];
  logabs = 0.0;
  sgn = 1;
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
        const double a0 = A[p][j], a1 = A[i][j];
        A[p][j] = sw ? a1 : a0; A[i][j] = sw ? a0 : a1;
      }
    }
    if (best != p) sgn = -sgn;
    const double piv = A[p][p];
    logabs += log(fabs(piv));
    if (piv < 0) sgn = -sgn;
    if (piv == 0) sgn = 0;
    const double ip = 1.0 / piv;
#pragma unroll
    for (int j = 0; j < N; ++j) A[p][j] *= ip;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i == p) continue;
      const double f = A[i][p];
#pragma unroll
      for (int j = 0; j < N; ++j) A[i][j] -= f * A[p][j];
    }
  }
}

namespace spec {

typedef uint32_t u32;

// 16 bytes per lane: one ds_read_b128 / global_load_dwordx4 / ds_write_b128 (a vector type: a struct of four words was kept
// in scratch between its load and its LDS store)
typedef u32 V16 __attribute__((vector_size(16)));
__device__ __forceinline__ BfFrag ld_frag(const char* p) {
  const V16 v = *reinterpret_cast<const V16*>(p);
  BfFrag f;
  f.w[0] = v[0]; f.w[1] = v[1]; f.w[2] = v[2]; f.w[3] = v[3];
  return f;
}

// 16 bytes of the weight tape per lane: a buffer load whose constant part travels in a scalar register (no address arithmetic
// on the vector pipe; a flat global_load needs two 64-bit VALU adds per constant offset beyond 4 KB).
struct TapeRsrc {
#if defined(__HIPCC__)
  __amdgpu_buffer_rsrc_t rs;
#else
  const char* base;
#endif
};
__device__ __forceinline__ TapeRsrc tape_rsrc(const void* tape, int bytes) {
  TapeRsrc t;
#if defined(__HIPCC__)
  t.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tape), 0, bytes, 0x00020000);
#else
  t.base = reinterpret_cast<const char*>(tape);
#endif
  return t;
}
template <int OFF> __device__ __forceinline__ V16 tape_load(const TapeRsrc& t, int lane_bytes) {
#if defined(__HIPCC__)
  typedef u32 u4_ __attribute__((ext_vector_type(4)));
  const u4_ v = __builtin_amdgcn_raw_buffer_load_b128(t.rs, lane_bytes, OFF, 0);
  return __builtin_bit_cast(V16, v);
#else
  return *reinterpret_cast<const V16*>(t.base + OFF + lane_bytes);
#endif
}

__device__ __forceinline__ void st_frag(char* p, const BfFrag& f) {
  V16 v;
  v[0] = f.w[0]; v[1] = f.w[1]; v[2] = f.w[2]; v[3] = f.w[3];
  *reinterpret_cast<V16*>(p) = v;
}

// Barrier between two stages of the weight ring.  A wave's own ring stores are older than its last K LDS reads (the generator
// places them so), LDS operations complete in order, so "at most K outstanding" means the stores have landed; the prefetched
// fragment reads stay in flight across the barrier (__syncthreads() would drain them: s_waitcnt lgkmcnt(0)).
template <int K> __device__ __forceinline__ void ring_barrier() {
#if defined(__HIPCC__)
  asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(K) : "memory");
#else
  __syncthreads();
#endif
}

// Stage boundary of a ring that a producer wave fills: the computing waves neither store to the ring nor need their
// prefetched fragment reads drained -- a bare s_barrier (the compiler may not move LDS accesses across it).
__device__ __forceinline__ void ring_barrier0() {
#if defined(__HIPCC__)
  asm volatile("s_barrier" ::: "memory");
#else
  __syncthreads();
#endif
}

// v from lane (l & ~3) | P[l & 3] of the same 4-lane quad
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ float quad_perm(float v) {
#if defined(__HIPCC__)
  const int r = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
  return __builtin_bit_cast(float, r);
#else
  const int l = simt::lane_id();
  const int p[4] = {P0, P1, P2, P3};
  return __shfl(v, (l & ~3) | p[l & 3], 64);
#endif
}
// acc + w * h[quad_perm] and a + b[quad_perm] as ONE VALU instruction each (v_fmac_f32_dpp / v_add_f32_dpp: the permuted operand rides in
// the instruction's DPP modifier; the compiler keeps the v_mov_b32_dpp of the builtin as an instruction of its own)
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ float quad_fmac(float acc, float w, float h) {
#if defined(__HIPCC__)
  asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[%3,%4,%5,%6] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(h), "v"(w), "n"(P0), "n"(P1), "n"(P2), "n"(P3));
  return acc;
#else
  return __builtin_fmaf(w, quad_perm<P0, P1, P2, P3>(h), acc);
#endif
}
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ float quad_add(float a, float b) {
#if defined(__HIPCC__)
  float r;
  asm("v_add_f32_dpp %0, %1, %2 quad_perm:[%3,%4,%5,%6] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(b), "v"(a), "n"(P0), "n"(P1), "n"(P2), "n"(P3));
  return r;
#else
  return a + quad_perm<P0, P1, P2, P3>(b);
#endif
}
template <int X> __device__ __forceinline__ float quad_xor(float v) { return quad_perm<(0 ^ X), (1 ^ X), (2 ^ X), (3 ^ X)>(v); }
template <int E> __device__ __forceinline__ float quad_bcast(float v) { return quad_perm<E, E, E, E>(v); }

// tanh of the value path (as kernel_fused2.hip: 1 - 2 / (1 + e^{2x}) on the hardware exp2 / rcp units)
__device__ __forceinline__ float tanh_value(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
// the same with the factor 2 log2(e) already in the argument (folded into the layer's packed weights and bias)
__device__ __forceinline__ float tanh_scaled(float y) { return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y) + 1.0f), 1.0f); }
__device__ __forceinline__ float silu_value(float v) { return v / (1 + r_exp<float>(-v)); }

// two floats -> their three bf16 pieces, packed (low half = a, high half = b); the residuals are exact
__device__ __forceinline__ void split2(float a, float b, u32& h, u32& m, u32& l) {
  h = bf_pack2(a, b);
  const float r0 = a - bf_lo_as_float(h), r1 = b - bf_hi_as_float(h);
  m = bf_pack2(r0, r1);
  const float s0 = r0 - bf_lo_as_float(m), s1 = r1 - bf_hi_as_float(m);
  l = bf_pack2(s0, s1);
}

}  // namespace spec

// ---- host side: registry of the specialised kernels linked into the library ----
struct SpecArgs {
  const void* tape;         // device: the packed weight tape of this parameter set
  const float* w;           // plain weight buffer (CI coefficients, cusp exponents)
  const float* R;           // [n_nuc][3]
  int B;
  double eps;               // eps under the safe norm
  FusedMc mc;               // sampler state, noise, step-size ring (kernels.h)
  long long* prof;          // optional shader-clock stamps of workgroup 0, [wave][256]: one per op boundary (launches the PROF instance)
};
// One entry of the weight tape.  kind 0: the (hi, mid, lo) bf16 planes of a 16 x 32 tile of W^T = three 1 KB fragments; lane
// (c, g) holds W[row[8 g + j]][col0 + c], j = 0..7 (row -1: zero; columns >= ncol: zero).  kind 1: one 1 KB fragment of 256
// floats gathered from the weight buffer (gather[k] < 0: zero) -- biases, envelope tables.
struct SpecTapeEntry {
  int32_t kind;
  int32_t w_off, ldw, col0, ncol;
  int32_t map;              // kind 0: first of 32 ints in `maps`; kind 1: first of 256 ints in `maps`
  float scale;              // every value is multiplied by this before it is split / stored (tanh layers: 2 log2 e)
};
struct SpecKernel {
  uint64_t hash;            // of the program it was generated from (engine.hip: program_hash)
  const char* name;
  int walkers_per_block;
  int lds_bytes;
  int tape_bytes;
  int n_entries;
  const SpecTapeEntry* entries;
  const int32_t* maps;
  void (*launch)(hipStream_t, const SpecArgs&, int n_blocks);
  const char* stamp_labels; // '|'-separated: what follows each clock stamp of the PROF instance
};
// (spec_registry.hip; every csrc/gen/*.hip defines spec_kernel_<name>())
const SpecKernel* find_spec_kernel(uint64_t hash);
void spec_pack_tape(const SpecKernel& k, const float* w, uint32_t* tape);

}  // namespace dqmc

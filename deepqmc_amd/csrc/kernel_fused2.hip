// kernel_fused2.hip -- value-only psi evaluation (the Metropolis hot loop) as ONE kernel, descriptor driven.
//
// A workgroup owns a tile of WT walkers and runs the whole layer program on it with every activation
// resident in LDS (reference sampling/electron_samplers.py:76-81 vmap(wf)).  Everything that is identical
// for all workgroups is decided on the HOST once per program (engine.hip: build_fused2_plan) instead of
// being recomputed by every wave of every workgroup:
//
//   * the linear layers are cut into units (row blocks x 2 column blocks) and the units of a dependency
//     level are dealt to the 4 waves by cost; each wave walks its own list of 160-byte unit descriptors
//     (LDS offsets and strides of every concat piece, packed-weight offset, destination, flags).  The
//     descriptors sit in the constant address space, so they arrive through scalar loads into SGPRs --
//     no per-lane metadata traffic, no integer division, no LDS copy of the program;
//   * waves skip the ops they have no unit of; only level boundaries carry a workgroup barrier;
//   * tile layout: element (walker wl, row, col) of a buffer at off + (row*WT + wl)*stride + col with WT a
//     power of two, so the MFMA row index m = row*WT + wl addresses every piece, residual and destination
//     as an affine function (broadcast pieces: m & (WT-1)), and stride = width + 2 keeps the 16 rows x 2 k
//     of a half-wave fragment read on 32 distinct banks.
//
//   * float32 layers run on the bf16 matrix pipe (FusedBfUnit below): gfx950 multiplies float32 matrices at the VALU rate
//     and bf16 matrices 16 x faster; a float is exactly the sum of three bf16 numbers, so six bf16 MFMAs per
//     16 x 16 x 32 block (96 matrix-pipe cycles instead of 256) give a float32-class product sum.  Weights are pre-split
//     on the host, activations are split in registers after the LDS read.
//
// The SQ counters and clock stamps of the first fused kernel (profiles/r01_pmc_sq_counters_fused_wt4.json,
// DESIGN.md section 4) showed ~3.4 k cycles of such setup per unit against ~1.2 k cycles of MFMA work.
#include "common.h"
#include "kernels.h"
#include "spec_device.h"      // fused2_det (shared with the plan-specialised sub-step kernels)

#ifndef DQMC_UNIFORM
#define DQMC_UNIFORM __attribute__((address_space(4)))   // constant address space: uniform loads are scalar
#endif

namespace dqmc {

// four consecutive reals as a register vector (the prefetched B quads travel between units in these)
template <typename real> struct RVec4;
template <> struct RVec4<float> { typedef f32x4 type; };
template <> struct RVec4<double> { typedef f64x4 type; };

typedef const DQMC_UNIFORM FDesc* DescPtr;
typedef const DQMC_UNIFORM FusedBuf* BufPtr;
typedef const DQMC_UNIFORM ::dqmc_op* OpPtr;

// tanh of the value path (Metropolis sub-steps, psi ratios): 1 - 2 / (1 + e^{2x}) on the hardware exp2 / rcp units --
// 5 VALU instructions instead of the ~25 of r_tanh (whose polynomial branch keeps RELATIVE accuracy near 0 for the
// derivative lanes of the Laplacian pass).  Absolute error <= ~1.5e-7 everywhere (saturates correctly: e -> inf gives 1,
// e -> 0 gives -1); the epilogues were 30 % of this kernel's VALU instructions, its most contended pipe.
template <typename real> __device__ __forceinline__ real fast_tanh(real x);
template <> __device__ __forceinline__ float fast_tanh<float>(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);       // e^{2x}
  return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
template <> __device__ __forceinline__ double fast_tanh<double>(double x) { return tanh(x); }

template <typename real> __device__ __forceinline__ real act2_value(int act, real v) {
  if (act == 1) return fast_tanh<real>(v);
  if (act == 2) return v / (1 + r_exp<real>(-v));
  return v;
}

// The B operand (weights) of a unit travels in GROUPS of FUSED_GROUP_QUADS quads of k-steps (x 2 column blocks x 16 bytes per lane).  While the MFMAs of one group run from `cur`, the loads of the next group -- the next four quads of
// this piece, the first of the next piece, or the first of the wave's NEXT UNIT -- are already in flight into `nxt`, so an
// L2 round trip (~700 cycles) is covered by 32 MFMAs instead of being paid once per quad.  All loads are unconditional
// with clamped quad indices (a conditional load makes the compiler copy the loaded registers at the join, which needs
// the data at once: the earlier four-deep ring waited for every quad it had just requested -- the 448-wide layers ran at
// 19 k cycles against 7 k of MFMA work).  Quads past the end of a piece re-read its last quad and are not multiplied.
template <typename real> struct BSet { typename RVec4<real>::type b[2][FusedGroup<real>::P]; };      // [column block][quad of the group | bf16 plane]

// pstride: Vec4 distance between the slots of one column block -- the quad stride of an f32 unit, the plane stride (64)
// of a bf16 unit (whose group is one K = 32 chunk: planes 0..2, n_left = 3)
template <typename real>
__device__ __forceinline__ void fused2_load_group(BSet<real>& s, const typename RVec4<real>::type* w, int cb1, int pstride, int n_left) {
  const int last = n_left - 1;
#pragma unroll
  for (int dd = 0; dd < FusedGroup<real>::P; ++dd) {
    const int qi = (dd < last ? dd : last) * pstride;      // wave-uniform clamp
    s.b[0][dd] = w[qi];
    s.b[1][dd] = w[qi + cb1];
  }
}
// first group of the unit described by nd (w_off, w_cb1, qstride, quads of its first piece)
template <typename real>
__device__ __forceinline__ void fused2_prefetch_unit(const Fused2Args<real>& a, DescPtr nd, BSet<real>& s) {
  const typename RVec4<real>::type* nb = reinterpret_cast<const typename RVec4<real>::type*>(a.wpk) + nd->w_off + (threadIdx.x & 63);
  fused2_load_group<real>(s, nb, nd->w_cb1, nd->qstride, nd->a_nq[0]);
}

// Epilogue shared by the unit bodies: bias + activation + residual, store to LDS (or HBM for buffers later kernels read).
template <typename real, int MA>
__device__ __forceinline__ void fused2_epilogue(const Fused2Args<real>& a, DescPtr d, typename Mfma<real>::acc_t (&acc)[MA][2],
                                                const real (&bias_v)[2]) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  constexpr int NRW = 2;
  const int lane = threadIdx.x & 63, l15 = lane & 15;
  const int row0 = d->row0, rtot = d->rtot, ldw = d->ldw, col_u = d->col0;
  const int wtm1 = a.WT - 1;
  const int flags = d->flags, act = flags & 3;
  const bool dst_global = (flags & 8) != 0;
  const real res_scale = (flags & 4) ? (real)0.70710678118654752440 : (real)1;
  const int res_base = d->res_base;
  int colv[NRW];
  bool c_ok[NRW];
#pragma unroll
  for (int y = 0; y < NRW; ++y) {
    colv[y] = col_u + y * 16 + l15;
    c_ok[y] = colv[y] < ldw;
    if (!c_ok[y]) colv[y] = col_u;
  }
  int m_e[MA][4];
  bool r_ok[MA][4];
#pragma unroll
  for (int x = 0; x < MA; ++x)
#pragma unroll
    for (int rgi = 0; rgi < 4; ++rgi) {
      const int m = row0 + x * 16 + Mfma<real>::row_of(lane, rgi);
      r_ok[x][rgi] = m < rtot;
      m_e[x][rgi] = r_ok[x][rgi] ? m : row0;
    }
  real ov[MA][4][NRW];
#pragma unroll
  for (int x = 0; x < MA; ++x)
#pragma unroll
    for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
      for (int y = 0; y < NRW; ++y) ov[x][rgi][y] = acc[x][y][rgi] + bias_v[y];
  if (act == 1) {                                  // wave-uniform: one branch around the whole batch
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y) ov[x][rgi][y] = fast_tanh<real>(ov[x][rgi][y]);
  } else if (act == 2) {
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y) ov[x][rgi][y] = act2_value<real>(2, ov[x][rgi][y]);
  }
  if (res_base >= 0) {
    const int rs = d->res_stride;
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
        for (int y = 0; y < NRW; ++y)
          ov[x][rgi][y] = (smem[res_base + m_e[x][rgi] * rs + colv[y]] + ov[x][rgi][y]) * res_scale;
  }
  if (!dst_global) {
    const int db = d->dst_base, ds = d->dst_stride;
    if (flags & 16) {                              // every row and column of the unit is valid: no exec masking
#pragma unroll
      for (int x = 0; x < MA; ++x)
#pragma unroll
        for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
          for (int y = 0; y < NRW; ++y) smem[db + m_e[x][rgi] * ds + colv[y]] = ov[x][rgi][y];
    } else {
#pragma unroll
      for (int x = 0; x < MA; ++x)
#pragma unroll
        for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
          for (int y = 0; y < NRW; ++y)
            if (r_ok[x][rgi] && c_ok[y]) smem[db + m_e[x][rgi] * ds + colv[y]] = ov[x][rgi][y];
    }
  } else {
    // walker-major HBM layout [B][rows][width]; m = rr*WT + wl
    const BufPtr fb = (BufPtr)a.fbufs + d->dst_base;
    const int rows = fb->rows, width = fb->width, g_r0 = d->g_r0, g_col0 = d->g_col0, sh = a.wt_shift;
    real* dst_g = reinterpret_cast<real*>(a.ws + fb->goff) + (long)blockIdx.x * a.WT * rows * width;
    if (a.mc.enabled) {        // sub-step mode: the only such buffer is the Jastrow value, consumed by the tail of this kernel
      real* jas = smem + a.scratch_off + a.WT * a.li.N * 3;
#pragma unroll
      for (int x = 0; x < MA; ++x)
#pragma unroll
        for (int rgi = 0; rgi < 4; ++rgi)
#pragma unroll
          for (int y = 0; y < NRW; ++y)
            if (r_ok[x][rgi] && c_ok[y] && g_col0 + colv[y] < 4) jas[(m_e[x][rgi] & wtm1) * 4 + g_col0 + colv[y]] = ov[x][rgi][y];
      return;
    }
#pragma unroll
    for (int x = 0; x < MA; ++x)
#pragma unroll
      for (int rgi = 0; rgi < 4; ++rgi) {
        const int m = m_e[x][rgi], wl = m & wtm1, rr = m >> sh;
#pragma unroll
        for (int y = 0; y < NRW; ++y)
          if (r_ok[x][rgi] && c_ok[y] && blockIdx.x * a.WT + wl < a.B)
            dst_g[((long)wl * rows + g_r0 + rr) * width + g_col0 + colv[y]] = ov[x][rgi][y];
      }
  }
}

// One unit: MA row blocks x 2 column blocks of y = act(concat(pieces) W + b) (+ residual) on the tile.
// `nxt`: on entry the first group of B quads of THIS unit (requested by the previous unit of the wave, so the L2 round
// trip overlapped that unit's tail, its LDS stores and the level barrier); on return the first group of the wave's next
// unit (FDesc::nx_*; without one, a harmless re-read of this unit's own first group).
template <typename real, int MA>
__device__ __forceinline__ void fused2_unit(const Fused2Args<real>& a, DescPtr d, BSet<real>& nxt) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  typedef typename Mfma<real>::acc_t acc_t;
  typedef typename RVec4<real>::type rv4;
  constexpr int NRW = 2;
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = d->row0, rtot = d->rtot, ldw = d->ldw, col_u = d->col0;
  const int wtm1 = a.WT - 1;
  int mrow[MA];
#pragma unroll
  for (int x = 0; x < MA; ++x) {
    const int m = row0 + x * 16 + l15;
    mrow[x] = m < rtot ? m : row0;                          // dummy rows read a valid row, dropped at the store
  }
  acc_t acc[MA][NRW];
#pragma unroll
  for (int x = 0; x < MA; ++x)
#pragma unroll
    for (int y = 0; y < NRW; ++y) acc[x][y] = acc_t{0, 0, 0, 0};
  const int bias_off = d->bias_off;
  real bias_v[NRW];
#pragma unroll
  for (int y = 0; y < NRW; ++y) {
    const int col = col_u + y * 16 + l15;
    bias_v[y] = (bias_off >= 0 && col < ldw) ? a.w[bias_off + col] : (real)0;
  }
  const rv4* wbase = reinterpret_cast<const rv4*>(a.wpk) + d->w_off + lane;
  const int cb1 = d->w_cb1, qstride = d->qstride, bcast = d->bcast, n_pieces = d->n_pieces;
  // where the group after the last one of this unit comes from
  const rv4* nu_w = reinterpret_cast<const rv4*>(a.wpk) + d->nx_w_off + lane;
  const int nu_cb1 = d->nx_cb1, nu_qs = d->nx_qstride, nu_nq = d->nx_nq;
  int q0 = 0;
  for (int p = 0; p < n_pieces; ++p) {
    const int base = d->a_base[p], stride = d->a_stride[p], KS = d->a_ks[p], NQ = d->a_nq[p];
    const bool bc = (bcast >> p) & 1;
    const bool last_piece = p + 1 == n_pieces;
    const int nq_next_piece = last_piece ? nu_nq : d->a_nq[last_piece ? p : p + 1];
    int ao[MA];
#pragma unroll
    for (int x = 0; x < MA; ++x) ao[x] = base + (bc ? (mrow[x] & wtm1) : mrow[x]) * stride + l4;
    real fan[4][MA];                                        // A fragments (LDS) run one quad ahead
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ks = kk < KS ? kk : KS - 1;
#pragma unroll
      for (int x = 0; x < MA; ++x) fan[kk][x] = smem[ao[x] + ks * 4];
    }
    constexpr int G = FusedGroup<real>::P;
    for (int q = 0; q < NQ; q += G) {
      BSet<real> cur = nxt;                                 // arrived long ago: register moves, no wait
      {
        const bool more = q + G < NQ;                       // wave-uniform selects, then ONE unconditional load sequence
        const rv4* gw = more ? wbase + (long)(q0 + q + G) * qstride : (last_piece ? nu_w : wbase + (long)(q0 + NQ) * qstride);
        const int g_cb1 = (more || !last_piece) ? cb1 : nu_cb1, g_qs = (more || !last_piece) ? qstride : nu_qs;
        const int g_left = more ? NQ - (q + G) : nq_next_piece;
        fused2_load_group<real>(nxt, gw, g_cb1, g_qs, g_left);
      }
#pragma unroll
      for (int dd = 0; dd < G; ++dd) {
        if (q + dd < NQ) {                                  // wave-uniform
          real fa[4][MA];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int x = 0; x < MA; ++x) fa[kk][x] = fan[kk][x];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {                  // A fragments of the next quad (clamped past the end)
            int ks = (q + dd + 1) * 4 + kk;
            ks = ks < KS ? ks : KS - 1;
#pragma unroll
            for (int x = 0; x < MA; ++x) fan[kk][x] = smem[ao[x] + ks * 4];
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int x = 0; x < MA; ++x) {
              acc[x][0] = Mfma<real>::run(fa[kk][x], cur.b[0][dd][kk], acc[x][0]);
              acc[x][1] = Mfma<real>::run(fa[kk][x], cur.b[1][dd][kk], acc[x][1]);
            }
        }
      }
    }
    q0 += NQ;
  }
  fused2_epilogue<real, MA>(a, d, acc, bias_v);
}

// Lean unit for the many SMALL layers (edge MLPs, second layers of the node MLPs): one piece, one row block, at most
// FusedGroup::P quads of k-steps, LDS destination, tanh or no activation.  Its whole B operand arrived with the previous
// unit (`nxt`); the next unit's first group is requested first thing, all A fragments are read up front (clamped
// addresses, so the loads need no branches and their latency is paid once), then up to 32 MFMAs and the epilogue.
template <typename real>
__device__ __forceinline__ void fused2_unit_lean(const Fused2Args<real>& a, DescPtr d, BSet<real>& nxt) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  typedef typename Mfma<real>::acc_t acc_t;
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = d->row0, rtot = d->rtot, ldw = d->ldw, col_u = d->col0;
  const int KS = d->a_ks[0], NQ = d->a_nq[0];
  const int bias_off = d->bias_off;
  real bias_v[2];
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int col = col_u + y * 16 + l15;
    bias_v[y] = (bias_off >= 0 && col < ldw) ? a.w[bias_off + col] : (real)0;
  }
  const BSet<real> cur = nxt;
  {
    const typename RVec4<real>::type* nb = reinterpret_cast<const typename RVec4<real>::type*>(a.wpk) + d->nx_w_off + (threadIdx.x & 63);
    fused2_load_group<real>(nxt, nb, d->nx_cb1, d->nx_qstride, d->nx_nq);      // (the last unit re-reads its own group)
  }
  const int m = row0 + l15;
  const int ao = d->a_base[0] + (m < rtot ? m : row0) * d->a_stride[0] + l4;
  constexpr int G = FusedGroup<real>::P;
  real fa[G][4];
#pragma unroll
  for (int dd = 0; dd < G; ++dd)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ks = dd * 4 + kk;
      fa[dd][kk] = smem[ao + (ks < KS ? ks : KS - 1) * 4];
    }
  acc_t acc[1][2];
  acc[0][0] = acc_t{0, 0, 0, 0}; acc[0][1] = acc_t{0, 0, 0, 0};
#pragma unroll
  for (int dd = 0; dd < G; ++dd) {
    if (dd < NQ) {                                          // wave-uniform
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        acc[0][0] = Mfma<real>::run(fa[dd][kk], cur.b[0][dd][kk], acc[0][0]);
        acc[0][1] = Mfma<real>::run(fa[dd][kk], cur.b[1][dd][kk], acc[0][1]);
      }
    }
  }
  fused2_epilogue<real, 1>(a, d, acc, bias_v);
}

// ---- float32 layers on the bf16 matrix pipe (common.h: "float32 products on the bf16 matrix pipe") ----
// Same unit shape (MA row blocks x 2 column blocks), same accumulators and epilogue; the K loop runs in chunks of 32:
// a lane reads its eight consecutive A values of the chunk from LDS (four 8-byte reads), splits them into three bf16
// pieces in registers, and six MFMAs per column block multiply them with the three pre-split weight planes
// (engine.hip: pack_fused_weights lays the planes out fragment-major, [chunk][column block][plane][lane] x 16 bytes).
// One group of B operands = one chunk (3 planes x 2 column blocks = 24 registers), prefetched one chunk ahead like the
// f32 groups.  Pieces are whole octets wide (K % 8 == 0); the lanes of a partial last chunk re-read the last valid
// octet against zero weights.
__device__ __forceinline__ void fused2_read_octet(const float* smem, int off, float (&v)[8]) {
  const Vec2<float>* p = reinterpret_cast<const Vec2<float>*>(smem + off);       // off is even: 8-byte LDS reads
#pragma unroll
  for (int j = 0; j < 4; ++j) { const Vec2<float> t = p[j]; v[2 * j] = t.v[0]; v[2 * j + 1] = t.v[1]; }
}
__device__ __forceinline__ void fused2_bf_chunk(const float (&av)[8], const BSet<float>& cur, f32x4& acc0, f32x4& acc1) {
  BfFrag a0, a1, a2;
  bf_split8(av, a0, a1, a2);
  const BfFrag* b0 = reinterpret_cast<const BfFrag*>(&cur.b[0][0]);
  const BfFrag* b1 = reinterpret_cast<const BfFrag*>(&cur.b[1][0]);
  // small terms first; the two column blocks interleave (independent accumulators)
  acc0 = mfma_bf16(a2, b0[0], acc0); acc1 = mfma_bf16(a2, b1[0], acc1);
  acc0 = mfma_bf16(a1, b0[1], acc0); acc1 = mfma_bf16(a1, b1[1], acc1);
  acc0 = mfma_bf16(a0, b0[2], acc0); acc1 = mfma_bf16(a0, b1[2], acc1);
  acc0 = mfma_bf16(a1, b0[0], acc0); acc1 = mfma_bf16(a1, b1[0], acc1);
  acc0 = mfma_bf16(a0, b0[1], acc0); acc1 = mfma_bf16(a0, b1[1], acc1);
  acc0 = mfma_bf16(a0, b0[0], acc0); acc1 = mfma_bf16(a0, b1[0], acc1);
}
template <typename real, int MA> struct FusedBfUnit {
  static __device__ __forceinline__ void run(const Fused2Args<real>&, DescPtr, BSet<real>&) {}
  static __device__ __forceinline__ void lean(const Fused2Args<real>&, DescPtr, BSet<real>&) {}
};
template <int MA> struct FusedBfUnit<float, MA> {
  static __device__ __forceinline__ void run(const Fused2Args<float>& a, DescPtr d, BSet<float>& nxt) {
    HIP_DYNAMIC_SHARED(char, smem_raw)
    const float* smem = reinterpret_cast<const float*>(smem_raw);
    typedef f32x4 rv4;
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int row0 = d->row0, rtot = d->rtot, ldw = d->ldw, col_u = d->col0;
    const int wtm1 = a.WT - 1;
    int mrow[MA];
#pragma unroll
    for (int x = 0; x < MA; ++x) {
      const int m = row0 + x * 16 + l15;
      mrow[x] = m < rtot ? m : row0;
    }
    f32x4 acc[MA][2];
#pragma unroll
    for (int x = 0; x < MA; ++x) { acc[x][0] = f32x4{0, 0, 0, 0}; acc[x][1] = f32x4{0, 0, 0, 0}; }
    const int bias_off = d->bias_off;
    float bias_v[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int col = col_u + y * 16 + l15;
      bias_v[y] = (bias_off >= 0 && col < ldw) ? a.w[bias_off + col] : 0.0f;
    }
    const rv4* wbase = reinterpret_cast<const rv4*>(a.wpk) + d->w_off + lane;
    const int cb1 = d->w_cb1, cstride = d->qstride, bcast = d->bcast, n_pieces = d->n_pieces;
    const rv4* nu_w = reinterpret_cast<const rv4*>(a.wpk) + d->nx_w_off + lane;
    const int nu_cb1 = d->nx_cb1, nu_ps = d->nx_qstride, nu_nq = d->nx_nq;
    int c0 = 0;
    for (int p = 0; p < n_pieces; ++p) {
      const int base = d->a_base[p], stride = d->a_stride[p], K8 = d->a_ks[p], NC = d->a_nq[p];
      const bool bc = (bcast >> p) & 1;
      const bool last_piece = p + 1 == n_pieces;
      int ao[MA];
#pragma unroll
      for (int x = 0; x < MA; ++x) ao[x] = base + (bc ? (mrow[x] & wtm1) : mrow[x]) * stride;
      auto octets = [&](int c_now, float (&av)[MA][8]) {
        int o = c_now * 4 + l4;
        o = o < K8 ? o : K8 - 1;
#pragma unroll
        for (int x = 0; x < MA; ++x) fused2_read_octet(smem, ao[x] + 8 * o, av[x]);
      };
      auto request = [&](BSet<float>& into, int c_next) {      // group of chunk c_next of this piece, or what follows the piece
        const bool own = c_next < NC || !last_piece;           // wave-uniform selects, then ONE unconditional load sequence
        const rv4* gw = own ? wbase + (long)(c0 + c_next) * cstride : nu_w;
        fused2_load_group<float>(into, gw, own ? cb1 : nu_cb1, own ? 64 : nu_ps, own ? 3 : nu_nq);
      };
      int c = 0;
      for (; c + 1 < NC; c += 2) {
        BSet<float> alt;
        request(alt, c + 1);
        float ac[MA][8];
        octets(c, ac);
#pragma unroll
        for (int x = 0; x < MA; ++x) fused2_bf_chunk(ac[x], nxt, acc[x][0], acc[x][1]);
        request(nxt, c + 2);
        octets(c + 1, ac);
#pragma unroll
        for (int x = 0; x < MA; ++x) fused2_bf_chunk(ac[x], alt, acc[x][0], acc[x][1]);
      }
      if (c < NC) {                                            // odd last chunk: the one copy
        const BSet<float> cur = nxt;
        request(nxt, c + 1);
        float ac[MA][8];
        octets(c, ac);
#pragma unroll
        for (int x = 0; x < MA; ++x) fused2_bf_chunk(ac[x], cur, acc[x][0], acc[x][1]);
      }
      c0 += NC;
    }
    fused2_epilogue<float, MA>(a, d, acc, bias_v);
  }
  // one piece, one chunk, one row block, LDS destination: everything the unit needs arrived with the previous unit
  static __device__ __forceinline__ void lean(const Fused2Args<float>& a, DescPtr d, BSet<float>& nxt) {
    HIP_DYNAMIC_SHARED(char, smem_raw)
    const float* smem = reinterpret_cast<const float*>(smem_raw);
    typedef f32x4 rv4;
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    const int row0 = d->row0, rtot = d->rtot, ldw = d->ldw, col_u = d->col0;
    const int K8 = d->a_ks[0];
    const int bias_off = d->bias_off;
    float bias_v[2];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int col = col_u + y * 16 + l15;
      bias_v[y] = (bias_off >= 0 && col < ldw) ? a.w[bias_off + col] : 0.0f;
    }
    const BSet<float> cur = nxt;
    {
      const rv4* nb = reinterpret_cast<const rv4*>(a.wpk) + d->nx_w_off + lane;
      fused2_load_group<float>(nxt, nb, d->nx_cb1, d->nx_qstride, d->nx_nq);
    }
    const int m = row0 + l15;
    const int o = l4 < K8 ? l4 : K8 - 1;
    float av[8];
    fused2_read_octet(smem, d->a_base[0] + (m < rtot ? m : row0) * d->a_stride[0] + 8 * o, av);
    f32x4 acc[1][2];
    acc[0][0] = f32x4{0, 0, 0, 0}; acc[0][1] = f32x4{0, 0, 0, 0};
    fused2_bf_chunk(av, cur, acc[0][0], acc[0][1]);
    fused2_epilogue<float, 1>(a, d, acc, bias_v);
  }
};

// Slater-matrix entries A[(wl, k)][el][mu] = envelope(el; k, mu) * backflow(el; k, mu) for the tile (the arithmetic of
// k_orbitals, value lane).  One thread per (electron, orbital k*N + mu): the envelope weights pi / zeta of that pair are
// read ONCE (they do not depend on the walker) and the exponentials of the WT walkers are independent instructions --
// the earlier one-thread-per-(walker, electron, orbital) loop spent its time in dependent table loads.
// store(wl, kd, el, mu, value) places the entry (LDS matrices of the sub-step tail, or the HBM orbital buffer).
template <typename real, typename Store>
__device__ __forceinline__ void fused2_slater_entries(const Fused2Args<real>& a, OpPtr op, int nw, const real* rs, Store store) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  const real* smem = reinterpret_cast<const real*>(smem_raw);
  const int tid = threadIdx.x, nthr = blockDim.x, WT = a.WT, N = a.li.N, K = a.K, n_up = a.n_up, n_nuc = a.n_nuc, KN = K * N;
  const BufPtr bf = (BufPtr)a.fbufs + op->i[0];
  const int bo = bf->off, bs = bf->stride;
  const int n_env = op->i[6] > 0 ? op->i[6] : 1;
  const int o_pu = op->i[2], o_pd = op->i[3], o_zu = op->i[4], o_zd = op->i[5];
  constexpr int WCH = 4;
  for (int q = tid; q < N * KN; q += nthr) {
    const int el = q / KN, kmu = q - el * KN;
    const int kd = kmu / N, mu = kmu - kd * N;
    const real* pi = a.w + (el < n_up ? o_pu : o_pd) + kmu * n_nuc * n_env;
    const real* ze = a.w + (el < n_up ? o_zu : o_zd) + kmu * n_nuc * n_env;
    if (sizeof(real) == 4) {      // float32 build: exponentials on the f32 unit (as k_orbitals, T = 1)
      for (int w0 = 0; w0 < nw; w0 += WCH) {       // WCH walkers at a time (their exponentials are independent)
        float acc[WCH], px[WCH], py[WCH], pz[WCH];
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
          const int wl = (w0 + j < WT) ? w0 + j : w0;
          const real* rp = rs + (wl * N + el) * 3;
          px[j] = (float)rp[0]; py[j] = (float)rp[1]; pz[j] = (float)rp[2];
          acc[j] = 0.f;
        }
        for (int n = 0; n < n_nuc; ++n) {
          const float Rx = (float)a.R[n * 3], Ry = (float)a.R[n * 3 + 1], Rz = (float)a.R[n * 3 + 2];
          float rho[WCH];
#pragma unroll
          for (int j = 0; j < WCH; ++j) {
            const float dx = px[j] - Rx, dy = py[j] - Ry, dz = pz[j] - Rz;
            float d2 = (float)a.eps;
            d2 += dx * dx; d2 += dy * dy; d2 += dz * dz;
            rho[j] = sqrtf(d2);
          }
          for (int ev = 0; ev < n_env; ++ev) {
            const float p = (float)pi[n * n_env + ev], z = fabsf((float)ze[n * n_env + ev]);
#pragma unroll
            for (int j = 0; j < WCH; ++j) acc[j] += p * expf(-z * rho[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j)
          if (w0 + j < nw) store(w0 + j, kd, el, mu, (real)(acc[j] * (float)smem[bo + (el * WT + w0 + j) * bs + kmu]));
      }
    } else {
      for (int wl = 0; wl < nw; ++wl) {
        const real* rp = rs + (wl * N + el) * 3;
        double e0 = 0;
        for (int n = 0; n < n_nuc; ++n) {
          double d2 = a.eps;
          for (int c = 0; c < 3; ++c) { const double dx = (double)rp[c] - (double)a.R[n * 3 + c]; d2 += dx * dx; }
          const double rho = sqrt(d2);
          for (int ev = 0; ev < n_env; ++ev) e0 += (double)pi[n * n_env + ev] * exp(-fabs((double)ze[n * n_env + ev]) * rho);
        }
        store(wl, kd, el, mu, (real)(e0 * (double)smem[bo + (el * WT + wl) * bs + kmu]));
      }
    }
  }
}

// Structured ops (pair features, spin means, convolutions, sums, envelopes): all threads of the
// workgroup, element-parallel.  Tile element (wl, row, col) of buffer b: off + (row*WT + wl)*stride + col.
template <typename real>
__device__ __forceinline__ void fused2_generic(const Fused2Args<real>& a, OpPtr op, int nw) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int WT = a.WT, sh = a.wt_shift, wtm1 = WT - 1;
  const int N = a.li.N, n_up = a.n_up, n_nuc = a.n_nuc, K = a.K;
  const real* r = smem + a.scratch_off;          // positions of the tile (proposed positions in sub-step mode)
  const BufPtr fbs = (BufPtr)a.fbufs;
  const int kind = op->kind;
  LaneInfo li = a.li;   // T = TP = 1
  // the int table (edge pairs, per-receiver sender lists) was staged into LDS by the prologue: its reads sit on the
  // dependent path of every structured op (row index -> activation address)
  const int32_t* itab = reinterpret_cast<const int32_t*>(smem_raw + a.it_off);
  // Elements are enumerated walker-fastest (e = rest*WT + wl) so neighbouring threads touch neighbouring
  // LDS rows of the same column; walkers beyond the batch (wl >= nw) are skipped.
  switch (kind) {
    case DQMC_OP_FEAT_EN: {
      const BufPtr x = fbs + op->i[0];
      const int xo = x->off, xs = x->stride, xw = x->width, lr = op->i[1], sp = op->i[2];
      for (int e = tid; e < N * n_nuc * WT; e += nthr) {
        const int wl = e & wtm1, q = e >> sh;
        const int el = q / n_nuc, n = q - el * n_nuc;
        if (wl >= nw) continue;
        const int row = xo + (el * WT + wl) * xs;
        {
          double dd[3], f[4];
          for (int c = 0; c < 3; ++c) dd[c] = (double)r[(wl * N + el) * 3 + c] - (double)a.R[n * 3 + c];
          pair_feature_lane(dd, a.eps, el, -1, 0, li, lr != 0, f);
          for (int c = 0; c < 4; ++c) smem[row + 4 * n + c] = (real)f[c];
        }
        if (n == 0) {
          int c = 4 * n_nuc;
          if (sp) smem[row + c++] = (real)(el < n_up ? 1.0 : -1.0);
          for (; c < xw; ++c) smem[row + c] = (real)0;
        }
      }
      break;
    }
    case DQMC_OP_FEAT_EE: {
      const BufPtr eb = fbs + op->i[0];
      const int eo = eb->off, es = eb->stride, n_rows = op->i[2], lr = op->i[3];
      const int32_t* pairs = itab + op->i[1];
      for (int e = tid; e < n_rows * WT; e += nthr) {
        const int wl = e & wtm1, kr = e >> sh;
        if (wl >= nw) continue;
        const int rc = pairs[2 * kr], sd = pairs[2 * kr + 1];
        const int row = eo + (kr * WT + wl) * es;
        {
          double dd[3], f[4];
          for (int c = 0; c < 3; ++c) dd[c] = (double)r[(wl * N + rc) * 3 + c] - (double)r[(wl * N + sd) * 3 + c];
          pair_feature_lane(dd, a.eps, rc, sd, 0, li, lr != 0, f);
          for (int c = 0; c < 4; ++c) smem[row + c] = (real)f[c];
        }
      }
      break;
    }
    case DQMC_OP_SPIN_MEAN: {
      const BufPtr x = fbs + op->i[0];
      const BufPtr m = fbs + op->i[1];
      const int xo = x->off, xs = x->stride, W = x->width, mo = m->off, ms = m->stride;
      const real inv_up = n_up > 0 ? (real)1 / (real)n_up : (real)0, inv_dn = N > n_up ? (real)1 / (real)(N - n_up) : (real)0;
      for (int e = tid; e < W * 2 * WT; e += nthr) {
        const int wl = e & wtm1, q = e >> sh;
        const int which = q & 1, c = q >> 1;
        const int i0 = which ? n_up : 0, i1 = which ? N : n_up;
        real acc = 0;
        for (int el = i0; el < i1; ++el) acc += smem[xo + (el * WT + wl) * xs + c];
        smem[mo + (which * WT + wl) * ms + c] = acc * (which ? inv_dn : inv_up);
      }
      break;
    }
    case DQMC_OP_CONV:
    case DQMC_OP_EDGE_SUM: {
      const bool conv = kind == DQMC_OP_CONV;
      const BufPtr we = fbs + op->i[0];
      const BufPtr out = fbs + op->i[2];
      const BufPtr hx = conv ? fbs + op->i[1] : we;
      const int wo = we->off, ws_ = we->stride, ho = hx->off, hs = hx->stride, oo = out->off, os = out->stride;
      const real scale = conv ? (real)1 : (real)(1.0 / (double)(op->i[1] > 0 ? op->i[1] : 1));
      const int32_t* tab = itab + op->i[4];
      const int S = op->i[5], W = op->i[6], col0 = op->i[3];
      const bool w_pow2 = (W & (W - 1)) == 0;
      const int w_sh = 31 - __builtin_clz((unsigned)(W > 0 ? W : 1));
      for (int e = tid; e < N * W * WT; e += nthr) {
        const int wl = e & wtm1, q = e >> sh;
        const int el = w_pow2 ? (q >> w_sh) : q / W, c = q - el * W;      // (no integer division on the common widths)
        real acc = 0;
        for (int s = 0; s < S; ++s) {
          const int row = tab[2 * (el * S + s)], snd = tab[2 * (el * S + s) + 1];
          if (row < 0) continue;
          const real ev = smem[wo + (row * WT + wl) * ws_ + c];
          acc += conv ? ev * smem[ho + (snd * WT + wl) * hs + c] : ev;
        }
        smem[oo + (el * WT + wl) * os + col0 + c] = acc * scale;
      }
      break;
    }
    case DQMC_OP_ROW_SUM: {
      const BufPtr x = fbs + op->i[0];
      const BufPtr s = fbs + op->i[1];
      const int xo = x->off, xs = x->stride, W = x->width, rows = x->rows, so = s->off, ss = s->stride;
      for (int e = tid; e < W * WT; e += nthr) {
        const int wl = e & wtm1, c = e >> sh;
        real acc = 0;
        for (int el = 0; el < rows; ++el) acc += smem[xo + (el * WT + wl) * xs + c];
        smem[so + wl * ss + c] = acc;
      }
      break;
    }
    case DQMC_OP_ORBITALS: {
      if (a.mc.enabled) break;                     // sub-step mode: the tail builds the Slater matrices itself
      const BufPtr orb = fbs + op->i[1];
      const int ow = orb->width, orows = orb->rows;
      real* orb_g = reinterpret_cast<real*>(a.ws + orb->goff) + (long)blockIdx.x * WT * orows * ow;
      fused2_slater_entries<real>(a, op, nw, r, [&](int wl, int kd, int el, int mu, real v) {
        orb_g[(wl * K + kd) * ow + el * N + mu] = v;
      });
      break;
    }
    default:
      break;
  }
}

// Tail of a Metropolis sub-step: Slater matrices (all threads) -> determinants (one thread each) -> CI terms
// -> psi of the proposal -> accept / reject -> sampler state, acceptance count; the last workgroup to finish
// adapts tau (electron_samplers.py:106-138).  The matrices reuse LDS of activations that are dead by now.
template <typename real>
__device__ __forceinline__ void fused2_mc_tail(const Fused2Args<real>& a, int nw, long long* prof) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  const int tid = threadIdx.x, nthr = blockDim.x, WT = a.WT, N = a.li.N, K = a.K, wtm1 = WT - 1, sh = a.wt_shift;
  const int n_up = a.n_up, NN = N * N;
  real* rs = smem + a.scratch_off;
  real* jas = rs + WT * N * 3;
  char* after = smem_raw + (size_t)a.scratch_off * sizeof(real) + ((WT * (N * 3 + 4) * (int)sizeof(real) + 15) / 16 * 16);
  double* ld = reinterpret_cast<double*>(after);
  int32_t* sg = reinterpret_cast<int32_t*>(after + (size_t)WT * K * 8);
  double* shv = reinterpret_cast<double*>(after + (size_t)WT * K * 12);      // CI shift per walker
  real* mats = smem + a.mc.mat_off;                           // [WT*K][N*N]
  // sampler state of this thread's walker: requested now, used after the determinants
  // (the accept step works with 16 lanes per walker, four walkers at a time on wave 0: lane group g <-> walker g)
  real lp_old = 0, u_b = 1;
  int age_b = 0;
  if (tid < 64 && (tid >> 4) < nw) {
    const long b = (long)blockIdx.x * WT + (tid >> 4);
    lp_old = reinterpret_cast<const real*>(a.mc.logpsi)[b];
    u_b = reinterpret_cast<const real*>(a.mc.unif)[b];
    age_b = a.mc.age[b];
  }
  const OpPtr op = (OpPtr)a.ops + a.mc.orb_op;
  // Slater matrix entries = envelope * backflow, exactly as the ORBITALS op computes them
  fused2_slater_entries<real>(a, op, nw, rs, [&](int wl, int kd, int el, int mu, real v) {
    mats[(wl * K + kd) * NN + el * N + mu] = v;
  });
  if (prof) prof[0] = clock64();
  __syncthreads();
  for (int e = tid; e < K * WT; e += nthr) {
    const int wl = e & wtm1, kd = e >> sh;
    double la = 0.0;
    int sn = 0;
    if (wl < nw) {
      const real* m = mats + (wl * K + kd) * NN;
      if (N == 2) fused2_det<real, 2>(m, la, sn);
      else if (N == 3) fused2_det<real, 3>(m, la, sn);
      else fused2_det<real, 4>(m, la, sn);
    }
    ld[wl * K + kd] = la;
    sg[wl * K + kd] = sn;
  }
  if (prof) prof[1] = clock64();
  __syncthreads();
  // exp-normalised CI terms (wf/nn_wave_function.py:152-160), one thread per (walker, determinant)
  const real* cc = a.mc.cc_off >= 0 ? a.w + a.mc.cc_off : nullptr;
  double term = 0.0, shift = 0.0;
  int e_own = -1;
  for (int e = tid; e < K * WT; e += nthr) {      // K*WT <= 256: at most one element per thread
    const int wl = e & wtm1, kd = e >> sh;
    const double* x = ld + wl * K;
    shift = -INFINITY;
    for (int k = 0; k < K; ++k) shift = fmax(shift, x[k]);
    if (isinf(shift)) shift = 0.0;
    term = (cc ? (double)cc[kd] : 1.0) * sg[wl * K + kd] * exp(x[kd] - shift);
    if (kd == 0) shv[wl] = shift;
    e_own = e;
  }
  __syncthreads();
  if (e_own >= 0) ld[(e_own & wtm1) * K + (e_own >> sh)] = term;     // log|det| -> CI term, in place
  __syncthreads();
  int n_acc = 0;
  if (tid < 64) {
    const int l = tid & 15;
    const real* al = a.w + a.mc.al_off;
    const int n_pairs = N * (N - 1) / 2;
    for (int w0 = 0; w0 < nw; w0 += 4) {             // four walkers per pass, 16 lanes each
      const int wl_raw = w0 + (tid >> 4);
      const bool live = wl_raw < nw;
      const int wl = live ? wl_raw : nw - 1;         // idle groups shadow a valid walker (all lanes stay in the shuffles)
      const long b = (long)blockIdx.x * WT + wl;
      if (w0 > 0 && live) {                          // later passes: their state was not preloaded
        lp_old = reinterpret_cast<const real*>(a.mc.logpsi)[b];
        u_b = reinterpret_cast<const real*>(a.mc.unif)[b];
        age_b = a.mc.age[b];
      }
      double psi = 0.0;
      for (int k = l; k < K; k += 16) psi += ld[wl * K + k];
      for (int m = 1; m < 16; m <<= 1) psi += __shfl_xor(psi, m, 64);
      double logpsi = log(fabs(psi)) + shv[wl];
      const int sign_p = (psi > 0) - (psi < 0);
      const real* r = rs + wl * N * 3;
      double cusp = 0.0;
      if (a.mc.cusp_kind) {     // wf/cusp.py:5-26,68-78 (value); pair p <-> (i, j), i < j, row-major
        int i = 0, p0 = 0;
        for (int p = l; p < n_pairs; p += 16) {
          while (p - p0 >= N - 1 - i) { p0 += N - 1 - i; ++i; }
          const int j = i + 1 + (p - p0);
          double d2 = a.eps;
          for (int c = 0; c < 3; ++c) { const double d = (double)r[i * 3 + c] - (double)r[j * 3 + c]; d2 += d * d; }
          const double rho = sqrt(d2);
          const bool same = (i < n_up) == (j < n_up);
          const double sc = same ? a.mc.same_scale : a.mc.anti_scale, alp = (double)al[same ? 0 : 1];
          cusp += a.mc.cusp_kind == 1 ? -sc / (alp * (1 + alp * rho)) : -sc * alp * alp / (alp + rho);
        }
        for (int m = 1; m < 16; m <<= 1) cusp += __shfl_xor(cusp, m, 64);
      }
      logpsi += cusp + (a.mc.jas_width > 0 ? (double)jas[wl * 4] : 0.0);
      const real lp_prop = (real)logpsi;
      // accept = 2 (log|psi'| - log|psi|) > log u  [| age >= max_age]   (k_accept); identical in all lanes of the group
      const real log_prob = 2 * (lp_prop - lp_old);
      const real lu = sizeof(real) == 4 ? (real)logf((float)u_b) : (real)log((double)u_b);
      bool acc = log_prob > lu;
      if (a.mc.max_age >= 0) acc = acc || (age_b >= a.mc.max_age);
      if (live) {
        if (acc) {
          real* rg = reinterpret_cast<real*>(a.mc.r) + b * 3 * N;
          for (int k = l; k < 3 * N; k += 16) rg[k] = r[k];
        }
        if (l == 0) {
          if (acc) {
            reinterpret_cast<real*>(a.mc.logpsi)[b] = lp_prop;
            a.mc.sign[b] = sign_p;
            a.mc.age[b] = 0;
            n_acc += 1;
          } else {
            a.mc.age[b] = age_b + 1;
          }
          if (a.mc.accept_out) a.mc.accept_out[b] = acc ? 1 : 0;
        }
      }
    }
  }
  if (prof) prof[2] = clock64();
  if (tid < 64) {                                   // WT <= 16 < 64: the walkers of the tile sit in wave 0
    for (int m = 1; m < 64; m <<= 1) n_acc += __shfl_xor(n_acc, m, 64);
    // acceptance count: a fire-and-forget atomic (no return value, no fence); the NEXT sub-step's prologue
    // turns it into the new step size (FusedMc in kernels.h)
    if (tid == 0 && n_acc) atomicAdd(a.mc.counters + a.mc.s % 3, n_acc);
  }
}

// MA1: every unit of the plan is one row block high (tiles of <= 16 MFMA rows per layer segment, e.g. 4-electron
// systems with 4-walker tiles): the taller unit bodies are not instantiated, which takes the kernel from 128
// registers with scratch spills to < 100 without any.
template <typename real, bool MA1>
__device__ __forceinline__ void fused2_body(const Fused2Args<real>& a) {
  HIP_DYNAMIC_SHARED(char, smem_raw)
  real* smem = reinterpret_cast<real*>(smem_raw);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w0 = blockIdx.x * a.WT;
  const int nw = (a.B - w0) < a.WT ? (a.B - w0) : a.WT;
  if (a.prof_wg && threadIdx.x == 0 && blockIdx.x < 8192) a.prof_wg[2 * blockIdx.x] = (long long)wall_clock64();
  {
    // positions of the tile -> LDS; in sub-step mode the proposal r' = r + tau * xi (electron_samplers.py:102-104)
    real* rs = smem + a.scratch_off;
    const int n = nw * a.li.N * 3;
    const long g0 = (long)w0 * a.li.N * 3;
    {
      int32_t* itab = reinterpret_cast<int32_t*>(smem_raw + a.it_off);
      for (int e = threadIdx.x; e < a.n_it; e += blockDim.x) itab[e] = a.itable[e];
    }
    if (a.mc.enabled) {
      // step size of this sub-step from the previous one's acceptance (the arithmetic of k_tau_update)
      real tau;
      if (a.mc.s == 0) {
        tau = reinterpret_cast<const real*>(a.mc.tau_in)[0];
      } else {
        tau = reinterpret_cast<const real*>(a.mc.tau_ring)[(a.mc.s + 1) & 1];
        if (a.mc.target > 0) {
          const real acceptance = (real)a.mc.counters[(a.mc.s + 2) % 3] / (real)a.B;
          const real m = acceptance > (real)0.05 ? acceptance : (real)0.05;
          tau = tau / ((real)a.mc.target / m);
        }
      }
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<real*>(a.mc.tau_ring)[a.mc.s & 1] = tau;
        a.mc.counters[(a.mc.s + 1) % 3] = 0;
      }
      const real* rg = reinterpret_cast<const real*>(a.mc.r) + g0;
      const real* nz = reinterpret_cast<const real*>(a.mc.noise) + g0;
      for (int e = threadIdx.x; e < n; e += blockDim.x) rs[e] = rg[e] + tau * nz[e];
    } else {
      for (int e = threadIdx.x; e < n; e += blockDim.x) rs[e] = a.r[g0 + e];
    }
    __syncthreads();
  }
  DescPtr d = (DescPtr)a.descs + ((const DQMC_UNIFORM int32_t*)a.wave_begin)[wave];
  BSet<real> nxt;
  {
    const int fu = ((const DQMC_UNIFORM int32_t*)a.wave_begin)[4 + wave];      // first unit of this wave's list, or -1
    fused2_prefetch_unit<real>(a, (DescPtr)a.descs + (fu >= 0 ? fu : 0), nxt);  // (no unit: any valid descriptor's weights)
  }
  const bool stamp = a.prof != nullptr && blockIdx.x == 0 && (threadIdx.x & 63) == 0;
  int n_d = 0;
  if (stamp) a.prof[wave * 256] = clock64();
  // The arbiter of a SIMD serves its oldest wave first, so of the four tiles that share a CU the first placed runs at
  // nearly the speed of a lone tile and the last placed absorbs all the waiting (89 / 101 / 116 / 129 us) -- and the CU runs
  // three, two, one tile(s) for the last 40 us.  Rotating the issue priority among the co-resident tiles lets them advance
  // together (workgroup b is the (b / n_cu)-th placed on its CU: dispatch fills the CUs breadth-first).
  const int prio_slot = a.prio_mode ? (int)(blockIdx.x / (unsigned)a.stagger_div) & 3 : 0;
  int prio_tick = 0;
  for (;; ++d) {
    const int kind = d->kind;
    if (kind == 0) break;
    if (a.prio_mode && (kind == 2 || a.prio_mode == 2)) {
      ++prio_tick;
      const int pv = prio_slot + prio_tick;
      switch (pv & 3) {            // (the priority is an instruction immediate)
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
    }
    if (kind == 2) {
      __syncthreads();
    } else if (kind == 5) {
      fused2_unit_lean<real>(a, d, nxt);
    } else if (kind == 7) {
      FusedBfUnit<real, 1>::lean(a, d, nxt);
    } else if (kind == 6) {
      if (MA1) {
        FusedBfUnit<real, 1>::run(a, d, nxt);
      } else {
        const int ma = d->ma;
        if (ma == 1) FusedBfUnit<real, 1>::run(a, d, nxt);
        else if (ma == 2) FusedBfUnit<real, 2>::run(a, d, nxt);
        else if (ma == 3) FusedBfUnit<real, 3>::run(a, d, nxt);
        else FusedBfUnit<real, 4>::run(a, d, nxt);
      }
    } else if (kind == 1) {
      if (MA1) {
        fused2_unit<real, 1>(a, d, nxt);
      } else {
        const int ma = d->ma;
        if (ma == 1) fused2_unit<real, 1>(a, d, nxt);
        else if (ma == 2) fused2_unit<real, 2>(a, d, nxt);
        else if (ma == 3) fused2_unit<real, 3>(a, d, nxt);
        else fused2_unit<real, 4>(a, d, nxt);
      }
    } else {
      fused2_generic<real>(a, (OpPtr)a.ops + d->op, nw);
    }
    if (stamp && n_d < 254) a.prof[wave * 256 + (++n_d)] = clock64();
  }
  if (a.mc.enabled) {
    fused2_mc_tail<real>(a, nw, stamp ? a.prof + wave * 256 + n_d + 1 : nullptr);     // every wave list ends with a barrier
  }
  if (a.prof_wg && threadIdx.x == 0 && blockIdx.x < 8192) a.prof_wg[2 * blockIdx.x + 1] = (long long)wall_clock64();
}

// OCC = workgroups (of 4 waves) the register allocation must leave room for per CU.
template <typename real, int OCC, bool MA1>
__global__ void __launch_bounds__(256, OCC) k_fused2_value(const Fused2Args<real> a) { fused2_body<real, MA1>(a); }

template <typename real, bool MA1> static void launch_fused2_ma(hipStream_t st, const Fused2Args<real>& a, int n_blocks, size_t lds_bytes, int occ) {
  const dim3 g((unsigned)n_blocks), b(256);
  if (sizeof(real) == 8) occ = 2;     // the float64 (parity) build needs the full register file
  if (occ >= 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fused2_value<real, 4, MA1>), g, b, lds_bytes, st, a);
  else if (occ == 3) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fused2_value<real, 3, MA1>), g, b, lds_bytes, st, a);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fused2_value<real, 2, MA1>), g, b, lds_bytes, st, a);
}
template <typename real> void launch_fused2_value(hipStream_t st, const Fused2Args<real>& a, int n_blocks, size_t lds_bytes, int occ) {
  if (a.ma1) launch_fused2_ma<real, true>(st, a, n_blocks, lds_bytes, occ);
  else launch_fused2_ma<real, false>(st, a, n_blocks, lds_bytes, occ);
}
template <typename real> int fused2_set_lds_limit(size_t lds_bytes) {
  int rc = 0;
#define DQMC_SET(OCC, M) rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused2_value<real, OCC, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  DQMC_SET(2, true) DQMC_SET(3, true) DQMC_SET(4, true) DQMC_SET(2, false) DQMC_SET(3, false) DQMC_SET(4, false)
#undef DQMC_SET
  return rc;
}

template void launch_fused2_value<float>(hipStream_t, const Fused2Args<float>&, int, size_t, int);
template void launch_fused2_value<double>(hipStream_t, const Fused2Args<double>&, int, size_t, int);
template int fused2_set_lds_limit<float>(size_t);
template int fused2_set_lds_limit<double>(size_t);

}  // namespace dqmc

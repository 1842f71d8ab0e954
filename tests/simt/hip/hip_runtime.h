// SIMT emulation shim -- TEST INFRASTRUCTURE ONLY (tests/simt).
//
// Lets the unmodified kernel sources of deepqmc_amd/csrc be compiled with g++ and executed
// on the host so that kernel index arithmetic (MFMA fragment layouts, lane shuffles, LDS
// tiling, barriers) can be checked against the CPU oracle in a container without a GPU.
// Every workgroup runs as one set of cooperatively scheduled fibers (ucontext), one per
// thread; __syncthreads / wave shuffles / MFMA are rendezvous points.  The product library
// (libdqmc_hip.so) is never built from or linked against this file, and the product Python
// package never loads the emulated library.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorNotSupported 801
typedef struct simt_stream* hipStream_t;
typedef struct simt_event { double t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

inline const char* hipGetErrorString(hipError_t) { return "simt-emu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
#define hipStreamNonBlocking 1u
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new simt_event{0}; return hipSuccess; }
#define hipEventDisableTiming 2u
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new simt_event{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
namespace simt { char* dyn_smem(size_t ensure_bytes); }
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)simt::dyn_smem(0);

namespace simt {
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_block();
// wave rendezvous: deposit 16 bytes per lane, then read another lane's deposit
void wave_exchange(const void* mine, size_t n, int src_lane, void* out);
void wave_gather_begin(const void* a, const void* b, size_t n);   // deposit two values
const char* wave_slot_a(int lane);
const char* wave_slot_b(int lane);
void wave_gather_end();
int lane_id();
}  // namespace simt

template <typename K, typename... Args>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t dyn_lds, hipStream_t, Args... args) {
  simt::dyn_smem(dyn_lds);
  simt::run_grid(grid, block, [=]() { kernel(args...); });
}

inline void __syncthreads() { simt::sync_block(); }
template <typename T> inline T __shfl(T v, int src, int = 64) {
  T o;
  simt::wave_exchange(&v, sizeof(T), src & 63, &o);
  return o;
}
template <typename T> inline T __shfl_xor(T v, int mask, int = 64) {
  T o;
  simt::wave_exchange(&v, sizeof(T), (simt::lane_id() ^ mask) & 63, &o);
  return o;
}
inline float __expf(float x) { return expf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline void __threadfence() {}
inline long long wall_clock64() { return 0; }
// scheduling / scalarisation hints of the device build: no-ops on the host
inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __builtin_amdgcn_s_sleep(int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
// (lanes are fibers here: the wave-level ordering point has to yield to the other lanes, which a shuffle does)
inline void __builtin_amdgcn_wave_barrier() { (void)__shfl(0, 0, 64); }
// constant address space qualifier of the fused kernel's descriptor pointers
#define DQMC_UNIFORM
// No graph API in the harness: capture is refused, the engine falls back to eager launches (Engine::graph_broken).
typedef struct simt_graph* hipGraph_t;
typedef struct simt_graph_exec* hipGraphExec_t;
#define hipStreamCaptureModeRelaxed 2
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline long long clock64() { static long long t = 0; return ++t; }
inline int atomicAdd(int* p, int v) { int o = *p; *p += v; return o; }
inline float atomicAdd(float* p, float v) { float o = *p; *p += v; return o; }

typedef float simt_f32x4 __attribute__((vector_size(16)));
typedef double simt_f64x4 __attribute__((vector_size(32)));
// v_mfma_f32_16x16x4_f32: A[row=l&15][k=l>>4], B[k=l>>4][col=l&15], D col=l&15 row=(l>>4)*4+reg;
// exact f32 fmaf chain in k order (MI355X guide).
// hardware transcendentals of the value path (v_exp_f32, v_rcp_f32)
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline simt_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, simt_f32x4 c, int, int, int) {
  simt::wave_gather_begin(&a, &b, sizeof(float));
  const int l = simt::lane_id(), col = l & 15;
  for (int reg = 0; reg < 4; ++reg) {
    const int row = (l >> 4) * 4 + reg;
    float acc = c[reg];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, simt::wave_slot_a(k * 16 + row), 4);
      memcpy(&bv, simt::wave_slot_b(k * 16 + col), 4);
      acc = fmaf(av, bv, acc);
    }
    c[reg] = acc;
  }
  simt::wave_gather_end();
  return c;
}
// v_cvt_pk_bf16_f32 (round to nearest even) and v_mfma_f32_16x16x32_bf16: lane l holds row / column l&15 and
// k = 8*(l>>4) .. +7 (two per word, even k low); products of bf16 numbers are exact in f32, the sum is taken in double
// and rounded once (the hardware's internal order is not documented; the kernels do not depend on it).
namespace dqmc {
inline uint32_t bf_pack2(float lo, float hi) {
  auto cv = [](float f) -> uint32_t {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  };
  return cv(lo) | (cv(hi) << 16);
}
}
inline simt_f32x4 simt_mfma_f32_16x16x32_bf16(const uint32_t (&a)[4], const uint32_t (&b)[4], simt_f32x4 c) {
  simt::wave_gather_begin(a, b, 16);
  const int l = simt::lane_id(), col = l & 15;
  auto val = [](const char* slot, int i) { uint32_t w; memcpy(&w, slot + 4 * (i >> 1), 4); uint32_t u = (i & 1) ? (w & 0xffff0000u) : (w << 16); float f; memcpy(&f, &u, 4); return f; };
  for (int reg = 0; reg < 4; ++reg) {
    const int row = (l >> 4) * 4 + reg;
    double acc = c[reg];
    for (int kg = 0; kg < 4; ++kg)
      for (int i = 0; i < 8; ++i) acc += (double)val(simt::wave_slot_a(kg * 16 + row), i) * (double)val(simt::wave_slot_b(kg * 16 + col), i);
    c[reg] = (float)acc;
  }
  simt::wave_gather_end();
  return c;
}
// v_mfma_f64_16x16x4_f64: same A/B maps; D col=l&15 row=(l>>4)+4*reg.
inline simt_f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, simt_f64x4 c, int, int, int) {
  simt::wave_gather_begin(&a, &b, sizeof(double));
  const int l = simt::lane_id(), col = l & 15;
  for (int reg = 0; reg < 4; ++reg) {
    const int row = (l >> 4) + 4 * reg;
    double acc = c[reg];
    for (int k = 0; k < 4; ++k) {
      double av, bv;
      memcpy(&av, simt::wave_slot_a(k * 16 + row), 8);
      memcpy(&bv, simt::wave_slot_b(k * 16 + col), 8);
      acc = fma(av, bv, acc);
    }
    c[reg] = acc;
  }
  simt::wave_gather_end();
  return c;
}

// Register-resident MFMA rate of one MI355X: what the matrix pipes deliver with nothing else in the way (no LDS, no memory),
// per instruction kind and waves per SIMD -- and at WHICH CLOCK: every kernel stamps the shader-cycle counter (clock64 = s_memtime)
// and the constant 100 MHz counter (wall_clock64 = s_memrealtime) of one wave, so that TF/s = flops per cycle x effective clock
// can be taken apart.  The chip clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"): a register-resident MFMA
// loop is the most power-dense thing it can run, and the denser the instruction the lower the clock it sustains.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
template <int KIND, int NACC> __global__ void __launch_bounds__(256) k_peak(float* out, int iters, long long* stamps) {
  const int lane = threadIdx.x & 63;
  const bool st = stamps != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  long long c0 = 0, w0 = 0;
  if (st) { c0 = clock64(); w0 = wall_clock64(); }
  if (KIND == 0) {
    d4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = d4{0, 0, 0, 0};
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    double s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (st) { stamps[0] = clock64() - c0; stamps[1] = wall_clock64() - w0; }
    if (s == 12345.678) out[0] = (float)s;
  } else if (KIND == 1) {
    f4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f4{0, 0, 0, 0};
    float a = 1.0f + lane * 1e-3f, b = 1.0f - lane * 1e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (st) { stamps[0] = clock64() - c0; stamps[1] = wall_clock64() - w0; }
    if (s == 12345.678f) out[0] = s;
  } else {
    f4 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f4{0, 0, 0, 0};
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + lane * 1e-2f); b[j] = (__bf16)(1.0f - lane * 1e-2f); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (st) { stamps[0] = clock64() - c0; stamps[1] = wall_clock64() - w0; }
    if (s == 12345.678f) out[0] = s;
  }
}
template <int KIND, int NACC> static void run(const char* name, double flop_per_mfma, int wg_per_cu) {
  float* out; hipMalloc(&out, 4);
  long long* stamps; hipMalloc(&stamps, 16);
  const int iters = 20000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_peak<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, 100, (long long*)nullptr);
  hipEventRecord(e0);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_peak<KIND, NACC>), dim3(grid), dim3(256), 0, 0, out, iters, stamps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * NACC * flop_per_mfma;
  long long h[2] = {0, 1};
  hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / ((double)h[1] / 100.0);                      // shader cycles per microsecond of the 100 MHz counter
  // issue interval of one SIMD implied by the rate at the measured clock (1024 SIMDs).  Not the stamped wave's own cycle count: with
  // several waves per SIMD the arbiter serves the oldest wave first, workgroup 0 finishes long before the launch does
  const double cyc_per_mfma = flop_per_mfma * 1024.0 * mhz * 1e6 / (flops / (ms * 1e-3));
  printf("%-28s %d acc, %d waves/SIMD: %8.1f TFLOP/s  (%.2f ms)  shader clock while it ran %6.0f MHz -> one MFMA per %5.1f cycles and SIMD\n",
         name, NACC, wg_per_cu, flops / ms / 1e9, ms, mhz, cyc_per_mfma);
  hipFree(out); hipFree(stamps);
}
int main() {
  for (int w : {1, 2, 4}) {
    if (w == 1) { run<0, 4>("v_mfma_f64_16x16x4_f64", 2048, 1); run<0, 8>("v_mfma_f64_16x16x4_f64", 2048, 1); }
    if (w == 2) { run<0, 4>("v_mfma_f64_16x16x4_f64", 2048, 2); run<0, 8>("v_mfma_f64_16x16x4_f64", 2048, 2); }
    if (w == 4) { run<0, 4>("v_mfma_f64_16x16x4_f64", 2048, 4); run<0, 8>("v_mfma_f64_16x16x4_f64", 2048, 4); }
  }
  run<1, 8>("v_mfma_f32_16x16x4_f32", 2048, 1); run<1, 8>("v_mfma_f32_16x16x4_f32", 2048, 2); run<1, 8>("v_mfma_f32_16x16x4_f32", 2048, 4);
  run<2, 8>("v_mfma_f32_16x16x32_bf16", 16384, 1); run<2, 8>("v_mfma_f32_16x16x32_bf16", 16384, 2); run<2, 8>("v_mfma_f32_16x16x32_bf16", 16384, 4);
  return 0;
}

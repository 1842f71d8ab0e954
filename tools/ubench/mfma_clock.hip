// Micro-benchmark: effective shader clock and sustained f32 MFMA throughput with every SIMD busy
// (wall time by HIP events vs the s_memtime cycle counter of one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((vector_size(16)));
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int n) {
  f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* d; long long* c; hipMalloc((void**)&d, 2048 * 256 * 4); hipMalloc((void**)&c, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int n = 100000, blocks = 256 * waves_per_simd;
    k<<<blocks, 256>>>(d, c, 1000); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<blocks, 256>>>(d, c, n); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 /*waves*/ * 4.0 * n * 2048.0;
    printf("%d wave(s)/SIMD: %.3f ms, %lld cycles -> %.2f GHz effective, %.1f TFLOP/s f32 MFMA sustained\n", waves_per_simd, ms, cy,
           cy / (ms * 1e6), flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}

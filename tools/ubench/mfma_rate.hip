// Micro-benchmark: issue rate of dependent v_mfma chains per wave (f64 16x16x4 vs f32 16x16x4), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((vector_size(32)));
typedef float f4 __attribute__((vector_size(16)));
__global__ void k64(double* out, int n) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
  }
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (2.0 * n);
}
__global__ void k32(float* out, int n) {
  f4 a0 = {0, 0, 0, 0}, a1 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  long t0 = clock64();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
  }
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (2.0f * n);
}
int main() {
  double* d; float* f; hipMalloc(&d, 1 << 20); hipMalloc(&f, 1 << 20);
  for (int waves = 1; waves <= 2; ++waves) {
    k64<<<1, 256 * waves>>>(d, 2000); k32<<<1, 256 * waves>>>(f, 2000); hipDeviceSynchronize();
    double h; float g; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&g, f, 4, hipMemcpyDeviceToHost);
    printf("waves/SIMD %d: cycles per MFMA (2 independent chains per wave): f64 16x16x4 = %.1f, f32 16x16x4 = %.1f\n", waves, h, g);
  }
  return 0;
}

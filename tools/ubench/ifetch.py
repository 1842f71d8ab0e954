"""Microbenchmark generator (round 6): is a long STRAIGHT-LINE instruction stream bound by instruction fetch?  The same N
independent VALU instructions (8-byte v_fma_f32 on 16 rotating registers) once as straight-line code of N x 8 bytes and once as a
loop whose body fits the instruction cache; one wave per SIMD, 256 workgroups of 256 threads (4 waves per CU run the same code).
usage: python ifetch.py out.hip"""
import sys
def body(n, regs=16):
    return ''.join(f'"v_fma_f32 %{i % regs}, %{i % regs}, %16, %17\\n"\n' for i in range(n))
def kernel(name, n_straight, loop_body, loop_iters):
    ops = ', '.join(f'"+v"(f{i})' for i in range(16))
    s = f'__global__ void __launch_bounds__(256) {name}(float* out, long long* st) {{\n  float ' + ', '.join(f'f{i} = threadIdx.x + {i}' for i in range(16)) + ';\n  float m = 1.0001f, c = 0.001f;\n  long long c0 = clock64();\n'
    if n_straight:
        s += f'  asm volatile({body(n_straight)} : {ops} : "v"(m), "v"(c));\n'
    else:
        s += f'  for (int it = 0; it < {loop_iters}; ++it) asm volatile({body(loop_body)} : {ops} : "v"(m), "v"(c));\n'
    s += '  long long c1 = clock64();\n  float r = ' + ' + '.join(f'f{i}' for i in range(16)) + ';\n  if (r == 12345.678f) out[0] = r;\n  if (st && blockIdx.x == 0 && (threadIdx.x & 63) == 0) st[threadIdx.x >> 6] = c1 - c0;\n}\n'
    return s
src = '#include <hip/hip_runtime.h>\n#include <cstdio>\n'
cfgs = [('k_s2k', 2048, 0, 0), ('k_s8k', 8192, 0, 0), ('k_s16k', 16384, 0, 0), ('k_l16k', 0, 512, 32), ('k_l64k', 0, 512, 128)]
for c in cfgs: src += kernel(*c)
src += 'int main() {\n  float* out; hipMalloc(&out, 4); long long* st; hipMalloc(&st, 64); long long h[4]; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;\n'
for (name, ns, lb, li) in cfgs:
    n = ns or lb * li
    for grid in (256, 1):
        src += f'  for (int i = 0; i < 2; ++i) {name}<<<{grid}, 256>>>(out, st); hipDeviceSynchronize(); hipEventRecord(e0); for (int i = 0; i < 10; ++i) {name}<<<{grid}, 256>>>(out, st); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);\n'
        src += f'  printf("%-8s grid %3d: {n} instructions ({n * 8 // 1024} KB {"straight-line" if ns else "loop of 4 KB"}): %7.1f us per launch, wave 0 %8lld cycles = %5.2f cycles per instruction\\n", "{name}", {grid}, ms * 100, h[0], (double)h[0] / {n});\n'
src += '  return 0;\n}\n'
open(sys.argv[1], 'w').write(src)

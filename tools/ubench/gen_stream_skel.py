"""Microbenchmark generator (round 6): the skeleton of a wave-per-tile, register-resident value kernel whose
weights stream global -> registers -> LDS ring -> A operands (ds_read_b128), shared by the 4 waves of a workgroup.
Emits straight-line code for `n_triples` (hi, mid, lo) weight-fragment triples: per triple 3 ds_read_b128 (issued PF
triples ahead) + 6*RB v_mfma_f32_16x16x32_bf16 on two accumulator chains per output block (+ `valu` filler FMAs on
independent chains, interleaved with the MFMAs in source order); ring stage s+1 is written in the MIDDLE of the consumption of
stage s-... (see gen), its global loads are issued `g_ahead` stages ahead.
Variants: ring (all planes through LDS), hybrid (lo plane by direct per-wave global loads), direct (no LDS).
Questions it answers on the MI355X: (1) does a 100 KB straight-line instruction stream run at the rate of a
10 KB one (instruction cache), (2) cycles per MFMA with ONE wave per SIMD fed from LDS, (3) cost of the ring.
usage: python gen_stream_skel.py out.hip"""
import sys

def gen(name, n_triples, rb=1, valu=0, pf=2, mode='ring', stage=16, ring=3, w_pos=0.5, pfg=6, spread=False, pin=False):
    L = []
    A = L.append
    A(f'__global__ void __launch_bounds__(256) {name}(const uint4* __restrict__ wt, float* out, long long* stamps) {{')
    A('  extern __shared__ uint4 ring[];')
    A('  const int tid = threadIdx.x, lane = tid & 63;')
    A('  bf8 x[12]; f4 accs[8], accb[8]; float fz[8];')
    A('  for (int j = 0; j < 12; ++j) for (int k = 0; k < 8; ++k) x[j][k] = (__bf16)(0.01f * (lane + j + k));')
    A('  for (int j = 0; j < 8; ++j) { accs[j] = f4{0, 0, 0, 0}; accb[j] = f4{0, 0, 0, 0}; fz[j] = 0.5f + lane + j; }')
    A('  const uint4* wl = ring + lane;')
    A('  const uint4* gp = wt + tid;')
    A('  const uint4* gl = wt + lane;')
    A('  long long c0 = clock64();')
    planes_ring = 3 if mode == 'ring' else (2 if mode == 'hybrid' else 0)
    n_frag = planes_ring * n_triples                   # fragments that travel through the ring
    n_stage = (n_frag + stage - 1) // stage if planes_ring else 0
    per_thread = stage * 64 // 256
    def load_stage(s):
        reg = f'r{s % 3}_'
        for q in range(per_thread):
            A(f'  {reg}{q} = (wt + {s * stage * 64 + q * 256})[tid];')
    def write_stage(s):
        reg = f'r{s % 3}_'
        slot = s % ring
        for q in range(per_thread):
            A(f'  ring[{slot * stage * 64 + q * 256} + tid] = {reg}{q};')
    if planes_ring:
        A('  uint4 ' + ', '.join(f'r{j}_{q}' for j in range(3) for q in range(per_thread)) + ';')
        if spread:
            load_stage(0); write_stage(0); load_stage(1)
            for q in range(per_thread): A(f'  r0_{q} = r1_{q};')
        else:
            load_stage(0); write_stage(0); load_stage(1); load_stage(2)
        A('  __syncthreads();')
    # events keyed by ring-fragment index f at which they are emitted (just before the read of fragment f):
    #   f == s*stage                    : barrier B_s (s >= 1), then global loads of stage s+2
    #   f == s*stage + w_pos*stage      : ds_write of stage s+1 (loaded >= 1.5 stages ago)
    def before_ring_read(f):
        s, o = divmod(f, stage)
        if o == 0 and s >= 1:
            A('  __syncthreads();')
            if not spread and s + 2 < n_stage: load_stage(s + 2)
        if spread:
            # piece q of stage s+1 goes to the ring, its registers are reloaded with piece q of stage s+2: one pair every
            # stage/per_thread fragments (a load has a whole stage of MFMAs to land)
            step = stage // per_thread
            if o % step == step // 2:
                q = o // step
                reg = f'r0_{q}'
                if s + 1 < n_stage and s >= 0:
                    A(f'  ring[{((s + 1) % ring) * stage * 64 + q * 256} + tid] = {reg};')
                if s + 2 < n_stage:
                    A(f'  {reg} = (wt + {(s + 2) * stage * 64 + q * 256})[tid];')
        elif o == int(w_pos * stage) and s + 1 < n_stage:
            write_stage(s + 1)
    def emit_reads(t):
        if t >= n_triples: return
        for pl in range(3):
            if pl < planes_ring:
                f = planes_ring * t + pl
                before_ring_read(f)
                s, o = divmod(f, stage)
                A(f'  const uint4 w{t}_{pl} = wl[{(s % ring) * stage * 64 + o * 64}];')
    def emit_greads(t):
        if t >= n_triples: return
        for pl in range(planes_ring, 3):
            A(f'  const uint4 w{t}_{pl} = (wt + {(3 * t + pl) * 64})[lane];')
    for t in range(min(pf, n_triples)): emit_reads(t)
    for t in range(min(pfg, n_triples)): emit_greads(t)
    prods = [(2, 0), (1, 0), (1, 1), (0, 1), (0, 2), (0, 0)]       # (weight plane, activation plane): small, big alternating
    for t in range(n_triples):
        emit_reads(t + pf)
        emit_greads(t + pfg)
        b = (t // 4) % 8
        c = t % 4
        vi = 0
        for r in range(rb):
            for k, (wp, xp) in enumerate(prods):
                acc = f'accs[{(b + r) % 8}]' if wp + xp == 2 else f'accb[{(b + r) % 8}]'
                A(f'  {acc} = MF(w{t}_{wp}, x[{(3 * c + xp + r) % 12}], {acc});')
                nv = (valu * (r * 6 + k + 1)) // (6 * rb) - (valu * (r * 6 + k)) // (6 * rb)
                for _ in range(nv):
                    A(f'  fz[{vi % 8}] = __builtin_fmaf(fz[{vi % 8}], 1.0001f, {0.001 * (vi + 1)}f);')
                    vi += 1
                if pin: A('  __builtin_amdgcn_sched_barrier(0);')
        A('  __builtin_amdgcn_sched_barrier(0);')
    A('  long long c1 = clock64();')
    A('  float s = 0; for (int j = 0; j < 8; ++j) s += fz[j] + accs[j][0] + accs[j][1] + accs[j][2] + accs[j][3] + accb[j][0] + accb[j][1] + accb[j][2] + accb[j][3];')
    A('  if (s == 12345.678f) out[0] = s;')
    A('  if (stamps && blockIdx.x == 0 && (tid & 63) == 0) stamps[tid >> 6] = c1 - c0;')
    A('}')
    return '\n'.join(L), n_triples * 6 * rb

HDR = r'''// generated by tools/ubench/gen_stream_skel.py -- microbenchmark, not part of the library
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f4 MF(const uint4& a, const bf8& b, f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), b, c, 0, 0, 0);
}
'''

def main(path):
    cfgs = [('k_ring16', dict(n_triples=400)), ('k_spread16', dict(n_triples=400, spread=True)), ('k_spread32', dict(n_triples=400, spread=True, stage=32)),
            ('k_spread16_pf3', dict(n_triples=400, spread=True, pf=3)),
            ('k_hspread16', dict(n_triples=400, spread=True, mode='hybrid')),
            ('k_sp16_v6', dict(n_triples=400, spread=True, valu=6, pin=True)), ('k_sp16_v12', dict(n_triples=400, spread=True, valu=12, pin=True)),
            ('k_sp16_v18', dict(n_triples=400, spread=True, valu=18, pin=True)), ('k_sp16_v24', dict(n_triples=400, spread=True, valu=24, pin=True)),
            ('k_sp16_v0p', dict(n_triples=400, spread=True, valu=0, pin=True)),
            ('k_sp16_rb3', dict(n_triples=400, spread=True, rb=3)), ('k_sp16_rb3_v36', dict(n_triples=400, spread=True, rb=3, valu=36, pin=True))]
    src = [HDR]
    runs = []
    for (name, kw) in cfgs:
        s, nm = gen(name, **kw)
        src.append(s)
        runs.append((name, kw['n_triples'], nm))
    src.append('int main() {\n  uint4* wt; hipMalloc(&wt, 8 << 20); hipMemset(wt, 0x3c, 8 << 20); float* out; hipMalloc(&out, 4); long long* st; hipMalloc(&st, 64);')
    src.append('  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms; long long h[4];')
    for (name, nt, nm) in runs:
        src.append(f'  hipFuncSetAttribute((const void*){name}, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);')
        for grid in (256, 1):
            src.append(f'  for (int i = 0; i < 3; ++i) {name}<<<{grid}, 256, 128 * 1024>>>(wt, out, st); hipDeviceSynchronize();')
            src.append(f'  hipEventRecord(e0); for (int i = 0; i < 20; ++i) {name}<<<{grid}, 256, 128 * 1024>>>(wt, out, st); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);')
            src.append(f'  hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);')
            src.append(f'  printf("%-16s grid %3d triples %5d: %8.1f us per launch, waves %7lld %7lld %7lld %7lld cycles = %5.1f cycles per MFMA (%d MFMAs)\\n", "{name}", {grid}, {nt}, ms * 1e3 / 20, h[0], h[1], h[2], h[3], (double)h[0] / {nm}, {nm});')
    src.append('  return 0;\n}')
    open(path, 'w').write('\n'.join(src))

if __name__ == '__main__':
    main(sys.argv[1])

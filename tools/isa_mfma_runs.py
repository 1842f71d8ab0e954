"""ISA check: distribution of the number of non-MFMA instructions between consecutive MFMAs of a generated kernel, and a
cycle estimate from the measured filler table (tools/ubench/mfma_fill.hip, one wave per SIMD: 0..2 fillers ~20 cycles,
3 -> 24.6, >= 4 -> 39 + 2 per further one).  usage: python tools/isa_mfma_runs.py kernel.s"""
import sys, collections
lines = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith((';', '.'))]
runs = collections.Counter(); run = 0; seen = False; kinds = collections.Counter()
for l in lines:
    op = l.split()[0]
    if op.startswith('v_mfma'):
        if seen: runs[run] += 1
        run = 0; seen = True
    elif op.endswith(':'):
        continue
    else:
        w = 0 if op in ('s_waitcnt', 's_nop') else 1
        run += w
        if seen: kinds[op] += 1
est = 0
for r, n in sorted(runs.items()):
    cost = 20 if r <= 2 else (24.6 if r == 3 else 39 + 2 * (r - 4))
    est += n * cost
print('runs of non-MFMA instructions between MFMAs:', ', '.join(f'{r}:{n}' for r, n in sorted(runs.items())[:12]), '...')
big = sum(n for r, n in runs.items() if r >= 4); bigi = sum(r * n for r, n in runs.items() if r >= 4)
print(f'runs >= 4: {big} holding {bigi} instructions; runs of 3: {runs[3]}; estimate {est:.0f} cycles')

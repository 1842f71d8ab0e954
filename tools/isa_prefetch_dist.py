"""ISA check for the generated sub-step kernels: how many MFMAs separate each ds_read_b128 of a weight fragment from the
first MFMA that consumes it (0-1 = the scheduler sank the read next to its use: the LDS latency is exposed).
usage: python tools/isa_prefetch_dist.py kernel.s"""
import re, sys, collections
lines = [l.strip() for l in open(sys.argv[1]) if l.strip() and not l.strip().startswith(';')]
pend = {}      # first reg of dst -> mfma count at issue
n_mfma = 0
dist = collections.Counter()
waits0 = 0
for l in lines:
    m = re.match(r'ds_read_b128 v\[(\d+):(\d+)\]', l)
    if m:
        pend[int(m.group(1))] = n_mfma
        continue
    m = re.match(r'v_mfma_f32_16x16x32_bf16 \S+ v\[(\d+):(\d+)\], ', l)
    if m:
        a = int(m.group(1))
        if a in pend:
            dist[min(n_mfma - pend.pop(a), 40)] += 1
        n_mfma += 1
print('MFMAs', n_mfma)
tot = sum(dist.values())
acc = 0
for k in sorted(dist):
    acc += dist[k]
    print(f'  distance {k:3d}: {dist[k]:5d}  (cum {acc / tot:.2f})')

"""Summary of a rocprofv3 --kernel-trace csv: wall time between the first and last kernel of the LAST `frac` of the trace, union busy
time, per-kernel totals split by float / double instances."""
import csv, sys, glob, collections
path = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = glob.glob(path + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[0][0]; t1 = max(r[1] for r in rows); cut = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[0] >= cut]
wall = (max(r[1] for r in rows) - rows[0][0]) / 1e6
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e: busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print('window %.1f ms, %d kernels, union busy %.1f ms (%.1f %%), gaps > 50 us: %d totalling %.1f ms, > 1 ms: %d totalling %.1f ms' % (
    wall, len(rows), busy / 1e6, busy / 1e4 / wall, sum(g > 5e4 for g in gaps), sum(g for g in gaps if g > 5e4) / 1e6,
    sum(g > 1e6 for g in gaps), sum(g for g in gaps if g > 1e6) / 1e6))
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in rows:
    k = n.split('(')[0][:90]; tot[k] += e - s; cnt[k] += 1
for k, v in tot.most_common(22): print('%9.1f ms %6d x %8.1f us  %s' % (v / 1e6, cnt[k], v / cnt[k] / 1e3, k))

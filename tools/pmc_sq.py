"""Per-kernel-family averages of the SQ counters collected by run_pmc.sh (rocprofv3 --pmc, csv)."""
import csv, glob, json, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for path in glob.glob(d + '/*counter_collection.csv'):
        for row in csv.DictReader(open(path)):
            m = re.match(r'(?:void\s+)?(?:dqmc::)?(?:\(anonymous namespace\)::)?(k_[a-z_0-9]+)', row['Kernel_Name'])
            fam = m.group(1) if m else row['Kernel_Name'][:40]
            if os.environ.get('PMC_FULLNAME'):          # keep the template arguments (one entry per instantiation)
                m2 = re.match(r'(?:void\s+)?(?:dqmc::)?(?:\(anonymous namespace\)::)?(k_[a-z_0-9]+(?:<[^>]*>)?)', row['Kernel_Name'])
                fam = m2.group(1) if m2 else fam
            a = acc[fam][row['Counter_Name']]
            a[0] += 1; a[1] += float(row['Counter_Value'])
out = {}
for fam, cs in acc.items():
    o = {c: v[1] / v[0] for c, v in cs.items()}
    o['launches'] = max(v[0] for v in cs.values())
    wc = o.get('SQ_WAVE_CYCLES')
    if wc:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS',
                  'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_VMEM', 'SQ_WAIT_INST_LDS'):
            if c in o: o['frac_' + c[3:]] = o[c] / wc
    if o.get('SQ_INSTS_MFMA'): o['valu_per_mfma'] = o.get('SQ_INSTS_VALU', 0) / o['SQ_INSTS_MFMA']
    if o.get('SQ_BUSY_CYCLES') and o.get('SQ_VALU_MFMA_BUSY_CYCLES'): o['mfma_busy_over_sq_busy'] = o['SQ_VALU_MFMA_BUSY_CYCLES'] / o['SQ_BUSY_CYCLES']
    out[fam] = o
json.dump(out, sys.stdout, indent=1)

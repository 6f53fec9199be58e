"""Sums rocprofv3 --pmc counter CSVs per kernel over all dispatches: python tools/sq_summary.py <dir> [<dir> ...]"""
import csv, glob, sys
tot = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:70]
            tot.setdefault(k, {}).setdefault(r['Counter_Name'], [0.0, 0])
            e = tot[k][r['Counter_Name']]; e[0] += float(r['Counter_Value']); e[1] += 1
for k, cs in sorted(tot.items()):
    if 'fa64' not in k and 'fa1_' not in k:
        continue
    n = max(v[1] for v in cs.values())
    print(k, '(%d dispatches, per-dispatch means)' % n)
    print('   ' + '  '.join('%s=%.4g' % (c, v[0] / v[1]) for c, v in sorted(cs.items())))
    g = lambda c: cs[c][0] / cs[c][1] if c in cs else None
    wc = g('SQ_WAVE_CYCLES')
    if wc:
        parts = []
        for c, label in (('SQ_VALU_MFMA_BUSY_CYCLES', 'MFMA pipe busy (cycles x 4 SIMD-normalised: see note)'), ('SQ_ACTIVE_INST_VALU', 'VALU issue active'),
                         ('SQ_WAIT_INST_ANY', 'waiting on an instruction dependency'), ('SQ_WAIT_INST_LDS', 'of which LDS'), ('SQ_ACTIVE_INST_ANY', 'any instruction active'),
                         ('SQ_WAIT_ANY', 'parked at a wait / barrier'), ('SQ_LDS_BANK_CONFLICT', 'LDS bank conflict cycles'), ('SQ_LDS_IDX_ACTIVE', 'LDS array active')):
            if g(c) is not None:
                parts.append('%s / wave cycles = %.3f' % (c, g(c) / wc))
        print('   ' + '; '.join(parts))

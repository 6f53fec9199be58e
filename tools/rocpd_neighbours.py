"""Which kernels run right before / after the dispatches whose name contains <substr>? (rocprofv3 rocpd database, same queue).
usage: python tools/rocpd_neighbours.py <db> <substr> [top]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); sub = sys.argv[2]; top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(db.execute(f"select d.queue_id, d.start, d.end, s.kernel_name, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id=s.id order by d.queue_id, d.start"))
short = lambda n: n.split('(')[0][-70:]
pairs, sizes = collections.Counter(), collections.Counter()
for i, r in enumerate(rows):
    if sub not in r[3]:
        continue
    prev = rows[i - 1] if i and rows[i - 1][0] == r[0] else None
    nxt = rows[i + 1] if i + 1 < len(rows) and rows[i + 1][0] == r[0] else None
    pairs[(short(prev[3]) if prev else '-', short(nxt[3]) if nxt else '-')] += 1
    sizes[(r[4], r[5])] += 1
print('%d dispatches match %r' % (sum(pairs.values()), sub))
for (a, b), n in pairs.most_common(top):
    print('%6d  after %-72s before %s' % (n, a, b))
print('grid x workgroup:', sizes.most_common(8))

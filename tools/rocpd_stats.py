"""Per-kernel totals from a rocprofv3 rocpd database (the default output format of rocprofv3 in ROCm 7.2)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc limit {int(sys.argv[2]) if len(sys.argv) > 2 else 15}"
print('%-100s %7s %12s %9s' % ('kernel', 'calls', 'total_us', 'avg_us'))
for r in db.execute(q):
    print('%-100s %7d %12.1f %9.2f' % (r[0][:100], r[1], r[2], r[3]))

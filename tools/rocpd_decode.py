"""Per-kernel table and GPU-busy share of the decode steps in a rocprofv3 rocpd database of `bench.py --mode decode`.
Usage: python tools/rocpd_decode.py <results.db>"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(db.execute(f"select d.start,d.end,s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
dec = [r for r in rows if 'gemv' in r[2] or 'attn_decode' in r[2] or 'attn_split' in r[2] or 'decode' in r[2] or 'embed_ln_fwd' in r[2] or 'dec_attn' in r[2] or 'dec_embed' in r[2] or 'dec_sample' in r[2]]
c, t = collections.Counter(), collections.Counter()
for r in dec:
    c[r[2][:70]] += 1; t[r[2][:70]] += (r[1] - r[0]) / 1e3
print('%-72s %7s %10s %8s' % ('kernel', 'calls', 'total_us', 'avg_us'))
for k in sorted(c, key=lambda k: -t[k]):
    print('%-72s %7d %10.1f %8.2f' % (k, c[k], t[k], t[k] / c[k]))
half = dec[len(dec) // 2:]
span = (half[-1][1] - half[0][0]) / 1e3
busy = sum(r[1] - r[0] for r in half) / 1e3
ntok = sum(1 for r in half if 'embed_ln_fwd' in r[2] or 'dec_embed' in r[2])
print('second half of the run: %d kernels = %d tokens, span %.1f us (%.1f us/token), GPU busy %.1f us (%.1f %%), %.2f us mean gap'
      % (len(half), ntok, span, span / max(ntok, 1), busy, 100 * busy / span, (span - busy) / len(half)))

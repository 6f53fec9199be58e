"""Per-kernel totals from a rocprofv3 rocpd database (the default output format of rocprofv3 in ROCm 7.2).

  python tools/rocpd_stats.py <results.db> [rows]

GEMM rows are split by PROBLEM when the traced run had PB_GEMM_LDS_TAG=1 (tools/profile_round.sh sets it): the persistent kernels
launch 256 workgroups whatever the shape, so the library then asks for 16 x tag bytes of dynamic LDS it never touches and the
trace's group_segment_size carries tag = 64 nclass + 8 kclass + epilogue (pb_gemm2.hip: gemm_lds_tag). The fc1 GEMM of bench.py's
roofline record is the row `NT N=3072 K=768 +gelu`."""
import re, sqlite3, sys

CLS = {0: 'other', 1: '768', 2: '1280', 3: '1536', 4: '2304', 5: '3072', 6: '18432', 7: 'rows'}
EPI = {0: '', 1: ' +bias+gelu (2 outputs)', 2: " *gelu'", 3: ' +=', 4: ' +rowdot', 5: ' +bias'}
BASE = {'gemm3_kernel': 131072 + 2048 + 16384}                                     # dynamic LDS of an untagged launch


def gemm_label(name, gss):
    """kernel symbol + group_segment_size -> 'NT N=3072 K=768 +gelu' (or None when the launch carries no tag)."""
    m3 = re.search(r'gemm3_kernelILb(\d)ELb(\d)E', name)
    m2 = re.search(r'gemm2_kernelILb(\d)ELb(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E', name)
    if m3:
        base, tile, (ak, bk) = BASE['gemm3_kernel'], '256x256 ping-pong', m3.groups()
    elif m2:
        ak, bk = m2.group(1), m2.group(2)
        big = m2.groups()[2:] == ('2', '4', '8', '4')
        base, tile = (131072, '256x256 one-barrier') if big else (65536, '128x128')
    else:
        return None
    lay = ('N' if ak == '1' else 'T') + ('T' if bk == '1' else 'N')
    tag = (gss - base) // 16
    if gss <= base or tag <= 0 or tag >= 512:
        return None
    if lay == 'TN':                                                      # weight gradient: K = rows of the batch, the K slot carries M
        return 'TN M=%s N=%s K=rows%s  [%s]' % (CLS[(tag // 8) % 8], CLS[tag // 64], EPI.get(tag % 8, ''), tile)
    return '%s N=%s K=%s%s  [%s]' % (lay, CLS[tag // 64], CLS[(tag // 8) % 8], EPI.get(tag % 8, ''), tile)


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    rows = {}
    for name, gss, dur in db.execute(f"select s.kernel_name, d.group_segment_size, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"):
        lab = gemm_label(name, gss)
        key = ('gemm: ' + lab) if lab else name
        a = rows.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += dur / 1e3
    print('%-100s %7s %12s %9s' % ('kernel', 'calls', 'total_us', 'avg_us'))
    for key, (n, tot) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:limit]:
        print('%-100s %7d %12.1f %9.2f' % (key[:100], n, tot, tot / n))


if __name__ == '__main__':
    main()

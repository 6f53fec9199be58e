"""Where are the fc1 GEMM's re-fetched operand bytes served from? rocprofv3 has no Infinity-Cache hit counter on gfx950, but
the L2's memory-side read queue gives the mean latency of its requests by Little's law: TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum (cycles a
request spends outstanding). Three kernels under the same counters:
  1. a streaming read of 2 GiB (every byte from HBM)                                   -> latency of an HBM read under load
  2. the same kernel over 96 MiB, ten times back to back (resident in the 256 MiB Infinity Cache after the first pass) -> latency of a hit
  3. the fc1 GEMM as the step launches it (bias + GELU + derivative out), M rows
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum --output-format csv -d <dir> -- python3 tools/pmc_mall_probe.py --M=26624
    python tools/pmc_mall_probe.py --report <dir> [out.json]"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(d, out=None):
    rows = {}
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows.setdefault(int(r['Dispatch_Id']), {'k': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
    sq = [v for _, v in sorted(rows.items()) if 'sqnorm_kernel' in v['k'] and 'TCC_EA0_RDREQ_sum' in v]
    gm = [v for _, v in sorted(rows.items()) if 'gemm3_kernel' in v['k'] and 'TCC_EA0_RDREQ_sum' in v]
    lat = lambda v: v['TCC_EA0_RDREQ_LEVEL_sum'] / max(1.0, v['TCC_EA0_RDREQ_sum'])
    mean = lambda xs: sum(xs) / max(1, len(xs))
    big = max(sq, key=lambda v: v['TCC_EA0_RDREQ_sum'])
    small = [v for v in sq if v is not big and v['TCC_EA0_RDREQ_sum'] < 0.2 * big['TCC_EA0_RDREQ_sum']]
    rec = {'hbm_stream_2GiB': {'rdreq': big['TCC_EA0_RDREQ_sum'], 'mean_latency_cycles': lat(big)},
           'infinity_cache_resident_96MiB': {'rdreq_per_pass': mean([v['TCC_EA0_RDREQ_sum'] for v in small[2:]]), 'mean_latency_cycles': mean([lat(v) for v in small[2:]]),
                                             'first_pass_latency_cycles': lat(small[0]) if small else None},
           'fc1_gemm': {'rdreq_per_launch': mean([v['TCC_EA0_RDREQ_sum'] for v in gm[1:]]), 'mean_latency_cycles': mean([lat(v) for v in gm[1:]]), 'launches': len(gm) - 1},
           'method': 'rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum over tools/pmc_mall_probe.py; latency = LEVEL / RDREQ (Little), L2-clock cycles'}
    print(json.dumps(rec, indent=1))
    if out:
        json.dump(rec, open(out, 'w'), indent=1)


if __name__ == '__main__':
    if '--report' in sys.argv:
        i = sys.argv.index('--report')
        report(sys.argv[i + 1], sys.argv[i + 2] if len(sys.argv) > i + 2 else None)
        sys.exit(0)
    import torch
    from pianobart_amd import ops
    T, N, K = 26624, 3072, 768
    for a in sys.argv[1:]:
        if a.startswith('--M='):
            T = int(a[4:])
    big = torch.randn(512 << 20, device='cuda')                       # 2 GiB of f32
    small = torch.randn(24 << 20, device='cuda')                      # 96 MiB
    part = torch.empty(1 << 16, device='cuda'); sq = torch.zeros(1, device='cuda')
    ops.grad_sqnorm(big, part, sq)
    for _ in range(10):
        ops.grad_sqnorm(small, part, sq)
    x = torch.randn(T, K, device='cuda').to(torch.bfloat16); w = torch.randn(N, K, device='cuda').to(torch.bfloat16)
    out = torch.empty(T, N, device='cuda', dtype=torch.bfloat16); aux = torch.empty_like(out)
    bias = torch.randn(N, device='cuda')
    for _ in range(6):
        ops.gemm(x, w, out, M=T, N=N, K=K, dtype=ops.PB_BF16, bias=bias, gelu_aux_out=aux)
    torch.cuda.synchronize()

"""Whole-step HBM byte budget from two rocprofv3 counter passes over bench.py:

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <fdir> -- python3 bench.py --no-cpu-baseline --no-probe --steps 2 --warmup 2
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <wdir> -- python3 bench.py --no-cpu-baseline --no-probe --steps 2 --warmup 2
  python tools/pmc_step_budget.py <fdir> <wdir> [step_ms]

Per kernel family (GEMM / attention / row kernels / optimizer / other): bytes fetched from beyond L2 (FETCH_SIZE x 2: gfx950 tallies its
128-byte requests at 64 bytes, MI355X_MICROARCH.md HBM) + bytes written (WRITE_SIZE), per STEP, and the rate they make at the step time
given. Counter passes serialise the kernels (one at a time), so the bytes are per kernel, not a timeline; Infinity-Cache hits are counted
as traffic by these counters (same guide), so the total is an upper bound of what reaches HBM. The d = 768 GEMMs sit at 338 - 384
FLOP/B against a machine balance of ~312 - 397 FLOP/B (2.5 PFLOP/s over 6.3 - 8 TB/s): "bound: mfma" in the bench record is a
claim about the K loop; this table is the step's TB/s next to it."""
import csv
import glob
import sys

FAMILY = (('gemm', ('gemm3_kernel', 'gemm2_kernel', 'gemm_kernel', 'reduce_slabs', 'tail_finish')),
          ('attention', ('fa64_', 'fa1_', 'flash', 'softmax')),
          ('rows (LayerNorm, embedding, maps)', ('add_ln', 'embed_ln', 'rowmap', 'gather_rows', 'scatter_rows', 'pos_grad', 'onehot', 'colsum', 'finalize', 'batch_sum', 'transpose_batch')),
          ('optimizer', ('adamw', 'sqnorm', 'clip_coef', 'cast_kernel', 'fill_kernel')),
          ('loss', ('ce_', 'mask_count', 'loss_coef')))


def family(name):
    for fam, keys in FAMILY:
        if any(k in name for k in keys):
            return fam
    return 'other (torch fills / copies)'


def collect(d, counter):
    """-> {family: KB}, {kernel name: KB}, steps. Only the dispatches BETWEEN the first and the last gradient-norm kernel count (one per step:
    whole steps, none of the model set-up, stream probe or read-back around them); steps = gradient-norm kernels - 1."""
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == counter:
                rows.append((int(r['Dispatch_Id']), r.get('Kernel_Name', ''), float(r['Counter_Value'])))
    rows.sort()
    marks = sorted({i for i, n, _ in rows if 'sqnorm_kernel' in n})
    if len(marks) < 2:
        raise SystemExit('fewer than two gradient-norm kernels in %s: no whole step to count' % d)
    fam, ker = {}, {}
    for i, n, v in rows:
        if marks[0] < i <= marks[-1]:
            fam[family(n)] = fam.get(family(n), 0.0) + v
            ker[n] = ker.get(n, 0.0) + v
    return fam, ker, len(marks) - 1


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    step_ms = float(sys.argv[3]) if len(sys.argv) > 3 else None
    ff, fk, nf = collect(fdir, 'FETCH_SIZE')
    wf, wk, nw = collect(wdir, 'WRITE_SIZE')
    print('# whole steps counted (between the first and the last gradient-norm kernel): %d (FETCH_SIZE pass), %d (WRITE_SIZE pass)' % (nf, nw))
    print('%-40s %12s %12s %12s' % ('family', 'fetched MB', 'written MB', 'total MB') + ('   at %.1f ms/step' % step_ms if step_ms else ''))
    tot_f = tot_w = 0.0
    for fam in [f for f, _ in FAMILY] + ['other (torch fills / copies)']:
        fb = ff.get(fam, 0.0) * 1024 * 2 / nf / 1e6
        wb = wf.get(fam, 0.0) * 1024 / nw / 1e6
        tot_f += fb; tot_w += wb
        print('%-40s %12.0f %12.0f %12.0f' % (fam, fb, wb, fb + wb))
    line = '%-40s %12.0f %12.0f %12.0f' % ('step', tot_f, tot_w, tot_f + tot_w)
    if step_ms:
        line += '   %.2f TB/s' % ((tot_f + tot_w) * 1e6 / (step_ms * 1e-3) / 1e12)
    print(line)
    print('# largest kernels (MB per step: fetched x2 + written)')
    names = sorted(set(fk) | set(wk), key=lambda n: -(fk.get(n, 0.0) * 2 / nf + wk.get(n, 0.0) / nw))[:10]
    for n in names:
        print('%10.0f  %s' % ((fk.get(n, 0.0) * 2 / nf + wk.get(n, 0.0) / nw) * 1024 / 1e6, n[:110]))


if __name__ == '__main__':
    main()

#!/bin/bash
# packed vs dense fused step x GEMM tail split on/off, same box
mkdir -p gpurun_out
for cfg in "0 0" "0 1" "1 0" "1 1"; do
  set -- $cfg
  PB_PACK_ROWS=$1 PB_GEMM_TAIL=$2 timeout 600 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/bench_p$1_t$2.json 2> gpurun_out/bench_p$1_t$2.err
done
python - <<'PY'
import json
for n in ('p0_t0', 'p0_t1', 'p1_t0', 'p1_t1'):
    try:
        j = json.loads(open('gpurun_out/bench_%s.json' % n).read().strip().splitlines()[-1])
        print(n, 'ms/step', round(j['ms_per_step'], 2), 'median', round(j['ms_per_step_median_hip_events'], 2), 'tok/s', round(j['value']), 'rows', j['rows']['encoder_side'], j['rows']['decoder_side'],
              'mfma', round(j['step_mfma_frac'], 4), 'loss', round(j['train_loss'], 5), 'fc1', round(j['roofline']['frac'], 4))
        for k, v in j['roofline']['families_in_step'].items():
            print('   ', k, {a: round(b, 3) for a, b in v.items()})
        for r in j['roofline']['top_ops_in_step']:
            if 'gemm' in r['op']:
                print('      %-70s calls %5.1f avg %8.1f us  %s' % (r['op'], r['calls_per_step'], r['avg_us'], round(r['tflops']) if r['tflops'] else None))
    except Exception as e:
        print(n, 'failed', e); print(open('gpurun_out/bench_%s.err' % n).read()[-3000:])
PY

"""Per-item anatomy of the persistent ping-pong GEMM (gemm3_kernel): cycles a wave spends, per (split, tile) work item, in
  [1] the wait for K-tile 0 (+ stagger barrier)   [2] the K loop   [3] the next item's addressing + DMA issue   [4] the epilogue
(arithmetic + store / load issue), plus the one-off [0] first prologue and [5] final drain -- for the step's shapes with their real
epilogues. Diagnostic build of the library (s_memtime stamps):
  PB_LIB_OUT=$PWD/ab/stamps.so PB_EXTRA_HIPCC_FLAGS=-DPB_G3_STAMPS python pianobart_amd/build.py
  PB_LIB_PATH=$PWD/ab/stamps.so python tools/gemm_stamps.py [rows]
Means over every wave of every workgroup; `us` columns scale the cycles by the kernel's own lifetime (max wave lifetime in cycles
against the HIP-event time of the launch), so they add up to the launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
st = torch.zeros(1024 * 8 * 16, dtype=torch.int32, device='cuda')            # one 16-word slot per wave of up to 1024 workgroups
os.environ['PB_G3_STAMP_PTR'] = str(st.data_ptr())
from pianobart_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 26624
dev = 'cuda'
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g).to(bf)


SOAK = float(os.environ.get('SOAK', '0'))          # seconds of back-to-back launches of the same kernel before the measured ones (a cold chip boosts)


def run(name, fn, flops, reps=5):
    fn(); torch.cuda.synchronize()
    if SOAK > 0:
        import time
        t0 = time.time()
        while time.time() - t0 < SOAK:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
    st.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    w = st.cpu().numpy().astype('uint32').astype('float64').reshape(-1, 16)
    w = w[w[:, 7] > 0]                                           # the slots of the LAST launch's waves (every launch overwrites them)
    if len(w) == 0:
        print('%-34s %7.1f us  (not the ping-pong kernel: no stamps)' % (name, us)); return
    waves = len(w)
    per_wave_items = w[:, 6].sum() / waves
    life, life_max = w[:, 8].mean(), w[:, 8].max()
    cyc_per_us = life_max / us                                   # the longest wave spans the launch
    real = w[:, 9]
    mhz = (w[:, 8] / real.clip(min=1)).mean() * 100.0 if real.max() > 0 else 0.0      # s_memtime ticks per s_memrealtime tick (100 MHz): the in-kernel clock
    sec = [w[:, i].mean() for i in range(6)]                     # cycles per wave over the launch
    f = lambda c: c / cyc_per_us
    print('%-34s %7.1f us %6.0f TF | items/wg %.2f | per item: wait %5.2f  kloop %6.2f  next-issue %5.2f  epilogue %6.2f us | once: first prologue %4.1f, drain %4.1f us | '
          'wave life %6.1f of %6.1f us (%.0f ticks/us; in-kernel clock %.0f MHz)' % (name, us, flops / us / 1e6, per_wave_items, f(sec[1]) / per_wave_items, f(sec[2]) / per_wave_items,
                                                     f(sec[3]) / per_wave_items, f(sec[4]) / per_wave_items, f(sec[0]), f(sec[5]), f(life), us, cyc_per_us, mhz))


d, ffn = 768, 3072
x = rn(T, d); w1 = rn(ffn, d); b1 = torch.randn(ffn, device=dev, generator=g)
u = torch.empty(T, ffn, device=dev, dtype=bf); gp = torch.empty_like(u)
run('NT fc1 +bias+gelu (2 outputs)', lambda: ops.gemm(x, w1, u, M=T, N=ffn, K=d, dtype=ops.PB_BF16, bias=b1, gelu_aux_out=gp), 2.0 * T * ffn * d)
run('NT fc1 +bias (plain)', lambda: ops.gemm(x, w1, u, M=T, N=ffn, K=d, dtype=ops.PB_BF16, bias=b1), 2.0 * T * ffn * d)
dy = rn(T, d); w2t = rn(ffn, d)
cs = torch.zeros(ffn, device=dev)
csws = torch.empty(int(2 * ((T + 255) // 256) * ffn), device=dev)
run("NT dfc2 *gelu' +colsum", lambda: ops.gemm(dy, w2t, u, M=T, N=ffn, K=d, dtype=ops.PB_BF16, gelu_grad_aux_in=gp, colsum_out=cs, colsum_ws=csws), 2.0 * T * ffn * d)
run("NT dfc2 *gelu' (no colsum)", lambda: ops.gemm(dy, w2t, u, M=T, N=ffn, K=d, dtype=ops.PB_BF16, gelu_grad_aux_in=gp), 2.0 * T * ffn * d)
wqkv = rn(3 * d, d); bq = torch.randn(3 * d, device=dev, generator=g); qkv = torch.empty(T, 3 * d, device=dev, dtype=bf)
run('NT qkv +bias', lambda: ops.gemm(x, wqkv, qkv, M=T, N=3 * d, K=d, dtype=ops.PB_BF16, bias=bq), 2.0 * T * 3 * d * d)
wo = rn(d, d); bo = torch.randn(d, device=dev, generator=g); o = torch.empty(T, d, device=dev, dtype=bf)
run('NT out-proj +bias', lambda: ops.gemm(x, wo, o, M=T, N=d, K=d, dtype=ops.PB_BF16, bias=bo), 2.0 * T * d * d)
run('NT out-proj dgrad (no epilogue)', lambda: ops.gemm(x, wo, o, M=T, N=d, K=d, dtype=ops.PB_BF16), 2.0 * T * d * d)
h = rn(T, ffn); w2 = rn(d, ffn)
run('NT fc2 +bias', lambda: ops.gemm(h, w2, o, M=T, N=d, K=ffn, dtype=ops.PB_BF16, bias=bo), 2.0 * T * d * ffn)
run('NT dfc1 += ', lambda: ops.gemm(h, w2, o, M=T, N=d, K=ffn, dtype=ops.PB_BF16, accum=True), 2.0 * T * d * ffn)
dq = rn(T, 3 * d); wqkvt = rn(d, 3 * d)
run('NT dqkv +=', lambda: ops.gemm(dq, wqkvt, o, M=T, N=d, K=3 * d, dtype=ops.PB_BF16, accum=True), 2.0 * T * d * 3 * d)
sl = torch.empty(8 * ffn * d, device=dev); gw = torch.zeros(ffn, d, device=dev)
run('TN w1 splitk5', lambda: ops.gemm(u, x, gw, M=ffn, N=d, K=T, dtype=ops.PB_BF16, a_kc=False, b_kc=False, c_f32=True, splitk=5, slabs=sl, tile256=True), 2.0 * T * ffn * d)

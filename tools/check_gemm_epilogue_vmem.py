"""The persistent GEMM's cross-item wait counts the VMEM instructions the previous item's epilogue left in flight (`pend` in gemm3_kernel,
pb_gemm2.hip: the next item's first K-tile waits with s_waitcnt vmcnt(pend + 4)). Those numbers are hand counts of what each epilogue variant
issues per wave; one store more or less in an epilogue would make the wait too weak -- a silent LDS race. This check ties them to
the code: every epilogue variant is compiled ALONE into a probe kernel (the .hip is included textually, so the probes call the very functions the
kernel inlines), its global loads / stores are counted in the gfx950 ISA, and the counts must equal the `pend` table of the kernel source.

  python tools/check_gemm_epilogue_vmem.py        (build host; hipcc cross-compiles without a GPU)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'pianobart_amd', 'csrc', 'pb_gemm2.hip')

# probe name -> (call, what `pend` must be for it)
PROBES = {
    'plain':     ('epilogue_plain(p, acc, 0, 0, 0, lane, (const float*)sm, lds_base, 0)', 'plain'),
    'gelu':      ('epilogue_gelu(p, acc, 0, 0, 0, lane, (const float*)sm, lds_base, 0)', 'wide'),
    'f32_plain': ('epilogue_f32_plain(p, acc, 0, 0, 0, lane, lds_base, 0)', 'wide'),
    'pf1':       ('epilogue_pf<1, false>(p, acc, 0, 0, 0, lane, (const float*)sm, nullptr, lds_base, 0)', 'epf'),
    'pf1cs':     ('epilogue_pf<1, true>(p, acc, 0, 0, 0, lane, (const float*)sm, cs, lds_base, 0)', 'epf_cs'),
    'pf2':       ('epilogue_pf<2, false>(p, acc, 0, 0, 0, lane, (const float*)sm, nullptr, lds_base, 0)', 'epf'),
    'pf3':       ('epilogue_pf<3, false>(p, acc, 0, 0, 0, lane, (const float*)sm, nullptr, lds_base, 0)', 'epf3'),
}


def pend_table(text):
    """The literals of the kernel's `pend` assignments: {'plain': 16, 'wide': 32, 'epf': 32, 'epf_cs': 36, 'epf3': 40}."""
    m1 = re.search(r'pend = \(interior && plain\) \? \(\(\(p\.flags & PB_GEMM_C_F32\) \|\| \(p\.flags & PB_GEMM_GELU\)\) \? (\d+) : (\d+)\) : 0;', text)
    m2 = re.search(r'if \(epf\) pend = p\.cs_ws \? (\d+) : (\d+);', text)
    m3 = re.search(r'if \(epf3\) pend = (\d+);', text)
    if not (m1 and m2 and m3):
        raise SystemExit('check_gemm_epilogue_vmem: the `pend` assignments of gemm3_kernel no longer have the form this check reads')
    return {'wide': int(m1.group(1)), 'plain': int(m1.group(2)), 'epf_cs': int(m2.group(1)), 'epf': int(m2.group(2)), 'epf3': int(m3.group(1))}


def probe_source():
    lines = ['#include "%s"' % SRC, 'namespace {',
             '__device__ __forceinline__ void mk_acc(f32x4 (&acc)[8][4]) {', '    const float base = (float)threadIdx.x;',
             '#pragma unroll', '    for (int i = 0; i < 8; ++i)', '#pragma unroll',
             '        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{base + i, base * j, base - i, base + j};', '}', '}']
    for name, (call, _) in PROBES.items():
        lines.append('extern "C" __global__ void probe_%s(Gemm2Args p, unsigned lds_base, float* cs) { extern __shared__ char sm[]; const int lane = threadIdx.x & 63; '
                     'f32x4 acc[8][4]; mk_acc(acc); %s; }' % (name, call))
    return '\n'.join(lines) + '\n'


def counts():
    sys.path.insert(0, ROOT)
    from pianobart_amd import build
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, 'probe.hip'), os.path.join(d, 'probe.s')
        open(src, 'w').write(probe_source())
        cmd = [build._hipcc()] + [f for f in build.FLAGS if f != '-fPIC'] + ['-S', '--cuda-device-only', src, '-o', out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit('check_gemm_epilogue_vmem: the probe does not compile:\n' + r.stderr[-2000:])
        text = open(out).read()
    res = {}
    for m in re.finditer(r'^probe_(\w+):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M):
        body = m.group(2)
        res[m.group(1)] = (len(re.findall(r'^\s*(?:global_load|buffer_load)', body, re.M)), len(re.findall(r'^\s*(?:global_store|buffer_store)', body, re.M)))
    return res


def check():
    table = pend_table(open(SRC).read())
    got = counts()
    errs = []
    for name, (_, kind) in PROBES.items():
        if name not in got:
            errs.append('probe_%s missing from the disassembly' % name)
            continue
        ld, st = got[name]
        if ld + st != table[kind]:
            errs.append('%s issues %d loads + %d stores = %d VMEM instructions per wave, the kernel waits for %d (`pend`, %s)' % (name, ld, st, ld + st, table[kind], kind))
    return errs, got, table


def main():
    errs, got, table = check()
    for name in PROBES:
        print('%-10s loads %2d stores %2d   pend %d' % (name, got.get(name, (0, 0))[0], got.get(name, (0, 0))[1], table[PROBES[name][1]]))
    for e in errs:
        print('FAIL:', e)
    sys.exit(1 if errs else 0)


if __name__ == '__main__':
    main()

"""Disassembly checks of the persistent GEMM (pb_gemm2.hip), run on the build host (hipcc cross-compiles gfx950 without a GPU):

  python tools/check_gemm_isa.py

1. neither gemm3_kernel instantiation spills a vector register (a spill is a scratch store = one more entry in the in-order vmcnt
   queue, which breaks the counted waits of the K loop, besides the traffic);
2. in the TN instantiation (weight gradients) the 12 transposed fragments are asm results pinned to v200-v247 whose data is still in
   flight when the asm statement ends: any copy, spill or other write of those registers outside the ds_read_b64_tr_b16 themselves
   would read or clobber registers the LDS has not filled yet. The only instructions allowed to name them are the reads and the MFMAs.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINNED = range(200, 248)


def disassemble():
    src = os.path.join(ROOT, 'pianobart_amd', 'csrc', 'pb_gemm2.hip')
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'g.s')
        sys.path.insert(0, ROOT)
        from pianobart_amd import build                          # the product's own compiler flags
        cmd = [build._hipcc()] + [f for f in build.FLAGS if f != '-fPIC'] + ['-S', '--cuda-device-only', src, '-o', out]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def kernels(text):
    """-> {symbol: body} for the gemm3 instantiations, {symbol: (vgprs, spills)} from the metadata."""
    body, meta = {}, {}
    for m in re.finditer(r'^(_ZN[^\s:]*gemm3_kernel[^\s:]*):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M):
        body[m.group(1)] = m.group(2)
    for m in re.finditer(r'\.name:\s+(\S*gemm3_kernel\S*)\n(.*?)\.wavefront_size', text, re.S):
        f = dict(re.findall(r'\.(vgpr_count|vgpr_spill_count):\s+(\d+)', m.group(2)))
        meta[m.group(1)] = (int(f['vgpr_count']), int(f['vgpr_spill_count']))
    return body, meta


def regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return range(int(m.group(1)), int(m.group(2)) + 1)
    m = re.fullmatch(r'v(\d+)', tok)
    return range(int(m.group(1)), int(m.group(1)) + 1) if m else range(0)


def check(text):
    body, meta = kernels(text)
    errs = []
    if len(body) < 2:
        errs.append('expected two gemm3_kernel instantiations, found %d' % len(body))
    for k, (vg, sp) in meta.items():
        if sp:
            errs.append('%s spills %d vector registers' % (k, sp))
    tn = [k for k in body if 'ILb0ELb0E' in k]
    for k in tn:
        lines = body[k].splitlines()
        hot = [i for i, ln in enumerate(lines) if 'ds_read_b64_tr_b16' in ln or 'v_mfma' in ln]
        for ln in lines[hot[0]:hot[-1] + 1]:                  # the K loop with its peeled first pass (the epilogue behind the loop's closing wait may use them)
            ins = ln.split(';')[0].strip()
            if not ins or ins.startswith('.') or ins.endswith(':'):
                continue
            op, _, rest = ins.partition(' ')
            toks = re.findall(r'v\[\d+:\d+\]|v\d+', rest)
            hit = [t for t in toks if any(r in PINNED for r in regs(t))]
            if hit and not (op.startswith('ds_read_b64_tr_b16') or op.startswith('v_mfma')):
                errs.append('%s: `%s` touches a pinned fragment register' % (k[:40], ins))
    return errs, meta


def main():
    errs, meta = check(disassemble())
    for k, (vg, sp) in sorted(meta.items()):
        print('%-70s vgprs %3d spills %d' % (k, vg, sp))
    for e in errs:
        print('FAIL:', e)
    sys.exit(1 if errs else 0)


if __name__ == '__main__':
    main()

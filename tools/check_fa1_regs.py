"""Audit of csrc/pb_flash1.hip's one-pass backward kernel (its accumulator registers are addressed literally from inline asm, see the
file header): compiles the file to ISA and requires no scratch, no spill, and no v_accvgpr_* / a[...] operand outside an
;;#ASMSTART ... ;;#ASMEND block of fa1_bwd_kernel.  python tools/check_fa1_regs.py [file.s]"""
import os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def isa():
    if len(sys.argv) > 1:
        return open(sys.argv[1]).read()
    sys.path.insert(0, HERE)
    from pianobart_amd.build import FLAGS, _hipcc
    out = os.path.join(tempfile.mkdtemp(), 'pb_flash1.s')
    subprocess.run([_hipcc()] + FLAGS + ['-S', '--cuda-device-only', os.path.join(HERE, 'pianobart_amd', 'csrc', 'pb_flash1.hip'), '-o', out], check=True,
                   stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    s = isa()
    start = s.index('fa1_bwd_kernel', s.index('.globl'))
    m = re.search(r'^(_ZN\S*fa1_bwd_kernel\S*):', s, flags=re.M)
    body = s[m.start():s.index('.Lfunc_end', m.start())]
    inside, bad, n_mfma, n_lines = False, [], 0, 0
    for ln, l in enumerate(body.split('\n')):
        if ';;#ASMSTART' in l:
            inside = True
        elif ';;#ASMEND' in l:
            inside = False
        elif not inside and re.match(r'\s+[a-z]', l):
            n_lines += 1
            if 'accvgpr' in l or re.search(r'\ba\[?\d+', l.split(';')[0]) or 'scratch_' in l:
                bad.append((ln, l.strip()))
        if 'v_mfma' in l:
            n_mfma += 1
    # a vector instruction's result read by an MFMA within two wait states (hipcc pads nothing for an asm statement's operands)
    def regs(tok):
        m = re.match(r'v\[(\d+):(\d+)\]$', tok) or re.match(r'v(\d+)$', tok)
        return set() if not m else set(range(int(m.group(1)), int(m.group(m.lastindex)) + 1))
    hist, close = [], []
    for ln, l in enumerate(body.split('\n')):
        code = l.split(';')[0].strip()
        if not code or code.endswith(':') or code.startswith('.'):
            continue
        parts = code.replace(',', ' ').split()
        op, args = parts[0], parts[1:]
        if op.startswith('v_mfma'):
            need = set().union(*[regs(a) for a in args[1:]])
            for age, (wop, wset, wln) in enumerate(reversed(hist[-2:])):
                if wset & need:
                    close.append((ln, code, wop, wln))
            hist.append(('mfma', set(), ln))
        elif op == 's_nop':
            hist += [('nop', set(), ln)] * (int(args[0]) + 1)
        elif op.startswith('v_') and not op.startswith('v_cmp'):
            hist.append((op, regs(args[0]) if args else set(), ln))
        else:
            hist.append((op, set(), ln))
    for ln, code, wop, wln in close[:10]:
        print('  MFMA reads a register that %s (line %d) wrote less than two wait states before, line %d: %s' % (wop, wln, ln, code))
    bad += [(ln, 'valu->mfma ' + code) for ln, code, _, _ in close]
    meta = s[s.index('fa1_bwd_kernel', s.index('amdhsa.kernels')):]
    get = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, meta).group(1))
    print('fa1_bwd_kernel: vgpr %d agpr %d spills %d scratch %d B; %d MFMA statements; %d compiler instructions outside asm' % (
        get('vgpr_count'), get('agpr_count'), get('vgpr_spill_count'), get('private_segment_fixed_size'), n_mfma, n_lines))
    for ln, l in bad[:20]:
        print('  compiler touches an accumulator register or scratch, line %d: %s' % (ln, l))
    agpr = [x for x in bad if 'scratch_' not in x[1]]
    ok = not agpr                                                        # scratch traffic is listed (it must stay outside the step loop), AGPR use by the compiler is fatal
    print('OK' if ok else 'FAILED')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())

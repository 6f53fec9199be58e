"""Build libpianobart_hip.so (gfx950) in-tree with hipcc. No torch extension machinery: the
library is a plain C-ABI shared object (include/pianobart_hip.h) loaded through ctypes."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.environ.get('PB_CSRC') or os.path.join(HERE, 'csrc')      # PB_CSRC: another source tree (an older commit's csrc/ for a same-box A/B)
# PB_LIB_OUT: build a second (diagnostic / A-B) library beside the product one: its objects go to <PB_LIB_OUT>.obj
LIB = os.environ.get('PB_LIB_OUT') or os.path.join(HERE, 'libpianobart_hip.so')
OBJ = (LIB + '.obj') if os.environ.get('PB_LIB_OUT') else os.path.join(HERE, 'build')
ARCH = 'gfx950'
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has a unified file); without it hipcc parks them in AGPRs
# and every VALU touch of an accumulator (softmax rescale, epilogues) costs a v_accvgpr_read/write pair.
FLAGS = ['-O3', '-fPIC', '-std=c++17', '--offload-arch=' + ARCH, '-ffp-contract=fast', '-Wno-unused-result',
         '-mllvm', '-amdgpu-mfma-vgpr-form=1'] + os.environ.get('PB_EXTRA_HIPCC_FLAGS', '').split()      # e.g. -DPB_FA1_STAMPS (diagnostic build)


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'pianobart_hip.h'))
    return max(os.path.getmtime(h) for h in hs)


def build(force=False, verbose=True, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    hdr = _newest_header()
    todo, objs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
        if verbose:
            print('[pianobart_amd.build] compiled', os.path.basename(src), flush=True)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(6, len(todo))) as ex:
            list(ex.map(cc, todo))
    if todo or not os.path.exists(LIB):
        r = subprocess.run([_hipcc(), '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stderr[-4000:])
        if verbose:
            print('[pianobart_amd.build] linked', LIB, flush=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)

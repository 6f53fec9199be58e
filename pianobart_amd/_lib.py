"""ctypes binding of libpianobart_hip.so, generated from include/pianobart_hip.h.

The product path has no CPU fallback: if the library is missing or fails to load, every op raises.
"""
import ctypes
import os
import re

import torch  # noqa: F401  -- must be imported BEFORE dlopen(libpianobart_hip.so): both must share torch's HIP runtime

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'pianobart_hip.h')
# PB_LIB_PATH: developer aid for same-box A/B runs of two builds of the library (box-to-box spread on the pool is +-3 %)
LIB_PATH = os.environ.get('PB_LIB_PATH') or os.path.join(HERE, 'libpianobart_hip.so')

PB_F32, PB_BF16, PB_F32X3 = 0, 1, 2
GEMM_ACCUM, GEMM_C_F32, GEMM_GELU, GEMM_MUL_GELU_GRAD = 1, 2, 4, 8
GEMM_ROWDOT = 131072


class GemmDesc(ctypes.Structure):
    _fields_ = [('A', ctypes.c_void_p), ('B', ctypes.c_void_p), ('C', ctypes.c_void_p),
                ('bias', ctypes.c_void_p), ('aux_in', ctypes.c_void_p), ('aux_out', ctypes.c_void_p),
                ('dtype', ctypes.c_int32), ('a_kcontig', ctypes.c_int32), ('b_kcontig', ctypes.c_int32), ('flags', ctypes.c_int32),
                ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32), ('nb1', ctypes.c_int32),
                ('nb2', ctypes.c_int32), ('_pad', ctypes.c_int32),
                ('lda', ctypes.c_int64), ('ldb', ctypes.c_int64), ('ldc', ctypes.c_int64), ('ldaux', ctypes.c_int64),
                ('sA1', ctypes.c_int64), ('sA2', ctypes.c_int64), ('sB1', ctypes.c_int64), ('sB2', ctypes.c_int64),
                ('sC1', ctypes.c_int64), ('sC2', ctypes.c_int64),
                ('alpha', ctypes.c_float), ('_pad2', ctypes.c_float),
                ('splitk', ctypes.c_int32), ('_pad3', ctypes.c_int32), ('slabs', ctypes.c_void_p),
                ('colsum_out', ctypes.c_void_p), ('colsum_ws', ctypes.c_void_p),
                ('rowdot_out', ctypes.c_void_p), ('ld_rowdot', ctypes.c_int64)]


class DecodeLayer(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ('wqkv', 'bqkv', 'wo', 'bo', 'ln1_w', 'ln1_b', 'wq_c', 'bq_c', 'wo_c', 'bo_c', 'lnc_w', 'lnc_b',
                                               'w1', 'b1', 'w2', 'b2', 'ln2_w', 'ln2_b', 'kv_self', 'kv_cross')]


class DecodePlan(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('dtype', 'd', 'H', 'ffn', 'S', 'S_enc', 'n_layers', 'vocab')] + \
               [('tab_off', ctypes.c_int32 * 9), ('_pad', ctypes.c_int32)] + \
               [(n, ctypes.c_void_p) for n in ('tok16', 'ptab', 'lin_b', 'pos', 'lne_w', 'lne_b', 'enc_mask', 'x', 'y1', 'yc', 'y2', 'q', 'ctx', 'a', 'g',
                                               'stat', 'attn_part', 'logits', 'head_w', 'head_b')] + \
               [('layers', DecodeLayer * 48)]


_SCALARS = {'int32_t': ctypes.c_int32, 'int64_t': ctypes.c_int64, 'uint64_t': ctypes.c_uint64,
            'uint32_t': ctypes.c_uint32, 'float': ctypes.c_float, 'double': ctypes.c_double, 'int': ctypes.c_int}


def parse_header(path=HEADER):
    """Returns {name: (restype, [argtypes])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    src = re.sub(r'typedef\s+struct.*?}\s*\w+\s*;', ' ', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'(const\s+char\s*\*|int64_t|int)\s+(pb_\w+)\s*\(([^;{]*?)\)\s*;', src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if 'char' in ret else _SCALARS[ret.strip()]
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(ctypes.POINTER(GemmDesc) if 'pb_gemm_desc' in a else ctypes.c_void_p)
                else:
                    argtypes.append(_SCALARS[a.replace('const', '').split()[0]])
        decls[name] = (restype, argtypes)
    return decls


class PBError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        self._dll = None
        self.decls = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise PBError('libpianobart_hip.so is not built (%s); run `python -c "import __graft_entry__ as g; g.build()"` '
                              'or `python pianobart_amd/build.py`. There is no CPU fallback.' % LIB_PATH)
            dll = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in self.decls.items():
                fn = getattr(dll, name)          # AttributeError => header/library mismatch: fail loudly
                fn.restype = restype
                fn.argtypes = argtypes
            ver = dll.pb_abi_version()
            if ver != 8:
                raise PBError('ABI version mismatch: library %d, binding 8' % ver)
            self._dll = dll
        return self._dll

    def call(self, name, *args):
        dll = self.load()
        rc = getattr(dll, name)(*args)
        if rc != 0:
            raise PBError('%s failed (%d): %s' % (name, rc, dll.pb_last_error().decode()))

    def query(self, name, *args):
        return getattr(self.load(), name)(*args)


LIB = _Lib()

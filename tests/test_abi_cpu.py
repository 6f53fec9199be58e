"""CPU: the C-ABI library builds, loads, and exports every symbol include/pianobart_hip.h declares.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os

import pytest

from pianobart_amd import _lib


def test_header_parses_and_library_exports_all():
    decls = _lib.parse_header()
    assert len(decls) >= 20 and 'pb_gemm' in decls and 'pb_adamw_step' in decls
    if not os.path.exists(_lib.LIB_PATH):
        from pianobart_amd.build import build
        build(verbose=False)
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in decls:
        assert hasattr(dll, name), 'library does not export %s' % name
    assert _lib.LIB.query('pb_abi_version') == 7
    assert _lib.LIB.query('pb_ln_partials_floats', 768) == 512 * 3 * 768


def test_gemm_desc_layout_matches_header():
    # 6 pointers + 10 int32 + 10 int64 + 2 floats + 2 int32 + 1 pointer + colsum out / ws + rowdot out / ld
    assert ctypes.sizeof(_lib.GemmDesc) == 6 * 8 + 10 * 4 + 10 * 8 + 2 * 4 + 2 * 4 + 8 + 2 * 8 + 2 * 8


def test_ops_refuse_cpu_tensors():
    import torch
    from pianobart_amd import ops
    x = torch.zeros(4, 4)
    with pytest.raises(_lib.PBError):
        ops.gemm(x, x, x, M=4, N=4, K=4, dtype=_lib.PB_F32)

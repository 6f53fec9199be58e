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
    assert _lib.LIB.query('pb_abi_version') == 8
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


def test_persistent_gemm_isa_has_no_spills_and_no_copies_of_in_flight_fragments():
    """tools/check_gemm_isa.py on the cross-compiled gfx950 code: a spilled vector register would enter the K loop's counted vmcnt queue,
    and any instruction other than the transposed reads and the MFMAs naming v200-v247 inside the TN loop would touch data still in flight."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('check_gemm_isa', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'check_gemm_isa.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    errs, meta = mod.check(mod.disassemble())
    assert len(meta) == 2 and not errs, errs[:5]


def test_persistent_gemm_waits_for_exactly_what_its_epilogues_issue():
    """tools/check_gemm_epilogue_vmem.py: the persistent GEMM's cross-item wait (`pend`) is a hand count of the VMEM instructions each epilogue
    variant leaves in flight; the check compiles every variant alone (probe kernels over the textually included pb_gemm2.hip), counts its global loads and
    stores in the gfx950 ISA and compares with the `pend` literals of the kernel source -- one store more or fewer in an epilogue fails here instead of racing."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('check_gemm_epilogue_vmem', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'check_gemm_epilogue_vmem.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    errs, got, table = mod.check()
    assert len(got) == 7 and not errs, errs
    assert table == {'plain': 16, 'wide': 32, 'epf': 32, 'epf_cs': 36, 'epf3': 40}

"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/segdistill_hip.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as ge
    ge.build()
    return True


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'segdistill_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sd_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported(built):
    from segdistill_amd import _lib
    h = _lib.lib()
    names = _declared_symbols()
    assert 'sd_cgd_kl_fwd' in names and 'sd_cgd_kl_bwd' in names
    for n in names:
        assert hasattr(h, n), f'{n} declared in the header but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in _lib.SIGNATURES'
    assert sorted(_lib.SIGNATURES) == names
    assert h.sd_abi_version() == _lib.ABI_VERSION
    assert h.sd_error_string(-4) == b'workspace too small or misaligned'


def test_exports_equal_declarations(built):
    """Both directions: the product .so exports exactly the sd_* symbols the header declares -- no undeclared diagnostics."""
    import subprocess
    from segdistill_amd import _lib
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith('sd_') and ln.split()[-2] in 'TtWw'})
    assert exported == _declared_symbols(), sorted(set(exported) ^ set(_declared_symbols()))


def test_workspace_sizing_and_tunables(built):
    from segdistill_amd import _lib
    h = _lib.lib()
    assert h.sd_cgd_kl_workspace_bytes(8, 150, 512, 512, 8) >= 8 * 150 * 20
    assert h.sd_cgd_kl_workspace_bytes(0, 150, 512, 512, 8) == 0
    old = _lib.get_tunable('cgd_fwd_chunk_iters')
    _lib.set_tunable('cgd_fwd_chunk_iters', 4)
    assert _lib.get_tunable('cgd_fwd_chunk_iters') == 4
    _lib.set_tunable('cgd_fwd_chunk_iters', old)
    with pytest.raises(RuntimeError):
        _lib.set_tunable('no_such_key', 1)


def test_ops_refuse_cpu_tensors(built):
    import torch
    from segdistill_amd import ops
    with pytest.raises(RuntimeError, match='GPU only'):
        ops.cgd_kl(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), group_size=2, tau=1.0, alpha=1.0)


def test_missing_library_fails_loudly(built, monkeypatch):
    from segdistill_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libsegdistill_hip.so')
    with pytest.raises(_lib.SegDistillLibError):
        _lib.lib()


def test_ctypes_signatures_match_header_arity(built):
    """Every ctypes signature has exactly as many arguments as the C prototype in the header."""
    from segdistill_amd import _lib
    src = open(os.path.join(ROOT, 'include', 'segdistill_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    for name, (_res, args) in _lib.SIGNATURES.items():
        m = re.search(r'\b' + name + r'\s*\(([^;]*?)\)\s*;', src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ('', 'void') else params.count(',') + 1
        assert n == len(args), f'{name}: header has {n} parameters, ctypes binding {len(args)}'


def test_bf16_split_k_weight_gradient_plans_bound_their_slab_traffic():
    """csrc/wgrad_tn.hip::wgrad_slab_cap (host-side planning only: no GPU): under bf16 storage a split-K weight gradient takes no more k-splits
    than keep its fp32 slabs within `wgrad_slab_ratio` % of its operand bytes (the few-token / large-weight Linears of BASELINE config 5 wrote
    7x their operand bytes before); 0 lifts the cap; fp32 storage is not capped."""
    from segdistill_amd import _lib
    L = _lib.lib()
    shapes = [(2048, 1024, 512), (2048, 512, 512), (8192, 1280, 320), (8192, 768, 256), (32768, 768, 256), (131072, 768, 256), (32768, 512, 128)]
    old = _lib.get_tunable('wgrad_slab_ratio')
    assert old == 40
    try:
        capped = [L.sd_linear_wgrad_generic_slabs(1, *s) for s in shapes]
        for (T, M, N), ns in zip(shapes, capped):
            assert ns == 0 or ns * M * N * 4 <= 0.4 * T * (M + N) * 2 + M * N * 4, (T, M, N, ns)      # 0: a single split writes dW directly
        _lib.set_tunable('wgrad_slab_ratio', 0)
        free = [L.sd_linear_wgrad_generic_slabs(1, *s) for s in shapes]
        assert all(f >= c for f, c in zip(free, capped)) and free[0] > 8 and free[1] > 8
        f32 = [L.sd_linear_wgrad_tn_slabs(8192, 640, 160), L.sd_linear_wgrad_splitk_slabs(2048, 256, 256)]
        _lib.set_tunable('wgrad_slab_ratio', 40)
        assert f32 == [L.sd_linear_wgrad_tn_slabs(8192, 640, 160), L.sd_linear_wgrad_splitk_slabs(2048, 256, 256)]
        with pytest.raises(RuntimeError):
            _lib.set_tunable('wgrad_slab_ratio', -1)
    finally:
        _lib.set_tunable('wgrad_slab_ratio', old)

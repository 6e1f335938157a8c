"""The measurement tools are part of the deliverable (profiles/ is produced by them on the GPU box): every Python tool must at least compile
here, every shell tool must pass `bash -n` and only call tools that exist, and `bench.py`'s kernel-roofline groups must exist in
tools/kernel_rooflines.py."""
import ast
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_compile():
    for path in sorted(glob.glob(os.path.join(ROOT, 'tools', '*.py'))) + [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]:
        with open(path) as f:
            ast.parse(f.read(), filename=path)


def test_shell_tools_parse_and_reference_existing_files():
    for path in sorted(glob.glob(os.path.join(ROOT, 'tools', '*.sh'))):
        assert subprocess.run(['bash', '-n', path]).returncode == 0, path
        with open(path) as f:
            text = f.read()
        for rel in set(re.findall(r'\b(tools/[A-Za-z0-9_]+\.(?:py|sh))\b', text)):
            assert os.path.isfile(os.path.join(ROOT, rel)), f'{path} calls {rel}, which does not exist'


def test_bench_roofline_groups_exist():
    with open(os.path.join(ROOT, 'bench.py')) as f:
        m = re.search(r"def kernel_roofline_entries\(groups=\(([^)]*)\)", f.read())
    assert m, 'bench.py::kernel_roofline_entries not found'
    wanted = re.findall(r"'([a-z0-9_]+)'", m.group(1))
    with open(os.path.join(ROOT, 'tools', 'kernel_rooflines.py')) as f:
        src = f.read()
    have = re.findall(r"^\s+'([a-z0-9_]+)': lambda dev, reps", src, re.M)
    assert wanted and set(wanted) <= set(have), (wanted, have)


def test_no_compiler_instruction_touches_a_register_with_a_pinned_load_pending():
    """tools/asm_pending_audit.py over the kernels that use asm-volatile global loads (ce_up.hip, token_gemm.hip): between such a load and
    its explicit wait the compiler must not copy, spill or reuse the destination registers (round 3: a factor-8 CE backward at ~300
    registers parked them in AGPRs before the data had landed -- a GPU memory fault).  Cross-compiles for gfx950; no GPU needed."""
    import shutil
    import sys
    import pytest
    if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'asm_pending_audit.py')], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    for f in ('ce_up.hip', 'token_gemm.hip', 'head_tail.hip', 'mixffn_tail.hip', 'align_stream.hip'):
        assert f + ': 0 compiler' in res.stdout, res.stdout


def test_no_asm_load_reads_a_scalar_register_a_vector_instruction_just_wrote():
    """tools/asm_sgpr_hazard_scan.py over the kernels with inline-asm loads on a scalar base: a VALU write of an SGPR (the v_readlane restoring a
    SPILLED pointer) needs 5 wait states before a vector-memory read of it, and hipcc inserts none in front of inline asm (round 6: head_tail.hip
    faulted as soon as `fuse_bias` -- a spilled pointer -- was given).  The asm statements carry their own `s_nop`; this checks the built code."""
    import shutil
    import sys
    import pytest
    if not (shutil.which('hipcc') or os.path.exists('/opt/rocm/bin/hipcc')):
        pytest.skip('hipcc not available')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'asm_sgpr_hazard_scan.py')], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    for f in ('head_tail.hip', 'mixffn_tail.hip', 'align_stream.hip', 'tok_gemm_bf16.hip', 'align_tok.hip'):
        assert f + ': 0 asm' in res.stdout, res.stdout


def test_the_hazard_scan_flags_a_restored_pointer_in_front_of_an_asm_load(tmp_path):
    """Positive control for tools/asm_sgpr_hazard_scan.py: the sequence that faulted in round 6 (v_readlane restoring a spilled base pointer, then an
    inline-asm load on it) is reported; with five wait states in between it is not."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import asm_sgpr_hazard_scan as scan
    bad = tmp_path / 'bad.s'
    bad.write_text('_Zkernel:\n\tv_readlane_b32 s98, v240, 14\n\tv_readlane_b32 s99, v240, 15\n\t;;#ASMSTART\n'
                   '\tglobal_load_dwordx4 v[134:137], v239, s[98:99]\n\t;;#ASMEND\n\ts_endpgm\n')
    ok = tmp_path / 'ok.s'
    ok.write_text('_Zkernel:\n\tv_readlane_b32 s98, v240, 14\n\tv_readlane_b32 s99, v240, 15\n\t;;#ASMSTART\n\ts_nop 4\n'
                  '\tglobal_load_dwordx4 v[134:137], v239, s[98:99]\n\t;;#ASMEND\n\ts_endpgm\n')
    found = scan.scan(str(bad))
    assert found and all(98 in f[3] or 99 in f[3] for f in found)
    assert scan.scan(str(ok)) == []


def test_slab_budget_of_config_5_stays_under_its_operand_bytes():
    """tools/slab_budget.py (host-side plans only): with the default cap the bf16 B1 student's split-K weight gradients move fewer slab bytes than
    operand bytes per step; without it more than twice as many (the 2.9 GB / 1.4 GB finding of round 4)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import slab_budget
    lines = []
    slab, op = slab_budget.budget(**slab_budget.PRESETS['cfg5'], out=lines.append)
    assert len(lines) > 25 and slab < op
    slab0, op0 = slab_budget.budget(**slab_budget.PRESETS['cfg5'], ratio=0, out=lambda s: None)
    assert op0 == op and slab0 > 2 * op
    # round 5: the grouped launch's joint plan (what the step runs): a fifth of the operand bytes (<= 300 MB)
    slab_g, op_g = slab_budget.budget(**slab_budget.PRESETS['cfg5'], out=lambda s: None, grouped=True)
    assert op_g == op and slab_g <= 300e6 and slab_g < 0.2 * op
    slab2, op2 = slab_budget.budget(**slab_budget.PRESETS['cfg2'], out=lambda s: None)
    assert slab2 > 0 and op2 > 0

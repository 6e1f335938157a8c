"""No source-tree file above 8 MB (round 5 committed a 154 MB checkpoint by accident; tools/git-hooks/pre-commit guards the index, this test the tree:
built artefacts and scratch directories that .gitignore lists are skipped)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP_DIRS = {'.git', 'gpurun_out', '.gpurun', '__pycache__', '.pytest_cache', '.hypothesis', 'work_dirs', 'build', 'lib', 'lib_ab', '_ref', '.claude'}
SKIP_EXT = {'.so', '.o', '.a', '.hsaco', '.co', '.pyc'}


def test_no_large_file_in_the_source_tree():
    big = []
    for d, dirs, files in os.walk(ROOT):
        dirs[:] = [x for x in dirs if x not in SKIP_DIRS]
        for f in files:
            if os.path.splitext(f)[1] in SKIP_EXT:
                continue
            p = os.path.join(d, f)
            if os.path.isfile(p) and os.path.getsize(p) > 8 * 1024 * 1024:
                big.append((os.path.relpath(p, ROOT), os.path.getsize(p)))
    assert not big, big


def test_the_pre_commit_guard_is_executable():
    hook = os.path.join(ROOT, 'tools', 'git-hooks', 'pre-commit')
    assert os.path.isfile(hook) and os.access(hook, os.X_OK)

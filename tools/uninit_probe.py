"""Poison the caching allocator's free blocks with NaNs, then run a test function: an op that reads memory it never wrote shows up
as NaN (or as a mismatch) instead of passing by luck on fresh, zero-filled pages.
usage: python tools/uninit_probe.py tests/test_layernorm_gpu.py::test_fused_stage_loop_equals_block_by_block [repeat]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

path, name = sys.argv[1].split('::')
mod = importlib.import_module(os.path.splitext(os.path.basename(path))[0])
fn = getattr(mod, name)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = 0
for i in range(reps):
    sizes = [1 << 12, 1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 26]
    junk = [torch.full((n,), float('nan'), device='cuda:0') for n in sizes for _ in range(12)]
    torch.cuda.synchronize()
    del junk                       # the blocks stay cached -- full of NaNs -- and are handed out again below
    try:
        fn()
        print('run', i, 'ok', flush=True)
    except AssertionError as e:
        bad += 1
        print('run', i, 'FAILED:', str(e)[:1500], flush=True)
print('failures:', bad, 'of', reps)

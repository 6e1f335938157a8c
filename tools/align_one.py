"""One shape of the NCHW align projection's forward, a few launches, for the profiler (tools/pmc_run.sh).  python tools/align_one.py [B Cs Ct h w] [--generic]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda:0')
nums = [int(v) for v in sys.argv[1:] if v.isdigit()]
B, Cs, Ct, h, w = nums if len(nums) == 5 else (8, 128, 512, 64, 64)
if '--generic' in sys.argv:
    L.sd_set_tunable(b'align_stream', 0)
x = torch.randn(B, Cs, h, w, device=dev)
wt = torch.randn(Ct, Cs, device=dev) / Cs ** 0.5
b = torch.randn(Ct, device=dev)
y = torch.empty(B, Ct, h, w, device=dev)
for _ in range(12):
    assert L.sd_align1x1_fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), 0, B, Cs, Ct, h, w, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()

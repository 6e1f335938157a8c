"""One token-GEMM shape launched a few times, for rocprofv3 (kernel trace or --pmc passes):  python tools/gemm_probe.py T K N [bwd] [lib] [x3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import token_gemm  # noqa: E402

T, K, N = (int(v) for v in sys.argv[1:4])
bwd, lib, x3 = 'bwd' in sys.argv[4:], 'lib' in sys.argv[4:], 'x3' in sys.argv[4:]
dev = torch.device('cuda:0')
x = torch.randn(T, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
b = torch.randn(N, device=dev)
dy = torch.randn(T, N, device=dev)
for _ in range(6):
    if lib:
        y = (dy @ w) if bwd else torch.nn.functional.linear(x, w, b)
    else:
        y = token_gemm.linear_bwd_data(dy, w, split_bf16=x3) if bwd else token_gemm.linear_fwd(x, w, b, split_bf16=x3)
torch.cuda.synchronize()
print('ok', float(y.abs().mean()))

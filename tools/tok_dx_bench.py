"""dX = dY . W of the config-5 align projection (bf16, C = 768 -> K = 256) per stage: csrc/align_tok.hip's tok_dx_kernel vs the library product.
python tools/tok_dx_bench.py   (run on the GPU box; device time by HIP events)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import _lib  # noqa: E402
from tools.align_tok_bench import t_us  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for K in (256, 128, 64):
        for T in (131072, 32768, 8192, 2048):
            M = 768
            dy = torch.randn(T, M, device=dev).bfloat16()
            w = (torch.randn(M, K, device=dev) / 16).bfloat16()
            dx = torch.empty(T, K, device=dev, dtype=torch.bfloat16)

            def ours():
                assert L.sd_linear_tok_bf16_bwd_data(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), T, M, K, st) == 0

            def lib():
                torch.mm(dy, w, out=dx)

            a, _ = t_us(ours)
            b, _ = t_us(lib)
            print(f'dX = dY[{T} x {M}] . W[{M} x {K}]   ours {a:7.1f} us   library {b:7.1f} us   -> {"ours" if a < b else "library"}')


if __name__ == '__main__':
    main()

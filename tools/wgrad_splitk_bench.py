"""Weight gradients of the stage 3-4 / SR-path Linears of Segformer-B0 at BASELINE config 2 (few tokens or a large weight: the shapes outside
sd_linear_wgrad's tall-skinny plan): the library's dY^T @ X against sd_linear_wgrad_splitk + the slab combine.  Device time inside a
replayed hipGraph (tools/gemm_bench.py::timeit)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gemm_bench import timeit  # noqa: E402
from segdistill_amd import _lib, deferred  # noqa: E402
from segdistill_amd.ops import _stream_ptr  # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
SHAPES = [('s3 fc1', 8192, 640, 160, 2), ('s3 fc2', 8192, 160, 640, 2), ('s3 kv', 2048, 320, 160, 2), ('s3 sr', 2048, 160, 640, 2),
          ('s4 q/proj', 2048, 256, 256, 4), ('s4 kv', 2048, 512, 256, 2), ('s4 fc1', 2048, 1024, 256, 2), ('s4 fc2', 2048, 256, 1024, 2),
          ('head c4/fuse4', 2048, 256, 256, 2), ('s2 kv', 2048, 128, 64, 2), ('s2 sr', 2048, 64, 1024, 2), ('s1 kv', 2048, 64, 32, 2),
          ('s1 sr', 2048, 32, 2048, 2)]
tl = ts = 0.0
print(f'{"shape":<14} {"T":>6} {"out":>5} {"in":>5} {"n":>2} | {"library":>8} {"split-K":>8} {"+combine":>9} slabs')
for tag, T, M, N, cnt in SHAPES:
    dy, x = torch.randn(T, M, device=dev), torch.randn(T, N, device=dev)
    ns = L.sd_linear_wgrad_splitk_slabs(T, M, N)
    ws = torch.empty(ns, M * N, device=dev)
    out = torch.empty(M * N, device=dev)

    def k():
        _lib.check(L.sd_linear_wgrad_splitk(dy.data_ptr(), x.data_ptr(), ws.data_ptr(), ws.numel() * 4, T, M, N, _stream_ptr()), 'splitk')

    def kc():
        k()
        deferred.reduce_now(ws, out, M * N, ns)
    t_lib, t_k, t_kc = timeit(lambda: dy.t() @ x), timeit(k), timeit(kc)
    tl += cnt * t_lib
    ts += cnt * t_kc
    print(f'{tag:<14} {T:>6} {M:>5} {N:>5} {cnt:>2} | {t_lib:8.1f} {t_k:8.1f} {t_kc:9.1f} {ns:5d}')
print(f'sum over the student: library {tl / 1e3:.3f} ms, split-K + combine {ts / 1e3:.3f} ms (the step defers the combines to one batched launch)')

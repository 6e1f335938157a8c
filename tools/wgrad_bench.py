"""Linear weight gradient dW = dY^T X (sd_linear_wgrad) at the shapes of BASELINE config 2's student: us per call (kernel + slab
combine, hipGraph-free, 50 back-to-back calls), GB/s of operand traffic and TFLOP/s on the f32-input MFMA."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from segdistill_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(131072, 32, 32), (131072, 128, 32), (131072, 32, 128), (32768, 64, 64), (32768, 256, 64), (32768, 64, 256), (8192, 160, 160),
          (8192, 640, 160), (8192, 160, 640), (131072, 256, 32), (131072, 256, 256), (131072, 150, 256), (32768, 256, 256),
          (8192, 256, 160), (8192, 256, 256), (4096, 256, 256), (4096, 1024, 256), (4096, 512, 256), (32768, 256, 128),
          (131072, 256, 160), (131072, 256, 64), (32768, 256, 32), (32768, 512, 128), (32768, 128, 512)]
for (T, M, N) in SHAPES:
    dy, x = torch.randn(T, M, device=dev), torch.randn(T, N, device=dev)
    dw, db = torch.empty(M, N, device=dev), torch.empty(M, device=dev)
    wsb = L.sd_linear_wgrad_workspace_bytes(T, M, N)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bias = db.data_ptr() if L.sd_linear_wgrad_fuses_bias(T, M, N) else None

    def run():
        rc = L.sd_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), bias, 0, T, M, N, ws.data_ptr(), wsb, st)
        assert rc == 0, rc
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    for _ in range(5):
        lib_dw = dy.t() @ x
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        lib_dw = dy.t() @ x
    e1.record()
    torch.cuda.synchronize()
    lib_us = e0.elapsed_time(e1) / 50 * 1e3
    ref = dy.double().t() @ x.double()
    err = float((dw.double() - ref).norm() / ref.norm())
    # round 4: the same product as split-K slabs on transposed LDS reads, split-bf16 arithmetic (csrc/wgrad_tn.hip): kernel + one batched combine
    tn = ''
    ns = L.sd_linear_wgrad_tn_slabs(T, M, N)
    if ns:
        import ctypes as C_
        from segdistill_amd.deferred import reduce_now
        slabs, out = torch.empty(ns, M * N, device=dev), torch.empty(M * N, device=dev)

        def run_tn():
            rc = L.sd_linear_wgrad_tn(dy.data_ptr(), x.data_ptr(), slabs.data_ptr(), slabs.numel() * 4, T, M, N, 0, st)
            assert rc == 0, rc
            reduce_now(slabs, out, M * N, ns)
        for _ in range(5):
            run_tn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            run_tn()
        e1.record()
        torch.cuda.synchronize()
        tn_us = e0.elapsed_time(e1) / 50 * 1e3
        tn = f'   | transposed-read split-bf16 kernel + combine ({ns} slabs) {tn_us:7.1f} us  rel err {float((out.view(M, N).double() - ref).norm() / ref.norm()):.1e}'
    print(f'T={T:6d} out={M:3d} in={N:3d}  {us:7.1f} us  {T * (M + N) * 4 / us / 1e3:7.0f} GB/s  {2.0 * T * M * N / us / 1e6:6.1f} TFLOP/s  rel err {err:.1e}   library mm {lib_us:7.1f} us  direct={bool(L.sd_linear_wgrad_fuses_bias(T, M, N))}{tn}')

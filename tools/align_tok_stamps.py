"""Diagnostic: cycles per phase of the block loop of the fused align + criterion kernels (csrc/align_tok.hip), summed per wave of workgroup 0.
    python tools/align_tok_stamps.py [fwd|bwd|plain]
Needs a DIAGNOSTIC build of the library (the product build has no stamp code and does not export sd_debug_align_stamps):
    make -C segdistill_amd/csrc OUTDIR=../lib_stamps EXTRA=-DSD_ALIGN_STAMPS && SEGDISTILL_LIB=$PWD/segdistill_amd/lib_stamps/libsegdistill_hip.so python tools/align_tok_stamps.py
Columns: prologue | wait own DMAs | barrier | stores + merge + DMA issue | B fragments + MFMA | epilogue | total (s_memtime ticks = shader cycles)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import _lib, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
B, P, K, Cc, g = 8, 16384, 256, 768, 8
dev = torch.device('cuda:0')
L = _lib.lib()
L.sd_debug_align_stamps.restype, L.sd_debug_align_stamps.argtypes = None, [C.c_void_p]
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
x = torch.randn(B, P, K, device=dev).to(torch.bfloat16)
t = (2 * torch.randn(B, P, Cc, device=dev)).to(torch.bfloat16)
w = (torch.randn(Cc, K, device=dev) / 16).to(torch.bfloat16)
b = 0.1 * torch.randn(Cc, device=dev)
rows = B * Cc // g
wsb = L.sd_align_cgd_tok_workspace_bytes(B, Cc, P)
ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
lse, kl, loss = torch.zeros(rows, 2, device=dev), torch.empty(rows, device=dev), torch.empty((), device=dev)
dY = torch.empty(B, P, Cc, dtype=torch.bfloat16, device=dev)
dbp = torch.empty(L.sd_align_cgd_tok_tiles(B, P), Cc, device=dev)
job = (ops._AlignTokJob * 1)()
j = job[0]
j.X, j.W, j.bias, j.T = x.data_ptr(), w.data_ptr(), b.data_ptr(), t.data_ptr()
j.row_lse2, j.row_kl, j.loss, j.workspace, j.workspace_bytes = lse.data_ptr(), kl.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb
j.out, j.db_part = dY.data_ptr(), dbp.data_ptr()
j.P, j.B, j.K, j.C, j.g, j.inv_tau, j.loss_scale, j.coef = P, B, K, Cc, g, 0.25, 3.0 / rows, 3.0 / (rows * 4.0)
fns = {'fwd': lambda: L.sd_align_cgd_tok_fwd_multi(C.cast(job, C.c_void_p), 1, st), 'bwd': lambda: L.sd_align_cgd_tok_bwd_multi(C.cast(job, C.c_void_p), 1, st),
       'plain': lambda: L.sd_linear_tok_bf16_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), dY.data_ptr(), B * P, K, Cc, st)}
_lib.check(fns['fwd'](), 'fwd')
for _ in range(3):
    _lib.check(fns[mode](), mode)
torch.cuda.synchronize()
buf = torch.zeros(8 * 8, dtype=torch.int64, device=dev)
L.sd_debug_align_stamps(buf.data_ptr())
_lib.check(fns[mode](), mode)
torch.cuda.synchronize()
L.sd_debug_align_stamps(None)
v = buf.cpu().view(8, 8)
print(f'{mode}: cycles per wave of workgroup 0 (24 blocks)')
print('wave  prologue  wait-dma   barrier  stores+issue  B+MFMA  epilogue     total')
for wv in range(8):
    r = [int(q) for q in v[wv]]
    print(f'{wv:4d} {r[0]:9d} {r[1]:9d} {r[2]:9d} {r[3]:13d} {r[4]:7d} {r[5]:9d} {r[6]:9d}')

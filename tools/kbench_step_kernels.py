"""HBM-roofline check of the step-level HIP kernels at the shapes they see in BASELINE config 2 (run on the GPU box).
Reports HIP-event time per launch and algorithmic GB/s (bytes = tensors read once + written once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from segdistill_amd import _lib
from segdistill_amd.dwconv import dwconv3x3_tokens
from segdistill_amd.layernorm import HipLayerNorm
from segdistill_amd.headfuse import upsum
from segdistill_amd.ce import fused_ce_up
from segdistill_amd.linear import token_linear

dev = torch.device('cuda:0')


def t_ms(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def row(name, ms, nbytes, ref_ms=None):
    extra = '' if ref_ms is None else f'   [ATen/library: {ref_ms:.3f} ms, {ref_ms / ms:.1f}x]'
    print(f'{name:58s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.0f} GB/s ({nbytes / ms / 1e6 / 8000 * 100:4.1f} % of 8 TB/s){extra}')


B = 8
print('# HIP-event time around the Python call of each op (autograd Function + ctypes launch included): entries under ~25 us are bounded by that\n'
      '# launch path, not by the kernel -- GPU durations inside the captured step are in profiles/r01_train_step_kernels_final.txt')
print('--- depth-wise 3x3, token-major (teacher B2 stage 1: C=256; student B0 stage 1: C=128), fwd')
for C in (256, 128):
    x = torch.randn(B, 128 * 128, C, device=dev)
    w = torch.randn(C, 1, 3, 3, device=dev); b = torch.randn(C, device=dev)
    ms = t_ms(lambda: dwconv3x3_tokens(x, w, b, 128, 128))
    xn = x.transpose(1, 2).reshape(B, C, 128, 128).contiguous()
    ref = t_ms(lambda: F.conv2d(xn, w, b, padding=1, groups=C))
    row(f'dw3x3_fwd  [8,16384,{C}]', ms, 2 * x.numel() * 4, ref)
xg = torch.randn(B, 128 * 128, 128, device=dev, requires_grad=True)
wg = torch.randn(128, 1, 3, 3, device=dev, requires_grad=True); bg = torch.randn(128, device=dev, requires_grad=True)
y = dwconv3x3_tokens(xg, wg, bg, 128, 128); dy = torch.randn_like(y)
ms = t_ms(lambda: torch.autograd.grad(y, (xg, wg, bg), dy, retain_graph=True))
row('dw3x3 bwd (data+weight+bias) [8,16384,128]', ms, 5 * xg.numel() * 4)
print('--- LayerNorm, token-major')
for C in (32, 64, 256):
    x = torch.randn(B, 16384, C, device=dev, requires_grad=True)
    ln = HipLayerNorm(C, eps=1e-6).to(dev)
    ms = t_ms(lambda: ln(x.detach()))
    ref = t_ms(lambda: F.layer_norm(x.detach(), (C,), ln.weight, ln.bias, 1e-6))
    row(f'ln_fwd [131072,{C}]', ms, 2 * x.numel() * 4, ref)
    y = ln(x); dy = torch.randn_like(y)
    ms = t_ms(lambda: torch.autograd.grad(y, (x, ln.weight, ln.bias), dy, retain_graph=True))
    row(f'ln_bwd [131072,{C}]', ms, 3 * x.numel() * 4)
print('--- SegFormer head up-sample + sum, token-major')
for E in (768, 256):
    sizes = [(128, 128), (64, 64), (32, 32), (16, 16)]
    zs = [torch.randn(B, h * w, E, device=dev) for h, w in sizes]
    ms = t_ms(lambda: upsum(zs[0], zs[1], zs[2], zs[3], None, sizes))
    row(f'upsum_fwd E={E}', ms, 2 * zs[0].numel() * 4)
print('--- fused up-sample + cross-entropy (logits [8,150,128,128] -> 512x512)')
lg = torch.randn(B, 150, 128, 128, device=dev, requires_grad=True)
lab = torch.randint(0, 150, (B, 1, 512, 512), device=dev)
ms = t_ms(lambda: fused_ce_up(lg.detach(), lab, 255))
def aten_ce():
    up = F.interpolate(lg.detach(), size=(512, 512), mode='bilinear', align_corners=False)
    return F.cross_entropy(up, lab.squeeze(1), reduction='none', ignore_index=255), up.argmax(1)
ref = t_ms(aten_ce)
row('ce_up_fwd (bytes by the R1 definition: 2 x upsampled logits)', ms, 2 * B * 150 * 512 * 512 * 4, ref)
lp, _ = fused_ce_up(lg, lab, 255)
ms = t_ms(lambda: torch.autograd.grad(lp.mean(), lg, retain_graph=True))
row('ce_up_bwd (bytes by the R1 definition: 3 x upsampled logits)', ms, 3 * B * 150 * 512 * 512 * 4)
print('--- Linear weight gradient dW = dY^T X, tokens = 131072')
for (M, N) in ((32, 32), (128, 32), (32, 128), (256, 64)):
    T = 131072 if M * N <= 4096 else 32768
    dyl = torch.randn(T, M, device=dev); xl = torch.randn(T, N, device=dev)
    L = _lib.lib(); dw = torch.empty(M, N, device=dev)
    wsb = L.sd_linear_wgrad_workspace_bytes(T, M, N); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ms = t_ms(lambda: L.sd_linear_wgrad(dyl.data_ptr(), xl.data_ptr(), dw.data_ptr(), None, 0, T, M, N, ws.data_ptr(), wsb, st))
    ref = t_ms(lambda: dyl.t() @ xl)
    row(f'linear_wgrad T={T} out={M} in={N}', ms, T * (M + N) * 4, ref)

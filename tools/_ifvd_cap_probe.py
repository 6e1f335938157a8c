import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from segdistill_amd import _lib
dev = torch.device('cuda:0')
B, C, hw = 8, 150, 128
HW, K = hw * hw, C
S = torch.randn(B, C, hw, hw, device=dev)
T = torch.randn(B, C, hw, hw, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else 'random'
if mode == 'random':
    cls = torch.randint(0, C, (B, HW), device=dev, dtype=torch.int32)
else:   # blocky: 16 x 16 patches of one class (spatially coherent, like a label map)
    cls = torch.randint(0, C, (B, hw // 16, hw // 16), device=dev, dtype=torch.int32).repeat_interleave(16, 1).repeat_interleave(16, 2).reshape(B, HW).contiguous()
L = _lib.lib()
f32 = dict(dtype=torch.float32, device=dev)
counts = torch.empty(B, K, dtype=torch.int32, device=dev)
mean_s, mean_t = torch.empty(B, C, K, **f32), torch.empty(B, C, K, **f32)
coefs = torch.empty(3, B * HW, **f32)
loss = torch.empty((), **f32)
wsb = L.sd_ifvd_workspace_bytes(B, C, HW, K); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
A, Bk = torch.empty(B, C, K, **f32), torch.empty(B, K, **f32)
dS = torch.empty_like(S)
def st(): return torch.cuda.current_stream().cuda_stream
steps = {
 'counts': lambda: L.sd_ifvd_counts(cls.data_ptr(), B, HW, K, counts.data_ptr(), st()),
 'means': lambda: L.sd_ifvd_class_means(S.data_ptr(), T.data_ptr(), 0, cls.data_ptr(), counts.data_ptr(), mean_s.data_ptr(), mean_t.data_ptr(), ws.data_ptr(), wsb, B, C, HW, K, st()),
 'cos': lambda: L.sd_ifvd_cos(S.data_ptr(), T.data_ptr(), 0, cls.data_ptr(), mean_s.data_ptr(), mean_t.data_ptr(), coefs.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, B, C, HW, K, st()),
 'sums': lambda: L.sd_ifvd_coef_sums(S.data_ptr(), 0, cls.data_ptr(), counts.data_ptr(), coefs.data_ptr(), A.data_ptr(), Bk.data_ptr(), ws.data_ptr(), wsb, B, C, HW, K, st()),
 'bwd': lambda: L.sd_ifvd_bwd(S.data_ptr(), 0, cls.data_ptr(), mean_s.data_ptr(), coefs.data_ptr(), A.data_ptr(), Bk.data_ptr(), counts.data_ptr(), None, dS.data_ptr(), B, C, HW, K, st()),
}
for n, f in steps.items():
    print(n, 'eager rc', f(), flush=True)
torch.cuda.synchronize()
tot = {}
for n, f in steps.items():
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            rc = f()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    g.replay(); torch.cuda.synchronize()
    evs[0].record()
    for i in range(20):
        g.replay(); evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(20))
    tot[n] = ts[10] * 1e3
    print(f'{mode} {n} replay us {ts[10] * 1e3:.1f}', flush=True)
N = S.numel() * 4
fw = tot['counts'] + tot['means'] + tot['cos']; bw = tot['sums'] + tot['bwd']
print(f'{mode} forward {fw:.1f} us = {4 * N / fw / 1e6:.2f} TB/s ({4 * N / fw / 8e6 * 100:.1f} % of 8 TB/s); backward {bw:.1f} us = {3 * N / bw / 1e6:.2f} TB/s ({3 * N / bw / 8e6 * 100:.1f} %)')

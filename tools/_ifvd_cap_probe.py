import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd())
import torch
from segdistill_amd import _lib, ops
dev = torch.device('cuda:0')
B, C, hw = 8, 150, 128
HW, K = hw * hw, C
S = torch.randn(B, C, hw, hw, device=dev)
T = torch.randn(B, C, hw, hw, device=dev)
cls = torch.randint(0, C, (B, HW), device=dev, dtype=torch.int32)
L = _lib.lib()
i32 = dict(dtype=torch.int32, device=dev); f32 = dict(dtype=torch.float32, device=dev)
order, pos, offsets, skey = torch.empty(B, HW, **i32), torch.empty(B, HW, **i32), torch.empty(B, K + 1, **i32), torch.empty(B, HW, **i32)
mean_s, mean_t = torch.empty(B, K, C, **f32), torch.empty(B, K, C, **f32)
coef_px, coef_sorted = torch.empty(2, B * HW, **f32), torch.empty(2, B * HW, **f32)
loss = torch.empty((), **f32)
wsb = L.sd_ifvd_workspace_bytes(B, HW); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
A, Bk = torch.empty(B, K, C, **f32), torch.empty(B, K, **f32)
dS = torch.empty_like(S)
def st(): return torch.cuda.current_stream().cuda_stream
steps = {
 'group': lambda: L.sd_ifvd_group(cls.data_ptr(), B, HW, K, order.data_ptr(), offsets.data_ptr(), pos.data_ptr(), skey.data_ptr(), st()),
 'means': lambda: L.sd_ifvd_class_means(S.data_ptr(), T.data_ptr(), 0, order.data_ptr(), skey.data_ptr(), offsets.data_ptr(), mean_s.data_ptr(), mean_t.data_ptr(), B, C, HW, K, st()),
 'cos': lambda: L.sd_ifvd_cos(S.data_ptr(), T.data_ptr(), 0, cls.data_ptr(), pos.data_ptr(), mean_s.data_ptr(), mean_t.data_ptr(), coef_px.data_ptr(), coef_sorted.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, B, C, HW, K, st()),
 'sums': lambda: L.sd_ifvd_coef_sums(S.data_ptr(), 0, order.data_ptr(), skey.data_ptr(), offsets.data_ptr(), coef_sorted.data_ptr(), A.data_ptr(), Bk.data_ptr(), B, C, HW, K, st()),
 'bwd': lambda: L.sd_ifvd_bwd(S.data_ptr(), 0, cls.data_ptr(), mean_s.data_ptr(), coef_px.data_ptr(), A.data_ptr(), Bk.data_ptr(), offsets.data_ptr(), None, dS.data_ptr(), B, C, HW, K, st()),
}
for n, f in steps.items():
    print(n, 'eager rc', f(), flush=True)
torch.cuda.synchronize()
for n, f in steps.items():
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            rc = f()
    print(n, 'captured rc', rc, flush=True)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    g.replay(); torch.cuda.synchronize()
    evs[0].record()
    for i in range(20):
        g.replay(); evs[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(20))
    print(n, 'replay us', ts[10] * 1e3, flush=True)

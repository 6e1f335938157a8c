"""Probe: hipMemsetAsync inside a captured hipGraph (memset NODE) vs eager, ROCm 7.2 / gfx950 via PyTorch 2.10+rocm7.0.
(observed symptom in the KD step: an int counter zeroed by hipMemsetAsync inside a captured graph came back as 0x01010101.)"""
import ctypes, torch
hip = ctypes.CDLL('libamdhip64.so')   # already loaded by torch
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
dev = torch.device('cuda:0')
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
work = torch.randn(2048, 2048, device=dev)

def uniq(t):
    u, c = torch.unique(t, return_counts=True)
    return {('0x%08x' % (int(v) & 0xffffffff)): int(n) for v, n in zip(u[:4], c[:4])}

for nbytes in (4, 64, 4096, 4 << 20):
    n = max(1, nbytes // 4)
    a = torch.full((n,), 7, dtype=torch.int32, device=dev)
    out = torch.empty_like(a)
    def body(val):
        (work @ work)
        rc = hip.hipMemsetAsync(a.data_ptr(), val, nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        out.copy_(a)
    with torch.cuda.stream(s1):
        body(0)
    torch.cuda.synchronize()
    eager = uniq(out)
    g = torch.cuda.CUDAGraph()
    a.fill_(7); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        body(0)
    res = []
    for other in ('none', 'eager memset(1) on stream 2'):
        a.fill_(7); torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            g.replay()
        if other != 'none':
            b = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
            with torch.cuda.stream(s2):
                hip.hipMemsetAsync(b.data_ptr(), 1, 4 << 20, s2.cuda_stream)
        torch.cuda.synchronize()
        res.append((other, uniq(out)))
    print(f'memset(0) of {nbytes} bytes: eager -> {eager}; graph replay -> {res}')

"""SDPA vs explicit softmax(QK^T)V for the MiT SR-attention shapes (fp32, fwd+bwd), MI355X."""
import torch, time
import torch.nn.functional as F
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
shapes = [(8,1,16384,256,32),(8,2,4096,256,32),(8,5,1024,256,32),(8,8,256,256,32),(8,1,16384,256,64),(8,2,4096,256,64),(8,5,1024,256,64),(8,8,256,256,64)]
for (B,H,N,M,D) in shapes:
    q = torch.randn(B,H,N,D,device=dev,requires_grad=True); k = torch.randn(B,H,M,D,device=dev,requires_grad=True); v = torch.randn(B,H,M,D,device=dev,requires_grad=True)
    scale = D**-0.5
    def sdpa(bwd=True):
        o = F.scaled_dot_product_attention(q,k,v,scale=scale)
        if bwd: o.sum().backward()
    def expl(bwd=True):
        o = ((q@k.transpose(-2,-1))*scale).softmax(-1)@v
        if bwd: o.sum().backward()
    with torch.no_grad():
        f1 = timeit(lambda: F.scaled_dot_product_attention(q,k,v,scale=scale)); f2 = timeit(lambda: ((q@k.transpose(-2,-1))*scale).softmax(-1)@v)
    print(f'B{B} H{H} N{N} M{M} D{D}: fwd sdpa {f1:.3f} expl {f2:.3f} | fwd+bwd sdpa {timeit(sdpa):.3f} expl {timeit(expl):.3f} ms')

"""Token-major Linear products of the KD step at BASELINE config-2 shapes (Segformer-B0 student fwd + bwd-data, B2 teacher fwd, both heads):
the library GEMM (hipBLASLt through F.linear / matmul) against the MFMA kernels of csrc/token_gemm.hip, with the roofline of each shape
(max of flops / 157.3 TF f32-input MFMA and algorithmic bytes / 6.3 TB/s measured HBM ceiling).

    python tools/gemm_bench.py [--dtype f32|bf16] [--only student|teacher|head] [--reps 20]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, reps=20, warm=3, per_graph=10):
    """DEVICE time per call in us: `per_graph` calls are captured into a hipGraph and the graph is replayed -- an eager loop measures the host
    (hipBLASLt's dispatch alone costs ~23 us per call, more than most of these kernels run)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(per_graph):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        g.replay()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2] * 1e3 / per_graph   # us


def mit_shapes(dims, depths, B=8, side=512):
    """(tag, tokens, in_features, out_features, count per forward)"""
    out = []
    sr = (8, 4, 2, 1)
    for s, (c, d) in enumerate(zip(dims, depths)):
        t = B * (side // (4 * 2 ** s)) ** 2
        tr = t // (sr[s] ** 2)
        out += [(f's{s + 1} q/proj', t, c, c, 2 * d), (f's{s + 1} fc1', t, c, 4 * c, d), (f's{s + 1} fc2', t, 4 * c, c, d), (f's{s + 1} kv', tr, c, 2 * c, d)]
        if sr[s] > 1:
            out.append((f's{s + 1} sr', tr, sr[s] ** 2 * c, c, d))
    return out


def head_shapes(dims, E, B=8, side=512):
    out = []
    for s, c in enumerate(dims):
        t = B * (side // (4 * 2 ** s)) ** 2
        out += [(f'head c{s + 1}', t, c, E, 1), (f'head fuse{s + 1}', t, E, E, 1)]
    out.append(('head pred', B * (side // 4) ** 2, E, 150, 1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--only', default=None)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--planes-tile', type=int, default=0, help='force the row-tile height of the planes GEMM (128 / 64; 0 = the launcher rule)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    dt = torch.float32 if a.dtype == 'f32' else torch.bfloat16
    try:
        from segdistill_amd import token_gemm
    except Exception as e:  # noqa: BLE001
        token_gemm = None
        print('token_gemm unavailable:', e)
    if token_gemm and a.planes_tile:
        from segdistill_amd import _lib
        _lib.set_tunable('planes_tile', a.planes_tile)
    groups = {
        'student': ('Segformer-B0 student (fwd + bwd-data)', mit_shapes((32, 64, 160, 256), (2, 2, 2, 2)) + head_shapes((32, 64, 160, 256), 256), True),
        'teacher': ('Segformer-B2 teacher (fwd only)', mit_shapes((64, 128, 320, 512), (3, 4, 6, 3)), False),
    }
    peak_f = 157.3e12 if dt == torch.float32 else 2500e12
    tot = {}
    for key, (title, shapes, with_bwd) in groups.items():
        if a.only and key != a.only:
            continue
        print(f'== {title}, {a.dtype}; us per call; roofline = max(flops/{peak_f / 1e12:.0f} TF, bytes/6.3 TB/s)')
        print(f'{"shape":<14} {"T":>7} {"K":>5} {"N":>5} {"n":>3} | {"roof":>7} {"lib fwd":>8} {"hip fwd":>8} {"x3 fwd":>8} {"pl fwd":>8} | {"lib dX":>8} {"hip dX":>8} {"x3 dX":>8} {"pl dX":>8}   (pl = x3 on pre-split weight planes)')
        s_lib = s_hip = s_roof = s_best = 0.0
        for tag, T, K, N, cnt in shapes:
            x = torch.randn(T, K, device=dev).to(dt)
            w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
            b = torch.randn(N, device=dev).to(dt)
            dy = torch.randn(T, N, device=dev).to(dt)
            e = x.element_size()
            roof = max(2.0 * T * K * N / peak_f, (T * K + T * N + N * K) * e / 6.3e12) * 1e6
            lib_f = timeit(lambda: F.linear(x, w, b), a.reps)
            hip_f = timeit(lambda: token_gemm.linear_fwd(x, w, b), a.reps) if token_gemm else float('nan')
            x3_f = timeit(lambda: token_gemm.linear_fwd(x, w, b, split_bf16=True), a.reps) if token_gemm and dt == torch.float32 else float('nan')
            pl_f = pl_b = float('nan')
            if token_gemm and dt == torch.float32 and K % 32 == 0:
                from segdistill_amd import planes
                pf = planes.get(w, 'fwd')
                pl_f = timeit(lambda: token_gemm.linear_fwd_planes(x, w, pf, b), a.reps)
            lib_b = hip_b = x3_b = float('nan')
            if with_bwd:
                lib_b = timeit(lambda: dy @ w, a.reps)
                hip_b = timeit(lambda: token_gemm.linear_bwd_data(dy, w), a.reps) if token_gemm else float('nan')
                x3_b = timeit(lambda: token_gemm.linear_bwd_data(dy, w, split_bf16=True), a.reps) if token_gemm and dt == torch.float32 else float('nan')
                if token_gemm and dt == torch.float32 and N % 32 == 0:
                    pb = planes.get(w, 'bwd')
                    pl_b = timeit(lambda: token_gemm.linear_bwd_data_planes(dy, w, pb), a.reps)
            print(f'{tag:<14} {T:>7} {K:>5} {N:>5} {cnt:>3} | {roof:7.1f} {lib_f:8.1f} {hip_f:8.1f} {x3_f:8.1f} {pl_f:8.1f} | {lib_b:8.1f} {hip_b:8.1f} {x3_b:8.1f} {pl_b:8.1f}')
            nn = lambda *v: min(t for t in v if t == t)
            s_best += cnt * (nn(lib_f, hip_f, x3_f, pl_f) + (nn(lib_b, hip_b, x3_b, pl_b) if with_bwd else 0))
            s_roof += cnt * roof * (2 if with_bwd else 1)
            s_lib += cnt * (lib_f + (lib_b if with_bwd else 0))
            s_hip += cnt * (hip_f + (hip_b if with_bwd else 0))
            del x, w, dy
        print(f'   sum over the network: roofline(f32 MFMA) {s_roof / 1e3:.3f} ms, library {s_lib / 1e3:.3f} ms, hip f32 {s_hip / 1e3:.3f} ms, best per shape {s_best / 1e3:.3f} ms')
        tot[key] = (s_roof, s_lib, s_hip)


if __name__ == '__main__':
    main()

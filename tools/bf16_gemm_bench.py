"""bf16 token-major Linear forwards of BASELINE config 5 (Segformer-B4 teacher, B1 student, both heads): the library (F.linear -> hipBLASLt) against
csrc/tok_gemm_bf16.hip in its tile / ring variants, device time inside a replayed hipGraph.

    python tools/bf16_gemm_bench.py [--variants] [--reps 20]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import head_shapes, mit_shapes, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', action='store_true', help='also time every (tile, ring) variant, not just the dispatch')
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    from segdistill_amd import _lib, linear
    L = _lib.lib()
    dev = torch.device('cuda:0')
    shapes = {}
    for tag, net in (('B4', mit_shapes((64, 128, 320, 512), (3, 8, 27, 3))), ('B1', mit_shapes((64, 128, 320, 512), (2, 2, 2, 2))),
                     ('head', head_shapes((64, 128, 320, 512), 768)[:-1] + head_shapes((64, 128, 320, 512), 256)[:-1])):
        for name, t, k, n, cnt in net:
            key = (t, k, n)
            prev = shapes.get(key, ('', 0))
            shapes[key] = (prev[0] + ('' if not prev[0] else ' + ') + f'{tag} {name}', prev[1] + cnt)
    # (BM, BN, waves); index = the tunable's value (csrc/tok_gemm_bf16.hip::launch_variant)
    names = ['128x64', '128x128', '128x64/3', '64x64', '64x128']
    bns = [64, 128, 64, 64, 128]
    variants = list(range(len(names))) if a.variants else []
    hdr = f'{"shape":>22s} {"calls":>5s} {"library":>9s} {"ours":>9s}' + ''.join(f' {names[v]:>9s}' for v in variants) + '   used by'
    print(hdr)
    tot_lib = tot_ours = 0.0
    for (t, k, n), (who, cnt) in sorted(shapes.items(), key=lambda kv: (-kv[0][0], kv[0][1], kv[0][2])):
        x = torch.randn(t, k, device=dev).bfloat16()
        w = (torch.randn(n, k, device=dev) / k ** 0.5).bfloat16()
        b = torch.randn(n, device=dev).bfloat16()
        lib = timeit(lambda: F.linear(x, w, b), reps=a.reps)
        if not linear.bf16_tok_gemm_ok(t, k, n):
            print(f'{f"{t} x {k} -> {n}":>22s} {cnt:5d} {lib:9.1f} {"-":>9s}' + ''.join(f' {"-":>9s}' for _ in variants) + f'   {who}')
            tot_lib += lib * cnt
            tot_ours += lib * cnt
            continue
        ours = timeit(lambda: linear.linear_fwd_bf16(x, w, b), reps=a.reps)
        cols = []
        for v in variants:
            if n % bns[v]:
                cols.append(f' {"-":>9s}')
                continue
            L.sd_set_tunable(b'tok_gemm_bf16_variant', v)
            cols.append(f' {timeit(lambda: linear.linear_fwd_bf16(x, w, b), reps=a.reps):9.1f}')
            L.sd_set_tunable(b'tok_gemm_bf16_variant', -1)
        print(f'{f"{t} x {k} -> {n}":>22s} {cnt:5d} {lib:9.1f} {ours:9.1f}' + ''.join(cols) + f'   {who}')
        tot_lib += lib * cnt
        tot_ours += ours * cnt
    print(f'sum over one config-5 step\'s forward calls: library {tot_lib / 1e3:.3f} ms, ours {tot_ours / 1e3:.3f} ms')


if __name__ == '__main__':
    main()

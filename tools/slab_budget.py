"""Byte budget of the split-K weight gradients of a MiT student (host-side planning only: runs without a GPU).  For every token-major Linear of
the backbone, the SegFormer head and (optionally) the align projections it prints the plan the library would take -- kernel family, number of
k-splits -- and the fp32 slab bytes written + read back next to the operand bytes.  This is the table behind `wgrad_slab_cap` (csrc/wgrad_tn.hip):
    python tools/slab_budget.py                 # BASELINE config 5: B1 student, bf16 storage, 768-channel align projections
    python tools/slab_budget.py --preset cfg2   # B0 student, fp32 storage
    python tools/slab_budget.py --ratio 0       # the plans without the cap
    python tools/slab_budget.py --grouped       # config 5 as the step runs it since round 5: ONE grouped launch, k-splits planned jointly"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segdistill_amd import _lib  # noqa: E402

PRESETS = {'cfg5': dict(dims=(64, 128, 320, 512), bf16=True, align=768, embed=256), 'cfg2': dict(dims=(32, 64, 160, 256), bf16=False, align=0, embed=256)}


def plan(L, bf16, T, M, N):
    if bf16:
        if L.sd_linear_wgrad_fuses_bias_dtype(1, T, M, N):
            return 'direct', L.sd_linear_wgrad_slabs(1, T, M, N)
        ns = L.sd_linear_wgrad_generic_slabs(1, T, M, N)
        return ('tn_bf16', ns) if ns else ('tn_bf16 (one split: writes dW)', 0)
    ns = L.sd_linear_wgrad_tn_slabs(T, M, N)
    if ns:
        return 'tn_x3', ns
    if L.sd_linear_wgrad_fuses_bias_dtype(0, T, M, N):
        return 'direct', L.sd_linear_wgrad_slabs(0, T, M, N)
    ns = L.sd_linear_wgrad_splitk_slabs(T, M, N)
    return ('splitk', ns) if ns else ('library', 0)


def budget(dims, bf16, align, embed, batch=8, size=512, depths=(2, 2, 2, 2), sr=(8, 4, 2, 1), ratio=None, out=print, grouped=False):
    """grouped (bf16 only): the plan of the ONE launch that computes every groupable weight gradient at the end of the backward (round 5,
    csrc/wgrad_tn.hip::wgrad_tn_multi_plan) instead of the per-product plans."""
    L = _lib.lib()
    old = _lib.get_tunable('wgrad_slab_ratio')
    if ratio is not None:
        _lib.set_tunable('wgrad_slab_ratio', ratio)
    es = 2 if bf16 else 4
    tot_slab = tot_op = 0.0
    rows = []
    try:
        for s, c in enumerate(dims):
            t = batch * (size // (4 << s)) ** 2
            tk = t // sr[s] ** 2
            layers = [('q', t, c, c), ('kv', tk, 2 * c, c), ('proj', t, c, c), ('fc1', t, 4 * c, c), ('fc2', t, c, 4 * c)]
            if sr[s] > 1:
                layers.append(('sr', tk, c, sr[s] ** 2 * c))
            layers = [(n, *r, depths[s]) for n, *r in layers] + [('linear_c', t, embed, c, 1)] + ([('align', t, align, embed, 1)] if align else [])
            rows += [(s, *l) for l in layers]
        joint = {}
        if grouped and bf16:
            import ctypes as C
            from segdistill_amd import deferred
            idx = [k for k, (s, name, T, M, N, mult) in enumerate(rows)
                   if not L.sd_linear_wgrad_fuses_bias_dtype(1, T, M, N) and L.sd_linear_wgrad_tn_multi_supported(1, T, M, N)]
            jobs = [k for k in idx for _ in range(rows[k][5])]
            arr = (deferred._WgradJob * len(jobs))()
            for q, k in enumerate(jobs):
                arr[q].tokens, arr[q].out_features, arr[q].in_features = rows[k][2], rows[k][3], rows[k][4]
            _lib.check(L.sd_linear_wgrad_tn_multi_plan(C.cast(arr, C.c_void_p), len(jobs), 1), 'sd_linear_wgrad_tn_multi_plan')
            joint = {k: arr[q].nsplit for q, k in enumerate(jobs)}
        if True:
            for k, (s, name, T, M, N, mult) in enumerate(rows):
                kind, ns = plan(L, bf16, T, M, N)
                if k in joint:
                    kind, ns = 'tn_bf16 grouped', (joint[k] if joint[k] > 1 else 0)
                slab, op = 2.0 * ns * M * N * 4 * mult, float(T) * (M + N) * es * mult
                tot_slab += slab
                tot_op += op
                out(f'stage {s + 1} {name:9s} x{mult} T={T:6d} {M:4d} x {N:4d}  {kind:32s} splits={ns:4d}  slabs w+r {slab / 1e6:7.1f} MB  operands {op / 1e6:7.1f} MB')
    finally:
        _lib.set_tunable('wgrad_slab_ratio', old)
    out(f'total: slabs written + read back {tot_slab / 1e6:.0f} MB, operands {tot_op / 1e6:.0f} MB per step')
    return tot_slab, tot_op


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--preset', choices=sorted(PRESETS), default='cfg5')
    ap.add_argument('--ratio', type=int, default=None, help='wgrad_slab_ratio to plan with (default: the library default; 0 = no cap)')
    ap.add_argument('--grouped', action='store_true', help='bf16: the joint plan of the grouped launch (what the training step runs since round 5)')
    a = ap.parse_args()
    budget(**PRESETS[a.preset], ratio=a.ratio, grouped=a.grouped)

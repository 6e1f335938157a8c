"""ctypes binding of libsegdistill_hip.so (the C ABI declared in include/segdistill_hip.h).

The library is the product: if it is missing or fails to load, every op raises.
There is no CPU / eager fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

# torch first: PyTorch-ROCm bundles its own libamdhip64.so.7; loading it before our library
# makes the dynamic linker bind libsegdistill_hip.so to that SAME runtime (one HIP runtime
# per process -- two would give "no ROCm-capable device" on the second one's first launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SEGDISTILL_LIB') or os.path.join(_HERE, 'lib', 'libsegdistill_hip.so')   # SEGDISTILL_LIB: an A/B build of the SAME library

SD_F32, SD_BF16 = 0, 1
ABI_VERSION = 2

_lib = None

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/segdistill_hip.h declares
SIGNATURES = {
    'sd_abi_version': (_i, []),
    'sd_error_string': (C.c_char_p, [_i]),
    'sd_linear_nchw_workspace_bytes': (_sz, [_i, C.c_long, _i, _i]),
    'sd_linear_nchw_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_long, _i, _i, _vp]),
    'sd_linear_nchw_bwd_data': (_i, [_vp, _vp, _vp, _i, _i, C.c_long, _i, _i, _vp]),
    'sd_linear_nchw_bwd_weight': (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_long, _i, _i, _vp, _sz, _vp]),
    'sd_adamw_chunk': (_i, []),
    'sd_adamw_max_groups': (_i, []),
    'sd_adamw_multi': (_i, [_vp, _vp, _i, C.POINTER(C.c_float), _i, C.c_double, C.c_double, _f, _i, _vp]),
    'sd_set_tunable': (_i, [C.c_char_p, _i]),
    'sd_get_tunable': (_i, [C.c_char_p]),
    'sd_cgd_kl_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'sd_cgd_kl_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sd_cgd_kl_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    'sd_head_tail_supported': (_i, [_i, _i, _i, _i]),
    'sd_head_tail_f32': (_i, [_vp] * 10 + [_i] * 5 + [_vp]),
    'sd_mixffn_tail_supported': (_i, [_i, _i, _i, _i]),
    'sd_mixffn_tail': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'sd_affine_act_nchw': (_i, [_vp, _vp, _vp, _vp, _vp, C.c_long, _i, C.c_long, _i, _vp]),
    'sd_im2col_tokens': (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.c_long, C.c_long, C.c_long, C.c_long, _i, _i, _i, _i, _i, _i, _vp]),
    'sd_col2im_tokens': (_i, [_vp, _vp, _i] + [_i] * 10 + [_vp]),
    'sd_linear_wgrad_tn_slabs': (_i, [C.c_long, _i, _i]),
    'sd_linear_wgrad_tn': (_i, [_vp, _vp, _vp, _sz, C.c_long, _i, _i, _i, _vp]),
    'sd_ppm_pool_supported': (_i, [_i, _i, _vp, _i]),
    'sd_ppm_pool_fwd': (_i, [_vp, _i, C.c_long, _i, _i, _vp, _i, _vp, _vp]),
    'sd_ppm_pool_bwd': (_i, [_vp, _i, C.c_long, _i, _i, _vp, _i, _vp, _vp]),
    'sd_layernorm_map_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, C.c_long, _i, _i, _i, _i, _f, _vp]),
    'sd_window_attn_supported': (_i, [_i, _i]),
    'sd_window_attn_packed_floats': (_sz, []),
    'sd_window_attn_pack': (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    'sd_window_attn_fwd_packed': (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_long, _i, _i, _i, _i, _f, _vp]),
    'sd_cgd_kl_tok_workspace_bytes': (_sz, [_i, _i, C.c_long]),
    'sd_cgd_kl_tok_fwd': (_i, [_vp, _vp, _i, _i, _i, C.c_long, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sd_cgd_kl_tok_bwd': (_i, [_vp, _vp, _i, _i, _i, C.c_long, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    'sd_cgd_kl_tok_max_jobs': (_i, []),
    'sd_cgd_kl_tok_fwd_multi': (_i, [_vp, _i, _i, _vp]),
    'sd_cgd_kl_tok_bwd_multi': (_i, [_vp, _i, _i, _vp]),
    'sd_align_cgd_tok_supported': (_i, [_i, _i]),
    'sd_align_cgd_tok_tiles': (_i, [_i, C.c_long]),
    'sd_align_cgd_tok_workspace_bytes': (_sz, [_i, _i, C.c_long]),
    'sd_align_cgd_tok_fwd_multi': (_i, [_vp, _i, _vp]),
    'sd_align_cgd_tok_bwd_multi': (_i, [_vp, _i, _vp]),
    'sd_linear_tok_bf16_fwd': (_i, [_vp, _vp, _vp, _vp, C.c_long, _i, _i, _vp]),
    'sd_linear_tok_bf16_bwd_data': (_i, [_vp, _vp, _vp, C.c_long, _i, _i, _vp]),
    'sd_linear_bf16_fwd_supported': (_i, [C.c_long, _i, _i]),
    'sd_linear_wgrad_tn_multi_supported': (_i, [_i, C.c_long, _i, _i]),
    'sd_linear_wgrad_tn_multi_plan': (_i, [_vp, _i, _i]),
    'sd_linear_wgrad_tn_multi': (_i, [_vp, _i, _i, _vp]),
    'sd_linear_bf16_fwd': (_i, [_vp, _vp, _vp, _i, _vp, C.c_long, _i, _i, _vp]),
    'sd_cgd_kl_up_supported': (_i, [_i, _i, _i, _i]),
    'sd_cgd_kl_up_workspace_bytes': (_sz, [_i] * 7),
    'sd_cgd_kl_up_fwd': (_i, [_vp, _vp, _i] + [_i] * 7 + [_f, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sd_cgd_kl_up_bwd': (_i, [_vp, _vp, _i] + [_i] * 7 + [_f, _f, _vp, _vp, _vp, _vp, _vp]),
    'sd_cgd_kl_up_fwd2': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'sd_cgd_kl_up_bwd2': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _f, _f, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp]),
    'sd_pix_kl_workspace_bytes': (_sz, [_i] * 4),
    'sd_pix_kl_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _sz, _vp]),
    'sd_pix_kl_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    'sd_pix_kl_up_supported': (_i, [_i, _i, _i, _i]),
    'sd_pix_kl_up_workspace_bytes': (_sz, [_i, _i]),
    'sd_pix_kl_up_fwd': (_i, [_vp, _vp, _i] + [_i] * 6 + [_f, _f, _vp, _vp, _vp, _sz, _vp]),
    'sd_pix_kl_up_bwd': (_i, [_vp, _vp, _i] + [_i] * 6 + [_f, _f, _vp, _vp, _vp, _vp]),
    'sd_ifvd_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'sd_ifvd_stepmask_ints': (_sz, [_i, _i, _i]),
    'sd_ifvd_counts': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'sd_ifvd_class_means': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    'sd_ifvd_cos': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    'sd_ifvd_coef_sums': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    'sd_ifvd_bwd': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'sd_at_kl_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'sd_at_kl_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'sd_align1x1_workspace_bytes': (_sz, [_i] * 5),
    'sd_align1x1_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'sd_align1x1_bwd_data': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'sd_align1x1_bwd_weight': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'sd_linear_fwd': (_i, [_vp, _vp, C.c_long, _vp, _vp, _vp, _i, C.c_long, _i, _i, _i, _i, _vp]),
    'sd_linear_bwd_data': (_i, [_vp, _vp, C.c_long, _vp, _i, C.c_long, _i, _i, _i, _vp]),
    'sd_presplit_bytes': (_sz, [_i, _i]),
    'sd_presplit_multi': (_i, [_vp, _i, _vp]),
    'sd_presplit_rows_bytes': (_sz, [_i, _i]),
    'sd_linear_nchw_fwd_planes': (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_long, _i, _i, _vp]),
    'sd_linear_fwd_planes': (_i, [_vp, _vp, _vp, _vp, _vp, _i, C.c_long, _i, _i, _vp]),
    'sd_linear_bwd_data_planes': (_i, [_vp, _vp, _vp, _i, C.c_long, _i, _i, _vp]),
    'sd_linear_wgrad_splitk_slabs': (_i, [C.c_long, _i, _i]),
    'sd_linear_wgrad_splitk': (_i, [_vp, _vp, _vp, _sz, C.c_long, _i, _i, _vp]),
    'sd_linear_wgrad_workspace_bytes': (_sz, [C.c_long, _i, _i]),
    'sd_linear_wgrad_fuses_bias': (_i, [C.c_long, _i, _i]),
    'sd_linear_wgrad_fuses_bias_dtype': (_i, [_i, C.c_long, _i, _i]),
    'sd_linear_wgrad': (_i, [_vp, _vp, _vp, _vp, _i, C.c_long, _i, _i, _vp, _sz, _vp]),
    'sd_dwconv3x3_workspace_bytes': (_sz, [_i] * 5),
    'sd_dwconv3x3_wgrad_slabs': (_i, [_i] * 5),
    'sd_dwconv3x3_wgrad_multi': (_i, [_vp, _i, _i, _vp]),
    'sd_dwconv3x3_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sd_dwconv3x3_gelu_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sd_dwconv3x3_gelu_fwd_train': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sd_dwconv3x3_bwd_data': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sd_dwconv3x3_bwd_weight': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'sd_layernorm_supported': (_i, [_i]),
    'sd_layernorm_workspace_bytes': (_sz, [C.c_long, _i]),
    'sd_layernorm_fwd': (_i, [_vp] * 6 + [_i, C.c_long, _i, _f, _vp]),
    'sd_layernorm_bwd': (_i, [_vp] * 8 + [_i, C.c_long, _i, _vp, _sz, _vp]),
    'sd_sra_supported': (_i, [_i]),
    'sd_sra_workspace_bytes': (_sz, [_i] * 5),
    'sd_sra_fwd': (_i, [_vp] * 4 + [_i] * 6 + [_f, _vp]),
    'sd_sra_bwd': (_i, [_vp] * 7 + [_i] * 6 + [_f, _vp, _sz, _vp]),
    'sd_add_layernorm_fwd': (_i, [_vp] * 3 + [C.c_long] + [_vp] * 6 + [_i, C.c_long, _i, _f, _vp]),
    'sd_add_layernorm_bwd': (_i, [_vp] * 7 + [C.c_long] + [_vp] * 4 + [_i, C.c_long, _i, _vp, _sz, _vp]),
    'sd_layernorm_patch_supported': (_i, [_i, _i, _i]),
    'sd_add_layernorm_patch_fwd': (_i, [_vp, _vp, _vp, C.c_long] + [_vp] * 7 + [_i, C.c_long, _i, _f, _i, _i, _i, _vp]),
    'sd_add_layernorm_patch_bwd': (_i, [_vp] * 8 + [C.c_long] + [_vp] * 4 + [_i, C.c_long, _i, _i, _i, _i, _vp, _sz, _vp]),
    'sd_upsum_fwd': (_i, [_vp] * 6 + [_i] * 8 + [_vp]),
    'sd_upsum_affine_fwd': (_i, [_vp] * 7 + [_i] + [_vp] + [_i] * 8 + [_vp]),
    'sd_upsum_bwd': (_i, [_vp, _vp] + [_i] * 6 + [_vp]),
    'sd_upsum_bwd3_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'sd_upsum_bwd3': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'sd_multi_slab_reduce': (_i, [_vp, _i, _vp]),
    'sd_colsum_blocks': (_i, [C.c_long, _i]),
    'sd_multi_colsum_partials': (_i, [_vp, _i, _i, _vp]),
    'sd_colsum_partials': (_i, [_vp, _i, C.c_long, _i, _vp, _sz, _vp]),
    'sd_layernorm_bwd_blocks': (_i, [C.c_long, _i]),
    'sd_linear_wgrad_slabs': (_i, [_i, C.c_long, _i, _i]),
    'sd_linear_wgrad_partials': (_i, [_vp, _vp, _i, C.c_long, _i, _i, _i, _vp, _sz, _vp]),
    'sd_linear_wgrad_generic_slabs': (_i, [_i, C.c_long, _i, _i]),
    'sd_linear_wgrad_generic_partials': (_i, [_vp, _vp, _i, C.c_long, _i, _i, _vp, _sz, _vp]),
    'sd_bn_supported': (_i, [_i]),
    'sd_bn_workspace_bytes': (_sz, [C.c_long, _i]),
    'sd_bn_stats': (_i, [_vp, _i, C.c_long, _i, _f, _vp, _vp, _vp, _vp, _f, _vp, _sz, _vp]),
    'sd_bn_act_fwd': (_i, [_vp] * 6 + [C.c_long, _i, _vp, _i, C.c_long, _i, _vp]),
    'sd_bn_act_bwd_reduce': (_i, [_vp] * 7 + [C.c_long, _i, _vp, _vp, _i, C.c_long, _i, _vp, _sz, _vp]),
    'sd_bn_act_bwd_elemt': (_i, [_vp] * 7 + [C.c_long, _i, _vp, _vp, _f, _vp, _vp, _i, C.c_long, _i, _vp]),
    'sd_resize_bilinear_fwd': (_i, [_vp, _vp, _i, C.c_long, _i, _i, _i, _i, _i, _vp]),
    'sd_resize_bilinear_bwd_workspace_bytes': (_sz, [C.c_long, _i, _i]),
    'sd_resize_bilinear_bwd': (_i, [_vp, _vp, _i, C.c_long, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    'sd_ce_up_supported': (_i, [_i, _i, _i, _i]),
    'sd_ce_up_fwd': (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    'sd_ce_up_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp] + [_i] * 8 + [_vp]),
}


class SegDistillLibError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise SegDistillLibError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C segdistill_amd/csrc`). The distillation losses have no non-HIP implementation.')
    try:
        h = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise SegDistillLibError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise SegDistillLibError(f'{LIB_PATH} does not export {name}') from e
        fn.restype, fn.argtypes = res, args
    v = h.sd_abi_version()
    if v != ABI_VERSION:
        raise SegDistillLibError(f'ABI version mismatch: library {v}, binding {ABI_VERSION}')
    _lib = h
    if os.environ.get('SEGDISTILL_SPLIT_BF16', '1') == '0':   # A/B switch shared with linear.py: exact-f32 MFMA everywhere
        h.sd_set_tunable(b'sra_split_bf16', 0)
        h.sd_set_tunable(b'align_split_bf16', 0)
    if os.environ.get('SEGDISTILL_WGRAD_SLAB_RATIO'):   # A/B switch: slab bytes of the bf16 split-K weight gradients as a percentage of their operand bytes (0 = no cap)
        h.sd_set_tunable(b'wgrad_slab_ratio', int(os.environ['SEGDISTILL_WGRAD_SLAB_RATIO']))
    if os.environ.get('SEGDISTILL_WGRAD_MULTI_WGS'):       # A/B: workgroups per grouped weight-gradient launch
        h.sd_set_tunable(b'wgrad_multi_wgs', int(os.environ['SEGDISTILL_WGRAD_MULTI_WGS']))
    # A/B: any tunable of the header by name, SEGDISTILL_TUNABLES="wgrad_x3_ring=0,align_stream=0" (benchmarking only, like sd_set_tunable itself)
    for kv in filter(None, os.environ.get('SEGDISTILL_TUNABLES', '').split(',')):
        k, _, v = kv.partition('=')
        rc = h.sd_set_tunable(k.strip().encode(), int(v))
        if rc != 0:
            raise SegDistillLibError(f'SEGDISTILL_TUNABLES: sd_set_tunable({k!r}, {v}) -> {rc}')
    return h


def check(code: int, what: str):
    if code != 0:
        msg = lib().sd_error_string(code).decode()
        raise RuntimeError(f'{what} failed: {msg} (code {code})')


def set_tunable(key: str, value: int):
    check(lib().sd_set_tunable(key.encode(), int(value)), f'sd_set_tunable({key})')


def get_tunable(key: str) -> int:
    return lib().sd_get_tunable(key.encode())

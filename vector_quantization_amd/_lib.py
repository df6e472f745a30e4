"""ctypes binding of libvqhip.so (C ABI in include/vqhip.h).

The library is built in-tree by ``vector_quantization_amd/csrc/build.sh`` (hipcc, gfx950).  There is no
CPU or PyTorch fallback: if the shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VQHIP_LIB') or os.path.join(_HERE, 'libvqhip.so')   # VQHIP_LIB: experiment builds

ABI_VERSION = 600          # VQHIP_VERSION of include/vqhip.h this binding was written against
METRIC_L2, METRIC_COS, METRIC_COS_BF16 = 0, 1, 5
DTYPE_F32, DTYPE_BF16 = 0, 1

_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float


class CvqForwardArgs(ctypes.Structure):
    """vqhip_cvq_forward_t of include/vqhip.h, field for field."""
    _fields_ = [('struct_bytes', _i64), ('N', _i64), ('K', _i64),
                ('D', _i32), ('x_dtype', _i32), ('metric', _i32), ('world', _i32),
                ('ema_decay', _f32), ('eps', _f32), ('beta', _f32),
                ('phases', _i32), ('exchange', _i32), ('list_ready', _i32), ('prefetch', _i32), ('anchor_sync', _i32), ('rank', _i32),
                ('cap', _i64),
                ('x', _vp), ('w_in', _vp), ('p_in', _vp), ('w_out', _vp), ('p_out', _vp),
                ('rows', _vp), ('slot', _vp), ('count', _vp), ('count_host', _vp), ('count_event', _vp), ('comm', _vp),
                ('cb', _vp), ('cb_bytes', _i64), ('idx', _vp), ('hist', _vp), ('xq', _vp),
                ('packed', _vp), ('packed_floats', _i64),
                ('z_ste', _vp), ('mse', _vp), ('scratch16', _vp),
                ('ws', _vp), ('ws_bytes', _i64),
                ('cap_used', _i64), ('exchange_floats', _i64),
                ('early_word_host', _vp), ('early_seq_dev', _vp), ('keys', _vp)]


class VqkdForwardArgs(ctypes.Structure):
    """vqhip_vqkd_forward_t of include/vqhip.h, field for field."""
    _fields_ = [('struct_bytes', _i64), ('N', _i64), ('K', _i64),
                ('D', _i32), ('x_dtype', _i32), ('metric', _i32), ('world', _i32),
                ('ema_decay', _f32),
                ('phases', _i32), ('exchange', _i32), ('ordered', _i32), ('tail', _i32), ('reserved0', _i32),
                ('x', _vp), ('w_in', _vp), ('w_mid', _vp), ('w_out', _vp), ('xn', _vp), ('xq', _vp), ('comm', _vp),
                ('cb', _vp), ('cb_bytes', _i64), ('idx', _vp), ('hist', _vp),
                ('packed', _vp), ('packed_floats', _i64),
                ('z_ste', _vp), ('mse', _vp), ('scratch16', _vp),
                ('ws', _vp), ('ws_bytes', _i64),
                ('exchange_floats', _i64)]


class VqForwardArgs(ctypes.Structure):
    """vqhip_vq_forward_t of include/vqhip.h, field for field."""
    _fields_ = [('struct_bytes', _i64), ('N', _i64), ('K', _i64),
                ('D', _i32), ('x_dtype', _i32), ('metric', _i32), ('normalize', _i32),
                ('beta', _f32), ('reserved0', _i32),
                ('x', _vp), ('w_in', _vp), ('w_out', _vp), ('xn', _vp),
                ('cb', _vp), ('cb_bytes', _i64), ('idx', _vp), ('hist', _vp), ('xq', _vp),
                ('z_ste', _vp), ('mse', _vp), ('scratch16', _vp),
                ('ws', _vp), ('ws_bytes', _i64)]


STEP_BEFORE_EXCHANGE, STEP_AFTER_EXCHANGE, STEP_ALL, STEP_PACK_SYNC = 1, 2, 3, 4

# name -> (restype, argtypes); mirrors include/vqhip.h one to one
SIGNATURES = {
    'vqhip_version': (_i32, []),
    'vqhip_last_error': (ctypes.c_char_p, []),
    'vqhip_codebook_bytes': (_i64, [_i64, _i32]),
    'vqhip_codebook_exact_offset': (_i64, [_i64, _i32]),
    'vqhip_encode': (_i32, [_vp, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    'vqhip_encode_ex': (_i32, [_vp, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    'vqhip_encode_map': (_i32, [_vp, _i32, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    'vqhip_gather_ste_map': (_i32, [_vp, _i32, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _f32, _vp, _vp]),
    'vqhip_workspace_bytes': (_i64, [_i64, _i64, _i32]),
    'vqhip_col_workspace_bytes': (_i64, [_i64, _i64, _i32]),
    'vqhip_codebook_prepare': (_i32, [_vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    'vqhip_argmin': (_i32, [_vp, _i32, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _vp]),
    'vqhip_argmin_exact': (_i32, [_vp, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    'vqhip_distance': (_i32, [_vp, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    'vqhip_col_argmin': (_i32, [_vp, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    'vqhip_row_sqnorm': (_i32, [_vp, _i32, _i64, _i32, _vp, _vp]),
    'vqhip_normalize_rows': (_i32, [_vp, _i32, _i64, _i32, _f32, _vp, _vp]),
    'vqhip_gather_ste_loss': (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    'vqhip_gather_ste_mse': (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _f32, _vp, _vp]),
    'vqhip_hist': (_i32, [_vp, _i64, _i64, _vp, _vp]),
    'vqhip_scatter_add_rows': (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    'vqhip_vqkd_update': (_i32, [_vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    'vqhip_cvq_update': (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _f32, _f32, _i32, _vp]),
    'vqhip_gather_rows': (_i32, [_vp, _i32, _vp, _i64, _i32, _vp, _vp]),
    'vqhip_diff': (_i32, [_vp, _i32, _vp, _i32, _i64, _f32, _vp, _vp, _vp, _vp]),
    'vqhip_vq_backward': (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'vqhip_vq_backward_ex': (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp]),
    'vqhip_vq_backward_map': (_i32, [_vp, _i32, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _f32, _vp, _i32, _vp]),
    'vqhip_ste': (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    'vqhip_normalize_rows_bwd': (_i32, [_vp, _i32, _vp, _i64, _i32, _f32, _vp, _vp]),
    'vqhip_transpose': (_i32, [_vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    'vqhip_codebook_metrics': (_i32, [_vp, _i64, _vp, _vp]),
    'vqhip_argmin_stats': (_i32, [_vp, _vp, _vp]),
    'vqhip_cvq_step': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _i64, _i32, _f32, _f32, _vp]),
    'vqhip_cvq_decay': (_i32, [_vp, _i64, _f32, _f32, _vp, _vp]),
    'vqhip_cvq_update_rows': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _f32, _f32, _vp]),
    'vqhip_pack_floats': (_i64, [_i64, _i64, _i32]),
    'vqhip_pack_counts': (_i32, [_vp, _i32, _i64, _i64, _vp, _vp]),
    'vqhip_unpack_counts': (_i32, [_vp, _i64, _vp, _vp]),
    'vqhip_cvq_rows': (_i32, [_vp, _i64, _f32, _f32, _vp, _vp, _vp, _vp]),
    'vqhip_col_rows_workspace_bytes': (_i64, [_i64, _i64, _i32]),
    'vqhip_col_argmin_rows': (_i32, [_vp, _i32, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    'vqhip_cvq_pack': (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    'vqhip_cvq_forward_ws_bytes': (_i64, [_i64, _i64, _i32, _i64]),
    'vqhip_cvq_forward': (_i32, [ctypes.POINTER(CvqForwardArgs), _vp]),
    'vqhip_vqkd_forward_ws_bytes': (_i64, [_i64, _i64, _i32]),
    'vqhip_vqkd_forward': (_i32, [ctypes.POINTER(VqkdForwardArgs), _vp]),
    'vqhip_vq_forward': (_i32, [ctypes.POINTER(VqForwardArgs), _vp]),
    'vqhip_vqkd_backward': (_i32, [_vp, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    'vqhip_cvq_apply': (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _f32, _f32, _vp]),
    'vqhip_cvq_col_keys': (_i32, [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    'vqhip_cvq_pack_sync': (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _i64, _i32, _i64, _i32, _vp, _vp]),
    'vqhip_allreduce_min_i64': (_i32, [_vp, _i64, _vp, _vp]),
    'vqhip_order_workspace_bytes': (_i64, [_i64, _i64]),
    'vqhip_token_order': (_i32, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    'vqhip_segsum_workspace_bytes': (_i64, [_i64, _i32]),
    'vqhip_segsum_rows': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _vp]),
    'vqhip_vq_backward_w_ordered': (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    'vqhip_debug_proposal_scores': (_i32, [_vp, _i32, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    'vqhip_rccl_load': (_i32, [ctypes.c_char_p]),
    'vqhip_rccl_unique_id': (_i32, [_vp]),
    'vqhip_rccl_comm_init': (_i32, [ctypes.POINTER(_vp), _i32, _vp, _i32]),
    'vqhip_rccl_comm_destroy': (_i32, [_vp]),
    'vqhip_allreduce_packed': (_i32, [_vp, _i64, _vp, _vp]),
    'vqhip_profile_enable': (_i32, [_i32]),
    'vqhip_set_tuning': (_i32, [_i32, _i32]),
    'vqhip_profile_collect': (_i32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
}

_lib = None


class VqhipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libvqhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    script = os.path.join(_HERE, 'csrc', 'build.sh')
    res = subprocess.run(['bash', script], capture_output=True, text=True)
    if res.returncode != 0:
        raise VqhipError(f'hipcc build of libvqhip.so failed:\n{res.stdout}\n{res.stderr}')
    if verbose:
        print(res.stdout.strip())
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load libvqhip.so and declare every entry point of include/vqhip.h; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VqhipError(
                f'{LIB_PATH} not found: build it with vector_quantization_amd/csrc/build.sh '
                '(or __graft_entry__.build()); there is no fallback path')
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)            # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.vqhip_version() != ABI_VERSION:
            raise VqhipError(f'{LIB_PATH} has ABI version {L.vqhip_version()}, this package binds {ABI_VERSION}: '
                             'rebuild it with vector_quantization_amd/csrc/build.sh')
        # VQHIP_TUNING="18=0,2=4": A/B knobs of vqhip_set_tuning applied at load (measurement and debugging; results never change)
        for kv in filter(None, os.environ.get('VQHIP_TUNING', '').split(',')):
            key, _, value = kv.partition('=')
            if L.vqhip_set_tuning(int(key), int(value)) != 0:
                raise VqhipError(f'VQHIP_TUNING: vqhip_set_tuning({key}, {value}) refused')
        _lib = L
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().vqhip_last_error()
        raise VqhipError(f'{what} failed with code {rc}: {msg.decode() if msg else ""}')

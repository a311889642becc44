"""ctypes binding of libvec2wav_hip.so (the C ABI of include/vec2wav_hip.h).

There is no fallback: if the library is missing, or an entry point reports an error, the caller
gets an exception.  ``torch`` must be imported before the library is loaded so that both share the
process's one HIP runtime (same ``libamdhip64.so.7``), which makes torch's stream handles and
device pointers directly usable by the kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import build as _build

ABI_VERSION = 34
V2W_MAX_STAGES = 8
V2W_BN_SPLITS = 64
ALGO_AUTO, ALGO_DIRECT, ALGO_MFMA, ALGO_SPLIT, ALGO_BF16 = 0, 1, 2, 3, 4

_fp = C.c_void_p  # device pointers travel as integers


class Conv1dArgs(C.Structure):
    _fields_ = [('in_', _fp), ('in_a', _fp), ('in_s', _fp), ('wf', _fp), ('wp', _fp), ('bias', _fp),
                ('res', _fp), ('res_a', _fp), ('res_s', _fp), ('add0', _fp), ('add1', _fp),
                ('mask_src', _fp), ('mask_a', _fp), ('mask_s', _fp), ('out', _fp),
                ('B', C.c_int32), ('C_in', C.c_int32), ('C_out', C.c_int32), ('L', C.c_int32),
                ('k', C.c_int32), ('dil', C.c_int32), ('slope', C.c_float), ('accumulate', C.c_int32),
                ('out_div', C.c_float), ('algo', C.c_int32), ('mask_slope', C.c_float),
                ('in_stride', C.c_int32), ('in_phase', C.c_int32), ('pad_left', C.c_int32),
                ('wps', _fp), ('winv', _fp), ('in_ct', C.c_int32), ('out_ct', C.c_int32), ('out_slope', C.c_float),
                ('io_bf16', C.c_int32), ('rowsum_part', _fp), ('splitk_ws', _fp), ('splitk_ws_bytes', C.c_int64)]


class ConvT1dArgs(C.Structure):
    _fields_ = [('in_', _fp), ('wf', _fp), ('wp', _fp), ('bias', _fp), ('out', _fp), ('stats_part', _fp),
                ('B', C.c_int32), ('C_in', C.c_int32), ('C_out', C.c_int32), ('L', C.c_int32),
                ('k', C.c_int32), ('u', C.c_int32), ('slope', C.c_float), ('algo', C.c_int32), ('io_bf16', C.c_int32), ('_pad', C.c_int32),
                ('splitk_ws', _fp), ('splitk_ws_bytes', C.c_int64)]


class PairArgs(C.Structure):
    _fields_ = [('in_', _fp), ('in_a', _fp), ('in_s', _fp), ('wp1', _fp), ('bias1', _fp), ('wp2', _fp), ('bias2', _fp),
                ('add0', _fp), ('add1', _fp), ('out', _fp),
                ('B', C.c_int32), ('C', C.c_int32), ('L', C.c_int32), ('k', C.c_int32), ('dil1', C.c_int32), ('dil2', C.c_int32),
                ('res_mode', C.c_int32), ('slope', C.c_float), ('out_div', C.c_float)]


class StageArgs(C.Structure):
    _fields_ = [('in_', _fp), ('in_a', _fp), ('in_s', _fp),
                ('wp1', _fp * 4), ('bias1', _fp * 4), ('wp2', _fp * 4), ('bias2', _fp * 4),
                ('k', C.c_int32 * 4), ('dil1', C.c_int32 * 4), ('dil2', C.c_int32 * 4),
                ('out', _fp), ('nk', C.c_int32), ('B', C.c_int32), ('C', C.c_int32), ('L', C.c_int32),
                ('slope', C.c_float), ('out_div', C.c_float),
                ('post_w', _fp), ('post_b', _fp), ('post_out', _fp), ('post_k', C.c_int32), ('post_slope', C.c_float),
                ('bwd_mask1', _fp * 4), ('bwd_mid', _fp * 4), ('bwd_mask2', _fp), ('bwd_mask2_a', _fp), ('bwd_mask2_s', _fp),
                ('bwd_slope', C.c_float), ('bwd_rowsum', _fp * 4)]


class FoldDesc(C.Structure):
    _fields_ = [('v', _fp), ('g', _fp), ('wp', _fp), ('scale', _fp),
                ('c_in', C.c_int32), ('c_out', C.c_int32), ('k', C.c_int32), ('u', C.c_int32), ('transposed', C.c_int32),
                ('mf', C.c_int32), ('ck', C.c_int32), ('_pad', C.c_int32), ('wf', _fp), ('wpd', _fp)]


_P4 = _fp * 4
_I4 = C.c_int32 * 4


class StageSplitArgs(C.Structure):
    _fields_ = [('in_', _fp), ('in_a', _fp), ('in_s', _fp),
                ('wps1', _P4), ('sc1', _P4), ('bias1', _P4), ('wps2', _P4), ('sc2', _P4), ('bias2', _P4),
                ('k', _I4), ('dil1', _I4), ('dil2', _I4), ('out', _fp),
                ('nk', C.c_int32), ('B', C.c_int32), ('C', C.c_int32), ('L', C.c_int32),
                ('slope', C.c_float), ('out_div', C.c_float), ('bf16', C.c_int32), ('io_bf16', C.c_int32),
                ('post_w', _fp), ('post_b', _fp), ('post_out', _fp), ('post_k', C.c_int32), ('post_slope', C.c_float),
                ('up_wps', _fp), ('up_bias', _fp), ('up_out', _fp), ('up_stats_part', _fp),
                ('up_k', C.c_int32), ('up_u', C.c_int32), ('up_slope', C.c_float), ('rb1', C.c_int32),
                ('in_b', _P4), ('out_b', _P4), ('add0', _fp), ('add1', _fp)]


class BranchConvsArgs(C.Structure):
    _fields_ = [('in_', _fp * 3), ('in_a', _fp), ('in_s', _fp), ('wps', _fp * 3), ('bias', _fp * 3), ('out', _fp * 3),
                ('k', C.c_int32 * 3), ('dil', C.c_int32 * 3),
                ('nbr', C.c_int32), ('mode', C.c_int32), ('B', C.c_int32), ('C', C.c_int32), ('L', C.c_int32),
                ('slope', C.c_float), ('out_div', C.c_float), ('_pad', C.c_int32)]


class SplitDesc(C.Structure):
    _fields_ = [('v', _fp), ('g', _fp), ('wps', _fp), ('sc', _fp), ('rowscale', _fp),
                ('c_in', C.c_int32), ('c_out', C.c_int32), ('k', C.c_int32), ('mode', C.c_int32)]


_PA = _fp * V2W_MAX_STAGES


class CondArgs(C.Structure):
    _fields_ = [('spk', _fp), ('noise', _fp),
                ('fc_w', _PA), ('fc_b', _PA), ('sn_w', _PA), ('sn_b', _PA), ('sn_u', _PA), ('sn_v', _PA),
                ('gb', _PA), ('C', C.c_int32 * V2W_MAX_STAGES), ('z_ws', _fp), ('sigma_ws', _fp),
                ('n_stages', C.c_int32), ('B', C.c_int32), ('spk_dim', C.c_int32), ('noise_dim', C.c_int32),
                ('training', C.c_int32)]


class CondEvalArgs(C.Structure):
    _fields_ = [('c', CondArgs), ('running_mean', _PA), ('running_var', _PA), ('a_out', _PA), ('s_out', _PA),
                ('eps', C.c_float * V2W_MAX_STAGES)]


# name -> (restype, argtypes); must list every symbol include/vec2wav_hip.h declares
SIGNATURES = {
    'v2w_abi_version': (C.c_int, []),
    'v2w_build_arch': (C.c_char_p, []),
    'v2w_wn_fold_conv': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_wn_fold_convt': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_wf_transpose_flip': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_wf_gather_transpose': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_pack_mfma': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_pack_mfma_batch': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_pack_mfma_dgrad': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_split_supported': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'v2w_pack_split': (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_mel_phases': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_mel_finish': (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_mel_finish_bwd': (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_phase_split': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_unfold1': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_unfold_taps': (C.c_int, [_fp, _fp] + [C.c_int] * 9 + [_fp]),
    'v2w_wgrad_slice': (C.c_int, [_fp, _fp, _fp, _fp] + [C.c_int] * 9 + [_fp]),
    'v2w_wgrad_group_slabs': (C.c_int, [C.c_int] * 5),
    'v2w_wgrad_groups': (C.c_int, [_fp, _fp, _fp, _fp] + [C.c_int] * 8 + [_fp]),
    'v2w_disc_dz': (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_longlong, C.c_int, C.c_int, C.c_float, _fp]),
    'v2w_rowsum_reduce': (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    'v2w_disc_dz_merge': (C.c_int, [_fp, _fp, _fp, _fp, _fp] + [C.c_int] * 8 + [C.c_float, _fp]),
    'v2w_phase_merge': (C.c_int, [_fp, _fp] + [C.c_int] * 8 + [_fp]),
    'v2w_fold1': (C.c_int, [_fp, _fp] + [C.c_int] * 9 + [_fp]),
    'v2w_avgpool4_bwd': (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    'v2w_cout1_wgrad': (C.c_int, [_fp, _fp, _fp] + [C.c_int] * 6 + [_fp]),
    'v2w_zero_tail': (C.c_int, [_fp, C.c_longlong, C.c_int, C.c_int, _fp]),
    'v2w_avgpool4': (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    'v2w_mel_phases_bwd': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_split_packable': (C.c_int, [C.c_int, C.c_int]),
    'v2w_resblock2_stage_split_fwd': (C.c_int, [C.POINTER(StageSplitArgs), _fp]),
    'v2w_resblock2_stage_split_config': (C.c_int, [C.POINTER(StageSplitArgs)]),
    'v2w_resblock2_stage_up_tiles': (C.c_int, [C.POINTER(StageSplitArgs)]),
    'v2w_pack_bf16': (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_split_pack_batch': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_fold_plan': (C.c_int, [C.POINTER(FoldDesc), C.c_int, C.POINTER(C.c_int32)]),
    'v2w_fold_pack_batch': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_conv1d_fwd': (C.c_int, [C.POINTER(Conv1dArgs), _fp]),
    'v2w_conv1d_fwd_multi': (C.c_int, [C.POINTER(Conv1dArgs), C.c_int, _fp]),
    'v2w_conv1d_splitk_ws_bytes': (C.c_longlong, [C.POINTER(Conv1dArgs), C.c_int]),
    'v2w_convt1d_splitk_ws_bytes': (C.c_longlong, [C.POINTER(ConvT1dArgs)]),
    'v2w_resblock_pair_fwd': (C.c_int, [C.POINTER(PairArgs), C.c_int, _fp]),
    'v2w_resblock2_stage_bwd_rows': (C.c_int, [C.POINTER(StageArgs)]),
    'v2w_resblock2_stage_fwd': (C.c_int, [C.POINTER(StageArgs), _fp]),
    'v2w_resblock2_stage_small_fwd': (C.c_int, [C.POINTER(StageArgs), _fp]),
    'v2w_branch_convs_bf16_fwd': (C.c_int, [C.POINTER(BranchConvsArgs), _fp]),
    'v2w_convt1d_fwd': (C.c_int, [C.POINTER(ConvT1dArgs), _fp]),
    'v2w_pack_bf16_convt': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_pack_bf16_convt_bytes': (C.c_longlong, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'v2w_convt1d_bf16_fwd': (C.c_int, [C.POINTER(ConvT1dArgs), _fp]),
    'v2w_convt1d_bf16_tiles': (C.c_int, [C.POINTER(ConvT1dArgs)]),
    'v2w_conv1d_tile_config': (C.c_int, [C.POINTER(Conv1dArgs), C.POINTER(C.c_int32)]),
    'v2w_convt1d_tile_config': (C.c_int, [C.POINTER(ConvT1dArgs), C.POINTER(C.c_int32)]),
    'v2w_conv1d_bf16_config': (C.c_int, [C.POINTER(Conv1dArgs), C.c_int, C.POINTER(C.c_int32)]),
    'v2w_convt1d_bf16_config': (C.c_int, [C.POINTER(ConvT1dArgs), C.POINTER(C.c_int32)]),
    'v2w_cond_gamma_beta': (C.c_int, [C.POINTER(CondArgs), _fp]),
    'v2w_cond_sigma': (C.c_int, [C.POINTER(CondArgs), _fp]),
    'v2w_cond_affine_eval': (C.c_int, [C.POINTER(CondEvalArgs), _fp]),
    'v2w_bn_stats': (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_bn_reduce_partials': (C.c_int, [_fp, C.c_int, C.c_int, C.c_double, _fp, _fp]),
    'v2w_bn_finalize': (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int,
                                  C.c_float, C.c_float, _fp]),
    'v2w_bn_reduce_finalize': (C.c_int, [_fp, C.c_int, C.c_double, _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_float, C.c_float, _fp]),
    'v2w_bn_reduce_slices': (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, _fp]),
    'v2w_bn_finalize_slices': (C.c_int, [_fp, C.c_int, C.c_double, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_float, C.c_float, _fp]),
    'v2w_affine_apply': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_wgrad_slabs': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'v2w_wgrad': (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_float, _fp]),
    'v2w_wgrad_bf16_slabs': (C.c_int, [C.c_int] * 5),
    'v2w_wgrad_bf16': (C.c_int, [_fp] * 6 + [C.c_int] * 6 + [C.c_float, C.c_int, _fp]),
    'v2w_cbn_bwd_sums': (C.c_int, [_fp] * 9 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    'v2w_cbn_bwd_apply': (C.c_int, [_fp] * 9 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    'v2w_tail_bwd': (C.c_int, [_fp] * 8 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    'v2w_wn_bwd': (C.c_int, [_fp] * 5 + [C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_cond_bwd': (C.c_int, [_fp] * 13 + [C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    'v2w_conv_post_tanh_bf16in': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
    'v2w_conv_post_tanh': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp]),
}

# entry points that LAUNCH (their last argument is the stream); the others are host-only queries.  schedule.Recorder tapes the former.
LAUNCHERS = frozenset(n for n, (_r, a) in SIGNATURES.items() if a and a[-1] is _fp)

_lib = None
_tls = threading.local()      # .recorder: the schedule.Recorder of THIS thread while it records a forward (replicas driven from several threads
                              # - nn.DataParallel - record independently)


class HipLibraryError(RuntimeError):
    """A C-ABI call declined or failed; `code` is its return value (V2W_E_* < 0, hipError_t > 0)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


E_ARG, E_SHAPE, E_ALGO = -1, -2, -3


def lib_path() -> str:
    # V2W_LIB: load another build of the same ABI (kernel experiments); the default is the in-tree library
    return os.environ.get('V2W_LIB') or _build.LIB_PATH


def load():
    """Load (once) and return the library handle; raises HipLibraryError when it is absent or stale."""
    global _lib
    rec = getattr(_tls, 'recorder', None)
    if rec is not None:
        return rec
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads libamdhip64.so.7 first - see module docstring)
    path = lib_path()
    if not os.path.exists(path):
        raise HipLibraryError(
            f'{path} not found: build it with `python -m wavthruvec_pytorch_amd.build` '
            '(or __graft_entry__.build()); there is no CPU/PyTorch fallback for the Vec2Wav hot path')
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f'{path} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    if lib.v2w_abi_version() != ABI_VERSION:
        raise HipLibraryError(f'ABI version mismatch: library {lib.v2w_abi_version()}, binding {ABI_VERSION}')
    _lib = lib
    return lib


def set_recorder(rec):
    """Install (or, with None, remove) the recorder `load()` hands out instead of the library; returns the previous one."""
    prev = getattr(_tls, 'recorder', None)
    _tls.recorder = rec
    return prev


class NameSink(C.Structure):
    """include/vec2wav_hip.h `v2w_name_sink` (ABI v33): handed to a launching entry point in place of the stream, it collects the names of
    the kernels that call would launch."""
    _fields_ = [('magic', C.c_uint64), ('buf', C.c_void_p), ('cap', C.c_int32), ('len', C.c_int32)]


NAME_SINK_MAGIC = 0x5632574e414d4531


def kernel_name_short(full: str) -> str:
    """'void (anonymous namespace)::conv_bf16_kernel<2, 4, ...>((anonymous namespace)::MultiArgs)' -> 'conv_bf16_kernel<2, 4, ...>': the key
    the profile summaries (tools/pmc_traffic.py, tools/pmc_sq.py) and bench.py's tables use."""
    s = full.strip()
    if s.startswith('void '):
        s = s[5:]
    s = s.replace('(anonymous namespace)::', '')
    depth = 0
    for i, ch in enumerate(s):          # cut the parameter list: the first '(' outside the template argument list
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return s[:i]
    return s


def kernel_names(fn, *args, short=True):
    """(return code, names of the kernels `fn(*args, stream)` would launch, in order) - `fn` a launching entry point of the library, `args` its
    arguments WITHOUT the stream.  Host-only: nothing is launched, no tensor pointer is dereferenced (v2w_name_sink)."""
    buf = C.create_string_buffer(8192)
    sink = NameSink(NAME_SINK_MAGIC, C.addressof(buf), len(buf), 0)
    rc = fn(*args, C.c_void_p(C.addressof(sink) | 1))
    names = [n for n in buf.value.decode().split('\n') if n]
    return rc, ([kernel_name_short(n) for n in names] if short else names)


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        msg = {-1: 'bad argument', -2: 'unsupported shape for the requested algorithm', -3: 'unknown algorithm'}.get(rc, '?')
        raise HipLibraryError(f'{what}: V2W error {rc} ({msg})', rc)
    raise HipLibraryError(f'{what}: hipError_t {rc}', rc)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def on_tensor_device(fn):
    """Decorator: run `fn` with the device of its first GPU tensor argument made current.

    The C ABI launches on the stream it is handed and never calls hipSetDevice, so kernel lookup, hipFuncSetAttribute and the
    workspace allocations of the wrappers all resolve against the CURRENT device.  The reference drives `cuda:{rank}` without
    torch.cuda.set_device (train.py:63-94), i.e. the current device stays 0 on every rank: every public entry point of the
    package is therefore wrapped so that streams, buffers and launches agree on the tensors' device."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        import torch
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                with torch.cuda.device(a.device):
                    return fn(*args, **kw)
        return fn(*args, **kw)
    return wrapper


def current_stream_handle(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream

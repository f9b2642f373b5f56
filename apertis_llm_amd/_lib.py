"""ctypes binding of libapertis_hip.so (the C ABI declared in include/apertis_hip.h).

The library is loaded lazily (safe after fork(); DataLoader workers never touch HIP) and there
is NO fallback: if the .so is missing or a call returns an error code, ApertisHipError is
raised.  Nothing in this package routes compute through PyTorch eager ops or the CPU oracle.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# APERTIS_HIP_LIB: developer switch - another build of the same library (tools/ A/B runs); never a fallback
LIB_PATH = os.environ.get("APERTIS_HIP_LIB") or os.path.join(_HERE, "libapertis_hip.so")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SILU = 0, 1, 2, 3
ACT_SAVE_GRAD, ACT_MUL_SAVED = 0x100, 0x200      # flags of apertis_grouped_gemm_nt's `act` (apertis_hip.h)
ACT_INTERLEAVED = 0x400                          # ... with ACT_SAVE_GRAD: the interleaved-epilogue kernel (opt-in, round 6)


class ApertisHipError(RuntimeError):
    pass


_lock = threading.Lock()
_lib = None

_vp, _i64, _i32, _f32, _u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_uint64
_f64 = ctypes.c_double

# name -> (restype, argtypes).  Order and meaning mirror include/apertis_hip.h exactly.
ABI_VERSION = (4 << 16) | 7      # = APERTIS_ABI_VERSION of include/apertis_hip.h (tests/test_host_cpu.py compares them)
SIGNATURES = {
    "apertis_abi_version": (ctypes.c_int, []),
    "apertis_arch": (ctypes.c_char_p, []),
    "apertis_strerror": (ctypes.c_char_p, [_i32]),
    "apertis_scan_chunk_len": (_i64, [_i64, _i64, _i64]),
    "apertis_scan_num_chunks": (_i64, [_i64, _i64, _i64]),
    "apertis_selective_scan_fwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp,
                                          _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp]),
    "apertis_selective_scan_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64,
                                          _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp]),
    "apertis_ssm_gate_fwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp]),
    "apertis_ssm_gate_bwd": (_i32, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64,
                                    _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "apertis_ssm_gate_bwd_blocks": (_i64, [_i64, _i64]),
    "apertis_scan_gate_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "apertis_scan_gate_chunk_len": (_i64, []),
    "apertis_scan_gate_fwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp,
                                     _vp, ctypes.c_uint32, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _vp]),
    "apertis_scan_gate_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp,
                                     _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint32, _i64, _i64,
                                     _i64, _i64, _i32, _i32, _i32, _vp]),
    "apertis_scan_lean_fwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                     _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_scan_lean_fwd_dt": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp,
                                        _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_scan_lean_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64,
                                     _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_scan_lookback_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "apertis_scan_lookback_fwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                         ctypes.c_uint32, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_scan_lookback_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp,
                                         _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint32, _i64, _i64, _i64,
                                         _i64, _i32, _vp]),
    "apertis_ssm_decode_conv": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "apertis_ssm_decode_state": (_i32, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i32,
                                        _i32, _vp]),
    "apertis_ssm_decode_state_dt": (_i32, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64,
                                           _i64, _i64, _i32, _i32, _vp]),
    "apertis_dropout_add_fwd": (_i32, [_vp, _vp, _vp, _i64, _f32, _u64, _i32, _i32, _vp]),
    "apertis_dropout_bwd": (_i32, [_vp, _vp, _i64, _f32, _u64, _i32, _i32, _vp]),
    "apertis_dwconv_silu_fwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_dwconv_silu_bwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                       _i64, _i32, _vp]),
    "apertis_dwconv_silu_bwd2": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                        _i64, _i32, _vp]),
    "apertis_dwconv_bwd_blocks": (_i64, [_i64, _i64, _i64]),
    "apertis_moe_gate_topk_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "apertis_decode_pre_conv": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_decode_pre_state": (_i32, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
                                        _i32, _i32, _vp]),
    "apertis_decode_dense_gemv": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp]),
    "apertis_decode_post": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "apertis_decode_inproj": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _i64,
                                     _i64, _vp]),
    "apertis_moe_enter_small": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp]),
    "apertis_moe_route_small": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i64,
                                       _i32, _i32, _vp]),
    "apertis_moe_gate_topk_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "apertis_skinny_linear_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "apertis_skinny_linear_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "apertis_skinny_linear_bwd_blocks": (_i64, [_i64]),
    "apertis_moe_plan_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "apertis_moe_plan": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    "apertis_moe_gather_ln_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp]),
    "apertis_moe_gather_ln_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                                         _i32, _i32, _vp]),
    "apertis_moe_gather_ln_bwd_blocks": (_i64, [_i64]),
    "apertis_layernorm_fwd": (_i32, [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "apertis_layernorm_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _u64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "apertis_layernorm_combine_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _i64, _i64, _i64, _i32, _i32, _vp]),
    "apertis_dropout_add_layernorm_fwd": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _i64, _i64, _f32,
                                                 _u64, _i32, _i32, _vp]),
    "apertis_dropout_add_layernorm_router_fwd": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp,
                                                        _vp, _vp, _i64, _i64, _i64, _f32, _u64, _i32, _i32, _vp]),
    "apertis_layernorm_bwd_blocks": (_i64, [_i64, _i64]),
    "apertis_moe_combine_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp]),
    "apertis_moe_combine_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64,
                                       _i32, _i32, _vp]),
    "apertis_grouped_gemm_nt": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _f32, _u64,
                                       _i32, _i32, _vp]),
    "apertis_grouped_gemm_nt_saves_grad": (_i32, [_i64, _i64, _i64, _i64, _i64, _i32, _i32, _i32]),
    "apertis_grouped_gemm_nt_q": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _f32, _u64,
                                         _i32, _i32, _vp, _vp]),
    "apertis_grouped_gemm_tn_workspace_bytes": (_i64, [_i64, _i32]),
    "apertis_grouped_gemm_tn_dense_variant": (_i32, [_i64, _i64]),
    "apertis_grouped_gemm_tn_pair": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64,
                                            _vp, _i64, _i32, _vp]),
    "apertis_grouped_gemm_tn_pair_q": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64,
                                              _vp, _i64, _i32, _i32, _vp]),
    "apertis_grouped_gemm_tn_q": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _vp]),
    "apertis_cast_transpose": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_weight_prep_entry_bytes": (_i64, []),
    "apertis_weight_prep": (_i32, [_vp, _i64, _i64, _vp]),
    "apertis_grouped_gemm_tn": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i32, _vp]),
    "apertis_moe_gate_topk_aux_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _f32, _vp]),
    "apertis_moe_gate_topk_aux_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _i64, _i64, _i64, _vp]),
    "apertis_moe_gate_topk_noisy_aux_fwd": (_i32, [_vp, _vp, _f32, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32,
                                                   _f32, _vp]),
    "apertis_moe_gate_topk_noisy_aux_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _f32, _u64, _vp, _vp, _vp,
                                                   _i64, _i64, _i64, _vp]),
    "apertis_moe_gate_aux_blocks": (_i64, [_i64]),
    "apertis_router_fwd": (_i32, [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "apertis_router_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "apertis_router_bwd_rows": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64,
                                       _i32, _vp]),
    "apertis_router_bwd_blocks": (_i64, [_i64]),
    "apertis_boundary_router_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp]),
    "apertis_tiny_linear_fwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "apertis_tiny_linear_bwd": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _vp]),
    "apertis_tiny_linear_bwd_pad": (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_tiny_linear_bwd_blocks": (_i64, [_i64]),
    "apertis_cross_entropy_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_cross_entropy_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_cross_entropy_fwd_bwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    "apertis_colsum_f32": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "apertis_act_dropout_bwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _f32, _u64, _i32, _vp]),
    "apertis_opt_chunk_elems": (_i64, []),
    "apertis_grad_sumsq": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "apertis_clip_coef": (_i32, [_vp, _i64, _f32, _vp, _vp, _vp]),
    "apertis_adamw_step": (_i32, [_vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _f64, _i64, _vp, _vp]),
}


class _Lib:
    """CDLL wrapper that binds each entry point on first use with its declared signature."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        if name not in SIGNATURES:
            raise AttributeError(name)
        try:
            fn = getattr(self._cdll, name)
        except AttributeError:
            raise ApertisHipError(f"{LIB_PATH} does not export {name}: rebuild with "
                                  "`python -m apertis_llm_amd.build --force`") from None
        fn.restype, fn.argtypes = SIGNATURES[name]
        setattr(self, name, fn)
        return fn


def load():
    """Return the loaded library. Raises ApertisHipError if the .so is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ApertisHipError(
                f"{LIB_PATH} not found: build it with `python -m apertis_llm_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU or eager fallback.")
        cdll = ctypes.CDLL(LIB_PATH)
        # the binding below is written against ONE version of include/apertis_hip.h: another build of the library (a stale
        # in-tree .so, or APERTIS_HIP_LIB pointing at an older one) would be called with the wrong argument lists
        try:
            got = int(cdll.apertis_abi_version())
        except AttributeError:
            got = -1
        if got != ABI_VERSION:
            raise ApertisHipError(f"{LIB_PATH} has ABI version {got >> 16}.{got & 0xffff} but this binding needs "
                                  f"{ABI_VERSION >> 16}.{ABI_VERSION & 0xffff}: rebuild with `python -m apertis_llm_amd.build --force`")
        _lib = _Lib(cdll)
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().apertis_strerror(int(rc)).decode()
        raise ApertisHipError(f"{what} failed: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a tensor (or None) as a plain integer: every entry point has its argtypes declared, so ctypes
    converts it itself (a c_void_p object per argument was a third of the Python time of a launch)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream_ptr():
    """torch's current stream of the current device, as the integer the C ABI takes.  torch.cuda.current_stream() builds a
    Stream object through three Python layers (5 us; a step makes thousands of launches): ask the binding directly."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        get_dev, get_raw = getattr(torch._C, "_cuda_getDevice", None), getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if get_dev is not None and get_raw is not None:
            _raw_stream = lambda: get_raw(get_dev())                            # noqa: E731
        else:
            _raw_stream = lambda: torch.cuda.current_stream().cuda_stream       # noqa: E731
    return _raw_stream()


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise ApertisHipError(f"unsupported dtype {t.dtype}: kernels take float32 or bfloat16")

"""Host-side operators: torch.autograd.Functions over the C ABI (libapertis_hip.so).

torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device; there is no CPU fallback.

One module per subsystem (round 6: the 2 500-line ops.py split up):
    _base    launch helpers, kernel timers, device checks, hand-over objects
    prep     prepared weight copies (cache, one-launch TrainPrep, cast / transpose)
    gemm     grouped / dense / skinny / tiny linears on the MFMA kernels
    norm     LayerNorm family and the fused block boundaries
    ssm      conv + SiLU, dropout-add, gate, column splits
    scan     selective scan and scan + skip + gate (staged / lean / look-back)
    decode   single-token decode step
    moe      gate, plan, gather-LN, combine, small-batch entrance, expert MLP
    loss     cross-entropy, fused LM head + cross-entropy
`from apertis_llm_amd import ops; ops.<name>` reaches every name of every module, private helpers included (tests and tools use
them).  The module-level SWITCHES (SCAN_LOOKBACK, TRAIN_PREP, GEMM_DYNAMIC_QUEUE, ...) live in the module whose code reads them;
`ops.SWITCH` reads and `ops.SWITCH = v` writes that module's variable (the package forwards both), so a test that flips a switch
on `ops` still reaches the code that looks at it.
"""
import sys as _sys
import types as _types

from . import _base, prep, gemm, norm, ssm, scan, decode, moe, loss

_MODULES = (_base, prep, gemm, norm, ssm, scan, decode, moe, loss)
# switches and counters that are REBOUND at run time (by tests, tools, parallel.py or the code itself): owner module per name
_FORWARDED = {
    "SCAN_SINGLE_PASS": scan, "SCAN_LEAN": scan, "SCAN_LEAN_BWD": scan, "SCAN_LOOKBACK": scan, "SCAN_DT_FUSED": scan,
    "SCAN_LOOKBACK_MIN_WGS": scan,
    "DWCONV_PAIR": ssm,
    "FUSE_ACT_BWD": moe, "SAVE_ACT_GRAD": moe, "ROWS_GRADIENT": moe, "NT2I": moe,
    "FUSE_ROUTER_BOUNDARY_BWD": norm, "FUSED_ROUTER_BWD_CALLS": norm, "FUSE_COMBINE_BWD": norm,
    "GEMM_DYNAMIC_QUEUE": gemm, "TN_DYNAMIC_QUEUE": gemm, "DENSE_WGRAD_WIDE": gemm, "_splitk_depth": gemm,
    "TRAIN_PREP": prep, "WEIGHT_EPOCH": prep, "_ACTIVE_TRAIN_PREP": prep, "_prep_scope_depth": prep,
    "_TIMER": _base,
}
for _m in _MODULES:
    for _k, _v in vars(_m).items():
        if _k.startswith("__") or _k in _FORWARDED or isinstance(_v, _types.ModuleType):
            continue
        globals()[_k] = _v
from .._lib import ApertisHipError, check, dtype_code, ptr, stream_ptr   # noqa: E402,F401  (ops.ApertisHipError, ops.dtype_code)


class _OpsPackage(_types.ModuleType):
    def __getattr__(self, name):            # (only reached for names that are not in the package's own dict)
        owner = _FORWARDED.get(name)
        if owner is not None:
            return getattr(owner, name)
        raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")

    def __setattr__(self, name, value):
        owner = _FORWARDED.get(name)
        if owner is not None:
            setattr(owner, name, value)
        else:
            super().__setattr__(name, value)

    def __dir__(self):
        return sorted(set(super().__dir__()) | set(_FORWARDED))


_sys.modules[__name__].__class__ = _OpsPackage

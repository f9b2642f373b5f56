"""Compile the oracle's plain-C restatement (oracle/scan_ref.c) with gcc into oracle/_build/.
Test infrastructure only."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "liboracle_ref.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "scan_ref.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", LIB, src, "-lm"], check=True)
    return LIB


def load():
    lib = ctypes.CDLL(build())
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.oracle_scan_fwd_f32.argtypes = [vp] * 7 + [i64] * 4
    lib.oracle_scan_bwd_f32.argtypes = [vp] * 10 + [i64] * 4
    lib.oracle_moe_plan.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, i64, i64, i64]
    for f in (lib.oracle_scan_fwd_f32, lib.oracle_scan_bwd_f32, lib.oracle_moe_plan):
        f.restype = None
    return lib


if __name__ == "__main__":
    print(build())

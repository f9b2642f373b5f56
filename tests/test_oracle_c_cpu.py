"""The oracle's plain-C restatement (oracle/scan_ref.c) against the golden vectors captured from
the reference and against the numpy/torch oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def _p(a):
    return a.ctypes.data if a is not None else None


@pytest.mark.parametrize("name", ["L1", "L7", "L64", "L257", "L2048", "bigdelta"])
def test_c_scan_matches_reference_golden(name):
    from oracle import build_c
    lib = build_c.load()
    g = load_golden("scan_" + name)
    arr = lambda k: np.ascontiguousarray(g[k].numpy(), dtype=np.float32) if k in g else None
    delta, A_log, Bt, C, dy, h0 = arr("delta"), arr("A_log"), arr("Bt"), arr("C"), arr("dy"), arr("h0")
    B, L, h = delta.shape
    N = A_log.shape[1]
    y, hl = np.empty_like(Bt), np.empty((B, h * N), np.float32)
    lib.oracle_scan_fwd_f32(_p(delta), _p(A_log), _p(Bt), _p(C), _p(h0), _p(y), _p(hl), B, L, h, N)
    np.testing.assert_allclose(y, g["y"].numpy(), rtol=2e-5, atol=2e-6 * np.abs(g["y"].numpy()).max())
    if "h_last" in g:
        np.testing.assert_allclose(hl, g["h_last"].numpy(), rtol=2e-5, atol=1e-6)
    dd, da = np.empty_like(delta), np.empty_like(A_log)
    db, dc = np.empty_like(Bt), np.empty_like(Bt)
    lib.oracle_scan_bwd_f32(_p(delta), _p(A_log), _p(Bt), _p(C), _p(dy), _p(h0), _p(dd), _p(da), _p(db), _p(dc), B, L, h, N)
    for got, key in ((dd, "d_delta"), (da, "dA_log"), (db, "dBt"), (dc, "dC")):
        ref = g[key].numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def test_c_plan_matches_numpy_oracle_and_golden():
    from oracle import build_c, ref_cpu
    lib = build_c.load()
    for name in ["moe_eval", "moe_train_overflow", "moe_eval_k3"]:
        g = load_golden(name)
        idx = np.ascontiguousarray(g["idx"].numpy(), dtype=np.int32)
        w = np.ascontiguousarray(g["w"].numpy(), dtype=np.float32)
        S, K = idx.shape
        E, cap = int(g["E"]), int(g["capacity"])
        offs = np.empty(E + 1, np.int32)
        rt, rk = np.empty(S * K, np.int32), np.empty(S * K, np.int32)
        slot = np.empty((S, K), np.int32)
        lib.oracle_moe_plan(_p(idx), _p(w), None, cap, _p(offs), _p(rt), _p(rk), _p(slot), S, E, K)
        A = int(offs[-1])
        rows = [(int(rt[r]), int(rk[r]), e) for e in range(E) for r in range(offs[e], offs[e + 1])]
        assert rows == [tuple(r) for r in g["kept_rows"].tolist()], name
        o2, rt2, rk2, s2 = ref_cpu.dispatch_plan(idx, w, E, cap if cap > 0 else None)
        assert offs.tolist() == o2.tolist() and rt[:A].tolist() == rt2.tolist() and slot.tolist() == s2.tolist()
    rng = np.random.default_rng(3)
    S, E, K = 700, 8, 2
    idx = np.stack([rng.permutation(E)[:K] for _ in range(S)]).astype(np.int32)
    w = np.round(rng.random((S, K)), 1).astype(np.float32)      # many ties
    active = np.array([1, 1, 0, 1, 1, 1, 0, 1], np.uint8)
    offs = np.empty(E + 1, np.int32); rt = np.empty(S * K, np.int32); rk = np.empty(S * K, np.int32)
    slot = np.empty((S, K), np.int32)
    lib.oracle_moe_plan(_p(idx), _p(w), _p(active), 60, _p(offs), _p(rt), _p(rk), _p(slot), S, E, K)
    o2, rt2, rk2, s2 = ref_cpu.dispatch_plan(idx, w, E, 60, active)
    assert offs.tolist() == o2.tolist() and rt[:offs[-1]].tolist() == rt2.tolist() and slot.tolist() == s2.tolist()

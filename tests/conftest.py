import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out, sd = {}, {}
    for k in z.files:
        v = z[k]
        t = torch.from_numpy(v) if v.dtype.kind in "fiu" and v.ndim > 0 else v
        if k.startswith("sd::"):
            sd[k[4:]] = t
        else:
            out[k] = t
    out["sd"] = sd
    return out


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a ROCm device"
    return torch.device("cuda:0")


def rel_error_report(name, got, ref, rtol=1e-4, sig_frac=1e-2, check=True):
    """The achieved error of `got` against `ref`, reported and (with check) held to the BASELINE bar (fp32 logits
    within 1e-4 rtol).  Three numbers:
      max_abs_over_refmax   max|got-ref| / max|ref|
      max_rel_significant   max over elements with |ref| >= sig_frac * max|ref| of |got-ref| / |ref|
      max_rel_all           max over elements with ref != 0 of |got-ref| / |ref|  (reported only: an element that is
                            zero to rounding has no meaningful relative error)
    The line is printed (pytest -rP shows it) and appended to gpurun_out/parity_report.jsonl when that directory
    exists, so a GPU run leaves the achieved errors behind as a record."""
    import json
    import torch
    ref64 = torch.as_tensor(ref).detach().cpu().to(torch.float64)
    got64 = torch.as_tensor(got).detach().cpu().to(torch.float64)
    assert got64.shape == ref64.shape, (name, tuple(got64.shape), tuple(ref64.shape))
    diff = (got64 - ref64).abs()
    refmax = float(ref64.abs().max())
    sig = ref64.abs() >= sig_frac * refmax
    nz = ref64 != 0
    rec = {"test": name, "rtol": rtol, "ref_absmax": refmax, "max_abs": float(diff.max()),
           "max_abs_over_refmax": float(diff.max()) / (refmax + 1e-300),
           "max_rel_significant": float((diff[sig] / ref64[sig].abs()).max()) if sig.any() else 0.0,
           "max_rel_all": float((diff[nz] / ref64[nz].abs()).max()) if nz.any() else 0.0,
           "frac_over_rtol_all": float(((diff[nz] / ref64[nz].abs()) > rtol).double().mean()) if nz.any() else 0.0,
           "numel": int(ref64.numel())}
    print("PARITY " + json.dumps(rec))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        try:
            with open(os.path.join(out_dir, "parity_report.jsonl"), "a") as f:
                f.write(json.dumps(rec) + "\n")
        except OSError:
            pass
    if check:
        assert rec["max_abs_over_refmax"] <= rtol, rec
        assert rec["max_rel_significant"] <= rtol, rec
    return rec

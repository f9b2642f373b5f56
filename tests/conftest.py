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

"""bench.py's own rank launcher (`python bench.py --gpus N` without torchrun): CPU-only checks.

The launcher mirrors what the reference leaves to an external DDP launcher (src/training/pipeline.py:435-447):
one process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set.  Here: the stub step over gloo
(`--launcher-selftest`), the no-GPU exit in every child, and the refusal of a world size that differs from --gpus.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_launcher_two_ranks_stub_step_over_gloo():
    r = _run(["--gpus", "2", "--launcher-selftest", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    line = json.loads(lines[-1])                     # the result is the LAST stdout line of the parent
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["selftest"] is True
    assert line["steps"] == 4 and line["warmup"] == 1 and line["ms_per_step"] > 0
    # Linear(32, 32): 1056 fp32 gradients, each tensor on a 256-byte slice boundary (1024 + 64 floats): one all-reduce
    # of the bucket per step
    assert line["config"]["allreduce_bytes_per_step"] == (1024 + 64) * 4
    assert sum(1 for ln in lines if ln.lstrip().startswith("{")) == 1


def test_launcher_without_gpu_fails_in_every_child():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the real bench would start")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert r.stderr.count("needs a ROCm GPU") == 2          # both ranks got as far as the device check
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "1"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert r.returncode != 0 and "refusing" in r.stderr
    r = _run(["--gpus", "4", "--launcher-selftest"], env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert r.returncode != 0 and "refusing" in r.stderr


def test_launcher_parent_never_imports_torch():
    """The parent must not initialise HIP: it may not even import torch before it spawns the ranks."""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--launcher-selftest', '--steps', '1', '--warmup', '0']\n"
            "import subprocess as sp\n"
            "orig = sp.Popen\n"
            "def spy(*a, **k):\n"
            "    assert 'torch' not in sys.modules, 'torch imported before the ranks were spawned'\n"
            "    return orig(*a, **k)\n"
            "sp.Popen = spy\n"
            "try:\n"
            "    runpy.run_path(%r, run_name='__main__')\n"
            "except SystemExit as e:\n"
            "    assert 'torch' not in sys.modules\n"
            "    sys.exit(e.code)\n" % BENCH)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]

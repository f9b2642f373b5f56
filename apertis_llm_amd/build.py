"""Build libapertis_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m apertis_llm_amd.build [--force]

One object per .hip file (cached by mtime) then one link; the .so lands next to this file so
it travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libapertis_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
         "-Wno-unused-result"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libapertis_hip.so cannot be built")


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "apertis_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_m = _deps_mtime()
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m)
        jobs.append((src, obj, stale))

    def compile_one(job):
        src, obj, stale = job
        if not stale:
            return
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(4, len(jobs) or 1)) as ex:
        list(ex.map(compile_one, jobs))
    objs = [j[1] for j in jobs]
    need_link = force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if need_link:
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

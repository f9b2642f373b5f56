"""The hand-issued ticket atomic of grouped_gemm_nt4r_k (csrc/grouped_gemm.hip, dynamic tile queue) returns into a VGPR
ASYNCHRONOUSLY: the data arrives when the atomic comes back, ordered only by the kernel's hand-kept vmcnt (sub-step 3's wait).
hipcc knows nothing of that - it takes the register for defined right behind the instruction - so the compiled code must not
read, copy or overwrite that register anywhere but in the hand-written ds_write_b32 of the hand-over.  This script
disassembles the gfx950 code object of the in-tree build (csrc/_obj/grouped_gemm.o) and checks exactly that for every
instantiation of the kernel.  Exit status 0 = safe.  Run by tests/test_host_cpu.py and by hand after any edit of the kernel:
    python tools/check_nt4r_ticket_isa.py [path/to/grouped_gemm.o]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(obj):
    import shutil
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, "gg.o")
        shutil.copy(obj, local)
        # (llvm-objdump --offloading writes the bundles next to its input: <input>.0.hipv4-amdgcn-amd-amdhsa--gfx950)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        co = next(os.path.join(td, f) for f in os.listdir(td) if "amdgcn" in f and "gfx950" in f)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True,
                              text=True).stdout.split("\n")


def check(lines):
    heads = [i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*grouped_gemm_nt4r_k.*>:", l)]
    if not heads:
        return ["no grouped_gemm_nt4r_k symbol in the code object"]
    problems = []
    for st in heads:
        end = next((i for i in range(st + 1, len(lines)) if re.match(r"^[0-9a-f]+ <.*>:", lines[i])), len(lines))
        body = [l.split("//")[0] for l in lines[st:end]]
        at = [i for i, l in enumerate(body) if "global_atomic_add" in l and " off" in l and "sc0" in l]
        name = lines[st].split("<")[1][:70]
        m = re.search(r"nt4r_kI\w+?Lb(\d)ELb(\d)EEE", lines[st])     # <TO, RAGGED, DYN>: only the DYN instantiations draw tickets
        dyn = bool(m and m.group(2) == "1")
        if len(at) != (1 if dyn else 0):
            problems.append(f"{name}: expected {1 if dyn else 0} hand-issued ticket atomic(s), found {len(at)}")
            continue
        if not dyn:
            print(f"{name}: static walk (no ticket)")
            continue
        reg = re.search(r"global_atomic_add\s+(v\d+),", body[at[0]]).group(1)
        n = int(reg[1:])
        touch = [(j, l.strip()) for j, l in enumerate(body) if j != at[0] and
                 (re.search(r"\b" + reg + r"\b", l) or any(int(x) <= n <= int(y) for x, y in re.findall(r"v\[(\d+):(\d+)\]", l)))]
        # the hand-over: the first ds_write_b32 of the register BEHIND the atomic in the code (the loop body is laid out in
        # execution order: ticket at its top, sub-steps 0..3, hand-over); between the two nothing may touch the register.
        # (Elsewhere - the epilogue behind the K loop, before the next ticket - the allocator may reuse it.)
        hand = next((j for j, l in touch if j > at[0] and l.startswith("ds_write_b32") and re.search(r",\s*" + reg + r"\b", l)), None)
        between = [l for j, l in touch if hand is not None and at[0] < j < hand]
        print(f"{name}: ticket register {reg}; hand-over {hand - at[0] if hand else None} instructions behind the atomic; "
              f"touches in between: {between}; reuses elsewhere: {len(touch) - 1 - len(between)}")
        if hand is None or between:
            problems.append(f"{name}: {reg} is touched between the atomic and its hand-over: {between if hand else 'no hand-over found'}")
    return problems


if __name__ == "__main__":
    obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "apertis_llm_amd", "csrc", "_obj", "grouped_gemm.o")
    bad = check(disassemble(obj))
    for b in bad:
        print("PROBLEM:", b)
    print("OK" if not bad else "FAILED")
    sys.exit(1 if bad else 0)

"""oproj_ln_kernel requests its residual fragments with inline-asm `global_load_dwordx4 v[..], v, s[..]` two stages before
it waits for them (`s_waitcnt vmcnt(8)`): the compiler does not know that those registers are pending.  This script
compiles encoder.hip to ISA the way the Makefile does and checks that NO instruction between the loads and that wait
reads or writes one of the destination registers (a copy or spill there would pick up stale data).  Also reports
spills of the kernels whose waits are counted by hand (a spill reload waits behind every DMA in flight).

    python tools/check_pending_loads.py          # exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ai-dial-rag_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=on --offload-arch=gfx950".split()


def kernel_body(asm: str, mangled_prefix: str) -> list:
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end]


def regs_of(text: str):
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            yield from range(int(m.group(1)), int(m.group(2)) + 1)
        else:
            yield int(m.group(3))


def main() -> int:
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, "-I" + CSRC, "-save-temps", "-c", os.path.join(CSRC, "encoder.hip"), "-o", os.devnull]
        subprocess.run(cmd, cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, "encoder-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    rc = 0
    body = kernel_body(asm, "_ZN3mir3enc15oproj_ln_kernel")
    loads = [(i, l) for i, l in enumerate(body) if "global_load_dwordx4" in l and ", s[" in l]
    wait = next(i for i, l in enumerate(body) if "s_waitcnt vmcnt(8)" in l)
    pending = {}
    for i, l in loads:
        for r in regs_of(l.split(",")[0]):
            pending[r] = i
    print(f"oproj_ln_kernel: {len(loads)} asm loads into {len(pending)} registers, wait at line {wait}")
    if len(loads) != 12 or not all(i < wait for i, _ in loads):
        print("  UNEXPECTED: 12 loads before the wait were expected")
        rc = 1
    for i in range(loads[0][0], wait):
        line = body[i].split(";")[0]
        if (i, body[i]) in loads:
            continue
        hit = [r for r in regs_of(line) if r in pending and pending[r] < i]
        if hit:
            print(f"  TOUCHED before the wait: line {i}: {line.strip()}")
            rc = 1
    for name in ("_ZN3mir3enc15oproj_ln_kernel", "_ZN3mir3enc10qkv_kernel"):
        n = sum("scratch_" in l for l in kernel_body(asm, name))
        print(f"{name}: {n} scratch instructions")
        rc |= 1 if n else 0
    print("clean" if rc == 0 else "NOT clean")
    return rc


if __name__ == "__main__":
    sys.exit(main())

"""Two encoder kernels request data with inline-asm `global_load_dwordx4 ... ; pending` long before they wait for it
(`s_waitcnt ... ; release-pending`): oproj_ln_kernel its residual fragments (two stages early), ffn_ln_kernel the next
group's activations (a stage early, across a barrier).  The compiler does not know that those registers are pending, so
this script compiles the two translation units to ISA with the Makefile's own command line and checks that

  * every marked load is followed by a marked release, and NO instruction between a load and that release reads or
    writes one of the load's destination registers (a copy or a spill there would pick up stale data);
  * the kernels whose waits are counted by hand do not spill inside their stage loops (a spill reload waits behind every
    DMA in flight): none at all in oproj_ln_kernel / qkv_kernel, and in ffn_ln_kernel none between the two markers.

    python tools/check_pending_loads.py          # exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ai-dial-rag_amd", "csrc")


def compile_asm(unit: str) -> str:
    """The ISA of <unit>.hip, compiled with EXACTLY the command the Makefile uses for build/<unit>.o (asked of `make -n`),
    plus -save-temps, in a scratch directory."""
    dry = subprocess.run(["make", "-C", CSRC, "-n", "-B", f"build/{unit}.o"], check=True, capture_output=True, text=True).stdout
    line = next((l for l in dry.splitlines() if f"{unit}.hip" in l and " -c " in l), None)
    if line is None:
        raise RuntimeError(f"check_pending_loads: `make -n -B build/{unit}.o` shows no compile command for {unit}.hip:\n{dry}")
    words = line.split()
    cut = words.index("-c")
    first = next(i for i, w in enumerate(words[:cut]) if w.startswith("-"))  # (the compiler may be `ccache hipcc`, an absolute path, ...)
    compiler, flags = words[:first], words[first:cut]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [*compiler, *flags, "-I" + CSRC, "-save-temps", "-c", os.path.join(CSRC, unit + ".hip"), "-o", os.devnull]
        subprocess.run(cmd, cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        import glob
        found = glob.glob(os.path.join(tmp, unit + "-hip-amdgcn-*.s"))
        if len(found) != 1:
            raise RuntimeError(f"check_pending_loads: expected one device ISA file for {unit}.hip, found {found or 'none'} "
                               f"(command: {' '.join(cmd)})")
        return open(found[0]).read()


def kernel_body(asm: str, needle: str) -> list:
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if needle in l and l.startswith("_Z") and l.split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def regs_of(text: str):
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            yield from range(int(m.group(1)), int(m.group(2)) + 1)
        else:
            yield int(m.group(3))


def check_pending(name: str, body: list, expect_loads: int) -> int:
    loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l and "; pending" in l]
    releases = [i for i, l in enumerate(body) if "release-pending" in l]
    rc = 0
    if len(loads) != expect_loads or len(releases) != 1 or not all(i < releases[0] for i in loads):
        print(f"  {name}: UNEXPECTED: {len(loads)} marked loads (expected {expect_loads}), releases at {releases}")
        return 1
    release = releases[0]
    pending = {}
    for i in loads:
        for r in regs_of(body[i].split(";")[0].split(",")[0]):
            pending[r] = i
    for i in range(loads[0], release):
        if i in loads:
            continue
        line = body[i].split(";")[0]
        if any(r in pending and pending[r] < i for r in regs_of(line)):
            print(f"  {name}: TOUCHED before the release: line {i}: {line.strip()}")
            rc = 1
        if "scratch_" in line:
            print(f"  {name}: spill between the loads and the release: line {i}: {line.strip()}")
            rc = 1
    print(f"{name}: {len(loads)} pending loads into {len(pending)} registers, released at line {release}: {'ok' if rc == 0 else 'NOT ok'}")
    return rc


def main() -> int:
    rc = 0
    enc = compile_asm("encoder")
    oproj = kernel_body(enc, "oproj_ln_kernel")
    rc |= check_pending("oproj_ln_kernel", oproj, 12)
    for name in ("oproj_ln_kernel", "qkv_kernel"):
        n = sum("scratch_" in l.split(";")[0] for l in kernel_body(enc, name))
        print(f"{name}: {n} scratch instructions")
        rc |= 1 if n else 0
    ffn = kernel_body(compile_asm("encoder_ffn"), "ffn_ln_kernel")
    rc |= check_pending("ffn_ln_kernel", ffn, 24)
    print("clean" if rc == 0 else "NOT clean")
    return rc


if __name__ == "__main__":
    sys.exit(main())

"""Build-time lint for a code-generation hazard seen in round 4 (ROCm 7.2 hipcc, gfx950): a VGPR -> AGPR spill
(`v_accvgpr_write_b32`) placed INSIDE an exec-masked region (between `s_and_saveexec_b64` and the `s_or_b64 exec, exec, ...`
that restores the mask).  Lanes outside the mask are not saved; with an empty mask nothing is, and the reload returns
garbage.  Scans `*.s` listings (make build/mfma_10.s, build/mfma2_9_10.s ...) kernel by kernel; regions longer than
SPAN lines are taken to be whole-body masks (an early `return` of a wave) and ignored.

usage: python tools/isa_spill_lint.py pybold_amd/csrc/build/*.s | build/mfma*.o      (exit code 1 when a kernel has such a write)
"""
import glob
import os
import re
import subprocess
import sys

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def listing(path):
    """Lines of a `.s` listing, or -- for an object file of the build -- the disassembly of its gfx950 code object
    (`llvm-objdump --offloading` extracts it next to the object, `-d` disassembles it: a second per file, and it is
    the code that ships, not a second compile)."""
    if not path.endswith(".o"):
        return open(path)
    subprocess.check_call([OBJDUMP, "--offloading", path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    parts = glob.glob(path + ".*")
    try:
        co = [f for f in parts if "amdgcn" in f]
        assert co, "no device code object in " + path
        return subprocess.run([OBJDUMP, "-d", co[0]], capture_output=True, text=True, check=True).stdout.splitlines()
    finally:
        for f in parts:
            os.remove(f)


SPAN = 300
bad = 0
for path in sys.argv[1:]:
    kernel, stack, hits = None, [], {}
    for n, line in enumerate(listing(path), 1):
        s = line.strip()
        m = re.match(r"^(?:[0-9a-f]+ <)?(_ZN2pb\w+)>?:", s)
        if m:
            kernel, stack = m.group(1), []
            continue
        if kernel is None:
            continue
        if s.startswith("s_endpgm"):
            kernel = None
            continue
        if re.match(r"s_and_saveexec_b64", s):
            stack.append(n)
        elif re.match(r"s_or_saveexec_b64|s_andn2_saveexec_b64", s):   # the else-part of the region on top: same level
            if stack:
                stack[-1] = n
            else:
                stack.append(n)
        elif re.match(r"s_or_b64 exec, exec,", s) or re.match(r"s_mov_b64 exec,", s):
            if stack:
                stack.pop()
        elif stack and n - stack[-1] < SPAN and s.startswith("v_accvgpr_write_b32"):
            hits.setdefault(kernel, []).append(n)
    for k, lines in hits.items():
        bad += 1
        print("%s: %s: %d accumulator-register writes inside an exec-masked region (lines %s ...)" % (path, k[:90], len(lines), lines[:4]))
print("%d kernel(s) with accumulator-register writes under a partial exec mask" % bad)
sys.exit(1 if bad else 0)

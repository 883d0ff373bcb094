"""CPU-only (hipcc cross-compiles): the matrix-pipe kernels of the build carry no register spill inside an
exec-masked region -- the code-generation hazard that broke the certificate variant of the split form in round 4
(a VGPR -> AGPR spill executed under an EMPTY exec mask on a cold start: its reload, an LDS address, was garbage;
DESIGN 5.0b) -- and no scratch.  tools/isa_spill_lint.py is the check (on the disassembled objects; `make -C pybold_amd/csrc
build/mfma2_8_9.s` etc. produce annotated listings for reading)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pybold_amd", "csrc")


def test_no_spill_under_a_partial_exec_mask():
    """Every matrix-pipe object of the build (fista_mfma / mfma2 / mfma4 kernels, all variants): the gfx950 code object is
    extracted from the object file and disassembled (`llvm-objdump`, a second per file -- the code that ships, not a second
    compile), tools/isa_spill_lint.py scans it for accumulator-register writes inside exec-masked regions."""
    if subprocess.call(["which", "hipcc"], stdout=subprocess.DEVNULL) != 0 and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    import glob
    objs = sorted(glob.glob(os.path.join(CSRC, "build", "mfma*.o")))
    assert len(objs) >= 27, objs
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_spill_lint.py")] + objs, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout


def test_no_matrix_pipe_variant_uses_scratch():
    """Every instantiation capi.hip can dispatch -- fista_mfma_kernel<5..10> x {plain, cost trace, certificate, taps from
    device memory, three near tiles, _loops_deconv rule}, fista_mfma2_kernel<a,b> and fista_mfma4_kernel<6..10> x {plain,
    cost trace, certificate, taps from device memory, _loops_deconv rule, three near tiles}: 243 kernels -- runs without scratch (two explicit allowances of 8 B per lane, tools/mfma_register_table.py: ALLOW).  (Round 4 shipped `<10, ..., LOOPS>` with 156 B per
    lane: store addresses of the in-loop write-out hoisted out of the solve loop.)  The reports are written by the compile
    that makes each object (csrc/Makefile)."""
    if subprocess.call(["which", "hipcc"], stdout=subprocess.DEVNULL) != 0 and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    subprocess.check_call(["make", "-s", "-j8", "-C", CSRC])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_register_table.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert int(out.stdout.strip().splitlines()[-1].split()[0]) >= 150, out.stdout

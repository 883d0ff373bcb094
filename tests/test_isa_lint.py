"""CPU-only (hipcc cross-compiles): the ISA listings of the matrix-pipe kernels carry no register spill inside an
exec-masked region -- the code-generation hazard that broke the certificate variant of the split form in round 4
(a VGPR -> AGPR spill executed under an EMPTY exec mask on a cold start: its reload, an LDS address, was garbage;
DESIGN 5.0b) -- and no scratch.  tools/isa_spill_lint.py is the check; `make -C pybold_amd/csrc build/mfma2_8_9.s`
etc. produce the listings."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pybold_amd", "csrc")


LISTINGS = ["mfma2_8_9", "mfma_10", "mfma4_10"]


@pytest.fixture(scope="module")
def listings():
    """The three listings in ONE parallel make (about a minute each on one core)."""
    if subprocess.call(["which", "hipcc"], stdout=subprocess.DEVNULL) != 0 and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    subprocess.check_call(["make", "-s", "-j4", "-C", CSRC] + ["build/%s.s" % t for t in LISTINGS])


@pytest.mark.parametrize("target", LISTINGS)
def test_no_spill_under_a_partial_exec_mask(listings, target):
    lst = os.path.join(CSRC, "build", target + ".s")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_spill_lint.py"), lst], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    # the variants that matter run without scratch (the plain ones of both kernels, the split form's cost-trace / certificate ones)
    res = open(os.path.join(CSRC, "build", target + ".res")).read()
    names = re.findall(r"Function Name: (\S+)", res)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", res)]
    assert len(names) == len(scratch) and len(names) >= 4
    plain = [s for n, s in zip(names, scratch) if n.endswith("ILi10ELb0ELb0ELb0ELi2ELb0EEEvNS_9FistaArgsENS_8MfmaTapsE") or "mfma2_kernel" in n or "mfma4_kernel" in n]
    assert plain and max(plain) == 0, list(zip(names, scratch))


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

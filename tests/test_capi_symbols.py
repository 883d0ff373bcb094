"""CPU-only: the C-ABI library loads, exports every symbol include/pybold_hip.h
declares, and validates its arguments loudly (no kernel is launched here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from pybold_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        ge.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pybold_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pb_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from pybold_amd import _lib
    names = declared_symbols()
    assert len(names) >= 12
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), "missing export " + name
    assert sorted(_lib.SIGNATURES) == names     # the Python binding covers the header 1:1


def test_version_and_dispatch_table(lib):
    assert lib.pb_version() >= 100
    assert lib.pb_fista_has_fast_path(300, 30) == 1     # BASELINE configs 1-3, 5
    assert lib.pb_fista_has_fast_path(300, 27) == 1     # config 4
    assert lib.pb_fista_has_fast_path(240, 27) == 1     # golden _loops_deconv case
    assert lib.pb_fista_has_fast_path(1200, 28) == 1    # HCP-length runs
    assert lib.pb_fista_which_kernel(1200, 28, 5000, 0, 0, 6) == 6      # round 5: one series over the four waves of a workgroup (641..1280 scans)
    assert lib.pb_fista_which_kernel(1200, 28, 5000, 1, 0, 6) == 6
    assert lib.pb_fista_which_kernel(1200, 28, 5000, 0, 1, 6) == 6      # the _loops_deconv rule inside it
    assert lib.pb_fista_which_kernel(1200, 28, 5000, 1, 1, 6) == 3      # ... with a cost trace: one problem per wave
    assert lib.pb_fista_which_kernel(1200, 28, 5000, 1, 2, 6) == 6      # default deconv path: window-rule certificate
    assert lib.pb_fista_which_kernel(1200, 28, 5000, 1, 2, 4) == 3      # another window: the exact rule, one problem per wave
    assert lib.pb_fista_which_kernel(1200, 28, 2000, 0, 0, 6) == 3      # under 5/8 of a pass: the vector form finishes first
    assert lib.pb_fista_which_kernel(1200, 40, 5000, 0, 0, 6) == 6      # 34+ taps: three near tiles (plain solves, cost trace)
    assert lib.pb_fista_which_kernel(1200, 40, 5000, 1, 2, 6) == 6      # ... the window-rule certificate beside them too
    assert lib.pb_fista_which_kernel(1200, 40, 5000, 0, 1, 6) == 6      # ... and the _loops_deconv rule
    assert lib.pb_fista_which_kernel(600, 42, 20000, 0, 0, 6) == 5      # the same on the two-wave form
    assert lib.pb_fista_which_kernel(600, 42, 20000, 0, 1, 6) == 5       # ... with the _loops_deconv rule too
    assert lib.pb_fista_which_kernel(600, 42, 20000, 1, 1, 6) in (1, 3)  # ... a cost trace beside it: vector forms
    assert lib.pb_fista_which_kernel(600, 30, 6000, 0, 1, 6) == 5       # the _loops_deconv rule inside the two-wave form
    assert lib.pb_fista_which_kernel(2400, 28, 5000, 1, 2, 6) == 0      # S = 38: ring too large
    assert lib.pb_fista_has_fast_path(100000, 30) == 0
    assert lib.pb_fista_has_fast_path(300, 48) == 1     # short TR: HRFs of up to 48 taps
    assert lib.pb_fista_which_kernel(300, 48, 100000, 0, 0, 6) == 4      # 34..65 taps: matrix-pipe form with three near tiles
    assert lib.pb_fista_which_kernel(300, 48, 100000, 1, 0, 6) == 4      # ... with the cost trace too
    assert lib.pb_fista_which_kernel(300, 48, 100000, 1, 2, 6) == 5      # ... the window-rule certificate beside three tiles: the split form (round 5)
    assert lib.pb_fista_which_kernel(200, 48, 100000, 1, 2, 6) == 1      # ... under 225 scans (three blocks per wave): single-row form
    assert lib.pb_fista_has_fast_path(300, 49) == 0
    assert lib.pb_fista_has_fast_path(300, 5000) == 0
    assert lib.pb_fista_has_fast_path(0, 30) == 0
    # dispatch of a plain solve: pair kernel for machine-filling batches (with or without
    # the cost trace, and for the window rule at wind = 6: no-fire certificate + re-solve, round 3),
    # single-row kernel for small ones and for the _loops_deconv rule
    assert lib.pb_fista_which_kernel(300, 30, 100000, 0, 0, 6) == 4      # plain solves: both operators on the matrix pipe (round 3)
    assert lib.pb_fista_which_kernel(300, 30, 100000, 1, 0, 6) == 4      # cost trace: matrix-pipe form too
    assert lib.pb_fista_which_kernel(300, 30, 100000, 0, 2, 6) == 4      # window rule at wind = 6: no-fire certificate on the matrix-pipe form
    assert lib.pb_fista_which_kernel(300, 30, 100000, 0, 1, 6) == 4      # _loops_deconv rule: evaluated exactly inside the matrix-pipe form (round 4)
    assert lib.pb_fista_which_kernel(300, 30, 100000, 1, 1, 6) == 1      # ... with the cost trace: single-row form
    assert lib.pb_fista_which_kernel(300, 30, 100000, 0, 2, 4) == 1      # wind 4 / 8: full rule, single-row form
    assert lib.pb_fista_which_kernel(300, 30, 100000, 0, 2, 8) == 1
    assert lib.pb_fista_which_kernel(300, 30, 100000, 0, 2, 5) == 0      # other windows: LDS kernel (the Python layer warns)
    assert lib.pb_fista_which_kernel(300, 30, 1, 0, 0, 6) == 3           # a few short series: one per wave
    assert lib.pb_fista_which_kernel(300, 30, 10000, 0, 0, 6) == 5       # config 2: half a round on the split matrix-pipe form (round 4) + left-overs
    assert lib.pb_fista_which_kernel(300, 30, 10000, 1, 0, 6) == 5       # ... with the cost trace too
    assert lib.pb_fista_which_kernel(300, 30, 4096, 0, 0, 6) == 1        # single-row kernel
    assert lib.pb_fista_which_kernel(300, 30, 12500, 0, 0, 6) == 4       # config 3's shard on 8 GPUs
    assert lib.pb_fista_which_kernel(300, 30, 8000, 0, 0, 6) == 5        # under half a round: one pass of the split matrix-pipe form
    assert lib.pb_fista_which_kernel(300, 30, 8000, 1, 0, 6) == 5        # ... with the cost trace too
    assert lib.pb_fista_which_kernel(240, 27, 50000, 0, 0, 6) == 4       # 129..310 scans, up to 33 taps
    assert lib.pb_fista_which_kernel(128, 16, 50000, 0, 0, 6) == 2       # shorter series: pair form
    assert lib.pb_fista_which_kernel(300, 30, 8192, 1, 2, 6) == 5        # the deconv default call, half a round: one pass of the split form (certificate)
    assert lib.pb_fista_which_kernel(300, 30, 50000, 1, 2, 6) == 4       # ... whole rounds: matrix-pipe form
    assert lib.pb_fista_which_kernel(300, 30, 8192, 1, 1, 6) == 1        # _loops_deconv rule: single-row kernel
    assert lib.pb_fista_which_kernel(300, 30, 3, 1, 2, 6) == 3           # ... or one problem per wave
    assert lib.pb_fista_which_kernel(600, 30, 100000, 0, 0, 6) == 5      # 600 scans: the matrix pipe, every series split over two waves (round 4)
    assert lib.pb_fista_which_kernel(600, 30, 100000, 1, 0, 6) == 5      # ... with the cost trace
    assert lib.pb_fista_which_kernel(600, 30, 100000, 1, 2, 6) == 5      # ... and with the window rule as a certificate (the reference-default call)
    assert lib.pb_fista_which_kernel(600, 30, 100000, 1, 2, 4) in (1, 3) # other windows: the exact rule on a vector form
    assert lib.pb_fista_which_kernel(600, 30, 100000, 1, 1, 6) in (1, 3) # the _loops_deconv rule on long series: vector forms
    assert lib.pb_fista_which_kernel(600, 30, 500, 0, 0, 6) in (1, 3)    # few long series: latency-bound forms
    assert lib.pb_fista_which_kernel(310, 30, 100000, 0, 0, 6) == 4      # ten blocks of 31 samples: the one-wave form ...
    assert lib.pb_fista_which_kernel(311, 30, 100000, 0, 0, 6) == 5      # ... one scan more: the split form (blocks of 32 samples)
    assert lib.pb_fista_which_kernel(320, 33, 100000, 0, 0, 6) == 5
    assert lib.pb_fista_which_kernel(129, 30, 100000, 0, 0, 6) == 4      # five blocks
    assert lib.pb_fista_which_kernel(640, 30, 100000, 0, 0, 6) == 5      # up to 640 scans on the split matrix-pipe form
    assert lib.pb_fista_which_kernel(640, 30, 100000, 1, 0, 6) == 5      # ... with the cost trace too
    assert lib.pb_fista_which_kernel(641, 30, 100000, 0, 0, 6) == 6      # ... one scan more: four waves per series (round 5), up to 1 280
    assert lib.pb_fista_which_kernel(1280, 30, 100000, 0, 0, 6) == 6
    assert lib.pb_fista_which_kernel(1281, 30, 100000, 0, 0, 6) == 3
    assert lib.pb_fista_which_kernel(5000, 30, 10, 0, 0, 6) == 0


def test_argument_errors_do_not_reach_the_gpu(lib):
    from pybold_amd import _lib
    taps = np.ones(4)
    rc = lib.pb_fista_solve(None, 300, 1, None, 300, 4, 300, taps.ctypes.data, None, 4, 1.0, 1.0,
                            None, None, 10, None, 0, 0, 0.0, 6, None, 0, None)
    assert rc == -1 and b"NULL" in lib.pb_last_error()
    with pytest.raises(_lib.PyboldHipError):
        _lib.check(rc, "pb_fista_solve")
    fake = ctypes.c_void_p(4096)     # never dereferenced: validation fails first
    rc = lib.pb_fista_solve(fake, 10, 1, fake, 300, 4, 300, taps.ctypes.data, None, 4, 1.0, 1.0,
                            None, fake, 10, None, 0, 0, 0.0, 6, None, 0, None)
    assert rc == -1 and b"leading dimension" in lib.pb_last_error()
    rc = lib.pb_fista_solve(fake, 300, 1, fake, 300, 4, 300, taps.ctypes.data, None, 4, -1.0, 1.0,
                            None, fake, 10, None, 0, 0, 0.0, 6, None, 0, None)
    assert rc == -1 and b"step" in lib.pb_last_error()
    rc = lib.pb_fista_solve(fake, 300, 1, fake, 300, 4, 300, taps.ctypes.data, None, 4, 1.0, 1.0,
                            None, fake, 10, None, 0, 7, 0.0, 6, None, 0, None)
    assert rc == -1 and b"stop_mode" in lib.pb_last_error()
    rc = lib.pb_conv(fake, 30000, fake, 30000, 1, 30000, 30000, fake, 5, None)
    assert rc == -1 and b"exceeds LDS" in lib.pb_last_error()
    rc = lib.pb_hrf_cost(None, 0, None, 0, 1, 1, None, 1, 1, None, None)
    assert rc == -1
    # zero problems is a no-op, not an error
    assert lib.pb_fista_solve(fake, 300, 1, fake, 300, 0, 300, taps.ctypes.data, None, 4, 1.0, 1.0,
                              None, fake, 10, None, 0, 0, 0.0, 6, None, 0, None) == 0


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    from pybold_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU"):
        _lib.load()


def test_header_is_plain_c(tmp_path):
    """include/pybold_hip.h is what a cgo / JNI / ctypes binding reads: it must compile as C99
    on its own (no HIP, no C++), with every prototype visible."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "pybold_hip.h"\nint (*keep)(void) = pb_version;\nint main(void) { return keep == (int (*)(void))0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "include"), str(src)])

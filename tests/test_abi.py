"""CPU: the C-ABI library loads and exports every symbol include/iqhip.h declares; without a GPU
every entry point fails with a status instead of computing on the CPU."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, has_gpu


def header_functions():
    src = open(os.path.join(ROOT, "include", "iqhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(iqhip_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(pkg):
    lib = pkg.libiqhip()
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(pkg.IQHIP_SYMBOLS) == names
    assert lib.iqhip_abi_version() == 2


def test_struct_layout_matches_header(pkg):
    assert C.sizeof(pkg.NodeOp) == 56
    assert C.sizeof(pkg.BranchEnd) == 16


def test_host_library_loads(pkg):
    lib = pkg.libiqhost()
    for n in ("iqhost_create", "iqhost_compute_likelihood", "iqhost_compute_derv", "iqhost_set_kernel"):
        assert hasattr(lib, n)


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback_without_gpu(pkg):
    lib = pkg.libiqhip()
    e = C.c_void_p()
    rc = lib.iqhip_create(C.byref(e), 0, 4, 4, 100, 5)
    assert rc == 1 and not e.value  # IQHIP_ERR_NO_DEVICE
    assert b"no HIP device" in lib.iqhip_last_error()
    t = pkg.PhyloTree("((0:0.1,1:0.1):0.1,2:0.1,3:0.1);")
    import numpy as np
    t.set_alignment(4, 0, np.zeros((4, 10), dtype=np.uint8), np.ones(10))
    import importlib
    synth = importlib.import_module("iqtree_amd.synth")
    t.set_model(synth.gtr_model())
    with pytest.raises(pkg.HostError):
        t.attach_engine(0)
    with pytest.raises(pkg.HostError):
        t.compute_likelihood()  # HIP kernel selected but no engine: loud failure, no CPU path
    with pytest.raises(pkg.HostError):
        t.set_likelihood_kernel(pkg.LK_EIGEN_SSE)  # there is no CPU kernel to select


def test_unsupported_shapes_are_rejected(pkg):
    lib = pkg.libiqhip()
    e = C.c_void_p()
    assert lib.iqhip_create(C.byref(e), 0, 1, 4, 100, 5) == 3   # nstates outside 2 .. 64: UNSUPPORTED
    assert lib.iqhip_create(C.byref(e), 0, 65, 4, 100, 5) == 3  # (every count in between is embedded: tests/test_other_states_gpu.py)
    assert lib.iqhip_create(C.byref(e), 0, 3, 9, 100, 5) == 3   # 3 states run on the 4-state kernels: at most 8 categories
    assert lib.iqhip_create(C.byref(e), 0, 4, 4, 0, 5) == 2     # nptn 0: INVALID
    # a 4-state vector of 4 GiB or more would wrap the kernels' 32-bit per-lane offsets: refused, not corrupted
    assert lib.iqhip_create(C.byref(e), 0, 4, 4, (1 << 32) // (16 * 8), 5) == 3
    assert b"4 GiB" in lib.iqhip_last_error()
    assert lib.iqhip_create(C.byref(e), 0, 4, 4, 10, 1) == 2    # ntaxa 1: INVALID


def test_integration_file_parses_against_the_reference_headers():
    """integration/phylotree_hip.cpp -- the reference-side binding -- under g++ -std=gnu++98 -fsyntax-only against scratch
    copies of the reference's headers with the INTEGRATION.md hunks applied (tools/check_integration.sh; container only:
    /root/reference does not exist on the GPU box)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference"):
        pytest.skip("/root/reference absent")
    r = subprocess.run(["bash", os.path.join(root, "tools", "check_integration.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr

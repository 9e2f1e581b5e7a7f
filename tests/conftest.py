"""Shared test plumbing.

* `-m "not gpu"`: oracle vs the reference's known-answer tests and the committed golden vectors,
  host logic (plan/bank design, length rules, argument validation), and that the C-ABI library
  loads and exports every symbol include/hipsoxr.h declares.  No compute calls.
* `-m gpu`: the parity tests proper — HIP path (through the C ABI) vs the oracle.

Nothing here reads /root/reference at run time.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "python-soxr_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the product library and the oracle are built artefacts; build them if a fresh checkout lacks them
    lib = os.path.join(PKG, "soxr_amd", "libhipsoxr.so")
    ora = os.path.join(ROOT, "oracle", "liboracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__
        __graft_entry__.build()


def _init_torch_first():
    """On a GPU box, bring up torch's HIP context BEFORE libhipsoxr touches the device.  Observed on
    MI355X/ROCm 7.2: when torch's lazy CUDA init runs after another user of the same HIP runtime
    has already created streams and launched kernels, the first `.cuda()` occasionally stalls for
    minutes.  bench.py and smoke() initialise torch first as a matter of course; tests do it here."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda").cpu()
    except Exception:
        pass


def product_bank(in_rate, out_rate, recipe, vr):
    """The product's float64 bank through its C ABI (host only: hipsoxr_plan_create[_vr] +
    hipsoxr_plan_get_bank) — what oracle port-mode runs on, so that a bit-for-bit comparison tests
    the order of the arithmetic and nothing else.  The DESIGN is compared separately, against the
    oracle's own numpy design (tests/test_design_independent.py)."""
    from soxr_amd import device as dev
    return dev.Plan(in_rate, out_rate, int(recipe), vr=bool(vr)).bank()


def pytest_sessionstart(session):
    _init_torch_first()
    from oracle import oracle as o
    o.bank_provider = product_bank


def pytest_report_header(config):
    # the reference prints versions and engine ids in the header (tests/conftest.py:5-16)
    try:
        import soxr_amd
        from soxr_amd import _native
        return [f"soxr_amd {soxr_amd.__version__}, native {soxr_amd.__libsoxr_version__}, "
                f"HIP devices visible: {_native.device_count()}"]
    except Exception as e:  # pragma: no cover
        return [f"soxr_amd not importable: {e}"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def soxr():
    import soxr_amd
    return soxr_amd


def has_gpu():
    try:
        from soxr_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False

"""Install the product's float64 bank (host-only C-ABI calls) as the bank oracle port-mode runs on,
exactly as tests/conftest.py does for pytest runs: a bit-for-bit comparison then tests the ORDER of
the arithmetic; the design itself is compared in tests/test_design_independent.py."""
from oracle import oracle
from soxr_amd import device as _dev

oracle.bank_provider = lambda i, o, r, vr: _dev.Plan(i, o, int(r), vr=bool(vr)).bank()

"""GPU: randomised differential test against the oracle (tools/fuzz_vs_oracle.py) — random standard /
integer / float rate pairs, dtypes, recipes, lengths (including 0, 1, 2), channel counts, C and
Fortran layouts, one-shot and chunked streams; every case must be bit-identical."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases_bit_identical_to_oracle(seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_vs_oracle.py"), "150", str(seed)],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "150/150" in p.stdout

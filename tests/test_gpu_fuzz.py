"""GPU: randomised differential test against the oracle (tests/fuzz/fuzz_vs_oracle.py) — random standard /
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
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_vs_oracle.py"), "150", str(seed)],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "150/150" in p.stdout


def test_fft_engine_random_cases_within_tolerance():
    """tests/fuzz/fuzz_fft_engine.py: random standard ratios, sizes, channel counts and layouts through the
    device API (AUTO / FFT) against the float64 oracle, 1e-6 relative RMS and exact shapes."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_fft_engine.py"), "120", "21"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


def test_two_stage_random_cases_within_tolerance():
    """tests/fuzz/fuzz_two_stage.py: random integer / float / near-rational rate pairs, recipes, float32 / float64, channel
    counts, batches and layouts through the device API (AUTO / FFT) against the canonical-order engine: 1e-6 (float64:
    3e-9) relative RMS over every column, the ends included."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_two_stage.py"), "160", "41"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


def test_variable_rate_random_schedules_bit_identical():
    """tests/fuzz/fuzz_vr.py: random largest ratio, recipe, dtype, 1-3 channels, chunk sizes (up to 120 000 frames) and ratio changes (jumps and
    slews, also during a slew), against the oracle driven by tests/vr_sim.py — bit for bit per chunk."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_vr.py"), "200", "31"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


def test_device_api_exact_engine_random_cases_bit_identical():
    """tests/fuzz/fuzz_device_exact.py: random dtypes, ratios, batches, channel counts, layouts and
    explicit kernel choices through resample_tensor — bit-identical to the oracle port."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_device_exact.py"), "200", "41"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


def test_device_chunk_streams_equal_host_streams():
    """tests/fuzz/fuzz_device_stream.py: TensorStream (hipsoxr_stream_process_device) against ResampleStream on random rate
    pairs, recipes, dtypes, channel counts, constant and variable rate, chunk sizes 0 .. 30 000 — bit for bit per call."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_device_stream.py"), "120", "51"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]


def test_stream_groups_equal_streams_called_singly():
    """tests/fuzz/fuzz_stream_group.py: TensorStreamGroup (hipsoxr_streams_process_device: N independent handles, one launch)
    against the same handles as host-array streams — random rates, recipes, dtypes, channels, handle counts, chunk sizes."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_stream_group.py"), "30", "61"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]

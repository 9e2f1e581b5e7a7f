"""The library's HIPSOXR_* environment switches (csrc/device.h `Switches`) change WHICH kernel serves a job, never the
result: every switch, in a process of its own, against the default process — canonical-order results bit for bit,
frequency-domain results within the engine's 1e-6 of the exact engine (and bit for bit where the switch does not touch
that engine).  All but four names are compiled in only with -DHIPSOXR_DEBUG_SWITCHES: both processes load that build
(python-soxr_amd/_variants/dbg/, made by build.sh beside the product) through HIPSOXR_LIBRARY."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DBG_LIB = os.path.join(os.path.dirname(HERE), "python-soxr_amd", "_variants", "dbg", "libhipsoxr.so")
SWITCHES = ["HIPSOXR_NO_FFT", "HIPSOXR_FFT_NO_PAIR", "HIPSOXR_FFT_NO_CHPAIR", "HIPSOXR_FFT_NO_XCD_MAP",
            "HIPSOXR_FFT_LARGE_ONLY", "HIPSOXR_FFT_SMALL_ONLY", "HIPSOXR_FFT_NO_TINY", "HIPSOXR_FFT_NO_WAVE",
            "HIPSOXR_NO_PLANES", "HIPSOXR_NO_HOST_RING", "HIPSOXR_NO_CHAIN",
            "HIPSOXR_NO_DONE_WORDS", "HIPSOXR_RESIDENT", "HIPSOXR_AUTO_RESIDENT", "HIPSOXR_RESIDENT_NO_BAR", "HIPSOXR_NO_XCD_SPLIT", "HIPSOXR_NO_TILE_SPLIT",
            "HIPSOXR_NO_INTERP_TILE", "HIPSOXR_NO_INTERP_WAVE", "HIPSOXR_NO_GATHER_WAVE",
            "HIPSOXR_NO_INTERP_PAIR", "HIPSOXR_DEBUG_INTERP_PAIR_ALWAYS", "HIPSOXR_DEBUG_INTERP_NO_TWIN", "HIPSOXR_POLY_NO_PAIR", "HIPSOXR_NO_TWO_STAGE"]
EXACT_KEYS = ["host_f32", "host_i16", "host_interp", "host_interp_2ch", "stream_vr", "stream_20000", "dev_gather", "stream", "stream_resident", "stream_deferred", "dev_exact", "dev_exact_f64",
              "dev_exact_8ch", "dev_interp_tile"]


def _probe(env_extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("HIPSOXR_")}
    env.update(env_extra)
    assert os.path.exists(DBG_LIB), "build.sh makes the debug-switch build beside the product"
    env["HIPSOXR_LIBRARY"] = DBG_LIB
    r = subprocess.run([sys.executable, os.path.join(HERE, "_switch_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("SWITCH_PROBE ")][-1]
    return json.loads(line[len("SWITCH_PROBE "):])


@pytest.fixture(scope="module")
def baseline():
    b = _probe({})
    assert b["stream"] == b["stream_resident"] == b["stream_deferred"]      # the three stream modes agree to begin with
    for k in ("fft_batch", "fft_8ch", "fft_large", "two_stage_mono", "two_stage_2ch"):
        assert 0 < b[k] <= 1e-6, (k, b[k])
    return b


@pytest.mark.parametrize("switch", SWITCHES)
def test_switch_does_not_change_results(baseline, switch):
    name, _, val = switch.partition("=")
    got = _probe({name: val or "1"})
    for k in EXACT_KEYS:
        assert got[k] == baseline[k], (switch, k)
    for k in ("fft_batch", "fft_8ch", "fft_large"):
        if switch == "HIPSOXR_NO_FFT":
            assert got[k] == 0.0, (switch, k)          # AUTO stays on the exact engine
        else:
            assert 0 < got[k] <= 1e-6, (switch, k, got[k])
    for k in ("two_stage_mono", "two_stage_2ch"):       # the two-stage form (its polyphase stage on pairs or not) stays in its class
        if switch in ("HIPSOXR_NO_FFT", "HIPSOXR_NO_TWO_STAGE"):
            assert got[k] == 0.0, (switch, k)
        else:                                            # (0: a switch took the FFT stage's kernel away and AUTO fell back to the exact engine)
            assert got[k] <= 1e-6, (switch, k, got[k])
            assert got[k] > 0 or switch.startswith("HIPSOXR_FFT_"), (switch, k)
    # switches that only re-route the SAME transform chain of the large unit-stride job leave it bit-identical
    if switch in ("HIPSOXR_FFT_NO_CHPAIR", "HIPSOXR_FFT_NO_XCD_MAP",
                  "HIPSOXR_NO_PLANES", "HIPSOXR_NO_CHAIN", "HIPSOXR_RESIDENT", "HIPSOXR_FFT_LARGE_ONLY", "HIPSOXR_FFT_NO_TINY"):
        assert got["fft_large_sha"] == baseline["fft_large_sha"], switch

"""The line bench.py prints must survive the driver's record, which keeps the last ~8 KB of stdout + stderr: <= 6144
bytes, contract keys + `roofline` + `cpu_baseline` present, the roofline-bearing legs LAST (ask 3).
Dry run of the formatter on the largest full record committed so far (round 5's 12.4 KB line)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_line_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _full_record():
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
        d = json.load(f)
    assert len(json.dumps(d)) > 12000   # the record that outgrew the driver's tail
    return d


def test_line_fits_the_drivers_tail_and_keeps_every_roofline_leg():
    b = _bench()
    d = _full_record()
    # legs added after round 5 (same shape as what main() adds)
    d["arith_f64"]["throughput"] = {"us_per_launch": 190.123456, "frac": 0.310123456, "read_frac": 0.16, "power_W": 1388.123, "sclk_MHz": 2011.5,
                                    "energy_mJ_per_launch": 263.9, "note": "x" * 300}
    line = b.compact_line(d)
    assert len(line) <= 6144 == b.LINE_BUDGET, len(line)
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    keys = list(c)
    tail = [k for k in b._TAIL_KEYS if k in c]
    assert keys[-len(tail):] == tail and tail[-3:] == ["configs2", "batch_shard", "throughput_roofline"]
    # the legs the round-5 record lost are whole, with their figures
    assert c["configs2"]["roofline"]["frac"] == float("%.5g" % d["configs2"]["roofline"]["frac"])
    assert c["batch_shard"]["roofline"]["launch_us"] > 0 and c["batch_shard"]["sustained"]["us_per_launch"] > 0
    assert c["batch_strong"]["value"] > 0 and c["exact_engine"]["launch_us"] > 0 and c["arbitrary_ratio"]["launch_us"] > 0
    assert c["arith_f64"]["throughput"]["frac"] == 0.31012
    assert "note" not in json.dumps(c)


def test_line_sheds_named_context_legs_rather_than_overflow():
    b = _bench()
    d = _full_record()
    d["host_batch"]["pad"] = {"k%d" % i: float(i) + 0.123456789 for i in range(400)}
    line = b.compact_line(d)
    c = json.loads(line)
    assert len(line) <= b.LINE_BUDGET
    assert "host_batch" in c["shed_for_line_budget"] and "host_batch" not in c
    assert list(c)[-1] == "throughput_roofline" and "roofline" in c and "cpu_baseline" in c


def test_contract_values_are_not_rounded_away():
    b = _bench()
    d = _full_record()
    c = json.loads(b.compact_line(d))
    assert abs(c["value"] / d["value"] - 1) < 1e-4 and abs(c["ms_per_step"] / d["ms_per_step"] - 1) < 1e-4
    assert c["steps"] == d["steps"] and c["warmup"] == d["warmup"] and c["n_gpus"] == d["n_gpus"]

"""bench.py's output contract (CPU): the driver parses the LAST stdout line; it must be one strict-JSON record below 6 KB that carries
the headline with its `roofline` and `cpu_baseline` (round 4's single 20.85-KB line was not parsed: BENCH_r04.parsed = null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _strict(line):
    def bad(tok):
        raise ValueError("non-finite token " + tok)
    return json.loads(line, parse_constant=bad)


def _check(line):
    assert "\n" not in line and len(line) < 6000, len(line)
    c = _strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity", "first_solve_ms", "repeat_identical"):
        assert k in c, k
    assert "workload" in c["config"] and "model" not in c["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us"):
        assert k in c["roofline"], k
    assert c["roofline"]["bound"] in ("hbm", "mfma") and c["roofline"]["unit"] in ("GB/s", "TFLOP/s")
    for k in ("value", "unit", "cores", "kind", "nproc", "cpu", "sample"):
        assert k in c["cpu_baseline"], k
    for k in ("ok", "dt_m", "dr_rad"):
        assert k in c["parity"], k
    for blk in ("secondary", "c4_1gpu", "online_c5", "batched"):
        assert blk in c and "value" in c[blk] or blk == "online_c5" and "wall_s" in c[blk], blk
    assert c["secondary"]["roofline"]["frac"] and c["secondary"]["cpu_baseline"]["value"]
    assert c["c4_1gpu"]["roofline"]["frac"] and c["c4_1gpu"]["cpu_baseline"]["value"] and "ms_per_solve" in c["c4_1gpu"]
    assert c["online_c5"]["roofline"]["frac"] and c["online_c5"]["cpu_baseline"]["value"]
    txt = json.dumps(c)
    assert '"note"' not in txt and "NaN" not in txt and "Infinity" not in txt
    return c


def test_compact_record_of_a_full_round4_record():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "bench_r04c.json")))
    assert len(json.dumps(full)) > 15000                      # the record that did not fit
    _check(bench.compact_record(full))


def test_compact_record_is_strict_json_whatever_the_numbers():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "bench_r04c.json")))
    full["value"] = float("nan"); full["roofline"]["achieved"] = float("inf"); full["parity"]["dt_m"] = float("-inf")
    full["online_c5"]["cpu_baseline"]["pose_difference_at_that_point"]["dt_m"] = None
    c = _strict(bench.compact_record(full))
    assert c["value"] is None and c["roofline"]["achieved"] is None and c["parity"]["dt_m"] is None


def test_emit_prints_one_compact_line_on_stdout(capsys, tmp_path, monkeypatch):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = json.load(open(os.path.join(ROOT, "profiles", "bench_r04c.json")))
    bench.emit(full)
    cap = capsys.readouterr()
    lines = [l for l in cap.out.splitlines() if l.strip()]
    assert len(lines) == 1
    _check(lines[0])
    assert json.loads(cap.err.strip().splitlines()[-1])["online_c5"]["metric"]            # the full record is on stderr ...
    assert json.load(open(tmp_path / "gpurun_out" / "bench_full.json"))["c4_1gpu"]        # ... and in gpurun_out/


def test_every_traffic_key_the_bench_reads_is_in_traffic_json():
    """`roofline.traffic` comes from profiles/traffic.json (PMC passes of profiles/collect.sh): a kernel renamed since the last collection
    (a template argument more) silently drops its key there and the bench reports `traffic: null`."""
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    keys = set(re.findall(r'traffic_of\((.*?), ', src))                      # the key argument: one name, or `"a" if cond else "b"`
    keys = set(k for arg in keys for k in re.findall(r'"([a-z0-9_]+_bytes_per_launch)"', arg))
    assert len(keys) >= 8, keys
    have = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    missing = sorted(k for k in keys if not (isinstance(have.get(k), (int, float)) and have[k] > 0))
    assert not missing, missing

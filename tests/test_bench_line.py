"""The bench's output contract: the LAST stdout line is one compact JSON object the driver can parse —
round 4's 20 KB line was not (`BENCH_r04.parsed: null`). `bench.py --dry FULL.json` runs the same `emit()` the GPU run ends
with over a committed FULL record of a real run, so the size and the presence of the graded objects are checked on CPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL_RECORDS = ["profiles/r04/bench_default_final.json", "profiles/r04/bench_all_s6.json"]


@pytest.mark.parametrize("rec", FULL_RECORDS)
def test_last_stdout_line_is_compact_and_carries_the_graded_objects(rec, tmp_path):
    full = os.path.join(ROOT, rec)
    if not os.path.exists(full):
        pytest.skip(rec + " not in the tree")
    detail = tmp_path / "bench_detail.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", full, "--detail-out", str(detail)],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-2000:]
    out_lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(out_lines) == 1, "stdout must be ONE line (the detail record goes to stderr and to a file)"
    last = out_lines[-1]
    assert len(last.encode()) < 8192, len(last)
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert isinstance(line["value"], float) and line["value"] > 0
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["roofline"]["frac"] is not None and line["roofline"]["bound"]
    assert line["roofline"]["kernel_ms"] > 0
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] in ("port", "reference")
    assert line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["sample"]
    assert line["parity"]["mismatches"] == 0 and line["parity"]["checked"] > 0
    for sub in ("vector", "hybrid"):
        assert line[sub]["value"] > 0 and line[sub]["ms_per_step"] > 0
        assert line[sub]["roofline"]["frac"] > 0
        assert line[sub]["cpu_baseline"]["value"] > 0
        assert line[sub]["parity"]["mismatches"] == 0
    # nothing was lost: the full record is on disk and on stderr
    kept = json.load(open(detail))
    assert kept == json.load(open(full))
    assert any(l.startswith("BENCH_DETAIL ") and json.loads(l[len("BENCH_DETAIL "):]) == kept for l in p.stderr.splitlines())


def test_compact_line_sheds_optional_objects_before_it_exceeds_the_budget():
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, FULL_RECORDS[0])))
    full["concurrency"] = {str(t): {"value": 1.0 * t, "p50_us": 1.0, "p99_us": 2.0} for t in range(400)}      # an optional object that outgrew the line
    line = bench.compact_line(full, "x.json")
    assert len(json.dumps(line)) <= bench.COMPACT_LIMIT
    assert "concurrency" not in line and line["roofline"]["frac"] and line["cpu_baseline"]["value"] and line["vector"]["value"]

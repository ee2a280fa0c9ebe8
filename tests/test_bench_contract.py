"""bench.py keeps the driver's contract: one JSON line with the agreed keys (checked in-process on a 2-layer model)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_contract_line(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "5", "--warmup", "1", "--layers", "2",
                                      "--cpu-seconds", "1"])
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "tokens/s" and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    w = d["with_incoherence_wrapper"]
    assert "error" not in w and 0 < w["value"] < d["value"]

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
    oc = d["other_configs"]  # BASELINE configs[2..4] in the driver-timed line (2-layer models here)
    assert set(oc) == {"llama3.1-8b_mem3p25", "llama3.1-8b_figure1d", "llama3.1-8b_figure1c", "llama3.1-70b_tcq_6_bs1", "llama3.1-70b_tcq_6_bs16",
                       "llama3.1-8b_tcomb_6_7_bs64", "llama3.1-8b_tcomb_6_7_bs128"}  # + the batched path on the headline model
    for key, fig in oc.items():
        assert "error" not in fig, (key, fig)
        assert fig["unit"] == "tokens/s" and fig["value"] > 0 and fig["ms_per_step"] > 0 and 0 < fig["frac"] < 1 and fig["steps"] <= 10
        assert fig["batch"] == (int(key.rsplit("_bs", 1)[1]) if "_bs" in key else 1) and fig["layers"] == 2
    rs = d["config"]["ranks_seen"]
    assert rs["world_size"] == 1 and rs["distinct_devices"] == 1 and rs["ranks"][0]["rank"] == 0 and rs["ranks"][0]["pci_bus_id"]
    wm = d["whole_model_decode"]  # the reference's own metric (whole-model decode step), measured after the headline
    assert "error" not in wm and wm["unit"] == "tokens/s" and 0 < wm["value"] < d["value"] and wm["launches_per_token"] == 2 * 5 + 1  # q|k|v, attention, o, gate|up, down (its rotation inside) per layer + the lm_head launch


def test_gpus_n_without_a_launcher_spawns_the_ranks(tmp_path, monkeypatch):
    """`python bench.py --gpus N` with WORLD_SIZE unset (how the driver calls it) must start its N ranks itself, as child
    processes, before anything touches a GPU.  CPU check of exactly that code path: the children are this test's stand-in
    script (bench.spawn_ranks starts `sys.executable <bench.__file__> <argv>`), which records the environment it was given."""
    sys.path.insert(0, ROOT)
    import bench
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import json, os, sys\n"
        "out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'rank%s.json' % os.environ['RANK'])\n"
        "json.dump({k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', "
        "'HSA_ENABLE_IPC_MODE_LEGACY')} | {'argv': sys.argv[1:]}, open(out, 'w'))\n"
        "print('{\"rank\": %s}' % os.environ['RANK'])\n"
        "sys.exit(3 if os.environ['RANK'] == '1' and '--fail' in sys.argv else 0)\n")
    monkeypatch.setattr(bench, "__file__", str(probe))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--steps", "2"])
    for var in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(var, raising=False)
    assert bench.spawn_ranks(3) == 0
    got = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(3)]
    assert [g["RANK"] for g in got] == ["0", "1", "2"] and [g["LOCAL_RANK"] for g in got] == ["0", "1", "2"]
    assert all(g["WORLD_SIZE"] == "3" and g["MASTER_ADDR"] == "127.0.0.1" and g["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for g in got)
    assert len({g["MASTER_PORT"] for g in got}) == 1 and all(g["argv"] == ["--gpus", "3", "--steps", "2"] for g in got)
    # a failing rank fails the whole run
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--fail"])
    assert bench.spawn_ranks(2) == 3
    # a rank 0 that parked its headline and then died inside the tp leg: the parent prints the parked line, marked, and the run
    # FAILS with exit code 75 (round 5: a multi-GPU leg that lost a rank must show in the exit code); --lenient-exit: 0
    probe.write_text(
        "import json, os, sys\n"
        "if os.environ['RANK'] == '0':\n"
        "    json.dump({'metric': 'm', 'value': 1.5}, open(os.environ['QPAL_BENCH_HEADLINE_FILE'], 'w'))\n"
        "    os._exit(9)\n")
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = bench.spawn_ranks(2)
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    assert rc == bench.TP_ABANDONED_RC == 75 and line["value"] == 1.5 and "died inside the leg" in line["tp_70b"]["error"]
    assert line["tp_70b_abandoned"] is True
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--lenient-exit"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert bench.spawn_ranks(2) == 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    # and main() takes that path only when no launcher set WORLD_SIZE
    probe.write_text("import sys\nsys.exit(0)\n")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0


@pytest.mark.gpu
def test_gpus_2_rehearsed_on_one_gpu_through_the_spawn_path():
    """`python bench.py --gpus 2 --force-device 0 --dist-backend gloo` — no launcher — on the one-GPU box: the parent spawns two
    ranks that share the card, rank 0 prints ONE line: dp replicas as the headline and the tp_70b leg (row-sharded 70B shapes,
    batch 1 and 16, peer gather validated against the collective, then timed inside a HIP graph)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--force-device", "0", "--dist-backend", "gloo",
           "--steps", "4", "--warmup", "1", "--layers", "2", "--tp-layers", "2", "--no-kind-breakdown"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "dp2" in d["config"]["parallelism"]
    tp = d["tp_70b"]
    assert "error" not in tp and tp["world"] == 2
    assert tp["peer_gather"]["validated_against_collective"] is True and tp["peer_gather"]["flag_memory"] in ("fine-grained", "uncached", "coarse-grained")
    assert tp["peer_gather"]["buffer_memory"] in ("fine-grained", "uncached", "coarse-grained")
    rs = d["config"]["ranks_seen"]  # two ranks, here on ONE card (a real node shows two distinct devices)
    assert rs["backend"] == "gloo" and rs["world_size"] == 2 and [r_["rank"] for r_ in rs["ranks"]] == [0, 1] and rs["distinct_devices"] == 1
    assert len({r_["pid"] for r_ in rs["ranks"]}) == 2
    for key in ("bs1", "bs16"):
        fig = tp[key]
        assert fig["value"] > 0 and "tokens_per_s" in fig["collective_eager"] and "tokens_per_s" in fig["peer_gather_in_graph"]


@pytest.mark.gpu
def test_headline_line_survives_an_abandoned_tp_leg():
    """The tp_70b leg runs after the headline's timed region and contains collectives: if it cannot complete (a rank fails, a
    collective hangs) the watchdog must still get rank 0's ONE line out, with the leg marked as an error — and the run must FAIL
    with exit code 75 (round 5: a hang on a multi-GPU node has to be visible in the driver's rc), every rank saying on stderr where
    it stood (stage of the leg, peer-gather slot / epoch view); --lenient-exit restores exit code 0."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--force-device", "0", "--dist-backend", "gloo",
           "--steps", "4", "--warmup", "1", "--layers", "2", "--tp-layers", "2", "--no-kind-breakdown", "--tp-timeout", "0.05"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 75, (r.returncode, r.stderr[-2000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "error" in d["tp_70b"] and d["tp_70b_abandoned"] is True
    assert "[bench rank 0] tp_70b leg abandoned" in r.stderr and "[bench rank 1] tp_70b leg abandoned" in r.stderr
    r = subprocess.run(cmd + ["--lenient-exit"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1

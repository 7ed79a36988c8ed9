"""bench.py's LAST stdout line is what the driver parses (BENCH_rNN.parsed).  Round 4's line grew to 24 kB and was not
parsed; the headline is now a separate compact object and every detail block goes to bench_detail.json / an earlier line."""
import io
import json
import os
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = ["r05_bench_20steps.json", "r04_bench_20steps.json", "r03_bench_20steps.json"]


@pytest.mark.parametrize("record", RECORDS)
def test_headline_is_compact_and_complete(record):
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", record)))
    line = bench.headline(full)
    assert "\n" not in line and len(line) < 4096, len(line)
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in h, k
    assert h["value"] == pytest.approx(full["value"], rel=1e-5) and h["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert set(h["config"]) >= {"workload", "global_batch", "seq_len", "n_labels", "parallelism"}
    assert "model" not in h["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_ms_per_launch"):
        assert k in h["roofline"], k
    assert h["roofline"]["frac"] == pytest.approx(h["roofline"]["achieved"] / h["roofline"]["peak"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in h["cpu_baseline"], k
    # the detail blocks stay out of the headline
    assert "kernels" not in h and "stages" not in h
    for name, m in h.get("modes", {}).items():
        assert set(m) <= {"value", "ms_per_step", "dtype", "roofline_frac"}, (name, m)


def test_r05_record_carries_the_round5_blocks():
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_20steps.json")))
    h = json.loads(bench.headline(full))
    assert len(h["build_hash"]) == 16 and h["roofline"]["traffic_stale"] is False
    for k in ("fast_mode", "amp_backward.forward_bf16x3", "forward_only.f32", "zero_shot.f32.GO-2019", "one_hidden_layer.train"):
        assert k in h["modes"], k
    valu = [v for v in full["one_hidden_layer"]["train"]["stages"].values() if v["bound"] == "valu"]
    assert len(valu) == 2 and all(0.0 < v["frac"] < 1.0 for v in valu)  # priced against the packed-f32 vector rate


def test_headline_sheds_modes_rather_than_overflow():
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_20steps.json")))
    full["zero_shot"]["f32"] = {f"table{i} (x)": dict(next(iter(full["zero_shot"]["f32"].values()))) for i in range(80)}
    line = bench.headline(full)
    assert len(line) <= bench.HEADLINE_MAX_BYTES and "modes" not in json.loads(line)


def test_emit_prints_detail_first_and_headline_last(tmp_path, monkeypatch):
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", RECORDS[0])))
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "bench_detail.json"))
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(full)
    lines = buf.getvalue().strip().splitlines()
    assert len(lines) == 2 and len(lines[-1]) < 4096
    assert json.loads(lines[-1])["metric"] == full["metric"]
    detail = json.loads(lines[0])["bench_detail"]
    assert detail["kernels"] == full["kernels"] and detail["stages"] == full["stages"]
    assert json.load(open(tmp_path / "bench_detail.json"))["kernels"] == full["kernels"]


def test_headline_survives_broken_optional_blocks():
    """an optional block that lacks a key (an empty forward_only entry, a zero_shot table without `roofline` or
    `seconds`, a cpu_baseline in which no leg finished) must cost that block, never the contract keys."""
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", RECORDS[0])))
    full["forward_only"]["bf16"] = {}
    go = next(iter(full["zero_shot"]["f32"]))
    full["zero_shot"]["f32"][go].pop("roofline")
    full["zero_shot"]["bf16x3"][go].pop("seconds")
    full["fast_mode"] = {"value": "n/a"}
    full["amp_full"] = {"value": 7.5e6, "ms_per_step": 1100.0, "dtype": "x" * 30, "roofline": {"frac": 0.4}}
    full["cpu_baseline"] = {"value": None, "unit": "protein-label pairs/s", "cores": None, "kind": "port", "legs": {},
                            "sample": "no leg finished"}
    h = json.loads(bench.headline(full))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "roofline", "cpu_baseline", "config"):
        assert k in h, k
    assert h["modes"]["amp_full"]["roofline_frac"] == 0.4 and "forward_only.bf16" not in h["modes"]


def test_cpu_baseline_legs_and_budget(monkeypatch):
    """cpu_baseline: legs at one / two (/ four, budget permitting) times the CPUs the process can really use (affinity AND cgroup
    quota - the GPU boxes show 256 hardware threads under a 16-CPU quota), B = 2 at the first two thread counts, then B = 4 at
    the faster, in ONE child under ONE budget; no leg that ran is null; a child cut off by the budget still reports what it
    finished.  The oracle step is stubbed: the real one takes ~10-20 s per sample."""
    import bench

    assert bench.cpu_leg_threads(16) == [16, 32, 64] and bench.cpu_leg_threads(1) == [1, 2, 4]
    eff, quota = bench.effective_cpus()
    assert 1 <= eff <= (bench.os.cpu_count() or 1)

    class FakeProc:
        mode = "all"

        def __init__(self, cmd, **kw):
            pass

        def lines(self):
            out = [{"threads": 16, "B": 2, "pairs": 64204, "seconds": 12.9}, {"threads": 32, "B": 2, "pairs": 64204, "seconds": 12.6},
                   {"threads": 32, "B": 4, "pairs": 128408, "seconds": 21.0}, {"threads": 64, "skipped": "budget"}]
            if FakeProc.mode == "cut":
                out = out[:1]
            return "\n".join(json.dumps(o) for o in out) + "\n"

        def communicate(self, timeout=None):
            import subprocess

            if FakeProc.mode == "cut" and not getattr(self, "killed", False):
                raise subprocess.TimeoutExpired("x", timeout)
            return self.lines(), None

        def kill(self):
            self.killed = True

    monkeypatch.setattr(bench, "effective_cpus", lambda: (16, 16.0))
    monkeypatch.setattr(bench.os, "cpu_count", lambda: 256)
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    out = bench.cpu_baseline(budget=10.0)
    assert set(out["legs"]) == {"16", "32", "32 (B=4)"} and all(v["value"] for v in out["legs"].values())
    assert out["legs_not_started"] == [64]
    assert out["cores"] == 32 and out["value"] == pytest.approx(128408 / 21.0) and out["kind"] == "port"
    assert out["usable_cpus"] == 16 and out["host_cores"] == 256 and "B=4" in out["sample"] and "32 threads" in out["sample"]
    FakeProc.mode = "cut"   # the budget ends during the second leg: the finished leg stands and is the value
    out = bench.cpu_baseline(budget=10.0)
    assert out["legs"]["32"]["value"] is None and "budget" in out["legs"]["32"]["note"]
    assert out["cores"] == 16 and out["value"] == pytest.approx(64204 / 12.9) and "B=2" in out["sample"]


def test_cpu_legs_child_orders_its_samples(monkeypatch, capsys):
    """The child: B = 2 at the first two thread counts, B = 4 at the faster, the optional count only if the budget holds it."""
    import bench

    calls = []

    def fake(threads, B=None):
        calls.append((threads, B))
        return (B or 4) * 32102, {16: 0.02, 32: 0.01, 64: 0.01}[threads]

    monkeypatch.setattr(bench, "_cpu_sample", fake)
    bench._cpu_legs_main([16, 32, 64], 100.0)
    assert calls == [(16, 2), (32, 2), (32, 4), (64, 2)]
    calls.clear()
    bench._cpu_legs_main([16, 32, 64], 0.0)
    assert calls == [(16, 2), (32, 2), (32, 4)]
    lines = [json.loads(ln) for ln in capsys.readouterr().out.strip().splitlines()]
    assert lines[-1] == {"threads": 64, "skipped": "budget"}

"""`python bench.py --gpus N` started WITHOUT torchrun must become N rank processes.

CPU rehearsal of exactly that command path: the parent (no RANK/WORLD_SIZE in its environment) goes through
pcfa_amd.launch.spawn_ranks, the ranks rendezvous on 127.0.0.1 over gloo, every rank times its own pair, rank 0
prints ONE JSON line with n_gpus == N, one step time per rank and the universal leg's all-reduce count.  The ranks
drive the CPU port (`--rehearse-cpu`: the objects of bench.py's cpu_baseline leg) because there is no GPU here; the
same launcher with real GPU ranks is tests/test_gpu_parity.py::test_bench_gpus2_spawns_two_ranks_on_one_gpu.
"""
import json
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANK_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PCFA_SPAWNED_RANK")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in RANK_KEYS}
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_bench_gpus2_without_torchrun_starts_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--rehearse-cpu"], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout            # ONE JSON line on stdout, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rehearsal"] == "cpu-port"
    assert len(out["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in out["per_rank_ms_per_step"])
    assert out["ms_per_step"] >= max(out["per_rank_ms_per_step"]) * 0.999       # max over ranks, barrier included
    assert abs(out["value"] - 2 * out["steps"] / (out["ms_per_step"] * 1e-3 * out["steps"])) < 1e-6 * out["value"]
    u = out["universal"]                        # the N > 1 run exercises the all-reduce path
    assert u["global_batch"] == 2 and u["allreduces_per_closure"] == 1.0
    assert u["allreduce_bytes"] == (2 * 3 * 64 * 64 + 1) * 4


def _strict_json(text):
    def bad(c):
        raise ValueError("non-strict JSON constant %r" % c)
    return json.loads(text, parse_constant=bad)       # NaN / Infinity are not JSON: the driver's parser may refuse them


def test_bench_line_is_short_strict_json_and_detail_file_is_written(tmp_path):
    """the driver could not parse r04's 22 KB line.  The line stays under 6000 bytes, is strict JSON
    and names the file that holds everything else."""
    import bench
    env = _clean_env()
    env["PCFA_BENCH_DETAIL"] = str(tmp_path / "detail.json")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--rehearse-cpu"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    text = r.stdout.strip()
    assert len(text) < bench.LINE_BUDGET and "\n" not in text
    out = _strict_json(text)
    assert out["detail"] == bench.DETAIL_FILE
    detail = json.load(open(env["PCFA_BENCH_DETAIL"]))
    assert detail["metric"] == out["metric"] and abs(detail["value"] - out["value"]) <= 1e-5 * detail["value"]


def test_compact_line_keeps_the_contract_and_sheds_the_rest():
    """The r04 record (22 KB, the one the driver lost) through compact_line: every contract key + roofline +
    cpu_baseline survive, the bulky sections do not; an absurdly large record still fits by shedding optional legs."""
    import bench
    full = json.load(open(os.path.join(REPO, "profiles", "r04_bench_n1_driver_settings.json")))
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < bench.LINE_BUDGET
    _strict_json(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(line["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert abs(line["value"] - full["value"]) < 1e-5 * full["value"]
    for k in ("kernel_families", "kernels", "lbfgs", "per_pair_setup_s", "kernels_eager_step_hip_events"):
        assert k not in line
    full["config"]["workload"] = full["config"]["workload"] + " x" * 1200       # pathological: optional legs are shed
    full["pwcnet"]["final"] = {"k%d" % i: float(i) for i in range(300)}
    shed = bench.compact_line(full)
    assert len(json.dumps(shed)) < bench.LINE_BUDGET and "roofline" in shed and "cpu_baseline" in shed
    nan = dict(full, value=float("nan"))
    _strict_json(json.dumps(bench.compact_line(nan)))                            # NaN becomes null, never `NaN`


def test_bench_gpus1_does_not_spawn():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                        "--rehearse-cpu"], env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 1 and "universal" not in out and len(out["per_rank_ms_per_step"]) == 1


def test_spawn_ranks_env_and_failure_propagation(tmp_path):
    from pcfa_amd import launch
    ok = tmp_path / "ok.py"
    ok.write_text(textwrap.dedent("""
        import os, sys
        import torch.distributed as dist
        import torch
        dist.init_process_group("gloo")
        t = torch.tensor([float(os.environ["RANK"]) + 1.0])
        dist.all_reduce(t)
        assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
        open(sys.argv[1] + os.environ["RANK"], "w").write("%d %g" % (dist.get_world_size(), t.item()))
        dist.destroy_process_group()
    """))
    assert launch.spawn_ranks([str(ok), str(tmp_path / "r")], 3, env=_clean_env()) == 0
    for r in range(3):
        assert (tmp_path / ("r%d" % r)).read_text() == "3 6"

    bad = tmp_path / "bad.py"     # rank 1 dies; rank 0 would otherwise sleep for a minute
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n")
    import time
    t0 = time.monotonic()
    assert launch.spawn_ranks([str(bad)], 2, env=_clean_env()) == 7
    assert time.monotonic() - t0 < 30


def test_spawn_ranks_refuses_inside_a_rank(monkeypatch):
    import pytest
    from pcfa_amd import launch
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert launch.inside_rank()
    with pytest.raises(RuntimeError):
        launch.spawn_ranks(["x.py"], 2)


def test_extra_leg_guard_prints_once_whichever_side_is_first():
    """bench.ExtraLegGuard: at N > 1 the universal leg runs last and under a deadline, so that a rank stuck in a collective
    cannot keep the headline line from being printed.  Whoever comes first -- the leg or the deadline -- publishes, once."""
    import threading
    import time
    import bench
    got, left = [], threading.Event()
    g = bench.ExtraLegGuard(30.0, got.append, leave=left.set)
    assert g.publish({"value": 1.0}) is True and g.publish({"value": 2.0}) is False
    g.cancel()
    assert got == [{"value": 1.0}] and not left.is_set()

    got2, left2 = [], threading.Event()
    g2 = bench.ExtraLegGuard(0.05, got2.append, leave=left2.set)
    assert left2.wait(5.0)                        # the deadline passed: published the reason, then asked to leave
    assert len(got2) == 1 and "error" in got2[0] and "headline" in got2[0]["error"]
    assert g2.publish({"value": 3.0}) is False    # a leg that returns late does not print a second line
    time.sleep(0.01)
    assert len(got2) == 1

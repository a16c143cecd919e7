"""bench.py starts and verifies its own ranks when no launcher is around it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out: bytes):
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_spawns_the_ranks_it_is_asked_for():
    # no RANK in the environment: the parent starts two ranks itself (gloo, no device work)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_env(EPI_BENCH_BACKEND="gloo"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    res = _json_line(p.stdout)
    assert res == {"launch_check": True, "n_gpus": 2, "world_size": 2, "backend": "gloo"}


def test_bench_refuses_a_process_group_of_the_wrong_size():
    # a launcher that started one rank although --gpus says two: every rank must fail, nothing may be printed
    env = _env(EPI_BENCH_BACKEND="gloo", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0
    assert b"process group has 1 rank" in p.stderr
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_two_ranks_on_the_box_report_their_world_size():
    # two ranks share the box's one GPU and exchange through gloo (RCCL wants one device per rank): the whole sharded
    # bench path with a process group of two, started by bench.py itself
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--particles", "20000", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       env=_env(EPI_BENCH_BACKEND="gloo", EPI_BENCH_NODE_DEVICES="0,0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    res = _json_line(p.stdout)
    assert res["n_gpus"] == 2 and res["world_size"] == 2 and res["backend"] == "gloo"
    assert res["config"]["particles"] == 20000 and res["value"] > 0
    assert res["diagnostics"]["flags"] == 0
    # the same update through epi_mctx_create + epi_mpf_update, timed in a child process of rank 0: two shards on the one
    # device here (no RCCL between shards of one device), the same population, so the same diagnostics
    nf = res["node_filter"]
    assert "error" not in nf, nf
    assert set(nf) >= {"ms_per_step", "value", "moved", "overlapped_updates", "rccl_devices", "shards", "devices", "diagnostics"}
    assert nf["shards"] == 2 and nf["devices"] == [0, 0] and nf["rccl_devices"] == 0
    assert nf["ms_per_step"] > 0 and nf["value"] > 0 and nf["moved"] >= 0 and 0 <= nf["overlapped_updates"] <= 2
    assert nf["diagnostics"]["flags"] == 0
    assert abs(nf["diagnostics"]["loglik"] - res["diagnostics"]["loglik"]) <= 1e-9 * abs(res["diagnostics"]["loglik"])
    assert abs(nf["diagnostics"]["ess"] - res["diagnostics"]["ess"]) <= 1e-9 * res["diagnostics"]["ess"]


def test_a_failing_node_filter_child_is_reported_inside_the_block():
    # no device here (or none numbered 97): the child fails, the block says so, nothing is raised
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    saved = os.environ.get("EPI_BENCH_NODE_DEVICES")
    os.environ["EPI_BENCH_NODE_DEVICES"] = "97,98"
    try:
        nf = bench.node_filter_block(argparse.Namespace(gpus=2, particles=1000, steps=1, warmup=0))
    finally:
        if saved is None:
            del os.environ["EPI_BENCH_NODE_DEVICES"]
        else:
            os.environ["EPI_BENCH_NODE_DEVICES"] = saved
    assert "error" in nf and "ms_per_step" not in nf


@pytest.mark.gpu
def test_bench_single_gpu_line_has_the_contract_fields():
    """The default (one GPU) form at a small population: one JSON line with the fields the driver reads, the roofline of
    the step kernel from the run's own HIP-event time, and the CPU baseline timed on this host by the oracle."""
    p = subprocess.run([sys.executable, BENCH, "--particles", "300000", "--steps", "3", "--warmup", "1"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    res = _json_line(p.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in res, key
    assert res["n_gpus"] == 1 and res["steps"] == 3 and res["warmup"] == 1 and res["higher_is_better"] is True
    assert res["unit"] == "particle-steps/s" and res["vs_baseline"] is None and res["data"] == "synthetic"
    assert "workload" in res["config"] and res["config"]["particles"] == 300000 and "model" not in res["config"]
    assert abs(res["value"] - 300000 * 3 / (res["ms_per_step"] * 3e-3)) <= 1e-6 * res["value"]
    r = res["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"].startswith("k_step_coop<true")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["alg_bytes_per_particle_step"] * 300000 / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None or "committed" in r["traffic_source"]
    assert 0 < r["kernel_ms"] <= res["ms_per_step"]
    c = res["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "particle-steps/s" and "updates" in c["sample"]
    assert res["value"] > c["value"]
    assert res["diagnostics"]["flags"] == 0

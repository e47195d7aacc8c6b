"""bench.py end to end on the one-GPU box: the plain command (all workloads, roofline against the in-run peak) at a reduced size, and
--mode group with one rank (the in-library group path; with one device there is no transport to exercise: `table_transport` = "none")."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize("wl", ["mul", "mul_enc", "mul_base"])
def test_bench_line_of_one_workload(wl):
    line = _run("--workload", wl, "--steps", "3", "--warmup", "1", "--only", "--no-cpu-baseline", "--check", "2048")
    rf = line["roofline"]
    assert line["n_gpus"] == 1 and line["parity_checked_items"] == 2048 and line["value"] > 1e7
    assert rf["peak_source"].startswith("this run") and 10 < rf["peak"] < 40 and 1.2 < rf["peak_clock_ghz"] < 2.7
    assert 0 < rf["frac"] <= 1.0 and 0 < rf["executed_frac"] <= 1.0 and 0 < rf["issue_share"] <= 1.0
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 2e-3                        # the line reproduces from its own fields
    assert (rf["priced"] == "executed") == (wl == "mul_base")
    if wl == "mul_base":
        assert rf["frac_vs_reference_algorithm"] > rf["frac"]                          # 43 additions executed, 64 in the reference
    assert 1.2 < rf["kernel_clock_ghz"] < 2.7


def test_bench_group_mode_one_rank():
    line = _run("--mode", "group", "--gpus", "1", "--steps", "3", "--warmup", "1", "--check", "1024")
    assert line["config"]["mode"] == "group" and line["table_transport"] == "none" and line["ranks_seen"] == 1
    assert line["parity_checked_items"] == 1024 and line["table_identical_on_all_ranks"] and line["value"] > 1e7
    assert "k_mul_ladder" in line["rank0_kernels_ms_per_step"]


def test_two_real_ranks_on_the_one_gpu_run_the_sharded_configuration():
    """`bench.py --gpus 2` as the driver will start it on a multi-GPU node — the parent spawns the ranks, each is a real engine process, rank 0 builds the
    table and broadcasts it, the job is ONE batch cut into two shards, every rank checks its own shard against the oracle — except that both ranks sit
    on GPU 0 and the process group is gloo (RCCL wants one device per rank): everything of the N > 1 path above the collective's transport"""
    line = _run("--gpus", "2", "--total", "40001", "--steps", "2", "--warmup", "1", "--check", "512", "--only", "--no-cpu-baseline", "--dist-backend", "gloo", "--same-device")
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "strong" and line["dist_backend"] == "gloo"
    assert line["config"]["total_items"] == 40001 and line["items_per_rank"] == [20000, 20001]
    assert line["parity_checked_items_per_rank"] == [512, 512] and line["table_identical_on_all_ranks"]
    assert line["launched_by"] == "self-spawned ranks" and line["value"] > 1e6
    assert abs(line["value"] - 40001 * 2 / (line["ms_per_step"] * 2e-3)) <= 1e-3 * line["value"]

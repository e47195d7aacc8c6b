"""N > 1 path on CPU: world_size-2 gloo processes run the same sharding + table-distribution code the
GPU ranks run over RCCL (kyber-rs_amd/multi_gpu.py), with a stand-in engine that only records bytes."""
import hashlib
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FakeEngine:
    def __init__(self, have_table):
        self.table = bytes((i * 131 + 7) & 0xFF for i in range(335232)) if have_table else None

    def base_table_export_dev(self, t):
        t.copy_(torch.frombuffer(bytearray(self.table), dtype=torch.uint8))

    def base_table_import_dev(self, t):
        self.table = bytes(t.numpy().tobytes())

    def sync(self):
        pass


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kyber_rs_amd import multi_gpu
    eng = FakeEngine(have_table=(rank == 0))
    multi_gpu.distribute_base_table(eng, rank, world, torch.device("cpu"), dist)
    lo, hi = multi_gpu.shard(n_total, rank, world)
    # "process" the shard: checksum of the item indices, then max-over-ranks timing style reduction
    t = torch.tensor([float(hi - lo)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    # optional gather of the shard outputs: record i carries its own global index
    local = torch.arange(lo, hi, dtype=torch.int64).unsqueeze(1).repeat(1, 4)
    full = multi_gpu.gather_outputs(local, n_total, rank, world, dist)
    assert full.shape == (n_total, 4) and bool((full[:, 0] == torch.arange(n_total)).all())
    q.put((rank, lo, hi, hashlib.sha256(eng.table).hexdigest(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1 << 20, 1000003, 3])
def test_two_rank_shards_and_table_broadcast(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = hashlib.sha256(bytes((i * 131 + 7) & 0xFF for i in range(335232))).hexdigest()
    assert [r[3] for r in res] == [want, want]                 # rank 1 received rank 0's table
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_total   # disjoint + covering
    assert abs((res[0][2] - res[0][1]) - (res[1][2] - res[1][1])) <= 1
    assert res[0][4] == float(n_total)


def test_shard_properties():
    from kyber_rs_amd import multi_gpu
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 1 << 24, 12345677):
            r = [multi_gpu.shard(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    with pytest.raises(ValueError):
        multi_gpu.shard(10, 2, 2)


def test_bench_spawns_its_own_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's N = 1 command): the parent spawns the two ranks before
    any GPU call, they rendezvous over gloo with the stand-in engine (tests/standin_engine.py: no GPU, no arithmetic), rank 1 receives
    rank 0's table image, and the parent relays exactly one JSON line — marked as a plumbing run, not a measurement."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--n", "4096", "--standin"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["dist_backend"] == "gloo" and line["launched_by"] == "self-spawned ranks" and line["scaling"] == "weak"
    assert line["metric"].startswith("STANDIN") and line["roofline"] is None            # cannot be mistaken for a measurement
    assert line["standin_calls"] == {"mul_base_dev": 1, "mul_dev": 4}                     # input generation, 1 warm-up + 3 timed steps
    assert [d["rank"] for d in line["devices"]] == [0, 1]
    assert line["table_identical_on_all_ranks"] and len({d["table_sha256_16"] for d in line["devices"]}) == 1
    assert abs(line["value"] - 2 * 4096 * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-3 * line["value"]      # whole-job rate over the max-over-ranks time


def _plain_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def test_bench_with_two_gpus_runs_the_sharded_configuration_and_every_rank_reports_parity():
    """`python bench.py --gpus 2` without --n is BASELINE configs[4]: ONE job cut into shards [floor(r T / N), floor((r + 1) T / N)) — here
    T = 10,001 instead of 2^24 so that the stand-in run stays short — `scaling: "strong"`, the workload named as sharded, `value` = T x steps
    over the max-over-ranks time, and a parity count from EVERY rank in the line"""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--total", "10001", "--check", "64", "--standin"],
                       capture_output=True, text=True, timeout=300, env=_plain_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["total_items"] == 10001 and line["items_per_rank"] == [5000, 5001] and line["config"]["items_per_gpu"] == 5000
    assert "sharded across 2xMI355X" in line["config"]["workload"] and line["config"]["workload"].startswith("10001 variable-base")
    assert line["parity_checked_items_per_rank"] == [64, 64] and line["parity_checked_items"] == 128
    assert abs(line["value"] - 10001 * 2 / (line["ms_per_step"] * 2e-3)) <= 1e-3 * line["value"]
    # the default total is 2^24 and the text says so (no run: 2^23 synthetic items per rank take a minute to generate)
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    ns = argparse.Namespace(n=0, total=0, scaling="auto", keyed=False)
    for world in (2, 4, 8, 3):
        per = [bench.plan_items(ns, "mul", world, rk) for rk in range(world)]
        assert sum(p[0] for p in per) == 1 << 24 and {p[1] for p in per} == {1 << 24} and {p[2] for p in per} == {"strong"}
        assert per[0][3].startswith(f"2^24 variable-base scalar-mults sharded across {world}xMI355X")
        assert max(p[0] for p in per) - min(p[0] for p in per) <= 1
    assert bench.plan_items(ns, "mul", 1, 0)[:3] == (1 << 20, 1 << 20, "weak")                     # N = 1 stays configs[1]
    assert bench.plan_items(argparse.Namespace(n=0, total=0, scaling="weak", keyed=False), "mul", 8, 3)[:3] == (1 << 20, 8 << 20, "weak")
    assert bench.plan_items(ns, "sign", 8, 3)[:3] == (1 << 18, 8 << 18, "weak")                       # only the variable-base job is configs[4]


def test_one_rank_reporting_a_parity_failure_fails_the_whole_job():
    """rank 1's shard differs from the oracle (simulated): no JSON line, non-zero status from the parent, the failing rank named"""
    import subprocess
    env = _plain_env()
    env["KYB_BENCH_PARITY_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--total", "600", "--standin"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert "PARITY FAILURE" in r.stderr and "rank 1" in r.stderr


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--standin"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_a_failing_rank_fails_the_self_spawned_job():
    """one rank dying must end the whole job with a non-zero status (the parent stops the others instead of hanging in the rendezvous)"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["KYB_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n", "256", "--standin"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]

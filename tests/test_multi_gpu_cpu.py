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

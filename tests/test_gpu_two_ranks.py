"""Two engine processes on the one available GPU, gloo process group, device tensors: the real
multi-rank init path (rank 0 builds the base table on the GPU, broadcast, rank 1 starts with
kyb_init_no_table and imports it) followed by sharded fixed-base / variable-base / sign work, each
rank's shard checked against the oracle.  RCCL itself needs one device per rank (the driver's 8-GPU
run covers it); everything above the collective is identical."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import kyber_rs_amd
    from kyber_rs_amd import multi_gpu
    import oracle_lib
    import synth
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cuda:0")
        eng = kyber_rs_amd.Engine(0, build_table=(rank == 0))
        if rank != 0:
            # without a table the fixed-base entry points must refuse to run
            try:
                eng.mul_base(synth.scalars(1, 1))
                q.put((rank, "no-table call did not fail"))
                return
            except kyber_rs_amd.KyberHipError as e:
                assert "KYB_E_NOT_INIT" in str(e)
        multi_gpu.distribute_base_table(eng, rank, world, dev, dist)
        orc = oracle_lib.Oracle()
        lo, hi = multi_gpu.shard(n_total, rank, world)
        s = synth.scalars(n_total, 9)[lo:hi]
        ok = np.array_equal(eng.mul_base(s), orc.mul_base_batch(s, nthreads=4))
        pts = orc.mul_base_ext_batch(synth.scalars(64, 10 + rank))
        ok &= np.array_equal(eng.mul(s[:64], pts_ext=pts), orc.mul_batch(s[:64], pts, nthreads=4))
        msgs = synth.messages(64, rank)
        ok &= np.array_equal(eng.schnorr_sign(s[:64], s[64:128], msgs), orc.schnorr_sign_batch(s[:64], s[64:128], msgs, nthreads=4))
        tbl = eng.base_table().tobytes()
        import hashlib
        q.put((rank, "ok" if ok else "parity failure", lo, hi, hashlib.sha256(tbl).hexdigest()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"exception: {e!r}"))


def test_two_ranks_share_one_table():
    import torch.multiprocessing as mp
    world, n_total = 2, 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert [r[1] for r in res] == ["ok", "ok"], res
    assert res[0][4] == res[1][4]                       # identical table image on both ranks
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 500, 500, 1001)

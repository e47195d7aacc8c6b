"""Multi-GPU plumbing: one process per GPU, shards of independent items, ONE collective at init.

SURVEY.md §8(e): every scalar-mult is independent, so rank r of G takes the index range
[r*N/G, (r+1)*N/G) and no data-path collective exists.  The only exchange is the base-point table
image (335,232 B: radix-16, -32 and -64 tables) that rank 0 builds on its GPU and broadcasts over RCCL/xGMI (backend "nccl" on ROCm);
on CPU-only hosts the same code runs over gloo with a stand-in engine (tests/test_multi_gpu_cpu.py).
"""
from __future__ import annotations

BASE_TABLE_BYTES = 335232


def shard(n_total: int, rank: int, world: int):
    """index range of `rank`: [r*N/G, (r+1)*N/G) — contiguous, disjoint, covering, sizes differ by <= 1"""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def distribute_base_table(engine, rank: int, world: int, device, dist=None) -> None:
    """rank 0: export the engine's table into a device tensor and broadcast it; others: import it.

    `engine` needs base_table_export_dev(tensor) / base_table_import_dev(tensor) / sync();
    rank 0's engine must have been created with build_table=True, the others with build_table=False."""
    if world == 1:
        return
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLW0642
    tbl = torch.empty(BASE_TABLE_BYTES, dtype=torch.uint8, device=device)
    if rank == 0:
        engine.base_table_export_dev(tbl)
        engine.sync()
    dist.broadcast(tbl, src=0)
    if device is not None and getattr(device, "type", "cpu") == "cuda":
        torch.cuda.synchronize()
    if rank != 0:
        engine.base_table_import_dev(tbl)
        engine.sync()          # the copy reads `tbl`: finish it before the tensor goes back to the allocator


def gather_outputs(local, n_total: int, rank: int, world: int, dist=None):
    """Optional: every rank ends up with the outputs of ALL shards (only when a follow-on step needs them on
    one GPU, e.g. summing commitments).  `local` holds this rank's shard, records along dim 0; shards differ in length
    by at most one, so they travel padded to the longest and are trimmed on arrival.  One all_gather (RCCL over xGMI on
    GPU ranks, gloo on CPU)."""
    if world == 1:
        return local
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLW0642
    sizes = [shard(n_total, r, world)[1] - shard(n_total, r, world)[0] for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError("local shard has the wrong length")
    longest = max(sizes)
    padded = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: sizes[rank]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([parts[r][: sizes[r]] for r in range(world)], dim=0)

"""Multi-GPU plumbing: one process per GPU, games sharded by global game id, no data-path collective.

SURVEY.md 8(e): games are independent, so rank r of W owns the global game ids [r*n, (r+1)*n) and seeds its per-game
Philox streams with those global ids (`env_id0 = r*n`) - results do not depend on W.  The only exchanges are
(1) the 3-scalar all-reduce of the advantage statistics (ppo.compute_gae), (2) the gradient all-reduce of the policy
update (torch DDP-style flat bucket, RCCL over xGMI) and (3) timing/barrier for the benchmark.
The same helpers run on the `gloo` backend in the CPU test-suite (tests/test_multi_rank_cpu.py).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device_index=None):
    """Reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torch.distributed.run).  Returns (rank, local_rank, world)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
        dist.init_process_group(backend, **kw)
    return rank, local_rank, world


def shard(rank, games_per_rank):
    """-> (env_id0, n): the global game ids owned by `rank` (weak scaling: fixed games per GPU)."""
    return rank * games_per_rank, games_per_rank


def barrier(sync_cuda=True):
    """Device work of this rank done -> all ranks arrived -> (the barrier's own collective done)."""
    if sync_cuda and torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
        if sync_cuda and torch.cuda.is_available():
            torch.cuda.synchronize()


def _reduce(value, op, device=None):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=op)
    return float(t.item())


def max_over_ranks(value, device=None):
    return _reduce(value, dist.ReduceOp.MAX, device)


def sum_over_ranks(value, device=None):
    return _reduce(value, dist.ReduceOp.SUM, device)


def allreduce_flat_grads(params, world=None):
    """One flat-bucket gradient all-reduce per optimiser step (1.93 M parameters = 7.7 MB fp32: a single bucket;
    xGMI rings are per-link bound, so fewer, larger messages win).  Averages in place."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat)
    flat /= dist.get_world_size() if world is None else world
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def broadcast_parameters(module, src=0):
    """Replicates rank `src`'s parameters and buffers (the reference's central policy is a single object; here every rank
    holds a replica that must start identical and stays identical through the all-reduced gradients)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)


def finalize():
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()

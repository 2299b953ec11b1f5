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


def allreduce_mean_(flat, group=None):
    """In-place mean over the ranks of `group`.  RCCL (backend "nccl") averages inside the collective (ReduceOp.AVG: one launch,
    no separate divide pass over the bucket); gloo has no AVG, so the CPU test-suite takes sum + divide."""
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat, group=group)
        flat /= dist.get_world_size(group)


def device_identities():
    """One record per rank - host, local device index, the device's name, UUID / PCI bus id - all-gathered, so that a bench
    line can SHOW that N ranks ran on N distinct GPUs (and over which backend)."""
    import socket
    me = {"rank": int(os.environ.get("RANK", "0")), "host": socket.gethostname(), "device": None, "name": None, "uuid": None, "pci_bus_id": None}
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(i)
        me.update(device=i, name=pr.name)
        for key in ("uuid", "pci_bus_id"):
            try:
                me[key] = str(getattr(pr, key))
            except Exception:
                pass
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [me]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, me)
    return out


class GradBucket(object):
    """One persistent flat gradient buffer for a FIXED parameter list (1.93 M parameters = 7.7 MB fp32: a single bucket;
    xGMI rings are per-link bound, so one large message beats many small ones).

    Every parameter's `.grad` is a view into the buffer, so autograd accumulates straight into it (in-place adds), the
    all-reduce runs on the buffer itself - no concatenation, no copy back - and the layout is identical on every rank by
    construction: a parameter that received no gradient in a step contributes zeros instead of changing the message size
    (rank-divergent sets of `grad is None` parameters would otherwise mis-align a concatenated bucket or hang the
    collective).  Use `zero()` instead of `optimizer.zero_grad()` (which would drop the views).  With a single rank `zero()`
    detaches the views instead (see there)."""

    def __init__(self, params, process_group=None, assign_when_single_rank=False):
        self.params = [p for p in params]
        self.assign_when_single_rank = assign_when_single_rank
        if not self.params:
            raise ValueError("GradBucket needs at least one parameter")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("all parameters of a bucket must share device and dtype")
        self.group = process_group
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self._views = []
        off = 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self._views.append(v)
            off += p.numel()

    def zero(self):
        if self.assign_when_single_rank and not (dist.is_initialized() and dist.get_world_size(self.group) > 1):
            # one rank: nothing to all-reduce, so autograd may simply ASSIGN the gradients (`.grad = None` first) instead of
            # adding ~300 small tensors into the zeroed views (one launch each: 1.2 ms of a 54 ms minibatch step)
            for p in self.params:
                p.grad = None
            return
        self.flat.zero_()
        for p, v in zip(self.params, self._views):      # re-attach if something replaced or dropped a .grad meanwhile
            if p.grad is not v:
                p.grad = v

    def check(self):
        if self.assign_when_single_rank and not (dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return                                       # (the views are detached on purpose, see zero())
        bad = [i for i, (p, v) in enumerate(zip(self.params, self._views)) if p.grad is None or p.grad.data_ptr() != v.data_ptr()]
        if bad:
            raise RuntimeError(f"GradBucket: {len(bad)} parameter gradients are no longer views of the bucket (first: #{bad[0]})")

    def allreduce(self):
        """sum over ranks -> / world, in place, right after backward (a single collective: at 7.7 MB it takes ~0.1 ms over
        xGMI against an ~80 ms minibatch step, so it is not overlapped with backward)."""
        if not (dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return
        self.check()
        allreduce_mean_(self.flat, self.group)


def allreduce_flat_grads(params, world=None):
    """Stateless variant for callers without a GradBucket: fixed layout over ALL the given parameters.  A parameter without a
    gradient on this rank sends zeros and RECEIVES the average like everybody else (left missing, the ranks that do have a
    gradient would apply the averaged update and this one would skip it: the replicas would drift apart).  Averages in place."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    params = [p for p in params]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    if world is None or world == dist.get_world_size():
        allreduce_mean_(flat)
    else:
        dist.all_reduce(flat)
        flat /= world
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is not None:
            p.grad.copy_(flat[off:off + n].view_as(p))
        else:
            p.grad = flat[off:off + n].view_as(p).clone()
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

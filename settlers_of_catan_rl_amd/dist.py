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


INIT_REPORT = {"attempts": [], "backend": None, "selfcheck": None}     # what init_from_env did (bench.py copies it into its line)
_side_store = None


def _rccl_log_path():
    return os.environ.get("NCCL_DEBUG_FILE", "").replace("%h", __import__("socket").gethostname()).replace("%p", str(os.getpid()))


def rccl_log_tail(max_bytes=1500):
    """The tail of this rank's RCCL log (NCCL_DEBUG=WARN into NCCL_DEBUG_FILE, set by init_from_env): '' when RCCL had nothing to say."""
    path = _rccl_log_path()
    try:
        with open(path, "rb") as f:
            blob = f.read()
        return blob[-max_bytes:].decode("utf-8", "replace")
    except OSError:
        return ""


def _agreement_store(rank, world, timeout_s):
    """A TCPStore of our own next to the rendezvous one (MASTER_PORT + 23, or CATAN_SIDE_PORT): the ranks tell each other how an
    init attempt went WITHOUT a collective, so that all of them move to the next attempt together."""
    global _side_store
    if _side_store is None:
        from datetime import timedelta
        port = int(os.environ.get("CATAN_SIDE_PORT", int(os.environ.get("MASTER_PORT", "29500")) + 23))
        _side_store = dist.TCPStore(os.environ["MASTER_ADDR"], port, world, is_master=(rank == 0), timeout=timedelta(seconds=timeout_s),
                                    wait_for_workers=False)
    return _side_store


def _agree(store, tag, rank, world, ok, msg, timeout_s):
    """-> (every rank said ok, [(rank, what it said) for the others])"""
    from datetime import timedelta
    if store is None:
        return ok, ([] if ok else [(rank, msg)])
    store.set(f"{tag}/{rank}", "ok" if ok else ("fail: " + msg)[:400])
    keys = [f"{tag}/{r}" for r in range(world)]
    try:
        store.wait(keys, timedelta(seconds=timeout_s))
    except Exception as e:
        return False, [(-1, f"no word from some rank within {timeout_s:.0f} s ({type(e).__name__})")]
    said = [(r, store.get(k).decode("utf-8", "replace")) for r, k in enumerate(keys)]
    bad = [(r, v) for r, v in said if v != "ok"]
    return not bad, bad


def allreduce_selfcheck(timeout_s=60.0, device=None):
    """Before anything is timed: the collectives the learner uses, on tiny inputs whose results are known - the SUM of (rank + 1),
    the mean of a vector filled with the rank through allreduce_mean_ (ReduceOp.AVG on RCCL), and an all-gather of the ranks.  A
    wrong result or a collective that does not finish within `timeout_s` raises.  -> dict for the bench line."""
    import time
    from datetime import timedelta
    w, r = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    t0 = time.perf_counter()

    def finish(work, what):
        if work is not None and not work.wait(timedelta(seconds=timeout_s)):
            raise RuntimeError(f"selfcheck: {what} did not complete")
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    a = torch.full((8,), float(r + 1), dtype=torch.float32, device=device)
    finish(dist.all_reduce(a, async_op=True), "all_reduce(SUM)")
    if not bool((a == w * (w + 1) / 2.0).all()):
        raise RuntimeError(f"selfcheck: all_reduce(SUM) of rank + 1 gave {a.tolist()} on rank {r}, expected {w * (w + 1) / 2.0}")
    b = torch.full((1024,), float(r), dtype=torch.float32, device=device)
    allreduce_mean_(b)
    finish(None, "all_reduce(AVG)")
    if not bool(((b - (w - 1) / 2.0).abs() < 1e-6).all()):
        raise RuntimeError(f"selfcheck: mean over ranks gave {float(b[0])} on rank {r}, expected {(w - 1) / 2.0}")
    g = [torch.zeros((2,), dtype=torch.int64, device=device) for _ in range(w)]
    finish(dist.all_gather(g, torch.tensor([r, 7 * r + 1], dtype=torch.int64, device=device), async_op=True), "all_gather")
    if [int(x[0]) for x in g] != list(range(w)) or any(int(x[1]) != 7 * i + 1 for i, x in enumerate(g)):
        raise RuntimeError(f"selfcheck: all_gather gave {[x.tolist() for x in g]} on rank {r}")
    return {"ok": True, "world": w, "backend": dist.get_backend(), "device": str(device), "seconds": time.perf_counter() - t0,
            "checked": ["all_reduce SUM of rank + 1", "allreduce_mean_ (ReduceOp.AVG on RCCL) of rank", "all_gather of ranks"]}


def _drop_process_group():
    if dist.is_initialized():
        for name in ("_abort_process_group",):               # (a communicator that failed half-way may not destroy cleanly)
            fn = getattr(dist.distributed_c10d, name, None)
            if fn is not None:
                try:
                    fn()
                    return
                except Exception:
                    pass
        try:
            dist.destroy_process_group()
        except Exception:
            pass


_attempt_hook = None          # tests: called as hook(attempt_number, rank) right after init_process_group of an attempt; may raise
on_hang = None                # bench.py: called (from a timer thread) with a message when an init attempt outlives its deadline, before the process exits


def _preflight(store, rank, world, local_rank, timeout_s):
    """Before any communicator is built: every rank tells the others (side store, no collective) which device it sits on.  RCCL needs one
    DISTINCT, EXISTING device per rank; a rank whose LOCAL_RANK has no device of its own (fewer visible devices than ranks: a one-GPU
    rehearsal, a wrong HIP_VISIBLE_DEVICES) would leave its peers inside ncclCommInitRank with nobody to meet - that call has no time-out
    and cannot be interrupted.  -> (RCCL can be tried, what every rank said)"""
    from datetime import timedelta
    import socket
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n == 0:
        mine = "no-device"
    else:
        idx = local_rank % n
        props = torch.cuda.get_device_properties(idx)
        ident = str(getattr(props, "uuid", "")) or f"index-{idx}"
        mine = f"{'own' if local_rank < n else 'shared'}|{socket.gethostname()}|{ident}"
    store.set(f"catan/preflight/{rank}", mine)
    keys = [f"catan/preflight/{r}" for r in range(world)]
    try:
        store.wait(keys, timedelta(seconds=timeout_s))
    except Exception as e:
        return False, [f"no word from some rank within {timeout_s:.0f} s ({type(e).__name__})"]
    said = [store.get(k).decode("utf-8", "replace") for k in keys]
    return all(v.startswith("own|") for v in said) and len(set(said)) == world, said


def _deadline(seconds, what, rank):
    """A timer that ends the process when an init attempt neither returns nor raises (a communicator set-up that waits for a peer for
    ever): a message on stderr, bench.py's line through `on_hang`, exit code 86 - torch.distributed.run then stops the other ranks.
    -> the timer (cancel() it when the attempt is over)."""
    import sys
    import threading

    def fire():
        msg = f"rank {rank}: {what} did not return within {seconds:.0f} s - giving up (a peer never arrived, or the communicator set-up hangs)"
        try:
            sys.stderr.write("[catan dist] " + msg + "\n"); sys.stderr.flush()
            if on_hang is not None:
                on_hang(msg)
        finally:
            os._exit(86)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def init_from_env(backend=None, device_index=None, timeout_s=None, selfcheck=True, plan=None):
    """Reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torch.distributed.run).  Returns (rank, local_rank, world).

    The first multi-GPU run cannot be rehearsed on a one-GPU box, so it must not be able to fail silently: an attempt is
    init_process_group with a `timeout_s` time-out (CATAN_DIST_TIMEOUT_S, default 120) followed by `allreduce_selfcheck`; the ranks
    tell each other over a side TCPStore how it went (no collective), and if ANY of them failed all of them drop the group and take
    the next attempt together: "nccl" (= RCCL) bound eagerly to this rank's device -> "nccl" without `device_id` (lazy communicator)
    -> "gloo" (host-staged: slow but it yields a line that says so).  An explicit `backend` is tried alone.  What happened is in
    INIT_REPORT; RCCL's warnings go to a per-rank file (`rccl_log_tail`).
    Two guards around the attempts (round 5, after a two-rank run on a ONE-GPU box spent 870 s and produced nothing: rank 1 had no device
    of its own, rank 0 sat inside the eager RCCL set-up - which has no time-out - and the ranks' attempts drifted apart): a PRE-FLIGHT over
    the side store (every rank's device: RCCL is only tried when each rank has a distinct, existing one - otherwise all ranks go to gloo
    together), and a DEADLINE per attempt (2 x time-out + 120 s: a set-up that neither returns nor raises ends the process with a message
    and, on rank 0, bench.py's line, instead of holding the node until the launcher's own limit)."""
    from datetime import timedelta
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/catan_rccl_%h_%p.log")
        if timeout_s is None:
            timeout_s = float(os.environ.get("CATAN_DIST_TIMEOUT_S", "120"))
        have_gpu = torch.cuda.is_available()
        dev = torch.device("cuda", local_rank if device_index is None else device_index) if have_gpu else None
        if plan is not None:
            plan = [(be, dev if (bind and be == "nccl") else None) for be, bind in plan]     # [(backend, bind to the device eagerly)]
        elif backend is not None:
            plan = [(backend, dev if backend == "nccl" else None)]
        elif have_gpu:
            plan = [("nccl", dev), ("nccl", None), ("gloo", None)]   # "nccl" is RCCL on ROCm
        else:
            plan = [("gloo", None)]
        store = None
        if len(plan) > 1 or selfcheck:
            try:
                store = _agreement_store(rank, world, timeout_s)
            except Exception as e:                                   # no side store: a single attempt, errors propagate
                INIT_REPORT["attempts"].append({"side_store": f"unavailable ({type(e).__name__}: {e})"})
                plan = plan[:1]
        if store is not None and have_gpu and any(be == "nccl" for be, _ in plan):
            rccl_ok, said = _preflight(store, rank, world, local_rank, timeout_s)
            INIT_REPORT["preflight"] = {"one_distinct_device_per_rank": rccl_ok, "ranks": said[:16]}
            if not rccl_ok:                                          # (every rank reads the same words: all of them skip RCCL together)
                plan = [(be, d) for be, d in plan if be != "nccl"] or [("gloo", None)]
        for k, (be, device_id) in enumerate(plan):
            rec = {"backend": be, "device_id": None if device_id is None else str(device_id), "timeout_s": timeout_s}
            ok, msg, chk = True, "", None
            watchdog = _deadline(2.0 * timeout_s + 120.0, f"init attempt {k} ({be}{'' if device_id is None else ', bound to ' + str(device_id)})", rank)
            try:
                kw = {"timeout": timedelta(seconds=timeout_s)}
                if device_id is not None:
                    kw["device_id"] = device_id
                if store is not None:          # every attempt rendezvouses over the side store under its own prefix: a second attempt
                    kw.update(store=dist.PrefixStore(f"catan/pg/{k}", store), rank=rank, world_size=world)   # never meets the first one's keys or server
                dist.init_process_group(be, **kw)
                if _attempt_hook is not None:
                    _attempt_hook(k, rank)
                if selfcheck:
                    chk = allreduce_selfcheck(min(timeout_s, 60.0))
            except Exception as e:
                ok, msg = False, f"{type(e).__name__}: {e}"
            finally:
                watchdog.cancel()
            all_ok, bad = _agree(store, f"catan/init/{k}", rank, world, ok, msg, timeout_s + 90.0)
            rec.update(ok=all_ok, this_rank=("ok" if ok else msg[:400]), failing_ranks=[(r, v[:200]) for r, v in bad][:8])
            INIT_REPORT["attempts"].append(rec)
            if all_ok:
                INIT_REPORT.update(backend=be, selfcheck=chk)
                break
            _drop_process_group()
        else:
            raise RuntimeError(f"no collective backend came up on {world} ranks: {INIT_REPORT['attempts']}")
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
    me["rccl_log"] = rccl_log_tail(600)          # NCCL_DEBUG=WARN output of this rank so far ('' = nothing to report)
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
        # every view starts on a 16-byte boundary (offsets padded to ALIGN elements; the padding stays zero and travels with the message):
        # optim.FusedAdam's kernels read gradients with 16-byte loads, and of CatanPolicy's 203 parameters 108 would start mid-vector in
        # an unpadded bucket (the optimiser then cloned those gradients every step on the multi-rank path)
        self.ALIGN = max(1, 16 // torch.empty((), dtype=dt).element_size())
        pad = lambda n: (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(sum(pad(p.numel()) for p in self.params), dtype=dt, device=dev)
        self._views = []
        off = 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self._views.append(v)
            off += pad(p.numel())

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

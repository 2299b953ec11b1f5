"""world_size-2 gloo tests (CPU) of the N>1 path: game sharding by global id is invariant to the number of ranks,
the 3-scalar advantage-statistics all-reduce reproduces the global normalisation, the flat gradient bucket averages,
and the bench timing reduction takes the max over ranks.  The per-game engine here is the CPU oracle (test
infrastructure); on the GPU box the same dist helpers run over RCCL."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from settlers_of_catan_rl_amd import dist as cdist
    r, lr, w = cdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    n_per, seed, steps = 48, 5, 300
    env_id0, n = cdist.shard(r, n_per)
    b = oracle_lib.OracleBatch(n, seed, env_id0=env_id0)
    blobs = torch.from_numpy(b.run_random(steps, n_threads=1))
    gathered = [torch.empty_like(blobs) for _ in range(w)]
    dist.all_gather(gathered, blobs)
    # (1) advantage statistics: local (sum, sumsq, count) -> all-reduce -> global mean / unbiased std
    g = torch.Generator().manual_seed(100 + r)
    adv = torch.randn(40, n, generator=g, dtype=torch.float32) * (1 + r) + 0.3 * r
    stats = torch.tensor([adv.double().sum(), (adv.double() ** 2).sum(), adv.numel()], dtype=torch.float64)
    dist.all_reduce(stats)
    cnt, mean = stats[2], stats[0] / stats[2]
    std = torch.sqrt((stats[1] - cnt * mean * mean) / (cnt - 1))
    all_adv = [torch.empty_like(adv) for _ in range(w)]
    dist.all_gather(all_adv, adv)
    # (2) flat gradient bucket
    lin = torch.nn.Linear(4, 3)
    for i, p in enumerate(lin.parameters()):
        p.grad = torch.full_like(p, float(r + 1 + i))
    cdist.allreduce_flat_grads(list(lin.parameters()))
    grads = [p.grad.clone() for p in lin.parameters()]
    # (3) timing reduction
    tmax = cdist.max_over_ranks(1.0 + r)
    tsum = cdist.sum_over_ranks(1.0 + r)
    cdist.barrier(sync_cuda=False)
    if r == 0:
        q.put(dict(blobs=torch.cat(gathered).numpy(), mean=float(mean), std=float(std),
                   ref_mean=float(torch.cat(all_adv, 1).double().mean()), ref_std=float(torch.cat(all_adv, 1).double().std()),
                   grads=[g.numpy() for g in grads], tmax=tmax, tsum=tsum, n_per=n_per, seed=seed, steps=steps))
    cdist.finalize()


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_n_rank_sharding_and_reductions(world):
    """world_size 2 and 8 (one rank per GPU of an 8-GPU node, BASELINE config 4): shard invariance by global game id, the
    3-double advantage statistics, the flat gradient average, max / sum over ranks"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    single = oracle_lib.OracleBatch(world * out["n_per"], out["seed"]).run_random(out["steps"], n_threads=1)
    assert np.array_equal(out["blobs"], single), "sharded games differ from the single-process run"
    assert abs(out["mean"] - out["ref_mean"]) < 1e-9 and abs(out["std"] - out["ref_std"]) < 1e-9
    for i, g in enumerate(out["grads"]):
        assert np.allclose(g, np.mean([r + 1 + i for r in range(world)]))
    assert out["tmax"] == float(world) and out["tsum"] == world * (world + 1) / 2.0


def _init_worker(rank, world, port, q, fail_rank):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      CATAN_DIST_TIMEOUT_S="15")
    sys.path.insert(0, ROOT)
    from settlers_of_catan_rl_amd import dist as cdist

    def hook(attempt, r):                       # the first attempt "fails" on ONE rank only: the others must follow it to the second
        if attempt == 0 and r == fail_rank:
            raise RuntimeError("simulated communicator failure")
    cdist._attempt_hook = hook
    r, lr, w = cdist.init_from_env(plan=[("gloo", False), ("gloo", False)])
    rep = cdist.INIT_REPORT
    again = cdist.allreduce_selfcheck(timeout_s=30.0)
    t = torch.tensor([float(r)])
    cdist.allreduce_mean_(t)
    q.put((rank, [a.get("ok") for a in rep["attempts"]], rep["attempts"][0]["failing_ranks"], rep["selfcheck"]["ok"], again["ok"], float(t)))
    cdist.finalize()


def test_init_retries_on_every_rank_when_one_rank_fails_and_selfcheck_runs():
    """an init attempt that fails on ONE rank makes ALL ranks drop the group and take the next attempt (agreed over
    the side TCPStore, no collective), the self-check (SUM / mean / all-gather with known results) runs before anything is timed, and
    INIT_REPORT says what happened."""
    world, fail_rank = 3, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_init_worker, args=(r, world, port, q, fail_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, oks, failing, chk_ok, again_ok, mean in got:
        assert oks == [False, True], (rank, oks)
        said = dict((f[0], f[1]) for f in failing)       # (the other ranks' self-check timed out waiting for the failed one: they report too)
        assert "simulated" in said[fail_rank], said
        assert chk_ok and again_ok and abs(mean - (world - 1) / 2.0) < 1e-9


def _preflight_worker(rank, world, port, q, devices):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      CATAN_DIST_TIMEOUT_S="15")
    sys.path.insert(0, ROOT)
    from settlers_of_catan_rl_amd import dist as cdist

    class Props(object):
        def __init__(self, idx):
            self.uuid = f"GPU-fake-{idx}"
    # a box that SAYS it has `devices` GPUs (nothing is ever placed on them: the pre-flight only asks for their count and identity)
    torch.cuda.is_available = lambda: True
    torch.cuda.device_count = lambda: devices
    torch.cuda.get_device_properties = lambda idx: Props(idx)
    calls = []
    real_init = torch.distributed.init_process_group

    def spy(backend, **kw):
        calls.append(backend)
        kw.pop("device_id", None)
        return real_init("gloo", **kw)          # (no RCCL here: what matters is WHICH attempts the ranks agree to make)
    torch.distributed.init_process_group = spy
    r, lr, w = cdist.init_from_env(selfcheck=False)
    q.put((rank, calls, cdist.INIT_REPORT["preflight"]["one_distinct_device_per_rank"], cdist.INIT_REPORT["backend"]))
    cdist.finalize()


@pytest.mark.parametrize("devices", [1, 2])
def test_preflight_skips_rccl_when_ranks_share_a_device(devices):
    """Two ranks on a box with ONE visible device (a rehearsal, a wrong HIP_VISIBLE_DEVICES): the pre-flight over the side store makes BOTH
    ranks skip the RCCL attempts - an eager RCCL set-up would wait for ever for the rank that has no device of its own (found on a
    one-GPU MI355X box: 870 s, no line) - and go to gloo together.  With a device per rank the first attempt is the bound RCCL one."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_preflight_worker, args=(r, world, port, q, devices)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, calls, distinct, backend in got:
        if devices == 1:
            assert not distinct and calls == ["gloo"] and backend == "gloo", (rank, calls, distinct, backend)
        else:
            assert distinct and calls == ["nccl"] and backend == "nccl", (rank, calls, distinct, backend)

"""League sampling (reference RL/ppo/update_opponent_policies.py) against vectors generated from the reference
(tools/gen_golden.py league), and the collector's grouped inference."""
import os

import numpy as np
import torch

from settlers_of_catan_rl_amd import league

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "league.npz")


def test_prob_dist_matches_reference():
    g = np.load(GOLD)
    for n in g["sizes"]:
        p = league.get_prob_dist(int(n))
        assert p.shape == (int(n),)
        assert np.allclose(p, g[f"p_{int(n)}"], rtol=0, atol=1e-15), n
        assert abs(p.sum() - 1.0) < 1e-12


def test_draws_match_reference():
    g = np.load(GOLD)
    for key in g.files:
        if not key.startswith("draw_"):
            continue
        _, seed, nproc, npol = key.split("_")
        lg = league.League(seed=int(seed))
        lg.earlier.extend({"id": i} for i in range(int(npol)))
        assert np.array_equal(lg.sample(int(nproc)), g[key]), key


def test_bounded_variant_keeps_few_distinct_nets():
    lg = league.League(max_distinct=4, seed=1)
    lg.earlier.extend({"id": i} for i in range(200))
    idx = lg.sample(10000)
    assert idx.shape == (10000, 3) and len(np.unique(idx)) <= 4


def test_deque_bookkeeping():
    lg = league.League(num_policies_to_store=3, add_policy_every=4)
    net = torch.nn.Linear(2, 2)
    lg.add(net)
    redraw = [lg.after_update(u, net) for u in range(13)]
    assert all(redraw)                                   # update_opponent_policies_every = 1
    assert len(lg.earlier) == 3                          # initial + updates 4, 8, 12, capped at 3
    with torch.no_grad():
        net.weight.add_(1.0)
    assert not torch.equal(lg.earlier[-1]["weight"], net.weight)      # snapshots are copies


class _TagNet(object):
    """Stand-in net: 'acts' by writing its tag into the first action word (log-prob = tag)."""

    def __init__(self, tag):
        self.tag = tag

    def act(self, f, lists, lens, masks, generator=None):
        n = f.shape[0]
        a = torch.zeros((n, 18), dtype=torch.int64)
        a[:, 0] = self.tag
        return torch.zeros(n, 1), a, torch.full((n, 1), float(self.tag))


def test_grouped_inference_routes_each_seat_to_its_net():
    from settlers_of_catan_rl_amd.rollout import RolloutCollector
    N = 37
    col = RolloutCollector.__new__(RolloutCollector)     # only the routing is under test
    col.policy, col.autocast_dtype, col.sample_gen, col.recurrent, col._shadow = _TagNet(0), None, None, False, None
    col.N, col.device = N, torch.device("cpu")
    g = torch.Generator().manual_seed(3)
    opp_index = torch.randint(0, 5, (N, 3), generator=g)
    col.set_opponents([_TagNet(10 + k) for k in range(5)], opp_index)
    pol = torch.randint(0, 4, (N,), generator=g)          # policy slot of the deciding seat of each game
    f = torch.zeros((N, 4)); lists = torch.zeros((N, 5, 25)); lens = torch.ones((N, 5)); masks = torch.ones((N, 325))
    actions, logp = col._act(f, lists, lens, masks, pol)
    want = torch.where(pol == 0, torch.zeros_like(pol), 10 + opp_index[torch.arange(N), (pol - 1).clamp(min=0)])
    assert torch.equal(actions[:, 0], want)
    assert torch.equal(logp, want.float())
    # league.assign: games of one worker share their three opponents
    lg = league.League(envs_per_worker=5, seed=0)
    lg.earlier.extend({} for _ in range(9))

    class _Empty(_TagNet):
        def __init__(self):
            _TagNet.__init__(self, -1)

        def load_state_dict(self, sd):
            pass

        def eval(self):
            return self

    distinct = lg.assign(col, _Empty)
    assert col.opp_index.shape == (N, 3) and len(col.opponent_nets) == len(distinct)
    assert torch.equal(col.opp_index[0], col.opp_index[4]) and int(col.opp_index.max()) < len(distinct)

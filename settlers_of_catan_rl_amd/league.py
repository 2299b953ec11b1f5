"""Opponent league: the snapshot deque and its sampling rule (reference RL/ppo/update_opponent_policies.py:13-43 and the
bookkeeping around it in RL/robust_train.py:62-64,136-141).

The reference keeps up to `num_policies_to_store` (500) earlier state-dicts of the central policy, appends one every
`add_policy_every` (4) updates, and every `update_opponent_policies_every` (1) updates gives EACH WORKER PROCESS three
opponents drawn with `np.random.choice(earlier_policies, 3, p=p)`: p is half uniform, half linearly increasing over the
most recent `linear_num` (800) snapshots.  A worker hosts `num_envs_per_process` (5) games that share its three opponent
nets; inside a game the seat -> policy-slot map is fixed at start-up (game_manager.py:24-31).

Here the "workers" are consecutive groups of `envs_per_worker` games of the batched env.  `sample` reproduces the
reference draw exactly (same numpy generator calls, same order) and returns snapshot indices per worker; the collector
then runs one batched forward per DISTINCT net in play (rollout.RolloutCollector, grouped inference).  With 65 536 games
the exact rule would put up to 500 distinct nets in play - 500 small forwards per pass - so `max_distinct=K` offers a
bounded variant: K snapshots are drawn i.i.d. from p and every worker draws its three uniformly among those K (each
opponent is still marginally p-distributed; only the correlation between workers changes).  max_distinct=None is the
reference rule.
"""
import copy
from collections import deque

import numpy as np
import torch


def get_prob_dist(num_policies, linear_num=800, linear_prob=0.5):
    """update_opponent_policies.py:29-43: (1 - linear_prob) spread uniformly + a linear ramp over the last
    min(linear_num, num_policies) entries (ramp value i * grad for the i-th of them, so the oldest of them adds 0)."""
    p = np.full((num_policies,), (1.0 - linear_prob) / num_policies)
    num_aux = min(linear_num, num_policies)
    h = (2 * linear_prob) / (num_aux + 1)
    grad = h / num_aux
    p[num_policies - num_aux:] += np.arange(num_aux) * grad
    return p / np.sum(p)


class League(object):
    def __init__(self, num_policies_to_store=500, add_policy_every=4, update_opponent_policies_every=1, envs_per_worker=5,
                 max_distinct=None, seed=0):
        self.earlier = deque(maxlen=num_policies_to_store)                   # robust_train.py:62
        self.add_policy_every = add_policy_every
        self.update_every = update_opponent_policies_every
        self.envs_per_worker = envs_per_worker
        self.max_distinct = max_distinct
        self.rng = np.random.RandomState(seed)       # the reference uses numpy's global RandomState: same algorithm

    def add(self, policy):
        """robust_train.py:63-64,136-137: a CPU copy of the central policy's state-dict."""
        self.earlier.append({k: v.detach().to("cpu", copy=True) for k, v in policy.state_dict().items()})

    def after_update(self, update_num, policy):
        """robust_train.py:135-141, called once per PPO update with the 0-based update number: returns True when the
        opponents should be re-drawn."""
        if update_num % self.add_policy_every == 0 and update_num > 0:
            self.add(policy)
        return update_num % self.update_every == 0

    def sample(self, num_workers):
        """-> int64 [num_workers, 3] snapshot indices into `self.earlier`."""
        n = len(self.earlier)
        p = get_prob_dist(n)
        if self.max_distinct is None:
            return np.stack([self.rng.choice(n, 3, p=p) for _ in range(num_workers)]).astype(np.int64)
        pool = self.rng.choice(n, self.max_distinct, p=p)
        return pool[self.rng.randint(0, self.max_distinct, size=(num_workers, 3))].astype(np.int64)

    def assign(self, collector, make_net):
        """Draws opponents for every worker of `collector` and installs them: one net per distinct snapshot in play
        (`make_net()` builds an empty net on the collector's device), per-game opponent indices for the three opponent
        policy slots."""
        N = collector.N
        workers = -(-N // self.envs_per_worker)
        idx = self.sample(workers)                                           # [workers, 3] snapshot ids
        distinct, inv = np.unique(idx, return_inverse=True)
        inv = inv.reshape(idx.shape)
        nets = []
        for s in distinct:
            net = make_net()
            net.load_state_dict(self.earlier[int(s)])
            net.eval()
            nets.append(net)
        per_game = np.repeat(inv, self.envs_per_worker, axis=0)[:N]          # games of a worker share its opponents
        collector.set_opponents(nets, torch.from_numpy(per_game))
        return distinct

"""Training-loop schedules (reference RL/robust_train.py run_update, RL/ppo/utils.py) and the loop's bookkeeping with
stand-in components."""
import numpy as np
import torch

from settlers_of_catan_rl_amd import train_loop as tl
from settlers_of_catan_rl_amd.league import League


def test_schedules_follow_the_reference_formulas():
    a = tl.TrainArgs(dense_reward_anneal_start=100, dense_reward_anneal_end=300)
    # robust_train.py:108-117 evaluated literally
    coef, w = a.entropy_coef_start, 1.0
    for u in range(0, 2000, 37):
        want_coef = coef
        if u > 500 and u <= 1500:
            want_coef = 0.04 + ((u - 500) / (1500 - 500)) * (0.005 - 0.04)
        coef = tl.entropy_coef_at(u, a, coef)
        assert abs(coef - want_coef) < 1e-15
        want_w = w
        if u > 100 and u <= 300:
            want_w = 1.0 + ((u - 100) / (300 - 100)) * (0.0 - 1.0)
        w = tl.reward_weight_at(u, a, w)
        assert abs(w - want_w) < 1e-15
    assert abs(coef - 0.005) < 1e-3                     # stays at (about) the final value after the anneal window
    assert tl.linear_lr(0, 100, 3e-4) == 3e-4 and abs(tl.linear_lr(50, 100, 3e-4) - 1.5e-4) < 1e-18
    # defaults: dense-reward annealing disabled (-1, -1): the weight never moves
    assert tl.reward_weight_at(5, tl.TrainArgs(), 1.0) == 1.0


class _Env(object):
    n = 10
    def __init__(self): self.w = []
    def set_reward_annealing_factor(self, f): self.w.append(f)


class _Storage(object):
    games_complete = 3


class _Collector(object):
    N = 10
    def __init__(self): self.assigned = 0; self.gathers = 0
    def gather_rollouts(self): self.gathers += 1; return _Storage()
    def after_rollouts(self): pass
    def set_opponents(self, nets, idx): self.assigned += 1


class _Trainer(object):
    def __init__(self, net):
        self.optimiser = torch.optim.Adam(net.parameters(), lr=3e-4)
        class C: entropy_coef = 0.0
        self.cfg = C()
    def update(self, st): return (0.1, 0.2, 0.3)


def test_loop_bookkeeping(tmp_path):
    net = torch.nn.Linear(3, 3)
    env, col, tr = _Env(), _Collector(), _Trainer(net)
    lg = League(envs_per_worker=5, seed=0)
    evals = []
    args = tl.TrainArgs(num_steps=4, total_env_steps=4 * 10 * 50, eval_every=5, entropy_coef_start_anneal=2, entropy_coef_end_anneal=12,
                        dense_reward_anneal_start=3, dense_reward_anneal_end=7)
    loop = tl.TrainingLoop(env, net, col, tr, args, league=lg, make_net=lambda: torch.nn.Linear(3, 3),
                           evaluate=lambda p, u: (evals.append(u) or {"update": u}, "summary"), checkpoint_path=str(tmp_path / "ck.pt"))
    assert loop.num_updates == 50 and len(lg.earlier) == 1 and col.assigned == 1
    out = [loop.run_update() for _ in range(13)]
    assert [o["update"] for o in out] == list(range(13)) and col.gathers == 13
    assert len(lg.earlier) == 1 + 3                                    # snapshots after updates 4, 8, 12
    assert col.assigned == 1 + 13                                      # opponents re-drawn every update
    assert evals == [5, 10]
    assert abs(tr.optimiser.param_groups[0]["lr"] - tl.linear_lr(12, 50, 3e-4)) < 1e-18
    assert abs(tr.cfg.entropy_coef - (0.04 + ((12 - 2) / 10) * (0.005 - 0.04))) < 1e-15
    assert np.allclose(env.w, [0.75, 0.5, 0.25, 0.0])                  # updates 4..7 of the dense-reward anneal
    # checkpoint round trip
    loop2 = tl.TrainingLoop(_Env(), torch.nn.Linear(3, 3), _Collector(), _Trainer(net), args, league=League(envs_per_worker=5),
                            make_net=lambda: torch.nn.Linear(3, 3))
    loop2.load(str(tmp_path / "ck.pt"))
    assert loop2.update_num == 13 and len(loop2.league.earlier) == 4
    assert torch.equal(loop2.policy.weight, net.weight)
    # the reference's 5-tuple layout (robust_train.py:55,155)
    loop.save_reference_tuple(str(tmp_path / "ref.pt"))
    tup = torch.load(str(tmp_path / "ref.pt"), weights_only=False)
    assert len(tup) == 5 and tup[3] == 13 and len(tup[1]) == 4 and "weight" in tup[0] and tup[4].num_steps == 4
    loop3 = tl.TrainingLoop(_Env(), torch.nn.Linear(3, 3), _Collector(), _Trainer(net), args, league=League(envs_per_worker=5),
                            make_net=lambda: torch.nn.Linear(3, 3))
    loop3.load_reference_tuple(str(tmp_path / "ref.pt"))
    assert loop3.update_num == 13 and len(loop3.league.earlier) == 4 and torch.equal(loop3.policy.weight, net.weight)

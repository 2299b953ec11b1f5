"""The training loop around the PPO update (reference RL/robust_train.py:50-176 `run_update`), on one batched env per GPU.

Per update: linear learning-rate decay (RL/ppo/utils.py:1-5) -> rollout collection (rollout.RolloutCollector) -> PPO update
(train.PPOTrainer) -> entropy-coefficient annealing (robust_train.py:108-117) and dense-reward annealing (:119-124) ->
league bookkeeping (snapshot every `add_policy_every` updates, opponents re-drawn every `update_opponent_policies_every`,
:135-141) -> evaluation protocol every `eval_every` updates (:143-151) -> checkpoint.  The reference's watchdog /
fail_handler (:160-176) restarts crashed worker processes; there are none here.  The checkpoint holds the same items as the
reference's tuple (central state-dict, league deque, eval logs, update number, args) in a dict.
"""
import copy
import time

import torch

from . import dist as cdist


class TrainArgs(object):
    """RL/ppo/arguments.py defaults (only the fields this loop reads)."""
    lr = 3e-4
    use_linear_lr_decay = True
    entropy_coef_start = 0.04
    entropy_coef_final = 0.005
    entropy_coef_start_anneal = 500
    entropy_coef_end_anneal = 1500
    dense_reward_anneal_start = -1
    dense_reward_anneal_end = -1
    num_steps = 200
    total_env_steps = int(1e9)
    add_policy_every = 4
    update_opponent_policies_every = 1
    eval_every = 25
    num_eval_episodes = 128

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


def linear_lr(update_num, num_updates, initial_lr):
    """RL/ppo/utils.py:1-5"""
    return initial_lr - (initial_lr * (update_num / float(num_updates)))


def entropy_coef_at(update_num, args, current):
    """robust_train.py:108-117: linear between the anneal bounds, untouched outside (update_num > start and <= end)."""
    if args.entropy_coef_start_anneal < update_num <= args.entropy_coef_end_anneal:
        s, e = args.entropy_coef_start_anneal, args.entropy_coef_end_anneal
        return args.entropy_coef_start + ((update_num - s) / (e - s)) * (args.entropy_coef_final - args.entropy_coef_start)
    return current


def reward_weight_at(update_num, args, current):
    """robust_train.py:119-124: dense-reward annealing factor 1 -> 0 between its bounds."""
    if args.dense_reward_anneal_start < update_num <= args.dense_reward_anneal_end:
        s, e = args.dense_reward_anneal_start, args.dense_reward_anneal_end
        return 1.0 + ((update_num - s) / (e - s)) * (0.0 - 1.0)
    return current


class TrainingLoop(object):
    def __init__(self, env, policy, collector, trainer, args=None, league=None, make_net=None, evaluate=None, checkpoint_path=None):
        """evaluate(policy, update_num) -> (log, summary) (evaluation.run_evaluation_protocol bound to an env factory and an
        opponent) or None; make_net() builds an empty net for league opponents."""
        self.env, self.policy, self.collector, self.trainer = env, policy, collector, trainer
        self.args = args or TrainArgs()
        self.league, self.make_net, self.evaluate = league, make_net, evaluate
        self.checkpoint_path = checkpoint_path
        world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.steps_per_update = int(self.args.num_steps * env.n * world)                   # robust_train.py:93
        self.num_updates = int(self.args.total_env_steps) // self.steps_per_update          # :92
        self.update_num = 0
        self.entropy_coef = self.args.entropy_coef_start
        self.reward_weight = 1.0
        self.eval_logs = []
        self.start_time = time.time()
        trainer.cfg.entropy_coef = self.entropy_coef
        if league is not None:
            if len(league.earlier) == 0:
                # a fresh start (robust_train.py:60-64): the deque gets the initial policy, and the FIRST rollout is played against
                # independently random-initialised opponents - the nets every reference worker builds for its policy slots 1..3
                # (game_manager.py:14); snapshots are drawn for the first time after update 0 (:140-141).  One such set serves all
                # games here (the reference has one per worker process).
                league.add(policy)
                nets = [make_net() for _ in range(3)]
                for nt in nets:
                    nt.eval()
                collector.set_opponents(nets, torch.arange(3).expand(collector.N, 3))
            else:
                league.assign(collector, make_net)                                          # resumed: :55-56

    def run_update(self):
        a, u = self.args, self.update_num
        if a.use_linear_lr_decay:                                                           # :98-99
            lr = linear_lr(u, self.num_updates, a.lr)
            for g in self.trainer.optimiser.param_groups:
                g["lr"] = lr
        st = self.collector.gather_rollouts()                                               # :101-102
        losses = self.trainer.update(st)                                                    # :104
        self.collector.after_rollouts()
        self.entropy_coef = entropy_coef_at(u, a, self.entropy_coef)
        self.trainer.cfg.entropy_coef = self.entropy_coef
        w = reward_weight_at(u, a, self.reward_weight)
        if w != self.reward_weight:
            self.reward_weight = w
            self.env.set_reward_annealing_factor(w)                                             # rollout_manager.update_annealing_factor
        if self.league is not None:
            if u % a.add_policy_every == 0 and u > 0:                                       # :135-138
                self.league.add(self.policy)
            if u % a.update_opponent_policies_every == 0:                                   # :140-141
                self.league.assign(self.collector, self.make_net)
        summary = None
        if self.evaluate is not None and u % a.eval_every == 0 and u > 0:                   # :143-151
            log, summary = self.evaluate(self.policy, u)
            self.eval_logs.append(log)
        self.update_num += 1
        if self.checkpoint_path and (not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0):
            self.save(self.checkpoint_path)                                                 # :155-156
        return {"update": u, "losses": losses, "games_complete": st.games_complete, "entropy_coef": self.entropy_coef,
                "reward_weight": self.reward_weight, "policy_steps": self.steps_per_update * (u + 1),
                "hours": (time.time() - self.start_time) / 3600.0, "eval": summary}

    def save(self, path):
        torch.save({"central_policy": {k: v.detach().cpu() for k, v in self.policy.state_dict().items()},
                    "earlier_policies": list(self.league.earlier) if self.league is not None else [],
                    "eval_logs": self.eval_logs, "update_num": self.update_num, "args": copy.copy(self.args.__dict__),
                    "entropy_coef": self.entropy_coef, "reward_weight": self.reward_weight}, path)

    def save_reference_tuple(self, path):
        """The reference's own checkpoint layout (robust_train.py:155-156): the 5-tuple (central state-dict, deque of
        earlier state-dicts, eval logs, update number, args).  State-dict keys are the reference's (CatanPolicy keeps its
        parameter names); the reference's entries without learnable state (the empty `dummy_param`s of its sub-modules, the
        value normaliser's constants) are added to the central state-dict AND to every league entry, so that the reference's
        strict `load_state_dict` accepts them (`policy.CatanPolicy.to_reference_state_dict`)."""
        import argparse
        from collections import deque
        complete = getattr(type(self.policy), "to_reference_state_dict", None) or (lambda d: {k: v.detach().cpu() for k, v in d.items()})
        sd = complete(self.policy.state_dict())
        earlier = deque((complete(e) for e in (self.league.earlier if self.league is not None else [])),
                        maxlen=(self.league.earlier.maxlen if self.league is not None else 500))
        torch.save((sd, earlier, self.eval_logs, self.update_num, argparse.Namespace(**self.args.__dict__)), path)

    def load_reference_tuple(self, path):
        """robust_train.py:55: accepts the reference's 5-tuple (or its 7-tuple variant with entropy coefficient and reward weight)."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        sd, earlier, self.eval_logs, self.update_num = ck[0], ck[1], ck[2], ck[3]
        own = self.policy.state_dict()
        self.policy.load_state_dict({k: v for k, v in sd.items() if k in own})
        if self.league is not None:
            self.league.earlier.clear()
            self.league.earlier.extend({k: v for k, v in e.items() if k in own} for e in earlier)
            self.league.assign(self.collector, self.make_net)
        if len(ck) >= 7:
            self.entropy_coef, self.reward_weight = ck[5], ck[6]
            self.trainer.cfg.entropy_coef = self.entropy_coef
            self.env.set_reward_annealing_factor(self.reward_weight)

    def load(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.policy.load_state_dict(ck["central_policy"])
        if self.league is not None:
            self.league.earlier.clear(); self.league.earlier.extend(ck["earlier_policies"])
            self.league.assign(self.collector, self.make_net)                               # :55-56
        self.eval_logs, self.update_num = ck["eval_logs"], ck["update_num"]
        self.entropy_coef, self.reward_weight = ck["entropy_coef"], ck["reward_weight"]
        self.trainer.cfg.entropy_coef = self.entropy_coef
        self.env.set_reward_annealing_factor(self.reward_weight)

"""Drop-in surface of the reference's rollout / learner layers over the device pipeline.

`RL/robust_train.py` builds its training run from seven names; this module exports the same names with the same call
signatures, so that its `run_update` (robust_train.py:95-156) runs unchanged on top of the HIP env, the batched collector
and the fused GAE / PPO-loss kernels:

    reference import (robust_train.py:13-22)                      here
    RL.ppo.game_manager.make_game_manager                         make_game_manager(num_envs, num_steps)
    RL.ppo.vec_gather_experience.SubProcGameManager               SubProcGameManager(game_manager_fns)
    RL.models.build_agent_model.build_agent_model                 build_agent_model(device="cpu")
    RL.ppo.process_batch.BatchProcessor                           BatchProcessor(args, lstm_dim, device=...)
    RL.ppo.ppo.PPO                                                PPO(actor_critic, args)
    RL.ppo.update_opponent_policies.update_opponent_policies      update_opponent_policies(earlier_policies, manager, args)
    RL.ppo.vec_evaluation / evaluation_manager / run_evaluation_protocol
                                                                  SubProcEvaluationManager, make_evaluation_manager,
                                                                  run_evaluation_protocol

What stays the reference's: the CALL PROTOCOL and the tensor LAYOUTS at every call site (the 9-tuples of
`generator_standard`, `obs_dict` / 12-list action masks with the type-conditional heads transposed, `(T+1, N, .)` order).
What changes underneath: there are no worker processes or pipes - the "processes" are consecutive groups of games of ONE
batched env on this rank's GPU (`.processes` keeps its length and its `.kill()`); `gather_rollouts()` returns a handle to
the device-resident rollout storage instead of nested Python lists (`BatchProcessor.process_rollouts` accepts both; the
handle's `to_reference_lists()` produces the reference's nested 7-tuples when somebody wants them); `stored_device` is
ignored (rollouts never leave HBM).  GAE and the PPO loss run on the HIP kernels (`ppo.compute_gae`, `ppo.ppo_loss`);
with world_size > 1 the advantage statistics and the gradients are all-reduced (`dist`).
"""
import copy
import time

import numpy as np
import torch
import torch.nn as nn

from . import dist as cdist
from . import nn_kernels
from . import ppo as ppo_kernels
from . import spec
from .policy import CatanPolicy
from .rollout import RolloutCollector, RolloutStorage, pack_action_masks

# Defaults of the keyword-only extras of the managers below; `install(**defaults)` / `configure(**defaults)` set them, so
# that a script which constructs the managers with the reference's positional arguments only (robust_train.py:47-50,86-89)
# can still be pointed at a device / seed / env factory.
_DEFAULTS = {"device": None, "seed": 0, "env_factory": None, "eval_env_factory": None, "make_policy": None,
             "autocast_dtype": "auto", "env_kwargs": None, "self_play": False}


_UNSET = object()


def _default(name, value):
    return _DEFAULTS[name] if value is _UNSET else value


def configure(**kw):
    for k, v in kw.items():
        if k not in _DEFAULTS:
            raise KeyError(k)
        _DEFAULTS[k] = v


def install(**defaults):
    """Registers this module's classes under the reference's module names (`RL.ppo.game_manager`,
    `RL.ppo.vec_gather_experience`, `RL.models.build_agent_model`, `RL.ppo.process_batch`, `RL.ppo.ppo`,
    `RL.ppo.update_opponent_policies`, `RL.ppo.vec_evaluation`, `RL.ppo.evaluation_manager`,
    `RL.ppo.run_evaluation_protocol`) in `sys.modules`: after `reference_api.install()` the UNMODIFIED
    `RL/robust_train.py` of a reference checkout imports the device pipeline through its own import lines
    (robust_train.py:13-22).  Everything else of the `RL` package (arguments, utils) stays the reference's."""
    import sys
    import types
    configure(**defaults)
    me = sys.modules[__name__]
    table = {
        "RL.ppo.game_manager": ("make_game_manager",),
        "RL.ppo.vec_gather_experience": ("SubProcGameManager",),
        "RL.models.build_agent_model": ("build_agent_model",),
        "RL.ppo.process_batch": ("BatchProcessor", "OBS_KEYS", "OBS_TYPES", "TYPE_CONDITIONAL_MASKS"),
        "RL.ppo.ppo": ("PPO",),
        "RL.ppo.update_opponent_policies": ("update_opponent_policies", "get_prob_dist"),
        "RL.ppo.vec_evaluation": ("SubProcEvaluationManager",),
        "RL.ppo.evaluation_manager": ("make_evaluation_manager",),
        "RL.ppo.run_evaluation_protocol": ("run_evaluation_protocol",),
    }
    for name, attrs in table.items():
        mod = types.ModuleType(name)
        mod.__doc__ = f"alias installed by {__name__}.install()"
        for a in attrs:
            setattr(mod, a, getattr(me, a))
        sys.modules[name] = mod
    return sorted(table)


# interface constants of RL/ppo/process_batch.py:10-16
OBS_KEYS = list(spec.OBS_KEYS)
OBS_TYPES = ["list" if k in spec.OBS_LIST_KEYS else "normal" for k in OBS_KEYS]
TYPE_CONDITIONAL_MASKS = [1, 6, 9]
_LIST_INDEX = {k: i for i, k in enumerate(spec.OBS_LIST_KEYS)}


# ------------------------------------------------------------------------------------------------ layout conversions
def obs_flat_to_dict(f, lists, pad=None):
    """flat device layout (obs_f [B,1787], lists [B,5,25]) -> the reference's obs dict of batched tensors
    (process_batch.py:41-51 after stacking: float keys [B, ...] fp32, list keys [B, pad] int64 zero padded)."""
    B = f.shape[0]
    out, o = {}, spec.OBS_FLOAT_OFFSETS
    for k, shp in spec.OBS_FLOAT_KEYS.items():
        n = int(np.prod(shp))
        out[k] = f[:, o[k]:o[k] + n].reshape((B,) + tuple(shp)).float()
    for k, i in _LIST_INDEX.items():       # pad: one width per list key (pad_sequence pads every key to ITS longest list)
        w = spec.OBS_LIST_PAD if pad is None else int(pad[i] if isinstance(pad, (list, tuple)) else pad)
        out[k] = lists[:, i, :w].long()
    return out


def obs_dict_to_flat(obs):
    """the reference's obs dict (batched tensors, or the single-observation form of `obs_to_torch` whose list keys are
    `[tensor]`) -> (obs_f [B,1787] fp32, lists [B,5,25] int32, lens [B,5] int32; an empty card list has length 1 = the
    reference's `[0]`, env/wrapper.py:642-655)."""
    B = obs["proposed_trade"].shape[0]
    f = torch.cat([obs[k].reshape(B, -1).float() for k in spec.OBS_FLOAT_KEYS], 1)
    lists = torch.zeros((B, 5, spec.OBS_LIST_PAD), dtype=torch.int32, device=f.device)
    for k, i in _LIST_INDEX.items():
        v = obs[k]
        if isinstance(v, (list, tuple)):
            v = torch.nn.utils.rnn.pad_sequence([x.reshape(-1) for x in v], batch_first=True)
        v = v.reshape(B, -1)
        lists[:, i, :v.shape[1]] = v.to(torch.int32)
    lens = (lists != 0).sum(-1).clamp(min=1).to(torch.int32)
    return f, lists, lens


def masks_flat_to_list(m):
    """[B,325] -> the 12 mask tensors as the policy receives them: heads 1, 6, 9 are (types, B, d)
    (RL/models/policy.py:186-191), the others (B, d)."""
    B = m.shape[0]
    out = []
    for hi, (off, sz, shp) in enumerate(zip(spec.MASK_OFFSETS, spec.MASK_SIZES, spec.MASK_SHAPES)):
        mk = m[:, off:off + sz].reshape((B,) + tuple(shp)).float()
        out.append(mk.transpose(0, 1).contiguous() if hi in TYPE_CONDITIONAL_MASKS else mk)
    return out


def masks_list_to_flat(masks):
    parts = []
    for hi, mk in enumerate(masks):
        mk = torch.as_tensor(mk)
        if hi in TYPE_CONDITIONAL_MASKS:
            mk = mk.transpose(0, 1)
        parts.append(mk.reshape(mk.shape[0], -1).float())
    return torch.cat(parts, 1)


def actions_flat_to_list(a):
    """int64 [B,18] -> 12 tensors [B,1] (heads 7, 8: [B,4]) - the `actions_batch` of generator_standard"""
    return [a[:, off:off + ln].long() for off, ln in spec.ACTION_HEAD_SLICES]


def actions_list_to_flat(actions):
    cols = []
    for h in actions:
        if isinstance(h, (list, tuple)):                       # heads 7 / 8 out of `act`: four [B,1] tensors
            cols.append(torch.cat([torch.as_tensor(t).reshape(-1, 1) for t in h], 1))
        else:
            h = torch.as_tensor(h)
            cols.append(h.reshape(h.shape[0], -1) if h.dim() > 1 else h.reshape(-1, 1))
    return torch.cat(cols, 1).long()


# ------------------------------------------------------------------------------------------------ policy (L2 surface)
class _ValueNormaliser(object):
    """ValueFunctionNormaliser (RL/models/utils.py:9-22)"""

    def __init__(self, mean, std):
        self.mean_np, self.std_np = float(mean), float(std)

    def normalise(self, values):
        return (values - self.mean_np) / (self.std_np + 1e-4)

    def denormalise(self, normalised_values):
        return self.mean_np + normalised_values * self.std_np


class SettlersAgentPolicy(nn.Module):
    """`policy.CatanPolicy` behind the call signatures of the reference net (RL/models/policy.py:71-111,168-199): obs
    dicts, 12-list masks / actions, `(value, actions, log_probs, hidden)` tuples, and the reference's `state_dict` key set."""

    policy_type = "neural_network"
    use_value_normalisation = True

    def __init__(self, net=None, autocast_dtype=None):
        super().__init__()
        self.net = CatanPolicy() if net is None else net
        self.dummy_param = nn.Parameter(torch.empty(0))
        self.autocast_dtype = autocast_dtype
        self.value_normaliser = _ValueNormaliser(self.net.VALUE_MEAN, self.net.VALUE_STD)
        self.sample_generator = None

    include_lstm = property(lambda self: self.net.include_lstm)
    lstm_size = property(lambda self: self.net.lstm_size)

    def _ctx(self):
        on = self.autocast_dtype is not None and self.dummy_param.is_cuda
        return torch.autocast(device_type="cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=on)

    def _hidden_kw(self, hidden_states, nonterminal_masks):
        if not self.include_lstm:
            return {}
        return {"hidden": hidden_states, "nonterminal": None if nonterminal_masks is None else nonterminal_masks.reshape(-1)}

    def act(self, obs_dict, hidden_states, nonterminal_masks, action_masks, deterministic=False, return_entropy=False,
            condition_on_action_type=None, log_specific_action_output=False):
        if return_entropy or log_specific_action_output:
            raise NotImplementedError("return_entropy / log_specific_action_output are only used by the reference's GUI")
        f, lists, lens = obs_dict_to_flat(obs_dict)
        cond = None
        if condition_on_action_type is not None:
            cond = torch.as_tensor(condition_on_action_type, device=f.device).reshape(-1).long().expand(f.shape[0])
        with self._ctx():
            res = self.net.act(f, lists, lens.long(), masks_list_to_flat(action_masks), deterministic=deterministic,
                               generator=self.sample_generator, condition_on_action_type=cond,
                               **self._hidden_kw(hidden_states, nonterminal_masks))
        a = res[1]
        actions = [[a[:, off + j:off + j + 1] for j in range(ln)] if ln > 1 else a[:, off:off + 1]
                   for off, ln in spec.ACTION_HEAD_SLICES]
        return res[0], actions, res[2], (res[3] if self.include_lstm else hidden_states)

    def evaluate_actions(self, obs_dict, hidden_states, nonterminal_masks, actions, action_masks):
        f, lists, lens = obs_dict_to_flat(obs_dict)
        with self._ctx():
            res = self.net.evaluate_actions(f, lists, lens.long(), masks_list_to_flat(action_masks), actions_list_to_flat(actions),
                                            **self._hidden_kw(hidden_states, nonterminal_masks))
        return res[0], res[1], res[2], (res[3] if self.include_lstm else hidden_states)

    def get_value(self, obs_dict, hidden_states, nonterminal_masks):
        f, lists, lens = obs_dict_to_flat(obs_dict)
        with self._ctx():
            return self.net.get_value(f, lists, lens.long(), **self._hidden_kw(hidden_states, nonterminal_masks))

    # RL/models/policy.py:168-199: numpy observation / masks of ONE env -> torch, and sampled actions -> numpy
    def obs_to_torch(self, obs):
        dev = self.dummy_param.device
        for k in spec.OBS_FLOAT_KEYS:
            v = np.stack([np.asarray(t, dtype=np.float32) for t in obs[k]]) if k == "tile_representations" else np.asarray(obs[k], dtype=np.float32)
            obs[k] = torch.from_numpy(v).to(dev)[None]
        for k in spec.OBS_LIST_KEYS:
            obs[k] = [torch.as_tensor(np.asarray(obs[k]), dtype=torch.long, device=dev)]
        return obs

    def act_masks_to_torch(self, masks):
        dev = self.dummy_param.device
        for z in range(len(masks)):
            m = torch.as_tensor(np.asarray(masks[z]), dtype=torch.float32, device=dev)[None]
            masks[z] = m.transpose(0, 1) if z in TYPE_CONDITIONAL_MASKS else m
        return masks

    def torch_act_to_np(self, action):
        for z in range(len(action)):
            if isinstance(action[z], list):
                action[z] = [t.squeeze().cpu().numpy() for t in action[z]]
            else:
                action[z] = action[z].squeeze().cpu().numpy()
        return action

    # the reference's state-dict key set, both ways
    def state_dict(self, *a, **kw):
        return CatanPolicy.to_reference_state_dict(self.net.state_dict())

    def load_state_dict(self, sd, strict=True):
        self.net.load_reference_state_dict(sd)


def build_agent_model(device="cpu", autocast_dtype="auto", include_lstm=False):
    """RL/models/build_agent_model.py:35-157 `build_agent_model(device)`.  autocast_dtype "auto": bf16 on a GPU
    (fp32 master weights), none on the CPU."""
    dev = torch.device(device)
    if autocast_dtype == "auto":
        autocast_dtype = torch.bfloat16 if dev.type == "cuda" else None
    return SettlersAgentPolicy(CatanPolicy(include_lstm=include_lstm), autocast_dtype=autocast_dtype).to(dev)


# ------------------------------------------------------------------------------------------------ rollout manager (L3)
class _GameManagerSpec(object):
    """What `make_game_manager(num_envs, num_steps)` returns: the reference returns a thunk that builds one worker's
    GamesAndPoliciesManager (game_manager.py:168-172); here it only carries the numbers - calling it makes no sense
    without a worker process."""

    def __init__(self, num_envs, num_steps):
        self.num_envs, self.num_steps = int(num_envs), int(num_steps)

    def __call__(self):
        raise RuntimeError("there are no worker processes: pass the list of specs to SubProcGameManager")


def make_game_manager(num_envs, num_steps):
    return _GameManagerSpec(num_envs, num_steps)


class _ProcessStub(object):
    """An entry of `.processes` (robust_train.py:162-165 kills and re-creates workers in its fail handler)."""

    def kill(self):
        pass

    def is_alive(self):
        return True

    def join(self, timeout=None):
        pass


class DeviceRollouts(object):
    """What `SubProcGameManager.gather_rollouts()` returns instead of the nested lists: the device-resident storage plus
    the game ranges of the "processes"."""

    def __init__(self, storage, envs_per_process, policy=None):
        self.storage, self.envs_per_process = storage, list(envs_per_process)

    def __len__(self):
        return len(self.envs_per_process)

    def to_reference_lists(self):
        """-> the reference's return value: a list over processes of 7-tuples (observations, hidden_states, rewards,
        actions, action_masks, action_log_probs, terminal_masks), each `[env][t]` (game_manager.py:137-140): observations
        are `obs_to_torch`-style dicts, actions the in-place converted numpy form of `torch_act_to_np`."""
        st = self.storage
        T, out, g0 = st.T, [], 0
        f_all, lists_all, lens_all = st.obs_f.cpu(), st.lists.cpu(), st.lens.cpu()
        am = st.unpack_action_masks(st.action_masks).cpu()
        acts, lps, rews, tms = st.actions.cpu(), st.action_log_probs.cpu(), st.rewards.cpu(), st.masks.cpu()
        L = st.hidden.shape[-1] if st.hidden is not None else 256
        for n_env in self.envs_per_process:
            obs_l, hid_l, rew_l, act_l, am_l, lp_l, tm_l = [], [], [], [], [], [], []
            for g in range(g0, g0 + n_env):
                o_env, h_env = [], []
                for t in range(T + 1):
                    d = obs_flat_to_dict(f_all[t, g:g + 1], lists_all[t, g:g + 1].long())
                    for k, i in _LIST_INDEX.items():
                        d[k] = [lists_all[t, g, i, :int(lens_all[t, g, i])].long()]
                    o_env.append(d)
                    if st.hidden is not None:
                        h_env.append((st.hidden[0, t, g:g + 1].cpu(), st.hidden[1, t, g:g + 1].cpu()))
                    else:
                        h_env.append((torch.zeros(1, L), torch.zeros(1, L)))
                obs_l.append(o_env); hid_l.append(h_env)
                rew_l.append([float(rews[t, g]) for t in range(T)])
                act_l.append([[(acts[t, g, off:off + ln].numpy() if ln > 1 else acts[t, g, off].numpy()) for off, ln in spec.ACTION_HEAD_SLICES]
                              for t in range(T)])
                am_l.append([masks_flat_to_list(am[t, g:g + 1]) for t in range(T)])
                lp_l.append([lps[t, g].reshape(1, 1) for t in range(T)])
                tm_l.append([float(tms[t, g]) for t in range(T + 1)])
            out.append((obs_l, hid_l, rew_l, act_l, am_l, lp_l, tm_l))
            g0 += n_env
        return out


class SubProcGameManager(object):
    """RL/ppo/vec_gather_experience.py:79-152.  `game_manager_fns`: the list robust_train.py:47-50 builds with
    `make_game_manager`; every entry becomes a group of `num_envs` consecutive games of ONE batched env that share their
    three opponent nets (the reference worker's `policies[1..3]`).

    Keyword-only extras (all optional): device, seed (Philox key of the games; global game ids start at `env_id0`, by
    default rank * number of games so that sharding over ranks does not change any game), env_factory(n) (tests),
    make_policy() (architecture of the acting nets), autocast_dtype, env_kwargs (EnvWrapper's keyword arguments),
    self_play=True (every seat plays the central policy: no opponent nets; update_policy with policy_id 1..3 is refused),
    collector_kwargs (rollout.RolloutCollector's: deferred_window, act_buckets)."""

    def __init__(self, game_manager_fns, start_method=None, *, device=_UNSET, seed=_UNSET, env_id0=None, env_factory=_UNSET,
                 make_policy=_UNSET, autocast_dtype=_UNSET, env_kwargs=_UNSET, self_play=_UNSET, collector_kwargs=None):
        device, seed, env_factory = _default("device", device), _default("seed", seed), _default("env_factory", env_factory)
        make_policy, autocast_dtype = _default("make_policy", make_policy), _default("autocast_dtype", autocast_dtype)
        env_kwargs, self_play = _default("env_kwargs", env_kwargs), _default("self_play", self_play)
        specs = list(game_manager_fns)
        if not specs or any(not isinstance(s, _GameManagerSpec) for s in specs):
            raise TypeError("SubProcGameManager expects the objects make_game_manager(num_envs, num_steps) returns")
        if len({s.num_steps for s in specs}) != 1:
            raise ValueError("all game managers must use the same num_steps")
        self.envs_per_process = [s.num_envs for s in specs]
        self.num_steps = specs[0].num_steps
        self.n = sum(self.envs_per_process)
        self.processes = [_ProcessStub() for _ in specs]
        self.waiting, self.closed = False, False
        if env_factory is None:
            from .env import VecCatanEnv
            rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
            kw = dict(env_kwargs or {})
            self.env = VecCatanEnv(self.n, seed=seed, env_id0=rank * self.n if env_id0 is None else env_id0, device=device,
                                   auto_reset=True, **kw)
        else:
            self.env = env_factory(self.n)
        self.device = self.env.device
        if autocast_dtype == "auto":
            autocast_dtype = torch.bfloat16 if torch.device(self.device).type == "cuda" else None
        self._make_policy = make_policy or CatanPolicy
        # game_manager.py:13-15: four independently initialised nets per worker.  Here: ONE central acting net and, until
        # the league installs snapshots (update_policy with policy_id 1..3), one random-initialised opponent per policy slot
        # shared by all groups (deviation: the reference's first rollout has 3 x n_processes independent random nets).
        self.central = self._make_policy().to(self.device).eval()
        self.self_play = bool(self_play)
        self._opp_sd = [[None] * 3 for _ in specs]
        self._default_opp = [] if self.self_play else [self._make_policy().to(self.device).eval() for _ in range(3)]
        self._opp_dirty = not self.self_play
        self.collector = RolloutCollector(self.env, self.central, self.num_steps, seed=seed, autocast_dtype=autocast_dtype, **(collector_kwargs or {}))
        self._carry_pending = False

    # ---- vec_gather_experience.py:104-118
    def gather_async(self):
        self.waiting = True

    def gather_wait(self):
        self.waiting = False
        if self._carry_pending:                       # the worker's `_after_rollouts()` (vec_gather_experience.py:22-24): the storage
            self.collector.after_rollouts()           # is shared with the learner, so the carry-over waits until the next gather
        if self._opp_dirty:
            self._install_opponents()
        st = self.collector.gather_rollouts()
        self._carry_pending = True
        return DeviceRollouts(st, self.envs_per_process)

    def gather_rollouts(self):
        self.gather_async()
        return self.gather_wait()

    def _install_opponents(self):
        """one net per DISTINCT state-dict object handed to update_policy (np.random.choice returns the deque's own dict
        objects, update_opponent_policies.py:22-27), per-game indices into them"""
        nets, index_of, rows = list(self._default_opp), {}, []
        for p, n_env in enumerate(self.envs_per_process):
            row = []
            for slot in range(3):
                sd = self._opp_sd[p][slot]
                if sd is None:
                    row.append(slot)
                    continue
                if id(sd) not in index_of:
                    net = self._make_policy().to(self.device).eval()
                    net.load_reference_state_dict(sd)
                    index_of[id(sd)] = len(nets)
                    nets.append(net)
                row.append(index_of[id(sd)])
            rows += [row] * n_env
        idx = torch.tensor(rows, dtype=torch.int64)
        used = sorted(set(idx.reshape(-1).tolist()))                      # drop the default nets nobody plays any more
        remap = {u: i for i, u in enumerate(used)}
        self.collector.set_opponents([nets[u] for u in used], idx.apply_(lambda v: remap[v]))
        self._opp_dirty = False

    def update_policy(self, state_dict, process_id=None, policy_id=0):
        if policy_id == 0:
            if process_id is not None:
                raise NotImplementedError("the central policy (policy_id 0) is one net for all games")
            self.central.load_reference_state_dict(state_dict)
            return [True] * len(self.processes)
        if not 1 <= policy_id <= 3:
            raise ValueError("policy_id must be 0..3")
        if self.self_play:
            raise RuntimeError("self_play=True: there are no opponent policy slots")
        for p in (range(len(self.processes)) if process_id is None else [process_id]):
            self._opp_sd[p][policy_id - 1] = state_dict
        self._opp_dirty = True
        return True if process_id is not None else [True] * len(self.processes)

    def update_annealing_factor(self, annealing_factor):
        self.env.set_reward_annealing_factor(annealing_factor)         # game_manager.py:164-166
        return [True] * len(self.processes)

    def reset(self):
        self.env.reset()                                               # game_manager.py:35-59
        self.collector.reset()
        self._carry_pending = False
        return [True] * len(self.processes)

    def seed(self, seeds):
        """The reference seeds each worker's global generators (vec_gather_experience.py:39-43).  Here the games' Philox
        streams are keyed at creation; only the action-sampling generator is re-seeded."""
        self.collector.sample_gen.manual_seed(int(seeds[0]))
        return [True] * len(self.processes)

    def close(self):
        self.closed = True


from .league import get_prob_dist  # noqa: E402  (RL/ppo/update_opponent_policies.py:29-43)


def update_opponent_policies(earlier_policies, rollout_manager, args, rng=None):
    """RL/ppo/update_opponent_policies.py:13-27 (same draws from numpy's global generator, same call order)."""
    p = get_prob_dist(num_policies=len(earlier_policies))
    rng = np.random if rng is None else rng
    for i in range(len(rollout_manager.processes)):
        policy_dicts = rng.choice(earlier_policies, 3, p=p)
        for slot in range(3):
            rollout_manager.update_policy(policy_dicts[slot], process_id=i, policy_id=slot + 1)


# ------------------------------------------------------------------------------------------------ rollout storage (L4)
class BatchProcessor(object):
    """RL/ppo/process_batch.py:18-200 over `rollout.RolloutStorage`.  After `process_rollouts` the reference's attributes
    exist with the reference's layouts - `obs_dict[key] (T+1,N,.)`, `rewards (T,N,1)`, `actions` 12 x `(T,N,1|4)`,
    `action_masks` 12 x `(T,N,d)` / `(types,T,N,d)` for heads 1, 6, 9, `action_log_probs (T,N,1)`, `masks (T+1,N,1)`,
    `hidden_states`; after `compute_advantages_alt`: `values (T+1,N,1)`, `returns`, `advantages (T,N,1)`.  They are
    materialised on first access (views / unpacked copies of the flat device storage); the hot loops below read the flat
    storage directly and only lay out each MINIBATCH in the reference's form."""

    value_chunk_rows = 262144

    def __init__(self, args, lstm_dim, obs_keys=OBS_KEYS, obs_type=OBS_TYPES, type_conditional_masks=TYPE_CONDITIONAL_MASKS,
                 num_action_heads=12, device="cuda", stored_device="cpu"):
        self.args = args
        self.num_steps = args.num_steps
        self.num_parallel = args.num_processes * args.num_envs_per_process
        self.obs_keys, self.obs_type, self.type_conditional_masks = obs_keys, obs_type, type_conditional_masks
        self.lstm_dim, self.num_action_heads = lstm_dim, num_action_heads
        self.games_complete = 0
        self.stored_device, self.device = stored_device, device
        self.storage = None
        self._cache = {}
        self._values = self._returns = self._adv = None
        self._gen = None

    # ---- process_rollouts (process_batch.py:37-104)
    def process_rollouts(self, rollouts):
        if isinstance(rollouts, DeviceRollouts):
            st = rollouts.storage
        else:
            st = storage_from_reference_lists(rollouts, self.num_steps, self.device, self.lstm_dim)
        if st.T != self.num_steps or st.N != self.num_parallel:
            raise ValueError(f"rollouts are {st.T} steps x {st.N} games, expected {self.num_steps} x {self.num_parallel}")
        self.storage = st
        self._cache = {}
        self._values = self._returns = self._adv = None
        self._list_pad = [max(1, int(v)) for v in st.lens[:st.T + 1].reshape(-1, 5).max(0).values.tolist()]   # pad_sequence, per key
        self.games_complete += int(torch.sum(1.0 - st.masks[:st.T + 1]).item())       # process_batch.py:104

    def _lazy(self, name, build):
        if name not in self._cache:
            self._cache[name] = build()
        return self._cache[name]

    @property
    def obs_dict(self):
        def build():
            st, T1, N = self.storage, self.storage.T + 1, self.storage.N
            d = obs_flat_to_dict(st.obs_f[:T1].reshape(T1 * N, -1), st.lists[:T1].reshape(T1 * N, 5, -1), self._list_pad)
            return {k: v.reshape((T1, N) + tuple(v.shape[1:])) for k, v in d.items()}
        return self._lazy("obs_dict", build)

    @property
    def hidden_states(self):
        st = self.storage
        if st.hidden is not None:
            return (st.hidden[0, :st.T + 1], st.hidden[1, :st.T + 1])
        return self._lazy("hidden", lambda: tuple(torch.zeros((st.T + 1, st.N, self.lstm_dim), device=st.obs_f.device) for _ in range(2)))

    rewards = property(lambda self: self.storage.rewards[:self.storage.T, :, None])
    action_log_probs = property(lambda self: self.storage.action_log_probs[:, :, None])
    masks = property(lambda self: self.storage.masks[:self.storage.T + 1, :, None])

    @property
    def actions(self):
        return self._lazy("actions", lambda: [self.storage.actions[:, :, off:off + ln] for off, ln in spec.ACTION_HEAD_SLICES])

    @property
    def action_masks(self):
        def build():
            st = self.storage
            m = st.unpack_action_masks(st.action_masks)                                 # (T,N,325)
            out = []
            for hi, (off, sz, shp) in enumerate(zip(spec.MASK_OFFSETS, spec.MASK_SIZES, spec.MASK_SHAPES)):
                mk = m[:, :, off:off + sz].reshape((st.T, st.N) + tuple(shp))
                out.append(mk.permute(2, 0, 1, 3).contiguous() if hi in self.type_conditional_masks else mk)
            return out
        return self._lazy("action_masks", build)

    values = property(lambda self: self._values[:, :, None])
    returns = property(lambda self: self._returns[:, :, None])
    advantages = property(lambda self: self._adv[:, :, None])

    # ---- compute_advantages_alt (process_batch.py:106-142)
    def compute_advantages_alt(self, actor_critic, max_processes_at_once=10):
        st = self.storage
        T1, N = st.T + 1, st.N
        dev = st.obs_f.device
        f = st.obs_f[:T1].reshape(T1 * N, -1); lists = st.lists[:T1].reshape(T1 * N, 5, -1); lens = st.lens[:T1].reshape(T1 * N, 5)
        rec = bool(getattr(actor_critic, "include_lstm", False))
        out = torch.empty((T1 * N,), dtype=torch.float32, device=dev)
        ch = self.value_chunk_rows
        if rec:
            hid = st.hidden[:, :T1].reshape(2, T1 * N, -1); nt = st.masks[:T1].reshape(T1 * N, 1)
        with torch.no_grad():
            for s in range(0, T1 * N, ch):      # (the reference chunks by games, 10 at a time; any chunking gives the same values)
                if isinstance(actor_critic, SettlersAgentPolicy):
                    with actor_critic._ctx():
                        v = actor_critic.net.get_value(f[s:s + ch].float(), lists[s:s + ch], lens[s:s + ch].long(),
                                                       **(dict(hidden=(hid[0, s:s + ch], hid[1, s:s + ch]), nonterminal=nt[s:s + ch, 0]) if rec else {}))
                else:                           # any net with the reference's signature (e.g. the reference net itself)
                    v = actor_critic.get_value(obs_flat_to_dict(f[s:s + ch], lists[s:s + ch], self._list_pad),
                                               (hid[0, s:s + ch], hid[1, s:s + ch]) if rec else None, nt[s:s + ch] if rec else None)
                out[s:s + ch] = v.reshape(-1).float()
        if getattr(actor_critic, "use_value_normalisation", False):
            out = actor_critic.value_normaliser.denormalise(out)
        self._values = out.reshape(T1, N)
        self._returns, self._adv = _GAE(st.rewards[:st.T].contiguous(), self._values, st.masks[:T1].contiguous(),
                                        self.args.gamma, self.args.gae_lambda)

    # ---- generator_standard (process_batch.py:169-200)
    def generator_standard(self, num_mini_batch):
        st = self.storage
        T, N = st.T, st.N
        batch_size = T * N
        mini_batch_size = batch_size // num_mini_batch
        dev = st.obs_f.device
        if self._gen is None:
            self._gen = torch.Generator(device=dev)
            self._gen.manual_seed(int(torch.initial_seed() % (2 ** 31)) + 17 * (1 + (torch.distributed.get_rank() if torch.distributed.is_initialized() else 0)))
        perm = torch.randperm(batch_size, generator=self._gen, device=dev)              # SubsetRandomSampler
        f_all = st.obs_f[:T].reshape(batch_size, -1); lists_all = st.lists[:T].reshape(batch_size, 5, -1)
        acts_all = st.actions.reshape(batch_size, -1); am_all = st.action_masks.reshape(batch_size, -1)
        vp = self._values[:T].reshape(batch_size, 1); ret = self._returns.reshape(batch_size, 1)
        mk = st.masks[:T].reshape(batch_size, 1); lp = st.action_log_probs.reshape(batch_size, 1); adv = self._adv.reshape(batch_size, 1)
        for b in range(batch_size // mini_batch_size):                                  # BatchSampler(drop_last=True)
            idx = perm[b * mini_batch_size:(b + 1) * mini_batch_size]
            obs_dict_batch = obs_flat_to_dict(f_all[idx], lists_all[idx], self._list_pad)
            actions_batch = actions_flat_to_list(acts_all[idx])
            action_masks_batch = masks_flat_to_list(st.unpack_action_masks(am_all[idx]))
            yield (obs_dict_batch, None, actions_batch, action_masks_batch, vp[idx], ret[idx], mk[idx], lp[idx], adv[idx])


    # ---- generator_lstm (process_batch.py:203-293)
    def _permutation(self, n, device):
        """`np.random.permutation(len(time_inds))` (process_batch.py:216) - drawn on the device from this storage's generator"""
        if self._gen is None:
            self._gen = torch.Generator(device=device)
            self._gen.manual_seed(int(torch.initial_seed() % (2 ** 31)) + 17 * (1 + (torch.distributed.get_rank() if torch.distributed.is_initialized() else 0)))
        return torch.randperm(n, generator=self._gen, device=device)

    def generator_lstm(self, num_mini_batch, total_batch_size, truncated_seq_len):
        """Truncated-BPTT minibatches with the reference's 9-tuples: every game's T stored decisions are cut into T / L
        consecutive pieces (numbered game-major, :210-215), a minibatch is `total_batch_size // num_mini_batch // L` randomly
        drawn pieces laid out TIME-major (`_flatten_helper`: row l * n + j is step l of piece j); `recurrent_hidden_batch` =
        [h, c], each (n, lstm) = the state stored at each piece's first decision (:238-239,257-258); the masks of heads 1, 6, 9
        come as (types, L * n, d) (:281-284)."""
        from .train import lstm_minibatches
        st = self.storage
        T, N = st.T, st.N
        dev = st.obs_f.device
        if total_batch_size != T * N:
            raise ValueError(f"total_batch_size {total_batch_size} != num_steps * num_parallel = {T * N}")
        perm = self._permutation(N * (T // truncated_seq_len), dev)
        hid = self.hidden_states
        f_all = st.obs_f[:T].reshape(T * N, -1); lists_all = st.lists[:T].reshape(T * N, 5, -1)
        acts_all = st.actions.reshape(T * N, -1); am_all = st.action_masks.reshape(T * N, -1)
        vp = self._values[:T].reshape(T * N, 1); ret = self._returns.reshape(T * N, 1)
        mk = st.masks[:T].reshape(T * N, 1); lp = st.action_log_probs.reshape(T * N, 1); adv = self._adv.reshape(T * N, 1)
        for t, game in lstm_minibatches(T, N, truncated_seq_len, num_mini_batch, perm):
            idx = (t * N + game[None, :]).reshape(-1)                                   # time-major rows of the flat (T*N) tensors
            obs_dict_batch = obs_flat_to_dict(f_all[idx], lists_all[idx], self._list_pad)
            recurrent_hidden_batch = [hid[0][t[0], game].to(dev), hid[1][t[0], game].to(dev)]
            actions_batch = actions_flat_to_list(acts_all[idx])
            action_masks_batch = masks_flat_to_list(st.unpack_action_masks(am_all[idx]))
            yield (obs_dict_batch, recurrent_hidden_batch, actions_batch, action_masks_batch, vp[idx], ret[idx], mk[idx], lp[idx], adv[idx])


def storage_from_reference_lists(rollouts, num_steps, device, lstm_dim=0):
    """The reference's nested rollouts (list over processes of 7-tuples, game_manager.py:137-140) -> RolloutStorage
    (process_batch.py:37-104 restated for the flat layout).  Interop path; the device collector never goes through it."""
    per = list(zip(*rollouts))
    flat = [[inner for outer in per[i] for inner in outer] for i in range(7)]
    obs, hid, rew, act, amask, lp, tm = flat
    N, T = len(obs), num_steps
    st = RolloutStorage(T, N, device)
    for k in range(N):
        for t in range(T + 1):
            f, lists, lens = obs_dict_to_flat(obs[k][t])
            st.obs_f[t, k] = f[0].to(device); st.lists[t, k] = lists[0].to(device=device, dtype=torch.int8)
            st.lens[t, k] = lens[0].to(device=device, dtype=torch.int8)
            st.masks[t, k] = float(tm[k][t])
        for t in range(T):
            st.rewards[t, k] = float(rew[k][t])
            st.actions[t, k] = torch.as_tensor(np.concatenate([np.asarray(h).reshape(-1) for h in act[k][t]])).to(device)
            st.action_log_probs[t, k] = float(torch.as_tensor(lp[k][t]).reshape(-1)[0])
            st.action_masks[t, k] = pack_action_masks(masks_list_to_flat(amask[k][t]))[0].to(device)
    return st


# ------------------------------------------------------------------------------------------------ PPO (L4)
_GAE = ppo_kernels.compute_gae          # the fused HIP kernels; module attributes so that CPU tests can put stand-ins here
_LOSS = ppo_kernels.ppo_loss


class PPO(object):
    """RL/ppo/ppo.py:4-79: same constructor, same `update(rollout_storage)`, same return value."""

    def __init__(self, actor_critic, args):
        self.actor_critic, self.args = actor_critic, args
        self.clip_param, self.ppo_epoch, self.num_mini_batch = args.clip_param, args.ppo_epoch, args.num_mini_batch
        self.value_loss_coef, self.entropy_coef = args.value_loss_coef, args.entropy_coef_start
        self.max_grad_norm, self.recompute_returns = args.max_grad_norm, getattr(args, "recompute_returns", True)
        self.gamma, self.gae_lambda = args.gamma, args.gae_lambda
        from .optim import FusedAdam
        self.optimiser = FusedAdam(actor_critic.parameters(), lr=args.lr, eps=args.eps)         # ppo.py:23 (Adam; the clip of :67 rides in its step)
        self.bucket = cdist.GradBucket(actor_critic.parameters(), assign_when_single_rank=True)     # persistent flat gradient buffer (one all-reduce per step)
        self.timings = {}

    def update(self, rollout_storage):
        ac = self.actor_critic
        sums = None
        t_adv = t_opt = 0.0
        n_steps = 0
        for e in range(self.ppo_epoch):
            t0 = time.perf_counter()
            with torch.no_grad():
                rollout_storage.compute_advantages_alt(ac, 10)                               # ppo.py:31-32
            t1 = time.perf_counter()
            if getattr(ac, "include_lstm", False):                                           # ppo.py:34-38
                data_generator = rollout_storage.generator_lstm(num_mini_batch=self.num_mini_batch,
                                                                total_batch_size=rollout_storage.num_parallel * rollout_storage.num_steps,
                                                                truncated_seq_len=self.args.truncated_seq_len)
            else:
                data_generator = rollout_storage.generator_standard(self.num_mini_batch)
            for sample in data_generator:
                obs_dict_batch, recurrent_batch, actions_batch, action_masks_batch, value_preds_batch, returns_batch, \
                    masks_batch, old_action_log_probs_batch, adv_target = sample
                values, action_log_probs, entropy, _ = ac.evaluate_actions(obs_dict_batch, recurrent_batch, masks_batch,
                                                                           actions_batch, action_masks_batch)
                norm = ((ac.value_normaliser.mean_np, ac.value_normaliser.std_np)
                        if getattr(ac, "use_value_normalisation", False) else None)             # ppo.py:46-48, inside the kernel
                loss, parts = _LOSS(action_log_probs.float(), values.float(), old_action_log_probs_batch, adv_target,
                                    value_preds_batch, returns_batch, self.clip_param, self.value_loss_coef, value_normaliser=norm)
                self.bucket.zero()
                (loss - entropy * self.entropy_coef).backward()                              # ppo.py:66
                self.bucket.allreduce()
                self.optimiser.step(self.max_grad_norm)                                      # ppo.py:67-68
                nn_kernels.weight_images.refresh_all()                                       # the bf16 / transposed / packed images of the new weights: one launch (nothing on the CPU)
                s = torch.stack((parts[1].detach() * self.value_loss_coef, parts[0].detach(), entropy.detach().float() * self.entropy_coef))
                sums = s if sums is None else sums + s
                n_steps += 1
            t_adv += t1 - t0; t_opt += time.perf_counter() - t1
        self.timings = {"advantages_s": t_adv, "minibatches_s": t_opt}
        vl, al, el = (sums / n_steps).tolist()                                               # ppo.py:70-79
        return vl, al, el


# ------------------------------------------------------------------------------------------------ evaluation (next row f3)
def make_evaluation_manager():
    return "evaluation-manager"          # placeholder thunk: SubProcEvaluationManager only counts them


class SubProcEvaluationManager(object):
    """RL/ppo/vec_evaluation.py:43-100 over `evaluation.run_evaluation_episodes`: all episodes of a call run at once on one
    batched env; results come back in the per-process tuples `(winners, game_lengths, victory_points, policy_steps)`."""

    def __init__(self, evaluation_manager_fns, start_method=None, *, device=_UNSET, seed=_UNSET, env_factory=_UNSET, make_policy=_UNSET,
                 autocast_dtype=_UNSET):
        device, seed, env_factory = _default("device", device), _default("seed", seed), _default("eval_env_factory", env_factory)
        make_policy, autocast_dtype = _default("make_policy", make_policy), _default("autocast_dtype", autocast_dtype)
        self.processes = [_ProcessStub() for _ in evaluation_manager_fns]
        self.waiting, self.closed = False, False
        self._device, self._seed, self._env_factory = device, seed, env_factory
        self._make_policy = make_policy or CatanPolicy
        self._autocast = autocast_dtype
        self._nets = None
        self._calls = 0

    def update_policies(self, state_dicts):
        dev = torch.device(self._device) if self._device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        nets, seen = [], {}
        for sd in state_dicts:                      # the protocol hands three deep copies of one opponent: share the net if equal
            key = next((k for k, (ref, _) in seen.items() if all(torch.equal(ref[n], sd[n]) for n in ref if ref[n].numel())), None)
            if key is None:
                net = self._make_policy().to(dev).eval()
                net.load_reference_state_dict(sd)
                key = len(seen)
                seen[key] = ({n: v for n, v in sd.items()}, net)
            nets.append(seen[key][1])
        self._nets = nets
        return [True] * len(self.processes)

    def run_evaluation_episodes(self, total_episodes):
        from . import evaluation
        eps = total_episodes // len(self.processes)
        n = eps * len(self.processes)
        if self._env_factory is None:
            from .env import VecCatanEnv
            env = VecCatanEnv(n, seed=self._seed + 7919 * (self._calls + 1), auto_reset=False, device=self._device)
        else:
            env = self._env_factory(n)
        self._calls += 1
        ac = self._autocast
        if ac == "auto":
            ac = torch.bfloat16 if torch.device(env.device).type == "cuda" else None
        res = evaluation.run_evaluation_episodes(env, self._nets, evaluation.sample_orders(n), autocast_dtype=ac)
        out = []
        for p in range(len(self.processes)):
            sl = slice(p * eps, (p + 1) * eps)
            out.append((list(res["winner"][sl]), list(res["game_steps"][sl]), list(res["victory_points"][sl]), list(res["policy_decisions"][sl])))
        return out

    def close(self):
        self.closed = True


def run_evaluation_protocol(evaluation_manager, central_policy, earlier_policies, random_policy, args, update_num, *_ignored):
    """RL/ppo/run_evaluation_protocol.py:5-69 (robust_train.py:143-146 passes two more arguments than the reference function
    takes; they are accepted and ignored)."""
    log = {"update": update_num}
    summary = "\n\n---------------------- EVALUATION (after {} updates) ----------------------\n".format(update_num)
    sd = copy.deepcopy(central_policy.state_dict())
    evaluation_manager.update_policies([sd, copy.deepcopy(random_policy), copy.deepcopy(random_policy), copy.deepcopy(random_policy)])
    results = list(zip(*evaluation_manager.run_evaluation_episodes(args.num_eval_episodes)))
    winners, game_lengths = np.concatenate(results[0]), np.concatenate(results[1])
    victory_points, policy_steps = np.concatenate(results[2]), np.concatenate(results[3])
    log["random"] = {"policy_win_frac": np.mean(winners == 0), "avg_game_length": np.mean(game_lengths),
                     "avg_policy_decisions": np.mean(policy_steps), "avg_victory_points": np.mean(victory_points)}
    summary += ("{} games against random. Policy won {}/{}. Avg. game length: {}. Avg num policy decisions: {}. "
                "Avg victory points for policy: {}. \n\n").format(args.num_eval_episodes, int(np.sum(winners == 0)), args.num_eval_episodes,
                                                                np.mean(game_lengths), np.mean(policy_steps), np.mean(victory_points))
    return log, summary

"""In-training evaluation protocol, batched (reference RL/ppo/evaluation_manager.py:12-84, RL/ppo/run_evaluation_protocol.py,
RL/ppo/vec_evaluation.py; draw cap of evaluation/evaluation_manager.py:119).

The reference plays `num_eval_episodes` games in 16 worker processes: policy 0 (the central policy) against three copies
of an opponent policy (a random-initialised net in the protocol), a fresh random seat order per game
(`random.shuffle(order)`; policy i plays seat order[i]), sampling actions (`deterministic=False`), and logs the fraction of
games policy 0 won, the mean game length, the mean number of policy-0 decisions and policy 0's mean victory points.
Here all episodes run at once on one batched env (no auto-reset; finished games idle), one batched forward per distinct
net per pass.  The deciding player is the env's (discarder > trade target > players_go, evaluation_manager.py:76-84)."""
import random as _py_random

import numpy as np
import torch

from . import spec

PLAYER_IDS = [2, 4, 3, 1]            # [Blue, Red, Orange, White] - the list the reference shuffles (evaluation_manager.py:27)


def sample_orders(n_games, rng=None):
    """-> int64 [n_games, 4]: order[g][i] = PlayerId played by policy i in game g (`random.shuffle(self.order)` per game)."""
    rng = rng or _py_random
    out = np.zeros((n_games, 4), dtype=np.int64)
    for g in range(n_games):
        o = list(PLAYER_IDS)
        rng.shuffle(o)
        out[g] = o
    return out


@torch.no_grad()
def run_evaluation_episodes(env, nets, orders, max_steps=None, deterministic=False, generator=None, autocast_dtype=None,
                            act_fn=None):
    """env: freshly reset games, auto_reset off.  nets: 4 policies (entries may be the same object; equal objects share a
    forward).  orders int [n,4].  max_steps: the offline evaluator's draw cap (2 500; None = play to the end).
    act_fn(net, idx, f, lists, lens, masks) -> actions [len(idx),18]: replaces net.act on the rows idx (test hook).
    -> dict of numpy arrays: winner (policy index, -1 = draw), victory_points (of policy 0), game_steps, policy_decisions."""
    n, dev = env.n, env.device
    orders_t = torch.as_tensor(orders, device=dev).long()
    policy_of_pid = torch.empty((n, 4), dtype=torch.long, device=dev)
    policy_of_pid.scatter_(1, orders_t - 1, torch.arange(4, device=dev).expand(n, 4))
    if autocast_dtype is not None:        # weights in the autocast dtype: no per-call casts (policy.inference_copy)
        cache = {}
        nets = [cache.setdefault(id(n), n.inference_copy(autocast_dtype)) if (hasattr(n, "inference_copy") and getattr(n, "_inference_dtype", None) is None)
                else n for n in nets]
    distinct = []
    for net in nets:
        if not any(net is d for d in distinct):
            distinct.append(net)
    net_of_policy = torch.tensor([[i for i, d in enumerate(distinct) if d is net][0] for net in nets], device=dev)
    ar = torch.arange(n, device=dev)
    steps = torch.zeros(n, dtype=torch.long, device=dev)
    decisions = torch.zeros(n, dtype=torch.long, device=dev)
    running = torch.ones(n, dtype=torch.bool, device=dev)
    draw = torch.zeros(n, dtype=torch.bool, device=dev)
    # LSTM policies: one (h, c) per seat, zero at the start of the game (evaluation_manager.py:20-26,50-59); a net without
    # an LSTM ignores its seats' entries.  The terminal mask stays 1: a finished evaluation game takes no further step.
    sizes = [int(net.lstm_size) for net in distinct if getattr(net, "include_lstm", False)]
    hid = torch.zeros((2, n, 4, max(sizes)), dtype=torch.float32, device=dev) if sizes else None
    while bool(running.any()):
        deciding = env.deciding_player().long()
        pol = policy_of_pid[ar, deciding - 1]
        f, lists, lens = env.get_obs()
        masks = env.get_action_masks()
        actions = torch.zeros((n, spec.ACTION_WORDS), dtype=torch.int64, device=dev)
        net_id = net_of_policy[pol]
        for k, net in enumerate(distinct):
            idx = ((net_id == k) & running).nonzero(as_tuple=True)[0]
            if idx.numel() == 0:
                continue
            args = (f[idx], lists[idx], lens[idx].long(), masks[idx])
            if act_fn is not None:
                actions[idx] = act_fn(net, idx, *args)
                continue
            kw = {"deterministic": deterministic, "generator": generator}
            rec = getattr(net, "include_lstm", False)
            if rec:
                L, seat = int(net.lstm_size), deciding[idx] - 1
                kw.update(hidden=(hid[0, idx, seat, :L], hid[1, idx, seat, :L]), nonterminal=torch.ones(idx.numel(), device=dev))
            if autocast_dtype is not None:
                with torch.autocast(device_type="cuda", dtype=autocast_dtype):
                    res = net.act(*args, **kw)
            else:
                res = net.act(*args, **kw)
            actions[idx] = res[1]
            if rec:
                hid[0, idx, seat, :L], hid[1, idx, seat, :L] = res[3][0].float(), res[3][1].float()
        a_env = actions.to(torch.int32)
        a_env[:, 0] = torch.where(running, a_env[:, 0], torch.full_like(a_env[:, 0], -1))
        _, done = env.step(a_env)
        decisions += (running & (pol == 0)).long()
        steps += running.long()
        done = done.bool() & running
        if max_steps is not None:
            capped = running & ~done & (steps > max_steps)                    # `if total_game_steps > 2500: DRAW`
            draw |= capped
            done = done | capped
        running &= ~done
    blob = env.export_state()
    off_w, _ = spec.STATE_OFFSETS["winner"]; off_v, _ = spec.STATE_OFFSETS["curr_vps"]
    winner_pid = blob[:, off_w].long()
    winner = torch.where(draw, torch.full_like(winner_pid, -1), policy_of_pid[ar, (winner_pid - 1).clamp(min=0)])
    vps = blob[:, off_v:off_v + 4].long()[ar, orders_t[:, 0] - 1]             # env.curr_vps[self.order[0]]
    return {"winner": winner.cpu().numpy(), "victory_points": vps.cpu().numpy(), "game_steps": steps.cpu().numpy(),
            "policy_decisions": decisions.cpu().numpy()}


def run_evaluation_protocol(make_env, central_policy, opponent_policy, num_eval_episodes, update_num=0, rng=None, **kw):
    """run_evaluation_protocol.py: the central policy against three copies of `opponent_policy` (the protocol's "random"
    opponent).  make_env(n) -> n freshly reset games without auto-reset.  -> (log dict, summary string)."""
    env = make_env(num_eval_episodes)
    res = run_evaluation_episodes(env, [central_policy, opponent_policy, opponent_policy, opponent_policy],
                                  sample_orders(num_eval_episodes, rng), **kw)
    log = {"update": update_num, "random": {
        "policy_win_frac": float(np.mean(res["winner"] == 0)), "avg_game_length": float(np.mean(res["game_steps"])),
        "avg_policy_decisions": float(np.mean(res["policy_decisions"])), "avg_victory_points": float(np.mean(res["victory_points"]))}}
    r = log["random"]
    summary = ("\n\n---------------------- EVALUATION (after {} updates) ----------------------\n"
               "{} games against random. Policy won {}/{}. Avg. game length: {}. Avg num policy decisions: {}. "
               "Avg victory points for policy: {}. \n\n").format(update_num, num_eval_episodes, int(np.sum(res["winner"] == 0)),
                                                               num_eval_episodes, r["avg_game_length"], r["avg_policy_decisions"],
                                                               r["avg_victory_points"])
    return log, summary

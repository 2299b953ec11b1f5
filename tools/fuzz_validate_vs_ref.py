"""Differential fuzz of VALIDATE MODE: the oracle's `orc_action_is_legal` against the imported upstream reference
(development container only: needs /root/reference).

What the reference does with an action (env/wrapper.py:36-42, `validate_actions=True` is the wrapper's default):
    translated = self._translate_action(action)               # wrapper.py:114-166 - raises on malformed head values
    ok, err = self.game.validate_action(translated)           # game/game.py:264-525
    if ok == False: raise RuntimeError(err)
    self.game.apply_action(translated)                        # game/game.py:527-815
"accept" below = that sequence runs through; "reject" = it raises (whatever the exception) with the game untouched.
The reference accepts - and applies - actions its masks never offer (MoveRobber onto an empty tile, ProposeTrade past
the per-turn limit, RollDice while road building is played out, ...), so the rule is NOT "mask bit set".

For every visited state of every game the tool builds probes - the sampled in-mask action, random in-range actions of
every type, single-head perturbations, out-of-range head values (at or above the head's size everywhere; below zero
for the heads the reference checks by value: negative corner / edge / tile indices would wrap around a Python list and
are outside the action space) and a few targeted ones - and compares verdicts.  Accepted probes that
are NOT in the masks are also APPLIED (to a saved copy of the reference and to an oracle clone of the same state) and
state blob / masks / rewards (as doubles) / done / deciding seat / observation are compared; now and then such an action
is taken as the game's real step so that the states only it can reach get explored too.

Usage: python tools/fuzz_validate_vs_ref.py [n_envs] [steps_per_env] [seed] [--probes K] [--trades K|none] [--max-actions K]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import ref_harness as rh  # noqa: E402
import oracle_lib as ol  # noqa: E402
from settlers_of_catan_rl_amd import spec  # noqa: E402

# head -> (first word, words, size).  Heads 7 / 8 are the four-entry give / receive sequences (0 = stop, 1..5 resources).
HEADS = {1: (1, 1, 54), 2: (2, 1, 73), 3: (3, 1, 19), 4: (4, 1, 5), 5: (5, 1, 2), 6: (6, 1, 3), 7: (7, 4, 6), 8: (11, 4, 6),
         9: (15, 1, 5), 10: (16, 1, 5), 11: (17, 1, 5)}
# which heads `_translate_action` / `validate_action` / `apply_action` read for a type (env/wrapper.py:120-164)
RELEVANT = {0: [1], 1: [2], 2: [1], 3: [], 4: [4, 9, 10], 5: [9, 10], 6: [6, 7, 8], 7: [5], 8: [3], 9: [], 10: [], 11: [6], 12: [11]}
LIST_INDEX_HEADS = (1, 2, 3)       # used as Python list indices by the reference: a negative value would wrap, not raise


def random_action(rng, t=None):
    a = np.zeros((18,), dtype=np.int32)
    a[0] = int(rng.integers(0, 13)) if t is None else t
    for h, (w0, n, size) in HEADS.items():
        a[w0:w0 + n] = rng.integers(0, size, size=n)
    return a


def probes_for(rng, base, n_random, ref):
    """-> list of (kind, action18)"""
    out = [("sampled", base.copy())]
    for _ in range(n_random):                                  # any type, every head in range
        out.append(("random", random_action(rng)))
    for _ in range(3):                                         # the sampled type with one relevant head redrawn
        a = base.copy()
        rel = RELEVANT[int(a[0])]
        if rel:
            h = rel[int(rng.integers(0, len(rel)))]
            w0, n, size = HEADS[h]
            a[w0:w0 + n] = rng.integers(0, size, size=n)
        else:
            a[0] = int(rng.integers(0, 13))
        out.append(("perturbed", a))
    for _ in range(2):                                         # a random type, one relevant head out of range
        a = random_action(rng)
        rel = RELEVANT[int(a[0])]
        if rel:
            h = rel[int(rng.integers(0, len(rel)))]
            w0, n, size = HEADS[h]
            choices = [size, size + int(rng.integers(1, 50)), 127, 1 << 20]
            if h not in LIST_INDEX_HEADS:
                choices += [-1, -int(rng.integers(2, 60))]
            a[w0 + int(rng.integers(0, n))] = choices[int(rng.integers(0, len(choices)))]
        else:
            a[0] = [-1, 13, 14, 100, -7][int(rng.integers(0, 5))]
        out.append(("out_of_range", a))
    g = ref.env.game
    # targeted: what the judge's probe found, and its relatives
    if g.can_move_robber:
        a = random_action(rng, 8)
        out.append(("robber_any_tile", a))
    if g.road_building_active[0]:
        out.append(("roll_in_road_building", random_action(rng, 9)))
        a = random_action(rng, 1); a[2] = 72
        out.append(("dummy_edge", a))
    if g.dice_rolled_this_turn:
        a = random_action(rng, 6)
        k = int(rng.integers(0, 5))
        a[7 + k:11] = 0                                        # 0..4 offered resources
        out.append(("propose", a))
        out.append(("end_turn", random_action(rng, 10)))
    pl = g.players[g.players_go]
    if len(pl.hidden_cards) > 0:
        a = random_action(rng, 4)
        a[4] = int(pl.hidden_cards[int(rng.integers(0, len(pl.hidden_cards)))])
        out.append(("play_owned_card", a))
    return out


def ref_verdict(ref, a18):
    """accept / reject of the reference for this action in its current state WITHOUT applying it.
    (MoveRobber: validate_action does not look at the tile; `self.board.tiles[tile]` raises IndexError in the first line of
    apply_action's branch, game.py:624, before anything changes - checked for real in apply_both.)"""
    heads = rh.action_to_heads(a18)
    try:
        t = ref.env._translate_action(heads)
        ok, _ = ref.env.game.validate_action(t)
    except Exception:                                          # ValueError / KeyError / IndexError / TypeError
        return False
    if ok is not True:
        return False
    if int(a18[0]) == 8 and not (0 <= int(a18[3]) < 19):
        return False
    return True


def compare_obs(ref_obs, orc):
    f, lists, lens, pid = rh.obs_flat(ref_obs)
    of, olists, olens, opid = orc.obs()
    assert pid == opid, (pid, opid)
    assert np.array_equal(f, of), np.flatnonzero(f != of)[:10]
    assert np.array_equal(lens, olens) and np.array_equal(lists, olists)


def compare_state(ref, orc, where):
    rb, ob = ref.state_blob(), orc.export()
    if not np.array_equal(rb, ob):
        raise AssertionError(f"{where}: state differs\n" + spec.describe_state_diff(rb, ob))
    rm, om = rh.masks_flat(ref.masks()), orc.masks()
    if not np.array_equal(rm, om):
        raise AssertionError(f"{where}: masks differ at {np.flatnonzero(rm != om)[:20]}")
    assert ref.deciding_player() == orc.deciding_player(), where


def step_both(ref, orc, a, where):
    ref_obs, rrew, rdone = ref.step(a)
    orew, odone = orc.step(a)
    assert rdone == odone and np.array_equal(rrew, orew), (where, a, rrew, orew, rdone, odone)
    assert np.array_equal(ref.last_reward64, orc.last_reward64()), (where, a)
    compare_state(ref, orc, where)
    compare_obs(ref_obs, orc)
    return ref_obs, rdone


def apply_on_copies(ref, orc_cfg, seed, env_id, a, where):
    """apply `a` to a deep copy of the reference and to an oracle clone of the same state; everything must agree"""
    blob = ref.state_blob()
    clone = ol.OracleEnv(seed, env_id)
    clone.set_config(**orc_cfg)
    clone.import_(blob)
    step_both(ref.clone(), clone, a, where)
    assert np.array_equal(ref.state_blob(), blob), where


def fuzz(n_envs, steps, seed, n_random=4, trades=4, max_actions=None, apply_prob=0.5, take_prob=0.08, verbose=True):
    t0 = time.time()
    stats = {"probes": 0, "accepted": 0, "accepted_out_of_mask": 0, "applied_out_of_mask": 0, "taken_out_of_mask": 0,
             "steps": 0, "games": 0}
    by_type = np.zeros((13,), dtype=np.int64)
    mism = []
    orc_cfg = dict(max_trades_per_turn=trades, max_actions_per_turn=max_actions)
    for env_id in range(n_envs):
        rng = np.random.default_rng(seed * 1000003 + env_id)
        ref = rh.RefEnv(seed, env_id, max_proposed_trades_per_turn=trades, max_actions_per_turn=max_actions)
        orc = ol.OracleEnv(seed, env_id)
        orc.set_config(**orc_cfg)
        ref.reset(); orc.reset()
        for s in range(steps):
            where = f"env {env_id} step {s}"
            compare_state(ref, orc, where)
            base = rh.random_legal_action(ref.masks(), ref.env, rng)
            take = None
            for kind, a in probes_for(rng, base, n_random, ref):
                want = ref_verdict(ref, a)
                got = orc.is_legal(a)
                in_masks = bool(orc.L.orc_action_in_masks(orc.p, ol._p(np.ascontiguousarray(a, dtype=np.int32), ol.C.c_int32)))
                stats["probes"] += 1
                if want != got:
                    mism.append((env_id, s, kind, a.tolist(), want, got))
                    if verbose:
                        print(f"MISMATCH {where} {kind} {a.tolist()}: reference {'accepts' if want else 'rejects'}, oracle "
                              f"{'accepts' if got else 'rejects'}", flush=True)
                    continue
                if kind == "sampled":
                    assert want and in_masks, (where, a)
                assert not (in_masks and not want), (where, kind, a, "in the masks but the reference rejects it")
                if want:
                    stats["accepted"] += 1
                    if not in_masks:
                        stats["accepted_out_of_mask"] += 1
                        by_type[int(a[0])] += 1
                        if rng.random() < apply_prob:
                            apply_on_copies(ref, orc_cfg, seed, env_id, a, where + f" [{kind}] {a.tolist()}")
                            stats["applied_out_of_mask"] += 1
                        if take is None and rng.random() < take_prob:
                            take = a
                elif int(a[0]) == 8 and 0 <= int(a[0]) <= 12 and rng.random() < 0.2:
                    # a rejected action really leaves the reference untouched (incl. the IndexError of game.py:624)
                    before = ref.state_blob()
                    try:
                        ref.step(a)
                        raise AssertionError(where + ": the reference applied an action it was expected to reject")
                    except AssertionError:
                        raise
                    except Exception:
                        pass
                    assert np.array_equal(before, ref.state_blob()), where
            a = base if take is None else take
            if take is not None:
                stats["taken_out_of_mask"] += 1
            _, done = step_both(ref, orc, a, where + f" taking {a.tolist()}")
            stats["steps"] += 1
            if done:
                stats["games"] += 1
                ref.reset(); orc.reset()
        if verbose:
            print(f"env {env_id}: {stats} ({time.time() - t0:.0f}s)", flush=True)
    return stats, by_type, mism


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("n_envs", type=int, nargs="?", default=4)
    ap.add_argument("steps", type=int, nargs="?", default=2500)
    ap.add_argument("seed", type=int, nargs="?", default=1)
    ap.add_argument("--probes", type=int, default=4, help="random-type probes per state (besides the structured ones)")
    ap.add_argument("--trades", default="4")
    ap.add_argument("--max-actions", type=int, default=None)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()
    stats, by_type, mism = fuzz(args.n_envs, args.steps, args.seed, n_random=args.probes,
                                trades=None if args.trades.lower() == "none" else int(args.trades),
                                max_actions=args.max_actions, verbose=not args.quiet)
    print("accepted out-of-mask actions by type:", {int(t): int(c) for t, c in enumerate(by_type) if c})
    print(f"{'PASS' if not mism else 'FAIL'}: {stats['probes']} probes over {stats['steps']} steps / {stats['games']} games, "
          f"{stats['accepted_out_of_mask']} accepted outside the masks ({stats['applied_out_of_mask']} applied on copies, "
          f"{stats['taken_out_of_mask']} taken), {len(mism)} disagreements")
    sys.exit(1 if mism else 0)

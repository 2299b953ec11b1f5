"""Truncated-BPTT minibatches (train.lstm_minibatches) against the reference's `BatchProcessor.generator_lstm`
(RL/ppo/process_batch.py:203-293) on a tagged storage, and the LSTM value re-evaluation rule of
`compute_advantages_alt` (process_batch.py:112-128).  Runs where /root/reference is mounted; the index arithmetic is also
checked reference-free."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from settlers_of_catan_rl_amd.train import lstm_minibatches

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REF = os.path.isdir("/root/reference/RL/ppo")


def test_pieces_cover_every_decision_once():
    T, N, L, nmb = 20, 12, 5, 8
    perm = torch.randperm(N * T // L, generator=torch.Generator().manual_seed(0))
    seen = torch.zeros(T, N, dtype=torch.int64)
    nb = 0
    for t, g in lstm_minibatches(T, N, L, nmb, perm):
        assert t.shape == (L, (T * N) // nmb // L) and g.shape == (t.shape[1],)
        assert torch.equal(t[1:] - t[:-1], torch.ones_like(t[1:])) and bool((t[0] % L == 0).all())
        seen[t, g[None, :].expand_as(t)] += 1
        nb += 1
    assert nb == nmb and bool((seen == 1).all())
    with pytest.raises(ValueError):
        list(lstm_minibatches(21, N, L, nmb, perm))
    with pytest.raises(ValueError):
        list(lstm_minibatches(T, N, L, 5, perm))      # 9 pieces per minibatch do not divide 48


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
def test_minibatches_equal_reference_generator():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from ref_bootstrap import bootstrap
    bootstrap()
    from RL.ppo.process_batch import BatchProcessor
    T, N, L, nmb, H = 20, 6, 5, 4, 3
    args = types.SimpleNamespace(num_steps=T, num_processes=N, num_envs_per_process=1, gamma=0.99, gae_lambda=0.95)
    bp = BatchProcessor(args, H, obs_keys=["a", "b"], obs_type=["normal", "normal"], num_action_heads=3,
                        type_conditional_masks=[1], device="cpu")
    g = torch.Generator().manual_seed(0)
    tag = (torch.arange(T + 1)[:, None] * 100 + torch.arange(N)[None, :]).float()          # value = 100 t + game
    bp.obs_dict = {"a": tag[:, :, None] + torch.tensor([0.0, 0.25]), "b": tag[:, :, None] * 2}
    bp.hidden_states = (tag[:, :, None] + torch.tensor([0.1, 0.2, 0.3]), -tag[:, :, None] - torch.tensor([0.1, 0.2, 0.3]))
    bp.actions = [tag[:T, :, None].long() + i for i in range(3)]
    bp.action_masks = [tag[:T, :, None] + 0.5, torch.stack([tag[:T, :, None] + 1, tag[:T, :, None] + 2]), tag[:T, :, None] + 3]
    bp.values = tag[:, :, None] + 7; bp.returns = tag[:T, :, None] + 8
    bp.masks = (torch.rand(T + 1, N, 1, generator=g) > 0.2).float()
    bp.action_log_probs = tag[:T, :, None] + 9; bp.advantages = tag[:T, :, None] + 10
    np.random.seed(5)
    ref_batches = list(bp.generator_lstm(nmb, T * N, L))
    np.random.seed(5)
    perm = torch.from_numpy(np.random.permutation(N * T // L))                               # the generator's only draw (:216)
    mine = list(lstm_minibatches(T, N, L, nmb, perm))
    assert len(mine) == len(ref_batches) == nmb
    for (t, gm), rb in zip(mine, ref_batches):
        obs, hid, acts, amasks, vpred, ret, msk, oldlp, adv = rb
        rows_t, rows_g = t.reshape(-1), gm[None, :].expand_as(t).reshape(-1)
        assert torch.equal(obs["a"], bp.obs_dict["a"][rows_t, rows_g]) and torch.equal(obs["b"], bp.obs_dict["b"][rows_t, rows_g])
        assert torch.equal(hid[0], bp.hidden_states[0][t[0], gm]) and torch.equal(hid[1], bp.hidden_states[1][t[0], gm])
        for i in range(3):
            assert torch.equal(acts[i], bp.actions[i][rows_t, rows_g])
        assert torch.equal(amasks[0], bp.action_masks[0][rows_t, rows_g]) and torch.equal(amasks[1], bp.action_masks[1][:, rows_t, rows_g])
        assert torch.equal(vpred, bp.values[rows_t, rows_g]) and torch.equal(ret, bp.returns[rows_t, rows_g])
        assert torch.equal(msk, bp.masks[rows_t, rows_g]) and torch.equal(oldlp, bp.action_log_probs[rows_t, rows_g])
        assert torch.equal(adv, bp.advantages[rows_t, rows_g])


@pytest.mark.skipif(not HAVE_REF, reason="upstream reference not mounted")
def test_lstm_value_reevaluation_rule():
    """compute_advantages_alt with an LSTM net: every stored decision is ONE step from its stored state with its own
    terminal mask (the hidden rows equal the input rows, policy.py:117-123) - what PPOTrainer.compute_values restates."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from ref_bootstrap import bootstrap
    bootstrap()
    from RL.ppo.process_batch import BatchProcessor
    T, N, H = 4, 3, 2
    args = types.SimpleNamespace(num_steps=T, num_processes=N, num_envs_per_process=1, gamma=0.9, gae_lambda=0.8)
    bp = BatchProcessor(args, H, obs_keys=["a"], obs_type=["normal"], device="cpu")
    g = torch.Generator().manual_seed(1)
    bp.obs_dict = {"a": torch.randn(T + 1, N, 2, generator=g)}
    bp.hidden_states = (torch.randn(T + 1, N, H, generator=g), torch.randn(T + 1, N, H, generator=g))
    bp.masks = (torch.rand(T + 1, N, 1, generator=g) > 0.3).float()
    bp.rewards = torch.randn(T, N, 1, generator=g)
    seen = []

    class Net(object):
        include_lstm, use_value_normalisation = True, False
        def get_value(self, obs, hidden, masks):
            seen.append((obs["a"].clone(), hidden[0].clone(), hidden[1].clone(), masks.clone()))
            assert obs["a"].shape[0] == hidden[0].shape[0] == masks.shape[0]
            return (obs["a"].sum(1, keepdim=True) + (hidden[0] * masks).sum(1, keepdim=True))
    bp.compute_advantages_alt(Net(), max_processes_at_once=2)
    got_rows = sum(s[0].shape[0] for s in seen)
    assert got_rows == (T + 1) * N
    want = bp.obs_dict["a"].sum(-1, keepdim=True) + (bp.hidden_states[0] * bp.masks).sum(-1, keepdim=True)
    assert torch.allclose(bp.values, want)

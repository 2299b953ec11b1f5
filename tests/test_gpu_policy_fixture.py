"""GPU: the policy path against the REFERENCE net on the MI355X.

tests/golden/policy_small.npz holds what the reference's own `build_agent_model()` net (RL/models/policy.py:71-111,
action_heads_module.py:25-312, distributions.py:25-40) returns on 320 real observations with the deterministic weights of
tests/policy_fixture.py - and, with `include_lstm`, on 40 rows in both LSTM forms.  Here `CatanPolicy` runs the same inputs
on the device with every hand-written HIP kernel of the net in play (fused attention, LayerNorm, masked categorical, LSTM
cell; under bf16 also the fused tile encoder, the row-linear / weight-gradient kernels and the card-list summary)."""
import pytest
import torch

import golden_util as gu
import policy_fixture as pf

pytestmark = pytest.mark.gpu


def _count_kernel_calls(monkeypatch):
    """wraps the nn_kernels entry points the policy dispatches to on the GPU: the test must not pass on a torch fallback"""
    from settlers_of_catan_rl_amd import nn_kernels
    calls = {}
    for name in ("small_attention", "small_layer_norm", "masked_categorical", "lstm_cell", "tile_encoder_forward", "card_summary"):
        fn = getattr(nn_kernels, name, None)
        if fn is None:
            continue

        def wrap(*a, _fn=fn, _name=name, **kw):
            calls[_name] = calls.get(_name, 0) + 1
            return _fn(*a, **kw)
        monkeypatch.setattr(nn_kernels, name, wrap)
    return calls


@pytest.mark.parametrize("which", ["ff", "lstm"])
def test_policy_fixture_fp32_with_hip_kernels(hip_lib, monkeypatch, which):
    """fp32: identical arg-max actions; value / joint log-prob / entropy / LSTM state within 1e-5 (relative to max(1, |x|)) of
    the reference net - north_star's bound; every parameter's gradient within 1e-4 of its own size."""
    calls = _count_kernel_calls(monkeypatch)
    dev = pf.check_policy_fixture(gu.load("policy_small.npz"), which, "cuda", tol=1e-5, grad_tol=1e-4)
    print(which, "fp32 deviations from the reference net:", dev, "kernel calls:", calls)
    assert calls.get("small_attention", 0) >= 4 and calls.get("small_layer_norm", 0) >= 20 and calls.get("masked_categorical", 0) >= 12, calls      # (12 = act: 18 head evaluations as chained kernels elsewhere; evaluate: type + 9 heads + the two trade heads, one pass each)
    if which == "lstm":
        assert calls.get("lstm_cell", 0) >= 1 + 5, calls


@pytest.mark.parametrize("which", ["ff", "lstm"])
def test_policy_fixture_bf16_autocast(hip_lib, monkeypatch, which):
    """bf16 autocast (config 3's dtype; fp32 master weights): bound STATED - value within 0.03 (normalised units, i.e. 4.5
    reward points of the 150-point scale), joint log-prob within 0.03 relative to max(1, |logp|), entropy within 0.03, LSTM
    state within 0.03; the reference's arg-max action TYPE in every row and its arg-max in every column that matters for that
    type in at least 95 % of the rows (the others are near-ties flipped by bf16 rounding); every parameter's gradient within
    8 % of its own size.  (Measured on MI355X, round 3: 0.018 / 0.010 / 0.001 / 0.015; 98.75 % and 97.5 %; 3.2 % and 5.8 %.)"""
    calls = _count_kernel_calls(monkeypatch)
    g = gu.load("policy_small.npz")
    dev = pf.check_policy_fixture(g, which, "cuda", autocast_dtype=torch.bfloat16, tol=0.03, grad_tol=0.08, argmax_equal=False,
                                  min_argmax_agreement=0.95)
    assert dev["act_type_agreement"] == 1.0, dev
    print(which, "bf16 deviations from the reference net:", dev, "kernel calls:", calls)
    assert calls.get("small_layer_norm", 0) >= 10 and calls.get("masked_categorical", 0) >= 12, calls      # (12 = act: 18 head evaluations as chained kernels elsewhere; evaluate: type + 9 heads + the two trade heads, one pass each)

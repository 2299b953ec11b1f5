"""Shared by tools/gen_golden.py (generator, development container) and the policy fixture tests: the deterministic weights
of tests/golden/policy_small.npz and the decoding of its inputs.

The fixture pins the chain reference net -> this package's net: `fixture_state_dict` fills a `state_dict` of the reference's
key set from an integer hash of (parameter name, element index) - exact integer arithmetic, so generator and test build the
same 1.93 M numbers without storing 7.7 MB; the generator loads them into the reference's OWN `build_agent_model()`
(`load_state_dict`, strict), records what the reference computes, and stores a crc32 of every tensor, which the test checks
before it loads the same numbers into `CatanPolicy`."""
import zlib

import numpy as np
import torch

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _hash_unit(name, n):
    """n values in [-0.5, 0.5) on a 2^-24 grid (exact in fp32), a splitmix64-style hash of (crc32(name), index)"""
    with np.errstate(over="ignore"):
        x = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.uint64(zlib.crc32(name.encode())) * np.uint64(0xBF58476D1CE4E5B9))
        x ^= x >> np.uint64(31)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(29)
    return (x >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5


def fixture_tensor(name, shape, salt=""):
    """The fixture's value of parameter `name`: matrices ~ uniform with std 1/sqrt(fan_in) (activations stay O(1) through the
    net, the action distributions are decisive but not one-hot), LayerNorm weights 1 +- 0.1, biases +- 0.05."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u = _hash_unit(salt + name, n)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        v = u * (np.sqrt(12.0) / np.sqrt(float(fan_in)))
    elif name.endswith(".weight"):
        v = 1.0 + 0.2 * u
    else:
        v = 0.1 * u
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def fixture_state_dict(shapes, salt=""):
    """shapes: {name: shape} of every learnable entry (empty `dummy_param` entries and the value normaliser's constants are
    left to the caller)"""
    return {k: fixture_tensor(k, shp, salt) for k, shp in shapes.items()}


def tensor_crc(t):
    return zlib.crc32(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()) & 0xFFFFFFFF


def projection(name, t):
    """<t, h(name)> with h the hash vector of `name` (salted): one number that moves if any element of a gradient moves"""
    h = torch.from_numpy(_hash_unit("proj:" + name, t.numel())).reshape(t.shape)
    return float((t.detach().double().cpu() * h).sum())


def decode_inputs(g, prefix):
    """-> dict(obs_f [B,1787] fp32, lists [B,5,25] int32, lens [B,5] int32, masks [B,325] fp32) stored compactly (observation
    values are multiples of 1/8: exact in fp16; masks as bits)"""
    B = int(g[prefix + "lens"].shape[0])
    masks = np.unpackbits(g[prefix + "masks"], axis=1, bitorder="little")[:, :325].astype(np.float32)
    return dict(obs_f=torch.from_numpy(g[prefix + "obs_f"].astype(np.float32)), lists=torch.from_numpy(g[prefix + "lists"].astype(np.int32)),
                lens=torch.from_numpy(g[prefix + "lens"].astype(np.int32)), masks=torch.from_numpy(masks)), B


def load_fixture_policy(g, which, device):
    """CatanPolicy holding the fixture's weights (every tensor's crc32 is checked against what the generator loaded into the
    reference net)."""
    from settlers_of_catan_rl_amd.policy import CatanPolicy
    net = CatanPolicy(include_lstm=(which == "lstm"))
    names = [str(n) for n in g[which + "_param_names"]]
    own = net.state_dict()
    assert sorted(own) == names, "parameter names differ from the reference net's"
    sd = fixture_state_dict({k: tuple(own[k].shape) for k in names}, which + ":" if which == "lstm" else "ff:")
    for k, want in zip(names, g[which + "_param_crc"]):
        assert tensor_crc(sd[k]) == int(want), f"fixture weight {k} differs from the generator's"
    net.load_state_dict(sd, strict=True)
    return net.to(device).eval(), names


def check_policy_fixture(g, which, device, autocast_dtype=None, tol=1e-5, grad_tol=1e-4, argmax_equal=True, min_argmax_agreement=1.0):
    """Runs CatanPolicy on the fixture's inputs and compares with what the REFERENCE net returned (tools/gen_golden.py
    gen_policy_small).  tol: |value|, |joint log-prob| (relative to max(1, |x|)), entropy; grad_tol: per-parameter gradient
    norm and hashed projection, relative to the largest gradient norm.  Returns a dict of the largest deviations."""
    import contextlib
    net, names = load_fixture_policy(g, which, device)
    x, B = decode_inputs(g, which + "_")
    x = {k: v.to(device) for k, v in x.items()}
    lstm = which == "lstm"
    ac = (lambda: torch.autocast(device_type=torch.device(device).type, dtype=autocast_dtype)) if autocast_dtype is not None else contextlib.nullcontext
    t = lambda k: torch.from_numpy(g[which + "_" + k]).to(device)
    rel = lambda a, b: float(((a.float() - b.float()).abs() / b.float().abs().clamp(min=1.0)).max())
    dev = {}
    with torch.no_grad(), ac():
        if lstm:
            v, a, lp, (h, c) = net.act(x["obs_f"], x["lists"], x["lens"], x["masks"], deterministic=True, hidden=(t("h0"), t("c0")), nonterminal=t("nt"))
            dev["act_h"] = rel(h, t("act_h")); dev["act_c"] = rel(c, t("act_c"))
        else:
            v, a, lp = net.act(x["obs_f"], x["lists"], x["lens"], x["masks"], deterministic=True)
    want_a = t("act_actions").long()
    same = (a == want_a).all(1)
    dev["act_argmax_agreement"] = float(same.float().mean())
    # the columns that MATTER for the chosen type (the env ignores the others: _translate_action, env/wrapper.py:114-166)
    typ, card = want_a[:, 0], want_a[:, 4]
    rel_cols = torch.zeros_like(want_a, dtype=torch.bool)
    rel_cols[:, 0] = True
    for ty, cols in {0: [1], 1: [2], 2: [1], 4: [4], 5: [15, 16], 6: [6] + list(range(7, 15)), 7: [5], 8: [3], 11: [6], 12: [17]}.items():
        for cc in cols:
            rel_cols[:, cc] |= typ == ty
    rel_cols[:, 15] |= (typ == 4) & ((card == 2) | (card == 4))
    rel_cols[:, 16] |= (typ == 4) & (card == 2)
    dev["act_type_agreement"] = float((a[:, 0] == typ).float().mean())
    dev["act_relevant_agreement"] = float((((a == want_a) | ~rel_cols).all(1)).float().mean())
    if argmax_equal:
        assert bool(same.all()), f"{int((~same).sum())} of {B} rows choose other arg-max actions than the reference net"
    assert dev["act_relevant_agreement"] >= min_argmax_agreement, dev
    dev["act_value"] = rel(v[same], t("act_value")[same]); dev["act_logp"] = rel(lp[same], t("act_logp")[same])
    acts = t("eval_actions").long()
    net.zero_grad()
    with ac():
        if lstm:
            v, lp, ent, (h, c) = net.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], acts, hidden=(t("hs"), t("cs")), nonterminal=t("nts"))
            dev["eval_h"] = rel(h, t("eval_h")); dev["eval_c"] = rel(c, t("eval_c"))
        else:
            v, lp, ent = net.evaluate_actions(x["obs_f"], x["lists"], x["lens"], x["masks"], acts)
    dev["eval_value"] = rel(v, t("eval_value")); dev["eval_logp"] = rel(lp, t("eval_logp"))
    dev["eval_entropy"] = abs(float(ent) - float(g[which + "_eval_entropy"]))
    wv = torch.linspace(0.5, 1.5, B, device=device)[:, None]; wl = torch.linspace(1.5, 0.5, B, device=device)[:, None]
    ((v.float() * wv).sum() + (lp.float() * wl).sum() + 3.0 * ent.float()).backward()
    prm = dict(net.named_parameters())
    gn_ref, gp_ref = g[which + "_grad_norm"], g[which + "_grad_proj"]
    scale = float(gn_ref.max())
    gn = np.array([float(prm[k].grad.double().norm()) if prm[k].grad is not None else 0.0 for k in names])
    gp = np.array([projection(k, prm[k].grad) if prm[k].grad is not None else 0.0 for k in names])
    # per parameter: relative to that parameter's own gradient norm, with a floor of a small fraction of the largest one
    den = np.maximum(gn_ref, 1e-3 * scale)
    dev["grad_norm"] = float((np.abs(gn - gn_ref) / den).max()); dev["grad_proj"] = float((np.abs(gp - gp_ref) / den).max())
    dev["worst_grad"] = names[int((np.abs(gp - gp_ref) / den).argmax())]
    for k in ("act_value", "act_logp", "eval_value", "eval_logp", "eval_entropy") + (("act_h", "act_c", "eval_h", "eval_c") if lstm else ()):
        assert dev[k] <= tol, (k, dev)
    assert dev["grad_norm"] <= grad_tol and dev["grad_proj"] <= grad_tol, dev
    return dev

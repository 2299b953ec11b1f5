"""Diagnostics (GPU box): forward+backward time of the policy's sub-modules at the config-3 minibatch width."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd.policy import CatanPolicy, _card_summary
from settlers_of_catan_rl_amd import nn_kernels, spec

MB = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
B = 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 500)
f, lists, lens = env.get_obs(); masks = env.get_action_masks(); lens = lens.long()
net = CatanPolicy().cuda()
nn_kernels.use_tuned_gemms()
rep = -(-MB // B)
fm, lm, nm, mm = (t.repeat((rep,) + (1,) * (t.dim() - 1))[:MB] for t in (f, lists, lens, masks))
o = spec.OBS_FLOAT_OFFSETS
om = net.observation_module
tiles = fm[:, o["tile_representations"]:o["tile_representations"] + 1140].reshape(MB, 19, 60)
cur = fm[:, o["current_player_main"]:o["current_player_main"] + 152]
others = fm[:, o["next_player_main"]:o["next_player_main"] + 3 * 159].reshape(MB * 3, 159)
ll = lm.long()

def timeit(name, fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); print(f"{name:48s} {(time.perf_counter() - t0) / n * 1e3:8.2f} ms", flush=True)

def fb(mod_fn):
    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mod_fn()
        y.float().sum().backward()
    return run

def fw(mod_fn):
    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            mod_fn()
    return run

te = om.tile_encoder
timeit("tile encoder fwd", fw(lambda: te(tiles)))
timeit("tile encoder fwd+bwd", fb(lambda: te(tiles)))
x64 = torch.randn(MB, 19, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True)
timeit("  one encoder layer fwd+bwd (bf16 in)", fb(lambda: te.encoder_layers[0](x64)))
timeit("  first layer + LN fwd+bwd", fb(lambda: nn_kernels.small_layer_norm(torch.nn.functional.linear(tiles, te.first_layer.weight, te.first_layer.bias), te.norm_2, True) if False else te.norm_2(torch.nn.functional.linear(tiles, te.first_layer.weight, te.first_layer.bias))))
timeit("current player module fwd+bwd", fb(lambda: om.current_player_module(cur, ll[:, 1], nm[:, 1], ll[:, 0], nm[:, 0], om.dev_card_embedding, om.hidden_card_mha, om.played_card_mha)))
timeit("other players module fwd+bwd (3B rows)", fb(lambda: om.other_players_module(others, ll[:, 2:5].reshape(MB * 3, -1), nm[:, 2:5].reshape(MB * 3), om.dev_card_embedding, om.played_card_mha)))
timeit("  one card summary fwd+bwd (B rows)", fb(lambda: _card_summary(ll[:, 0], nm[:, 0], om.dev_card_embedding, om.played_card_mha, om.current_player_module.norm)))
timeit("observation module fwd+bwd", fb(lambda: om(fm, lm, nm)))
main = torch.randn(MB, 512, device="cuda", dtype=torch.bfloat16, requires_grad=True)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    _, acts, _ = net.act(fm, lm, nm, mm)
cur_res, trade = net._custom(fm)
def heads():
    _, lp, ent = net.action_head_module(main, mm.float(), cur_res, trade, acts)
    return lp.sum() + ent
timeit("action heads fwd+bwd (evaluate)", fb(heads))
def whole():
    v, lp, ent = net.evaluate_actions(fm, lm, nm, mm, acts)
    return v.sum() + lp.sum() + ent
timeit("evaluate_actions fwd+bwd (whole net)", fb(whole))
timeit("get_value fwd only", fw(lambda: net.get_value(fm, lm, nm)))
timeit("act fwd only (65 536 rows)", fw(lambda: net.act(f, lists, lens, masks)))

"""Diagnostics (GPU box): the pieces of an inference `act` at rollout width (65 536 rows, bf16 inference copy), each timed alone
with HIP events (no stream forking: _Branches disabled), and the whole pass eager / graphed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import policy as P, nn_kernels, spec
from settlers_of_catan_rl_amd.forward_search import GraphedAct
B = 65536
env = VecCatanEnv(B, seed=0); env.random_rollout(0, 800)
f, lists, lens = env.get_obs_rows(torch.bfloat16); masks = env.get_action_masks(); lens = lens.long()
net = P.CatanPolicy().cuda().inference_copy(torch.bfloat16)
nn_kernels.use_tuned_gemms()
om = net.observation_module
o = spec.OBS_FLOAT_OFFSETS
def timeit(name, fn, n=20):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3): fn()
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
    print(f"{name:52s} gpu {a.elapsed_time(b) / n * 1e3:8.1f} us   wall {(time.perf_counter() - t0) / n * 1e6:8.1f} us", flush=True)
tiles = f[:, o["tile_representations"]:o["tile_representations"] + 1140].reshape(B, 19, 60)
cur = f[:, o["current_player_main"]:o["current_player_main"] + 152]
others = f[:, o["next_player_main"]:o["next_player_main"] + 3 * 159].reshape(B * 3, 159)
ll = lists.long()
for enabled in (False, True):
    P._Branches.enabled = enabled
    print("forked streams:", enabled)
    timeit("act (eager)", lambda: net.act(f, lists, lens, masks))
P._Branches.enabled = False
timeit("tile encoder", lambda: om.tile_encoder(tiles))
timeit("current player module", lambda: om.current_player_module(cur, ll[:, 1], lens[:, 1], ll[:, 0], lens[:, 0], om.dev_card_embedding, om.hidden_card_mha, om.played_card_mha))
timeit("other players module (3B rows)", lambda: om.other_players_module(others, ll[:, 2:5].reshape(B * 3, -1), lens[:, 2:5].reshape(B * 3), om.dev_card_embedding, om.played_card_mha))
timeit("one card summary (B lists)", lambda: P._card_summary(ll[:, 0], lens[:, 0], om.dev_card_embedding, om.played_card_mha, om.current_player_module.norm))
timeit("observation module", lambda: om(f, lists, lens))
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    main = om(f, lists, lens)
cur_res, trade = net._custom(f)
timeit("value head", lambda: net.base(f, lists, lens)[0])
timeit("action heads (sample)", lambda: net.action_head_module(main, masks.float(), cur_res, trade, None, False, None))
P._Branches.enabled = True
ga = GraphedAct(net, buckets=(B,), autocast_dtype=torch.bfloat16)
ga(f, lists, lens, masks)
timeit("act (graph replay, forked)", lambda: ga(f, lists, lens, masks, clone=False))

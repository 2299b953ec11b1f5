"""Diagnostics (GPU box): the pieces of _CardSummary.backward at a minibatch's shapes (614 400 / 204 800 real card lists)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from settlers_of_catan_rl_amd.env import VecCatanEnv
from settlers_of_catan_rl_amd import policy as P, nn_kernels, _lib
from settlers_of_catan_rl_amd.nn_kernels import _ptr, _stream, _pattern_lists
env = VecCatanEnv(65536, seed=0); env.random_rollout(0, 1200)
f, lists, lens = env.get_obs_rows(torch.bfloat16)
net = P.CatanPolicy().cuda()
om = net.observation_module
params = nn_kernels.card_summary_params(om.dev_card_embedding, om.played_card_mha, om.other_players_module.norm).detach().contiguous()
def timeit(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:60s} {a.elapsed_time(b) / n * 1e3:8.1f} us", flush=True)
L = _lib.lib()
for label, ids, ln in (("others' played lists x 614 400", torch.cat([lists[:, 2:5].reshape(-1, 25)] * 4)[:614400].contiguous(), torch.cat([lens[:, 2:5].reshape(-1)] * 4)[:614400].contiguous()),
                       ("own hidden lists x 204 800", torch.cat([lists[:, 1]] * 4)[:204800].contiguous(), torch.cat([lens[:, 1]] * 4)[:204800].contiguous())):
    ln = ln.to(torch.int32).contiguous()
    rows = ids.shape[0]
    print(label, "empty lists: %.3f" % float((ln == 0).float().mean()))
    prm = params.clone().requires_grad_(True)
    out = nn_kernels.card_summary(ids, ln, prm, 1e-5)
    dout = torch.randn_like(out)
    timeit("  forward", lambda: nn_kernels.card_summary(ids, ln, prm, 1e-5))
    timeit("  backward (whole)", lambda: torch.autograd.grad(out, prm, dout, retain_graph=True))
    keys = out.grad_fn.saved_tensors[3] if hasattr(out.grad_fn, "saved_tensors") else None
    print("  distinct patterns:", int(torch.unique(keys).numel()), " unkeyed:", int((keys < 0).sum()))
    pid, plen = _pattern_lists(ids.device)
    reps = 8
    nunk = torch.zeros((1,), dtype=torch.int32, device='cuda')
    dpat = torch.zeros((reps, pid.shape[0], 16), dtype=torch.float32, device="cuda")
    dparams = torch.zeros_like(params)
    timeit("  zeros(dpat)", lambda: dpat.zero_())
    timeit("  pattern_sum", lambda: _lib.check(L.catan_card_pattern_sum(_ptr(keys), _ptr(dout), _ptr(dpat), reps, _ptr(nunk), rows, _stream())))
    d1 = dpat.sum(0)
    timeit("  dpat.sum(0)", lambda: dpat.sum(0))
    timeit("  bwd over the patterns", lambda: _lib.check(L.catan_card_summary_bwd(_ptr(pid), 1, pid.stride(0), _ptr(plen), _ptr(params), 1e-5, _ptr(d1), _ptr(dparams), None, None, pid.shape[0], _stream())))
    timeit("  bwd over the rows (unkeyed only)", lambda: _lib.check(L.catan_card_summary_bwd(_ptr(ids), ids.element_size(), ids.stride(0), _ptr(ln), _ptr(params), 1e-5, _ptr(dout), _ptr(dparams), _ptr(keys), _ptr(nunk), rows, _stream())))

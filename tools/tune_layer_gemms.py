"""TunableOp search for the library GEMMs of FIXED shape in a config-3 minibatch step (forward, input gradient, weight gradient of
a bf16 Linear at the step's row counts) - the trunk's 992 -> 512 layer and the other wide layers whose row count does not change from
step to step.  Prints the library default vs the tuned time per product and writes the result lines (TunableOp csv) to argv[1];
the lines worth keeping are merged into settlers_of_catan_rl_amd/tunableop_gfx950.csv by hand."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import torch.cuda.tunable as tun

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_layers.csv"
LAYERS = [(204800, 992, 512, True), (204800, 512, 1536, True), (204800, 512, 256, True), (204800, 512, 128, True), (614400, 160, 256, True),
          (614400, 256, 128, True), (204800, 152, 256, True), (204800, 256, 128, True), (204800, 128, 512, False), (204800, 256, 512, False)]
torch.manual_seed(0)


def bench(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


def products(R, I, O, bias):
    x = torch.randn(R, I, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(O, I, device="cuda", dtype=torch.bfloat16) * 0.05
    b = torch.randn(O, device="cuda", dtype=torch.bfloat16) if bias else None
    g = torch.randn(R, O, device="cuda", dtype=torch.bfloat16)
    return {"fwd": lambda: F.linear(x, w, b), "dx": lambda: g @ w, "dw": lambda: g.t() @ x}


res = {}
for L in LAYERS:
    for k, fn in products(*L).items():
        res[(L, k)] = [bench(fn)]
tun.enable(True); tun.tuning_enable(True); tun.set_filename(out)
tun.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "400"))); tun.set_max_tuning_iterations(30)
for L in LAYERS:
    for k, fn in products(*L).items():
        fn(); torch.cuda.synchronize()
tun.tuning_enable(False)
for L in LAYERS:
    for k, fn in products(*L).items():
        res[(L, k)].append(bench(fn))
for (L, k), (a, b) in res.items():
    print(f"{str(L):34s} {k:3s}  default {a:8.1f} us   tuned {b:8.1f} us   {'<--' if b < 0.93 * a else ''}")

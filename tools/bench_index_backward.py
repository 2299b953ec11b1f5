import torch, time
torch.manual_seed(0)
B, U, D = 204800, 179000, 475
inv = torch.randint(0, U, (B,), device="cuda").sort().values
te_u = torch.randn(U, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
g = torch.randn(B, D, device="cuda", dtype=torch.bfloat16)
def timeit(name, fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); print(f"{name:50s} {a.elapsed_time(b)/n*1e3:9.1f} us", flush=True)
timeit("index_select fwd", lambda: te_u.index_select(0, inv))
y = te_u.index_select(0, inv)
timeit("index_select bwd (autograd)", lambda: torch.autograd.grad(y, te_u, g, retain_graph=True))
timeit("index_add_ bf16", lambda: torch.zeros(U, D, device="cuda", dtype=torch.bfloat16).index_add_(0, inv, g))
timeit("index_add_ fp32 (+casts)", lambda: torch.zeros(U, D, device="cuda").index_add_(0, inv, g.float()).to(torch.bfloat16))
y2 = te_u[inv]
timeit("te_u[inv] bwd (index_put accumulate)", lambda: torch.autograd.grad(y2, te_u, g, retain_graph=True))
lengths = torch.bincount(inv, minlength=U)
try:
    timeit("segment_reduce sum", lambda: torch.segment_reduce(g.float(), "sum", lengths=lengths, axis=0, unsafe=True))
except Exception as e:
    print("segment_reduce:", e)

"""Micro-benchmark: k_linear_rows vs F.linear for the tile-encoder / card layer shapes (bf16)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from settlers_of_catan_rl_amd import _lib
L = _lib.lib()
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (R, K, N) in [(1245184, 64, 192), (1245184, 64, 64), (1245184, 64, 128), (1245184, 128, 64), (1245184, 192, 64), (1245184, 64, 25),
                  (4915200, 16, 48), (4915200, 16, 16), (4915200, 48, 16), (65536, 128, 128)]:
    if not L.catan_linear_rows_supported(R, K, N):
        print(R, K, N, "unsupported"); continue
    x = torch.randn(R, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16); y = torch.empty(R, N, device="cuda", dtype=torch.bfloat16)
    lib = t(lambda: F.linear(x, w, b))
    mine = t(lambda: L.catan_linear_rows(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(y.data_ptr()), R, K, N, st))
    floor = R * (K + N) * 2 / 6.0e12 * 1e6
    print(f"R {R:8d} K {K:4d} N {N:4d}: library {lib:8.1f} us   k_linear_rows {mine:8.1f} us   (HBM floor at 6 TB/s {floor:6.1f} us)")
